"""Build the in-tree HIP library (gfx950).  `python -m multi_car_racing_amd.build` or __graft_entry__.build()."""
import os, subprocess, sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "_lib", "libmcr_hip.so")
# translation units and their extra flags (mcr_view.hip explains why the raster is built without the SLP vectoriser)
# mcr_hip.hip: rounds 3-4 tuned the SLP vectoriser's threshold for this unit (its packed f32 forms paid in k_dynamics' velocity sweeps: off 135 us,
# threshold 5 112.6 us).  Since round 5 the sweeps are written on pairs by hand (k_dynamics.h: joint_velocity) and the vectoriser only costs: same
# box, alternating, threshold 5 / 12 / off: N=2 18.63 / 18.67 / 18.70 M env-steps/s, N=4 12.75 / 12.76 / 13.13, N=8 7.02 / 7.04 / 7.22,
# --actions drive 5.17 / 5.20 / 5.41 (k_collide 52 / 42 / 42 us, resume chain 72 / 69 / 68 us; the contact chains' scalar sweeps lose their
# half-packed forms and the register shuffling that came with them).  Same IEEE operations either way (-ffp-contract=off): the parity suite is the proof.
SOURCES = [("mcr_hip.hip", ["-fno-slp-vectorize"] + os.environ.get("MCR_HIP_CFLAGS", "").split()), ("mcr_view.hip", ["-fno-slp-vectorize"]), ("mcr_host.cpp", []), ("mcr_world.cpp", [])]


def deps():
    """every file the library is built from: all of csrc/ plus the public header"""
    import glob
    out = [f for pat in ("*.h", "*.hip", "*.cpp", "*.inc") for f in glob.glob(os.path.join(CSRC, pat))]
    return out + [os.path.join(HERE, "..", "include", "mcr.h"), os.path.abspath(__file__)]      # (this file: the per-file flags)

# -ffp-contract=off: host (x86-64) and gfx950 must round identically (numerics)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-value"] + os.environ.get("MCR_EXTRA_CFLAGS", "").split()   # (build-time diagnostics only)


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in deps())


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    objdir = os.path.join(os.path.dirname(LIB), "obj")
    os.makedirs(objdir, exist_ok=True)
    objs, procs = [], []
    for src, extra in SOURCES:                       # the translation units compile side by side
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        cmd = [hipcc] + FLAGS + extra + ["-c", "-o", obj, os.path.join(CSRC, src)]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
