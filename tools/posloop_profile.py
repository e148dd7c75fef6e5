"""Where a position sweep of the contact chain goes (build with MCR_EXTRA_CFLAGS=-DMCR_POSLOOP_PROFILE python -m multi_car_racing_amd.build --force):
clock of lane 0 per segment, summed over the sweeps of a wavefront.  GPU only, diagnostics.   N=8 python tools/posloop_profile.py"""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_car_racing_amd.vec_env import VecMultiCarRacing
from multi_car_racing_amd import _lib
B, N = 4096, int(os.environ.get("N", "8"))
env = VecMultiCarRacing(B, N, seed=0, auto_reset=True, streams=2)
env.reset()
g = torch.Generator(device="cuda"); g.manual_seed(1)
pool = torch.rand((64, B, N, 3), device="cuda", generator=g); pool[..., 0] = pool[..., 0] * 2 - 1
if int(os.environ.get("DRIVE", "0")): pool[..., 0] *= 0.1; pool[..., 1] = 1.0; pool[..., 2] = 0.0      # bench.py --actions drive
G = 1
while G < N: G *= 2
nb = (B * G + 63) // 64
ns = (B + 1) // 2
buf = np.zeros((nb + 2 * ns) * 8, np.uint64)
VEL = int(os.environ.get("VEL", "0"))          # 1: the velocity sweeps instead (debug bit 16)
_lib.check(env.L.mcr_debug_set(env.h, 256 | (65536 if VEL else 0)))
rows = []
for k in range(600):
    env.step(pool[k % 64])
    if k >= 200 and k % 10 == 0:
        _lib.check(env.L.mcr_debug_read_dynamics_stamps(env.h, _lib.ptr(buf), len(buf)))
        st = buf.reshape(nb + 2 * ns, 8)[nb:nb + ns]
        if VEL:
            for r in st[(st[:, 0] > 0) & (st[:, 5] > 0)]:
                rows.append((180, int(r[5] & np.uint64(0xffffffff)), int(r[5] >> np.uint64(32)), int(r[6] & np.uint64(0xffffffff)), int(r[6] >> np.uint64(32)), int(r[7])))
            continue
        ok = (st[:, 0] > 0) & ((st[:, 5] >> np.uint64(48)) > 0)
        for r in st[ok]:
            n = int(r[5] >> np.uint64(48))
            rows.append((n, int(r[5] & np.uint64((1 << 48) - 1)), int(r[6] & np.uint64(0xffffffff)), int(r[6] >> np.uint64(32)), int(r[7] & np.uint64((1 << 48) - 1)), int(r[7] >> np.uint64(48))))
d = np.array(rows, np.float64)
print(f"N={N}: {len(d)} contact wavefronts, sweeps per wavefront mean {d[:, 0].mean():.1f}; ticks per sweep (2.1 GHz: 1000 ticks = 0.48 us)")
names = ["4 joints", "exchange out + barrier", "leader: contacts (cc_velocity)", "exchange in"] if VEL else ["exchange out + barrier", "leader: contacts (cc_position)", "exchange in + 4 joint corrections", "island bookkeeping + 2 barriers"]
for i, n in enumerate(names):
    per = d[:, 1 + i] / d[:, 0]
    print(f"   {n:36s} mean {per.mean():8.0f}  median {np.median(per):8.0f}")
print(f"   {'total':36s} mean {(d[:, 1:5].sum(1) / d[:, 0]).mean():8.0f}")
print("by the env's manifold count (wavefronts; ticks per sweep by segment; total):")
for c in sorted(set(d[:, 5].astype(int))):
    m = d[d[:, 5] == c]
    print(f"   {c:2d} manifolds: {len(m):6d}   " + "  ".join(f"{(m[:, 1 + i] / m[:, 0]).mean():7.0f}" for i in range(4)) + f"   {(m[:, 1:5].sum(1) / m[:, 0]).mean():8.0f}   sweeps {m[:, 0].mean():.1f}")
