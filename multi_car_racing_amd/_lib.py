"""ctypes binding of include/mcr.h.  Fails loudly if the HIP library is missing — there is no CPU fallback."""
import ctypes, os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MCR_LIB_PATH") or os.path.join(HERE, "_lib", "libmcr_hip.so")      # (MCR_LIB_PATH: another build of the same ABI, for A/B measurements)

MAX_AGENTS = 8
TILE_CAP = 512
QUAD_CAP = 768
MT_WORDS = 625

_vp, _i, _d = ctypes.c_void_p, ctypes.c_int, ctypes.c_double


class Config(ctypes.Structure):
    _fields_ = [("num_envs", ctypes.c_int32), ("num_agents", ctypes.c_int32), ("device", ctypes.c_int32),
                ("obs_enabled", ctypes.c_int32), ("auto_reset", ctypes.c_int32), ("backwards_flag", ctypes.c_int32),
                ("use_ego_color", ctypes.c_int32), ("car_contacts", ctypes.c_int32), ("max_episode_steps", ctypes.c_int32),
                ("num_streams", ctypes.c_int32), ("h_ratio", ctypes.c_double),
                ("skid_particles", ctypes.c_int32), ("fresh_world", ctypes.c_int32)]


# every symbol include/mcr.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "mcr_last_error": (ctypes.c_char_p, []),
    "mcr_version": (ctypes.c_char_p, []),
    "mcr_create": (_i, [ctypes.POINTER(Config), ctypes.POINTER(_vp)]),
    "mcr_destroy": (_i, [_vp]),
    "mcr_episode_bytes": (ctypes.c_size_t, []),
    "mcr_mt_seed": (None, [_vp, ctypes.c_uint32]),
    "mcr_mt_seed_by_array": (None, [_vp, _vp, _i]),
    "mcr_mt_random_sample": (_d, [_vp]),
    "mcr_mt_choice_cw": (_i, [_vp]),
    "mcr_mt_car_order": (None, [_vp, _i, _vp]),
    "mcr_episode_generate": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "mcr_episodes_generate": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _i]),
    "mcr_episodes_generate_rows": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i]),
    "mcr_episode_unpack": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mcr_stage_episodes": (_i, [_vp, _vp, _i, _vp, _vp]),
    "mcr_reset": (_i, [_vp, _vp, _vp, _vp]),
    "mcr_step": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mcr_poll_consumed": (_i, [_vp, _vp, _i, _vp]),
    "mcr_refill_start": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    "mcr_refill_stop": (_i, [_vp]),
    "mcr_refill_wait": (_i, [_vp]),
    "mcr_refill_lag": (_i, [_vp]),
    "mcr_refill_hold": (_i, [_vp, _i]),
    "mcr_refill_generated": (ctypes.c_longlong, [_vp]),
    "mcr_refill_debug": (_i, [_vp, _vp]),
    "mcr_get_state": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mcr_set_bodies": (_i, [_vp, _vp]),
    "mcr_get_env_state": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mcr_get_positions": (_i, [_vp, _vp]),
    "mcr_mass_props": (None, [_vp]),
    "mcr_sincos_host": (None, [ctypes.c_float, _vp, _vp]),
    "mcr_sincos_device": (_i, [_vp, _vp, _vp, _vp, _i, _vp]),
    "mcr_timing_enable": (_i, [_vp, _i]),
    "mcr_debug_set": (_i, [_vp, _i]),
    "mcr_debug_read_view_scratch": (_i, [_vp, _i, _vp, _i]),
    "mcr_set_episode_stats": (_i, [_vp, _vp, _vp]),
    "mcr_set_terminal_obs": (_i, [_vp, _vp, _vp, _vp, _i]),
    "mcr_read_rollout_stats": (_i, [_vp, _vp, _i]),
    "mcr_render": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "mcr_debug_read_contact_counts": (_i, [_vp, _vp]),
    "mcr_debug_read_proxy_ids": (_i, [_vp, _i, _vp, _i]),
    "mcr_debug_read_env_records": (_i, [_vp, _vp, _i]),
    "mcr_debug_read_partition": (_i, [_vp, _vp, _vp]),
    "mcr_debug_next_verdicts": (_i, [_vp, _i, _vp]),
    "mcr_debug_read_counters": (_i, [_vp, _vp]),
    "mcr_debug_read_counters8": (_i, [_vp, _vp]),
    "mcr_debug_read_verdict_mismatches": (_i, [_vp, _vp]),
    "mcr_concurrent_collide": (_i, [_vp]),
    "mcr_step_ordering": (_i, [_vp]),
    "mcr_step_ordering_for": (_i, [_vp, _vp]),
    "mcr_bind_stream": (_i, [_vp, _vp]),
    "mcr_debug_overlap": (_i, [_vp, _i, _vp, _vp, _i, _vp]),
    "mcr_status": (_i, [_vp, _vp, _i]),
    "mcr_world_create": (_vp, [_i]),
    "mcr_world_destroy": (None, [_vp]),
    "mcr_world_reset": (_i, [_vp, _vp]),
    "mcr_world_step": (_i, [_vp, _vp]),
    "mcr_world_proxy_ids": (_i, [_vp, _vp, _i]),
    "mcr_debug_read_dynamics_stamps": (_i, [_vp, _vp, _i]),
    "mcr_timing_read": (_i, [_vp, _vp, _vp]),
    "mcr_set_step_graph": (_i, [_vp, _i]),
    "mcr_state_blob_bytes": (ctypes.c_size_t, [_vp]),
    "mcr_get_state_blob": (_i, [_vp, _i, _vp]),
    "mcr_set_state_blob": (_i, [_vp, _i, _vp]),
    "mcr_synth_actions": (_i, [_vp, _vp, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, _vp]),
    "mcr_synth_actions_block": (_i, [_vp, _vp, ctypes.c_uint64, ctypes.c_uint32, _i, ctypes.c_uint32, _vp]),
    "mcr_synth_actions_host": (None, [_vp, _i, _i, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32]),
}

_lib = None


class McrError(RuntimeError):
    pass


class McrWarning(UserWarning):
    """the build works, but not the way its numbers were measured (e.g. a second handle on a device orders its streams with events)"""


def load():
    """dlopen the in-tree library.  Import torch FIRST in GPU processes so both share one libamdhip64."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise McrError(f"{LIB_PATH} is missing: run `python -m multi_car_racing_amd.build` "
                       "(hipcc --offload-arch=gfx950). There is no CPU fallback for the step path.")
    try:
        # PyTorch-ROCm bundles its own libamdhip64.so.7; load it FIRST so that this library (same soname)
        # binds to the same HIP runtime instance that owns torch's device memory and streams.
        import torch  # noqa: F401
    except Exception:
        pass
    L = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(L, name)           # AttributeError here == header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def effective_cpus():
    """CPUs this process may actually use: min(affinity mask, cgroup v2/v1 CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, n)


def check(rc, what=""):
    if rc < 0:
        raise McrError(f"{what} failed ({rc}): {load().mcr_last_error().decode()}")
    return rc


def ptr(a):
    """host numpy array -> void*"""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.c_void_p)


def episode_bytes():
    return int(load().mcr_episode_bytes())


def unpack_episode(blob):
    """host blob (uint8 array) -> dict(track (T,3) x,y,beta; quads (P,4,2) f32; quad_meta (P,) u32; spawn (8,3); cw)"""
    L = load()
    T, P, cw = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    L.mcr_episode_unpack(ptr(blob), ctypes.byref(T), ctypes.byref(P), ctypes.byref(cw), None, None, None, None, None)
    track = np.zeros((T.value, 3)); quads = np.zeros((P.value, 8), np.float32); meta = np.zeros(P.value, np.uint32)
    spawn = np.zeros((MAX_AGENTS, 3)); alpha = np.zeros(T.value)
    L.mcr_episode_unpack(ptr(blob), None, None, None, ptr(track), ptr(quads), ptr(meta), ptr(spawn), ptr(alpha))
    return dict(T=T.value, P=P.value, cw=bool(cw.value), track=track, alpha=alpha, quads=quads.reshape(-1, 4, 2),
                quad_meta=meta, spawn=spawn)
