"""CPU: the C-ABI library loads without a GPU and exports every symbol include/mcr.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "mcr.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mcr_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(lib):
    L = lib.load()
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/mcr.h but not exported by libmcr_hip.so"
    assert set(names) == set(lib.SYMBOLS), "multi_car_racing_amd/_lib.py binds a different symbol set than the header declares"


def test_version_and_sizes(lib):
    L = lib.load()
    assert b"gfx950" in L.mcr_version()
    assert lib.episode_bytes() % 16 == 0 and lib.episode_bytes() > 80000
    assert ctypes.sizeof(lib.Config) == 56          # include/mcr.h: 10 x int32, double h_ratio, skid_particles, fresh_world


def test_library_contains_gfx950_code_object():
    so = os.path.join(ROOT, "multi_car_racing_amd", "_lib", "libmcr_hip.so")
    data = open(so, "rb").read()
    assert b"gfx950" in data and b"k_dynamics" in data and b"k_view" in data and b"k_collide" in data


def test_missing_library_fails_loudly(lib, monkeypatch):
    from multi_car_racing_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libmcr_hip.so")
    import pytest
    with pytest.raises(_lib.McrError):
        _lib.load()


def test_step_path_refuses_cpu(lib):
    import pytest, torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from multi_car_racing_amd import McrError
    from multi_car_racing_amd.vec_env import VecMultiCarRacing
    with pytest.raises(McrError):
        VecMultiCarRacing(4, 2)


def test_argument_validation_without_a_gpu(lib):
    """The C layer never throws or exits: bad arguments come back as negative status codes with a message, before any
    HIP call is made (so this runs on a CPU-only box)."""
    L = lib.load()
    h = ctypes.c_void_p()
    for B, N in ((0, 2), (4, 0), (4, 9), (-1, 2)):
        cfg = lib.Config(B, N, 0, 1, 1, 1, 0, 1, 1000, 1, 0.25)
        assert L.mcr_create(ctypes.byref(cfg), ctypes.byref(h)) < 0 and b"out of range" in L.mcr_last_error()
    assert L.mcr_create(None, ctypes.byref(h)) < 0
    assert L.mcr_step(None, None, None, None, None, None, None) < 0
    assert L.mcr_reset(None, None, None, None) < 0
    assert L.mcr_render(None, 0, 600, 400, None, None) < 0
    assert L.mcr_set_episode_stats(None, None, None) < 0
    assert L.mcr_destroy(None) < 0


def test_synth_actions_host_is_counter_based(lib):
    """mcr_synth_actions_host: a pure function of (seed, global env, agent, t) — slicing the batch (env_offset) or asking
    again gives the same values; ranges follow the action_space bounds (multi_car_racing.py:162-165)."""
    import ctypes
    import numpy as np
    L = lib.load()
    full = np.zeros((64, 3, 3), np.float32)
    L.mcr_synth_actions_host(lib.ptr(full), 64, 3, ctypes.c_uint64(7), ctypes.c_uint32(11), ctypes.c_uint32(100))
    part = np.zeros((16, 3, 3), np.float32)
    L.mcr_synth_actions_host(lib.ptr(part), 16, 3, ctypes.c_uint64(7), ctypes.c_uint32(11), ctypes.c_uint32(120))
    assert np.array_equal(part, full[20:36])
    other = np.zeros_like(full)
    L.mcr_synth_actions_host(lib.ptr(other), 64, 3, ctypes.c_uint64(7), ctypes.c_uint32(12), ctypes.c_uint32(100))
    assert not np.array_equal(other, full)
    big = np.zeros((4096, 2, 3), np.float32)
    L.mcr_synth_actions_host(lib.ptr(big), 4096, 2, ctypes.c_uint64(1), ctypes.c_uint32(0), ctypes.c_uint32(0))
    assert (big[..., 0] >= -1).all() and (big[..., 0] < 1).all() and (big[..., 1:] >= 0).all() and (big[..., 1:] < 1).all()
    assert abs(big[..., 0].mean()) < 0.03 and abs(big[..., 1].mean() - 0.5) < 0.02 and abs(big[..., 2].mean() - 0.5) < 0.02
    assert abs(np.corrcoef(big[:, 0, 1], big[:, 1, 1])[0, 1]) < 0.06
