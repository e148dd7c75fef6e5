"""Host-side episode generation rate (tracks/s) vs thread count: the refill budget of the auto-reset path."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_car_racing_amd import _lib
import ctypes
L = _lib.load()
n = 2048
mt_t = np.zeros((n, _lib.MT_WORDS), np.uint32); mt_d = np.zeros((n, _lib.MT_WORDS), np.uint32)
for e in range(n):
    L.mcr_mt_seed(_lib.ptr(mt_t[e]), ctypes.c_uint32(e)); L.mcr_mt_seed(_lib.ptr(mt_d[e]), ctypes.c_uint32(e + 2 ** 31))
blobs = np.empty((n, _lib.episode_bytes()), np.uint8); info = np.zeros((n, 12), np.int32)
print("effective cpus", _lib.effective_cpus(), "logical", os.cpu_count())
for th in (1, 2, 4, 8, 15, 32):
    t0 = time.perf_counter()
    _lib.check(L.mcr_episodes_generate(_lib.ptr(mt_t), _lib.ptr(mt_d), n, 2, 2, _lib.ptr(blobs), _lib.ptr(info), th))
    dt = time.perf_counter() - t0
    print(f"threads {th:3d}: {n / dt:9.0f} episodes/s  ({dt / n * 1e3 * th:.3f} ms of thread time per episode, mean retries {info[:, 2].mean():.2f})")
