"""Host CPU cost of one batched step (no resets in the measured window): wall and CPU time of env.step() calls issued back to
back (the device queue absorbs them), of the consumed-episode poll alone, and of a bare mcr_step.  usage: python tools/host_cost.py [N]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multi_car_racing_amd.vec_env import VecMultiCarRacing
from multi_car_racing_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = 4096
env = VecMultiCarRacing(B, N, seed=3, auto_reset=True, max_episode_steps=100000, car_contacts=True, streams=2)
env.reset()
a = torch.zeros((B, N, 3), device="cuda"); a[..., 1] = 0.3
for _ in range(50): env.step(a)
torch.cuda.synchronize()
def timed(fn, n):
    c0, t0 = time.thread_time(), time.perf_counter()
    for _ in range(n): fn()
    c1, t1 = time.thread_time(), time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6, (c1 - c0) / n * 1e6
print("ordering", env.L.mcr_step_ordering(env.h))
for rep in range(2):
    w, c = timed(lambda: env.step(a), 300); print(f"env.step()           wall {w:6.1f} us  cpu(thread) {c:6.1f} us per call")
ids = np.zeros(B, np.int32)
w, c = timed(lambda: env.L.mcr_poll_consumed(env.h, _lib.ptr(ids), B, None), 2000); print(f"mcr_poll_consumed    wall {w:6.1f} us  cpu {c:6.1f} us")
st = torch.cuda.current_stream()
args = (env.h, ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(env.obs.data_ptr()), ctypes.c_void_p(env.reward.data_ptr()), ctypes.c_void_p(env.done.data_ptr()), ctypes.c_void_p(env.truncated.data_ptr()), ctypes.c_void_p(st.cuda_stream))
for rep in range(2):
    w, c = timed(lambda: env.L.mcr_step(*args), 300); print(f"bare mcr_step        wall {w:6.1f} us  cpu {c:6.1f} us")
env.close()
