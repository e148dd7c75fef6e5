"""Per-phase latency of the raster kernel from in-kernel s_memtime stamps (debug bit 32). Not a test."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multi_car_racing_amd.vec_env import VecMultiCarRacing
from multi_car_racing_amd import _lib
B, N = 4096, int(os.environ.get("N", "2"))
env = VecMultiCarRacing(B, N, seed=1, use_random_direction=True, auto_reset=True)
env.reset()
pool = torch.rand((64, B, N, 3), device="cuda"); pool[..., 0] = pool[..., 0] * 2 - 1
for k in range(80): env.step(pool[k % 64])
_lib.check(env.L.mcr_debug_set(env.h, 32 | int(os.environ.get('DBG', 0))))
for k in range(3): env.step(pool[k])
torch.cuda.synchronize()
names = ["prologue (agent 0: env-level loads) / hand-over", "wait: barrier", "clear + label + candidates (cull, set-up, scan)", "wait: barrier",
         "next fetch + task fill", "wait: barrier", "span fill", "wait: barrier", "resolve + write-out"]
rows = []
for v in range(0, B * N, 97):
    buf = np.zeros(16, np.uint64)
    env.L.mcr_debug_read_view_scratch(env.h, v, _lib.ptr(buf), 128)
    if buf[:9].sum() > 0:
        rows.append(np.concatenate([buf[:10].astype(np.int64), [v % N], buf[13:16].astype(np.int64)]))
d = np.array(rows)
print("views sampled", len(d), " rounds per view: mean %.2f" % d[:, 9].mean(), " (clock ticks of thread 0, summed over the view's rounds)")
for i, nme in enumerate(names):
    print(f"{nme:>52}: median {np.median(d[:, i]):8.0f} ticks  mean {d[:, i].mean():8.0f}  p90 {np.percentile(d[:, i], 90):8.0f}")
print(f"{'total per view':>52}: median {np.median(d[:, :9].sum(1)):8.0f} ticks   (agent 0: {np.median(d[d[:, 10] == 0][:, :9].sum(1)):.0f}, later agents: {np.median(d[d[:, 10] > 0][:, :9].sum(1)):.0f})")
print('wavefronts 1..3 reach the candidates barrier (ticks after the view\'s start; wavefront 0: the clear + candidates row above):', [int(np.median(d[:, 11 + i])) for i in range(3)])
env.close()
