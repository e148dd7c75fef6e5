"""Where and when the raster's workgroups ran (debug bit 32 stamps): views per CU, start/end spread. Not a test."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multi_car_racing_amd.vec_env import VecMultiCarRacing
from multi_car_racing_amd import _lib
B, N = 4096, 2
env = VecMultiCarRacing(B, N, seed=1, use_random_direction=True, auto_reset=True)
env.reset()
pool = torch.rand((64, B, N, 3), device="cuda"); pool[..., 0] = pool[..., 0] * 2 - 1
for k in range(80): env.step(pool[k % 64])
_lib.check(env.L.mcr_debug_set(env.h, 32))
for k in range(3): env.step(pool[k])
torch.cuda.synchronize()
rows = []
for v in range(B * N):
    buf = np.zeros(13, np.uint64)
    env.L.mcr_debug_read_view_scratch(env.h, v, _lib.ptr(buf), 104)
    rows.append(buf.astype(np.int64))
d = np.array(rows)
end = d[:, 10]; t0 = end.min()
end_us = (end - t0) / 100.0
wg = d[:, 11] & 0xffffffff; vs = d[:, 11] >> 32
hw = d[:, 12] & 0xffffffff; xcc = (d[:, 12] >> 32) & 0xf
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
cuid = xcc * 1000 + se * 100 + sh * 16 + cu
dur = d[:, :9].sum(1)
print("views", len(d), " last view ends %.1f us after the first" % end_us.max())
print("views per workgroup: ", collections.Counter(collections.Counter(wg.tolist()).values()))
per_cu = collections.Counter(cuid.tolist())
print("distinct CUs", len(per_cu), " views per CU: min %d max %d" % (min(per_cu.values()), max(per_cu.values())), " histogram", sorted(collections.Counter(per_cu.values()).items()))
wgs_per_cu = collections.Counter()
for c, w in set(zip(cuid.tolist(), wg.tolist())): wgs_per_cu[c] += 1
print("workgroups per CU histogram", sorted(collections.Counter(wgs_per_cu.values()).items()))
print("end time of each workgroup's LAST view: p10 %.1f p50 %.1f p90 %.1f max %.1f us" % tuple(np.percentile([end_us[wg == w].max() for w in np.unique(wg)], [10, 50, 90, 100])))
print("end time of each workgroup's FIRST view: p10 %.1f p50 %.1f p90 %.1f max %.1f us" % tuple(np.percentile([end_us[wg == w].min() for w in np.unique(wg)], [10, 50, 90, 100])))
print("per-view ticks by view number in the workgroup:", {int(k): int(np.median(dur[vs == k])) for k in np.unique(vs)[:14]})
env.close()
