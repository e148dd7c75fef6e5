"""Clock stamps of lane 0 at the phase boundaries of the step's contact pass (k_collide, debug bit 15) in a steady-state rollout:
where does an env's wavefront spend its time?  GPU only, diagnostics.   N=8 python tools/collide_phases.py"""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_car_racing_amd.vec_env import VecMultiCarRacing
from multi_car_racing_amd import _lib
B, N = 4096, int(os.environ.get("N", "2"))
env = VecMultiCarRacing(B, N, seed=0, auto_reset=True, streams=2)
env.reset()
g = torch.Generator(device="cuda"); g.manual_seed(1)
pool = torch.rand((64, B, N, 3), device="cuda", generator=g); pool[..., 0] = pool[..., 0] * 2 - 1
if int(os.environ.get("DRIVE", "0")): pool[..., 0] *= 0.1; pool[..., 1] = 1.0; pool[..., 2] = 0.0      # bench.py --actions drive
buf = np.zeros(B * 8, np.uint64)
_lib.check(env.L.mcr_debug_set(env.h, 32768))
names = ["fixtures + proxies (all 8 per car)", "car<->car broadphase contacts", "tile candidates + overlap tests", "tile contact state + event replay",
         "result stores", "car<->car manifolds + island order"]
rows = []
for k in range(700):
    env.step(pool[k % 64])
    if k >= 300 and k % 10 == 0:
        _lib.check(env.L.mcr_debug_read_dynamics_stamps(env.h, _lib.ptr(buf), len(buf)))
        st = buf.reshape(B, 8).astype(np.int64)
        ok = (st[:, 0] > 0) & (st[:, 6] > st[:, 0])
        rows.append(np.diff(st[ok, :7], axis=1))
d = np.concatenate(rows)
tot = d.sum(1)
print(f"N={N}: {len(d)} env passes; clock ticks of the shader clock (100 ticks ~ 0.05 us at 2.1 GHz)")
for i, n in enumerate(names):
    print(f"   {n:44s} mean {d[:, i].mean():8.0f}  median {np.median(d[:, i]):8.0f}  p99 {np.percentile(d[:, i], 99):8.0f}  ({100 * d[:, i].sum() / tot.sum():4.1f} %)")
print(f"   {'total':44s} mean {tot.mean():8.0f}  median {np.median(tot):8.0f}  p99 {np.percentile(tot, 99):8.0f}")
