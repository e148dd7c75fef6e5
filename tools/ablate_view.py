"""Within-process ablation of the raster kernel (mcr_debug_set bits) -> per-phase cost. Not a test."""
import sys, os, ctypes, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multi_car_racing_amd.vec_env import VecMultiCarRacing
from multi_car_racing_amd import _lib

B, N = int(os.environ.get("B", 4096)), int(os.environ.get("N", 2))
env = VecMultiCarRacing(B, N, seed=1, use_random_direction=True, auto_reset=True, streams=int(os.environ.get("STREAMS", 0)), car_contacts=bool(int(os.environ.get("CONTACTS", 1))))
env.reset()
pool = torch.rand((64, B, N, 3), device="cuda"); pool[..., 0] = pool[..., 0] * 2 - 1
if int(os.environ.get("CONST", 0)): pool[:] = pool[0]
_k = [0]
def nxt():
    _k[0] += 1; return pool[_k[0] % 64]
for _ in range(80): env.step(nxt())
torch.cuda.synchronize()
names = {0: "full", 31: "-all"} if int(os.environ.get("QUICK", 0)) else {0: "full", 1: "-flags", 2: "-road shade", 4: "-cars", 8: "-writeout", 16: "-cull", 31: "-all", 18: "-cull-road", 26: "-cull-road-writeout"}
res = {}
for rnd in range(3):
    for mask in names:
        _lib.check(env.L.mcr_debug_set(env.h, mask))
        env.timing(31)
        for _ in range(30): env.step(nxt())
        ms, n = env.timing_read(); env.timing(0)
        res.setdefault(mask, []).append(ms / np.maximum(n, 1))
_lib.check(env.L.mcr_debug_set(env.h, 0))
for mask, v in res.items():
    v = np.array(v)
    print(f"{names[mask]:>22}: view {np.median(v[:,2])*1e3:8.1f} us   collide {np.median(v[:,0])*1e3:6.1f} us  dynamics {np.median(v[:,1])*1e3:6.1f} us  reset-pass {np.median(v[:,3])*1e3:5.1f}+{np.median(v[:,4])*1e3:5.1f} us")
env.close()
