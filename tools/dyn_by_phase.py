"""Per-step kernel times as a function of the episode step (all envs in phase): shows which part of an episode is
expensive for which kernel.  GPU only."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_car_racing_amd.vec_env import VecMultiCarRacing
B, N = 4096, 2
env = VecMultiCarRacing(B, N, seed=0, auto_reset=True, car_contacts=bool(int(os.environ.get("CONTACTS", "1"))))
env.reset()
from multi_car_racing_amd import _lib
_lib.check(env.L.mcr_debug_set(env.h, int(os.environ.get("DEBUG", "0"))))
g = torch.Generator(device="cuda"); g.manual_seed(1)
pool = torch.rand((64, B, N, 3), device="cuda", generator=g); pool[..., 0] = pool[..., 0] * 2 - 1
rows = []
for k in range(int(os.environ.get("STEPS", "1100"))):
    env.timing(7)
    env.step(pool[k % 64])
    torch.cuda.synchronize()
    ms, nl = env.timing_read()
    rows.append(ms[:3])
rows = np.array(rows)
for a, b in [(0, 5), (5, 10), (10, 20), (20, 30), (30, 40), (40, 50), (50, 60), (60, 80), (80, 100), (100, 150), (150, 200), (200, 300), (300, 500), (500, 800), (800, 995), (995, 1005), (1005, 1050), (1050, 1100)]:
    if b <= len(rows):
        r = rows[a:b].mean(0)
        print(f"steps {a:4d}-{b:4d}: collide {r[0]*1e3:7.1f} us  dynamics {r[1]*1e3:7.1f} us  view {r[2]*1e3:7.1f} us")
env.close()
