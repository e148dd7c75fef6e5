"""Minimal step loop for profilers: python tools/step_loop.py [steps] [B] [N] [obs] [debug bits]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multi_car_racing_amd.vec_env import VecMultiCarRacing
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
N = int(sys.argv[3]) if len(sys.argv) > 3 else 2
obs = int(sys.argv[4]) if len(sys.argv) > 4 else 1
env = VecMultiCarRacing(B, N, seed=1, use_random_direction=True, auto_reset=True, obs=bool(obs))
env.reset()
if len(sys.argv) > 5:
    from multi_car_racing_amd import _lib
    _lib.check(env.L.mcr_debug_set(env.h, int(sys.argv[5])))
act = torch.rand((B, N, 3), device="cuda"); act[..., 0] = act[..., 0] * 2 - 1
for _ in range(steps): env.step(act)
torch.cuda.synchronize()
env.close()
