import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np, time
from multi_car_racing_amd.vec_env import VecMultiCarRacing
env = VecMultiCarRacing(64, 2, seed=3, max_episode_steps=int(os.environ.get("L", 20)), auto_reset=True)
env.reset(); env.wait_refills()
print("after reset: generated", env.episodes_generated)
a = torch.rand((64,2,3), device='cuda')
nd=0
for k in range(int(os.environ.get("K", 100))):
    o,r,d,i = env.step(a); nd += int(d.sum().item())
    if k%int(os.environ.get("E", 20))==int(os.environ.get("E", 20))-1:
        env.wait_refills(); print(k, "done so far", nd, "generated", env.episodes_generated, "active", None)
es = env.get_env_state(); print("t", es["t"][:8])
env.close()
