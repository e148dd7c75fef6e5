import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from multi_car_racing_amd.sharded import ShardedVecEnv
B=4096
env = ShardedVecEnv(B, 2, seed=0, rank=0, world_size=1, device=torch.device("cuda",0), obs=True, auto_reset=True, use_random_direction=True)
env.reset()
pool = torch.rand((64, B, 2, 3), device="cuda"); pool[..., 0] = pool[..., 0]*2-1
tot = torch.zeros((), device="cuda", dtype=torch.int64)
for k in range(1064):
    o, r, d, i = env.step(pool[k % 64]); tot += d.sum()
    if k in (63, 900, 998, 999, 1000, 1001, 1063):
        torch.cuda.synchronize(); env.wait_refills(); print(k, "done total", int(tot.item()), "generated", env.env.episodes_generated)
env.close()
