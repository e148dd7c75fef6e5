"""Distribution of touching car<->car fixture pairs per env over a steady-state rollout (GPU only, diagnostics)."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_car_racing_amd.vec_env import VecMultiCarRacing
from multi_car_racing_amd import _lib
B, N = 4096, int(os.environ.get("N", "2"))
env = VecMultiCarRacing(B, N, seed=0, auto_reset=True)
env.reset()
g = torch.Generator(device="cuda"); g.manual_seed(1)
pool = torch.rand((64, B, N, 3), device="cuda", generator=g); pool[..., 0] = pool[..., 0] * 2 - 1
if int(os.environ.get("DRIVE", "0")): pool[..., 0] *= 0.1; pool[..., 1] = 1.0; pool[..., 2] = 0.0      # bench.py --actions drive
cnt = np.zeros(B, np.int32); hist = np.zeros(32, int); mx = []
for k in range(1000):
    env.step(pool[k % 64])
    if k % 10 == 9:
        _lib.check(env.L.mcr_debug_read_contact_counts(env.h, _lib.ptr(cnt)))
        h = np.bincount(cnt, minlength=32)[:32]
        if k >= 300: hist += h; mx.append(cnt.max())
print("manifolds per env (samples from steps 300..1000):", {i: int(v) for i, v in enumerate(hist) if v})
print("max per sampled step: mean %.1f" % np.mean(mx), "values", sorted(set(mx)))
env.close()
