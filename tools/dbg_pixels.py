import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle as O
from tests.util import random_actions
from tests.test_gpu_parity import _make, _oracles, _rear_end_setup
N=4; B, seed = 5, 60 + N
env = _make(B, N, seed, contacts=True); env.reset()
orcs = _oracles(O, B, N, seed, contacts=True)
_rear_end_setup(env, orcs)
rng = np.random.RandomState(4)
for k in range(160):
    a = random_actions(rng, B, N, 0.0)
    a[:, 0, 1] = 0.0; a[:, 0, 2] = 0.8 if k < 60 else 0.0
    a[:, 1, 0] *= 0.2; a[:, 1, 1] = 1.0
    obs, rew, done, _ = env.step(torch.from_numpy(a).cuda())
    for e, o in enumerate(orcs): o.step(a[e], render=False)
    ob = obs.cpu().numpy()
    es = env.get_env_state()
    for e, o in enumerate(orcs):
        eo = o.env_state()
        if not np.array_equal(es["driving_backward"][e], eo["driving_backward"]) or not np.array_equal(es["driving_on_grass"][e], eo["driving_on_grass"]):
            print("FLAG MISMATCH step", k, "env", e, es["driving_backward"][e], eo["driving_backward"], es["driving_on_grass"][e], eo["driving_on_grass"])
    for e, o in enumerate(orcs):
        oo, amb = o.render_with_mask()
        d = (oo != ob[e]).any(-1) & (amb == 0)
        if d.any():
            for ag in range(N):
                ys, xs = np.nonzero(d[ag])
                if len(ys):
                    print(f"step {k} env {e} agent {ag}: {len(ys)} px; first {[(int(y),int(x), ob[e,ag,y,x].tolist(), oo[ag,y,x].tolist()) for y,x in list(zip(ys,xs))[:6]]}")
            print(" positions", o.positions().tolist())
            sys.exit(0)
print("no mismatch")
