"""First-contact GPU diagnostic: parity summary vs the oracle + rough timings (not a test)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import oracle as O
from multi_car_racing_amd.vec_env import VecMultiCarRacing
from tests.util import oracle_episode, random_actions

B, N, seed = 8, 2, 100
env = VecMultiCarRacing(B, N, seed=seed, use_random_direction=False, auto_reset=False, max_episode_steps=0, async_refill=False)
obs = env.reset().cpu().numpy()
orcs = []
for e in range(B):
    ep = oracle_episode(O, N, seed, e)
    o = O.OracleEnv(N); oo = o.reset(ep); orcs.append(o)
    d = (oo != obs[e]); _, amb = o.render_with_mask()
    print(f"env {e}: reset obs mismatched px {d.any(-1).sum()} (ambiguous {amb.sum()}, mismatch outside amb {(d.any(-1) & (amb == 0)).sum()})")
st = env.get_state()
for e in range(B):
    so = orcs[e].state()
    print("reset bodies equal", np.array_equal(st["bodies"][e], so["bodies"]), "wheels", np.array_equal(st["wheels"][e], so["wheels"]))
rng = np.random.RandomState(0)
first_bad = None
for k in range(300):
    a = random_actions(rng, B, N, 0.3)
    obs, rew, done, info = env.step(torch.from_numpy(a).cuda())
    rw = rew.cpu().numpy(); dn = done.cpu().numpy()
    for e in range(B):
        oo, r, d, _ = orcs[e].step(a[e], render=(k % 50 == 49))
        if not np.array_equal(r, rw[e]) or bool(dn[e]) != d:
            if first_bad is None: first_bad = (k, e, r, rw[e], d, dn[e])
    if k % 50 == 49:
        st = env.get_state(); o = obs.cpu().numpy()
        eqb = [np.array_equal(st["bodies"][e], orcs[e].state()["bodies"]) for e in range(B)]
        eqw = [np.array_equal(st["wheels"][e], orcs[e].state()["wheels"]) for e in range(B)]
        eqj = [np.array_equal(st["joints"][e], orcs[e].state()["joints"]) for e in range(B)]
        mm = []
        for e in range(B):
            oo, amb = orcs[e].render_with_mask()
            d = (oo != o[e]).any(-1)
            mm.append((int(d.sum()), int((d & (amb == 0)).sum())))
        print(f"step {k+1}: bodies {eqb} wheels {all(eqw)} joints {all(eqj)} px mismatch (all, outside-amb) {mm}")
        if not all(eqb):
            e = eqb.index(False)
            print(" max abs diff", np.abs(st["bodies"][e] - orcs[e].state()["bodies"]).max())
print("first reward/done mismatch:", first_bad)
es = env.get_env_state()
print("tvc gpu", es["tile_visited_count"].tolist(), "oracle", [o.env_state()["tile_visited_count"].tolist() for o in orcs])
# timing at B=4096
env.close()
for obs_on in (True, False):
    env = VecMultiCarRacing(4096, 2, seed=1, use_random_direction=True, auto_reset=True, obs=obs_on)
    t0 = time.time(); env.reset(); torch.cuda.synchronize(); print("reset 4096 envs: %.3f s" % (time.time() - t0))
    act = torch.rand((4096, 2, 3), device="cuda"); act[..., 0] = act[..., 0] * 2 - 1
    for _ in range(20): env.step(act)
    torch.cuda.synchronize()
    env.timing(7)
    t0 = time.time()
    for _ in range(200): env.step(act)
    torch.cuda.synchronize(); dt = time.time() - t0
    ms, n = env.timing_read()
    print(f"obs={obs_on}: {200 * 4096 / dt:.0f} env-steps/s, {dt / 200 * 1e3:.3f} ms/step; kernel ms/launch collide {ms[0] / max(n[0], 1):.4f} dynamics {ms[1] / max(n[1], 1):.4f} view {ms[2] / max(n[2], 1):.4f}  launches {n.tolist()}")
    env.close()
