"""The reference calls reset() on ONE b2World (multi_car_racing.py:138, 341); the build treats every episode as the first episode of a
fresh world.  What does that change?  CPU only: the oracle with a literal b2DynamicTree carried through the resets of an env
(orc_set_world_mode 1: proxy ids come off the tree's free list) against the oracle in the mode the kernels implement (mode 0), on the SAME
second episode: first episode = `first_steps` steps of driving on track A, then reset() onto track B and `steps` steps with identical
actions.  Not a test.   python tools/world_reuse_effect.py [episodes] [N]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from tests.util import oracle_episode

episodes = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2
first_steps, steps = 150, 120
rng = np.random.RandomState(7)
n_spawn = n_ret = n_visit = n_scr_car = n_scr_mixed = n_t_shorter = 0
for e in range(episodes):
    epA, epB = oracle_episode(O, N, 20000, e, use_random_direction=True), oracle_episode(O, N, 60000, e, use_random_direction=True)
    envs = []
    for mode in (0, 1):
        o = O.OracleEnv(N); o.set_world_mode(mode); envs.append(o)
        o.reset(epA, render=False)
    r = np.random.RandomState(1000 + e)
    for k in range(first_steps):
        a = np.stack([r.uniform(-0.4, 0.4, N), np.ones(N), np.zeros(N)], -1).astype(np.float32)
        for o in envs: o.step(a, render=False)
    # second episode
    sp = []
    for o in envs:
        o.reset(epB, render=False)
        sp.append(o.env_state()["reward"].copy())
    tid, fid = envs[1].proxy_ids()
    flat = fid.ravel()
    if not np.all(np.diff(flat) > 0): n_scr_car += 1
    if tid.max() > flat.min(): n_scr_mixed += 1
    if len(epB["track"]) < len(epA["track"]): n_t_shorter += 1
    ret = [np.zeros(N), np.zeros(N)]
    vis = [None, None]
    for k in range(steps):
        a = np.stack([r.uniform(-0.4, 0.4, N), np.ones(N), np.zeros(N)], -1).astype(np.float32)
        for i, o in enumerate(envs):
            _, rw, d, _ = o.step(a, render=False); ret[i] += rw
    st = [o.env_state() for o in envs]
    n_spawn += int(np.sum(np.asarray(sp[0]) != np.asarray(sp[1])))
    n_ret += int(np.sum(ret[0] != ret[1]))
    n_visit += int(not np.array_equal(st[0]["tile_visited_count"], st[1]["tile_visited_count"]))
    for o in envs: o.close()
print(f"second episodes: {episodes} at N={N}  (track B shorter than track A: {n_t_shorter})")
print(f"  car proxy ids not ascending in creation order: {n_scr_car}   some tile id above some car id: {n_scr_mixed}")
print(f"  (episode, car) pairs whose spawn-step reward differs from the fresh-world oracle: {n_spawn} of {episodes * N}")
print(f"  (episode, car) pairs whose {steps}-step return differs: {n_ret} of {episodes * N}")
print(f"  episodes whose tile_visited_count differs after {steps} steps: {n_visit}")
