"""The reference calls reset() on ONE b2World (multi_car_racing.py:138, 173-181, 341); VecMultiCarRacing's default treats every episode as the
first episode of a fresh world.  What does that change?  CPU only: the oracle with a literal b2DynamicTree carried through the resets of an
env (orc_set_world_mode 1: proxy ids come off the tree's free list, mcr.py:113-120 then sees Box2D's callback order and a car<->car contact
its fixtureA) against the oracle in fresh-world mode (mode 0), same tracks, same actions, FULL episodes (TimeLimit `steps`), episodes 1..5
back to back as an auto-resetting VecEnv runs them.  Not a test.
   python tools/world_reuse_effect.py [envs] [N] [policy: drive|random] [steps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from tests.util import oracle_episode

envs_n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2
policy = sys.argv[3] if len(sys.argv) > 3 else "drive"
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
EPISODES = 5
thr = os.cpu_count() or 1


def actions(r, k):
    if policy == "drive":                   # bench.py --actions drive: gas 1, brake 0, steering noise +-0.1
        return np.stack([r.uniform(-0.1, 0.1, (envs_n, N)), np.ones((envs_n, N)), np.zeros((envs_n, N))], -1).astype(np.float32)
    return np.stack([r.uniform(-1, 1, (envs_n, N)), r.uniform(0, 1, (envs_n, N)), r.uniform(0, 1, (envs_n, N))], -1).astype(np.float32)


worlds = []
for mode in (0, 1):
    es = []
    for e in range(envs_n):
        o = O.OracleEnv(N); o.set_world_mode(mode); es.append(o)
    worlds.append(es)
print(f"# tools/world_reuse_effect.py: {envs_n} envs x {EPISODES} episodes of <= {steps} steps, N={N}, policy {policy}; oracle world mode 1 (one b2World, the reference) vs mode 0 (fresh world per episode)")
print("# episode | (env, car) step rewards that ever differ | (env,car) returns differing | envs: tile_visited_count differs at the end | done step differs | final poses differ | envs with a car<->car contact (mode 1)")
for ep in range(EPISODES):
    for es in worlds:
        for e, o in enumerate(es):
            o.reset(oracle_episode(O, N, 20000 + 7919 * ep, e, use_random_direction=True), render=False)
    r = np.random.RandomState(100 + ep)
    ret = [np.zeros((envs_n, N)), np.zeros((envs_n, N))]
    rdiff = np.zeros((envs_n, N), bool)
    done_at = [np.full(envs_n, steps), np.full(envs_n, steps)]
    contact = np.zeros(envs_n, bool)
    alive = [np.ones(envs_n, bool), np.ones(envs_n, bool)]
    final = [None, None]
    for k in range(steps):
        a = actions(r, k)
        rw = []
        for i, es in enumerate(worlds):
            _, _, rew, done = O.step_batch(es, a, None, threads=thr)
            rew = np.where(alive[i][:, None], rew, 0.0)          # an env that is done stays where it is (the oracle has no auto-reset): ignore it
            ret[i] += rew
            newly = done & alive[i]
            done_at[i][newly] = k
            alive[i] &= ~done
            rw.append(rew)
        rdiff |= (rw[0] != rw[1]) & (alive[0] | alive[1])[:, None]
        if k % 10 == 0:
            contact |= np.array([o.num_car_contacts() > 0 for o in worlds[1]])
    st = [[o.env_state() for o in es] for es in worlds]
    bodies = [[o.state()["bodies"] for o in es] for es in worlds]
    tvc = sum(int(not np.array_equal(a_["tile_visited_count"], b_["tile_visited_count"])) for a_, b_ in zip(*st))
    poses = sum(int(not np.array_equal(a_, b_)) for a_, b_ in zip(*bodies))
    print(f"  {ep + 1} | {int(rdiff.sum())} of {envs_n * N} | {int((ret[0] != ret[1]).sum())} | {tvc} of {envs_n} | {int((done_at[0] != done_at[1]).sum())} | {poses} | {int(contact.sum())}", flush=True)
