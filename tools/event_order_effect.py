"""How many (episode, car) rewards change between the order rounds 1-2 DEFINED for same-step begin events (tile^, car^,
wheel^) and Box2D's own order (broadphase model, oracle/mcr_oracle.cpp: later FindNewContacts batch first, then descending
proxy ids)?  CPU oracle only.  Prints per N: resets, (episode, car) pairs whose spawn-step reward differs, and — over a short
drive with both cars steering alike — pairs whose return after `steps` steps differs."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O
from tests.util import oracle_episode

resets = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 120
for N in (2, 4):
    spawn_diff = ret_diff = pairs = drive_pairs = 0
    a = O.OracleEnv(N); b = O.OracleEnv(N); b.L.orc_set_event_order(b.h, 1)
    rng = np.random.RandomState(N)
    for e in range(resets):
        ep = oracle_episode(O, N, 1000 * N, e, use_random_direction=True)
        a.reset(ep, render=False); b.reset(ep, render=False)
        ra, rb = a.env_state()["reward"].copy(), b.env_state()["reward"].copy()
        spawn_diff += int((ra != rb).sum()); pairs += N
        if e < resets // 10:
            ta, tb = ra.copy(), rb.copy()
            for k in range(steps):
                act = np.zeros((N, 3), np.float32); act[:, 1] = 0.5; act[:, 0] = rng.uniform(-0.1, 0.1)
                _, r1, _, _ = a.step(act, render=False); _, r2, _, _ = b.step(act, render=False)
                ta += r1; tb += r2
            ret_diff += int((np.abs(ta - tb) > 1e-9).sum()); drive_pairs += N
    print(f"N={N}: {resets} resets: spawn-step reward differs for {spawn_diff} of {pairs} (episode, car) pairs; "
          f"return after {steps} steps differs for {ret_diff} of {drive_pairs}")
