"""Box2D solves an island's joints and contacts in the order its depth-first search met them (b2World::Solve); the build DEFINES an order
(contacts ascending by (carA, fixA, carB, fixB), joints 3,2,1,0 per car — what the DFS gives for a car entered through its wheel 3 or its
hull).  How often do the two differ, and what does it change?  CPU oracle only: island order 1 (DFS, oracle/mcr_oracle_contacts.inc:
island_dfs) against island order 0 (what the kernels implement) on the same rollouts with rear-end collisions.  Not a test.
   python tools/island_order_effect.py [episodes] [steps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from tests.util import oracle_episode

episodes = int(sys.argv[1]) if len(sys.argv) > 1 else 200
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
for N in (2, 4, 8):
    contact_steps = joint_diff = contact_diff = any_diff = diverged = 0
    first_div = []
    for e in range(episodes):
        ep = oracle_episode(O, N, 4000 + N, e, use_random_direction=True)
        a, b = O.OracleEnv(N), O.OracleEnv(N); a.set_island_order(0); b.set_island_order(1)
        a.reset(ep, render=False); b.reset(ep, render=False)
        rng = np.random.RandomState(e)
        div = None
        for k in range(steps):
            act = np.stack([rng.uniform(-0.3, 0.3, N), rng.uniform(0.2, 1.0, N), np.zeros(N)], -1).astype(np.float32)
            act[N // 2:, 1] = 1.0                                    # the cars of the back rows floor it: rear-end collisions
            a.step(act, render=False); b.step(act, render=False)
            if b.num_car_contacts() > 0:
                contact_steps += 1
                d = b.island_diff()
                joint_diff += d & 1; contact_diff += (d >> 1) & 1; any_diff += int(d != 0)
            if div is None and not np.array_equal(a.state()["bodies"], b.state()["bodies"]):
                div = k
        if div is not None:
            diverged += 1; first_div.append(div)
        a.close(); b.close()
    print(f"N={N}: {episodes} episodes x {steps} steps: {contact_steps} env-steps with touching car<->car contacts; the DFS order differs from the "
          f"defined one in {any_diff} of them (a car's joint order in {joint_diff}, the contact order in {contact_diff}); "
          f"{diverged} episodes end up with different body states (first difference at step {int(np.median(first_div)) if first_div else '-'} median)")
