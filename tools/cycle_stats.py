"""Do the velocity sweeps of an env with touching car<->car contacts become PERIODIC (state after sweep k == state after sweep k - p, bit for bit)?
From there on the remaining sweeps are determined, and a wavefront that holds one env could stop.  CPU oracle only (orc_debug_cycle).  Not a test.
   python tools/cycle_stats.py [episodes] [steps] [N]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from tests.util import oracle_episode

episodes = int(sys.argv[1]) if len(sys.argv) > 1 else 60
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
N = int(sys.argv[3]) if len(sys.argv) > 3 else 2
for contacts_only in (1, 0):
    hist = np.zeros((181, 9), np.int64)
    L = O.lib()
    L.orc_debug_cycle.restype = None
    L.orc_debug_cycle.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    for e in range(episodes):
        ep = oracle_episode(O, N, 4000 + N, e, use_random_direction=True)
        a = O.OracleEnv(N); a.reset(ep, render=False)
        L.orc_debug_cycle(a.h, hist.ctypes.data_as(ctypes.c_void_p), contacts_only)
        rng = np.random.RandomState(e)
        for k in range(steps):
            act = np.stack([rng.uniform(-0.1, 0.1, N), np.ones(N), np.zeros(N)], -1).astype(np.float32)      # bench.py --actions drive
            a.step(act, render=False)
        L.orc_debug_cycle(a.h, None, 0)
        a.close()
    tot = hist.sum()
    print(f"N={N} {'env-steps with touching contacts' if contacts_only else 'all env-steps'}: {tot}")
    if tot == 0: continue
    print("   period shares:", {p: round(float(hist[:, p].sum()) / tot, 3) for p in range(9)}, "(0: not periodic within 180 sweeps)")
    cum = np.cumsum(hist[:, 1:].sum(1)) / tot
    print("   periodic by sweep:", {k: round(float(cum[k]), 3) for k in (5, 10, 20, 40, 60, 80, 100, 120, 150, 179)})
