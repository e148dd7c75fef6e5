"""Soak test of the step's stream orderings: a three-chain handle (phase words where kernels overlap) and a single-stream handle
step the same actions for many steps with TimeLimit resets; rewards / dones every step, observations and full state every 64th
step must be bit-identical.  usage: python tools/soak_orderings.py [N] [steps] [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multi_car_racing_amd.vec_env import VecMultiCarRacing
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
# ASYNC=1: both handles with the native refill service (the default of VecMultiCarRacing) instead of staging inside step()
kw = dict(seed=77, use_random_direction=True, auto_reset=True, max_episode_steps=250, car_contacts=True, async_refill=bool(int(os.environ.get("ASYNC", "0"))))
a = VecMultiCarRacing(B, N, streams=2, **kw)
b = VecMultiCarRacing(B, N, streams=1, **kw)
print("ordering of the three-chain handle:", a.L.mcr_step_ordering(a.h), "contact pass beside the dynamics:", a.L.mcr_concurrent_collide(a.h))
oa, ob = a.reset(), b.reset()
assert torch.equal(oa, ob)
g = torch.Generator(device="cuda"); g.manual_seed(9)
pool = torch.rand((128, B, N, 3), device="cuda", generator=g); pool[..., 0] = pool[..., 0] * 2 - 1
if N > 1: pool[:, :, 1, 1] = torch.clamp(pool[:, :, 1, 1] + 0.4, max=1.0)      # car 1 is faster: rear-ends happen
if int(os.environ.get("DRIVE", "0")): pool[..., 0] *= 0.1; pool[..., 1] = 1.0; pool[..., 2] = 0.0      # bench.py --actions drive: ~1000 contact envs per step
bad = torch.zeros((), dtype=torch.int64, device="cuda")
t0 = time.perf_counter(); resets = 0
from multi_car_racing_amd import _lib
cnt = np.zeros(B, np.int32); peak = 0; seen = []          # contact envs per step, sampled (how full the contact chain's launch was)
for k in range(steps):
    act = pool[(k * 7) % 128]
    o1, r1, d1, _ = a.step(act); o2, r2, d2, _ = b.step(act)
    bad += (~torch.equal(r1, r2)) + (~torch.equal(d1, d2)) if False else ((r1 != r2).any() | (d1 != d2).any()).to(torch.int64)
    if k % 64 == 63:
        bad += (o1 != o2).any().to(torch.int64)
        resets += int(d1.sum())
        _lib.check(a.L.mcr_debug_read_contact_counts(a.h, _lib.ptr(cnt))); n_c = int((cnt > 0).sum()); peak = max(peak, n_c); seen.append(n_c)
        if int(bad) != 0:
            print("MISMATCH by step", k); break
sa, sb = a.get_state(), b.get_state()
state_ok = all(np.array_equal(sa[key], sb[key]) for key in sa)
print(f"N={N} B={B}{' DRIVE' if int(os.environ.get('DRIVE', '0')) else ''}: {k + 1} steps in {time.perf_counter() - t0:.1f} s, mismatching comparisons: {int(bad)}, final state identical: {state_ok}, verdict mismatches: {a.verdict_mismatches()}, counters {a.debug_counters().tolist()}, contact envs per sampled step: mean {np.mean(seen) if seen else 0:.0f} peak {peak}")
a.close(); b.close()
sys.exit(0 if int(bad) == 0 and state_ok else 1)
