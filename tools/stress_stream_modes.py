"""Long differential run of the two launch topologies (single stream vs contact side stream + deferral): rewards and
dones every step, observations and full state periodically.  Any difference is a bug (race, list handling, resume).
usage: stress_stream_modes.py [B] [N] [steps] [time_limit]"""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_car_racing_amd.vec_env import VecMultiCarRacing
B, N, steps, L = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 4096), (2, 2), (3, 1500), (4, 200)))
kw = dict(seed=3, auto_reset=True, max_episode_steps=L, use_random_direction=True, car_contacts=True)
a1 = VecMultiCarRacing(B, N, streams=1, **kw); a2 = VecMultiCarRacing(B, N, streams=2, **kw)
o1, o2 = a1.reset(), a2.reset()
assert torch.equal(o1, o2)
g = torch.Generator(device="cuda"); g.manual_seed(9)
for k in range(steps):
    a = torch.rand((B, N, 3), device="cuda", generator=g); a[..., 0] = a[..., 0] * 2 - 1; a[..., 2] *= 0.4
    a[:, N - 1, 1] = 1.0
    o1, r1, d1, i1 = a1.step(a); o2, r2, d2, i2 = a2.step(a)
    if not (torch.equal(r1, r2) and torch.equal(d1, d2)):
        bad = torch.nonzero((r1 != r2).any(1) | (d1 != d2)).flatten().tolist()
        raise SystemExit(f"step {k}: envs {bad[:8]} differ")
    if k % 50 == 49:
        assert torch.equal(o1, o2), f"obs differ at step {k}"
        assert torch.equal(i1["episode_return"], i2["episode_return"]) and torch.equal(i1["episode_length"], i2["episode_length"])
s1, s2 = a1.get_state(), a2.get_state()
for key in s1:
    assert np.array_equal(s1[key], s2[key]), key
c = a2.debug_counters()
print(f"OK: B={B} N={N} {steps} steps identical; deferred {c[0]} resumed {c[1]} contact-env routings {c[2]}; episodes {a2.rollout_stats()[0]:.0f}")
a1.close(); a2.close()
