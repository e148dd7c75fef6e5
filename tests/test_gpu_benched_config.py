"""MI355X: the BENCHED configurations against the oracle (VERDICT r01 "configs untested").

* B=4096, N=2, contacts on, streams=2, device-side auto-reset, async host refill — exactly what bench.py times —
  with a random sample of envs followed by per-env oracles across TimeLimit resets and masked resets;
* N=8 at B=4096 (BASELINE configs[3]): size-independent properties, batch independence and sampled oracles incl. pixels;
* the freeze/thaw safety net of the auto-reset (an env whose next episode was not staged in time).
All comparisons go through the C-ABI (VecMultiCarRacing -> mcr_step)."""
import os

import numpy as np
import pytest

from tests.util import oracle_episode  # noqa: F401  (kept for symmetry with the other parity files)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


class _Follower:
    """Per-env oracle that follows global env g of a VecMultiCarRacing through its episodes: same RNG streams
    (vec_env.py docstring), TimeLimit counted here (the oracle is the bare env), next episode = next draw."""

    def __init__(self, O, N, seed, g, max_steps, use_random_direction=True, contacts=True):
        s = (seed + g) % 2 ** 32
        self.O, self.N, self.g, self.max_steps, self.urd = O, N, g, max_steps, use_random_direction
        self.tr, self.gr = np.random.RandomState(s), np.random.RandomState((s + 2 ** 31) % 2 ** 32)
        self.o = O.OracleEnv(N, car_contacts=contacts)
        self.first_obs = None
        self.new_episode()

    def new_episode(self):
        ep = self.O.new_episode(self.N, self.tr, self.gr, use_random_direction=self.urd)
        self.first_obs = self.o.reset(ep)
        self.first_amb = self.o.last_amb
        self.steps = 0

    def after_step(self, done):
        """bookkeeping after the batched oracle step: returns (done incl. TimeLimit, truncated)"""
        self.steps += 1
        trunc = False
        if self.max_steps > 0 and self.steps >= self.max_steps:
            trunc = not done
            done = True
        return done, trunc


def _cmp_pixels(got, want, amb, what, budget=14):
    d = (got != want).any(-1)
    bad = int((d & (amb == 0)).sum())
    if bad:
        where = np.argwhere(d & (amb == 0))[:6]
        detail = "; ".join(f"view {a} row {r} col {c}: got {got[a, r, c].tolist()} want {want[a, r, c].tolist()}" for a, r, c in where)
        raise AssertionError(f"{what}: {bad} unambiguous pixels differ: {detail}")
    assert int(d.sum()) <= budget * got.shape[0], f"{what}: {int(d.sum())} edge pixels differ"


def _cmp_state(env, followers, idx, what):
    st = env.get_state(); es = env.get_env_state()
    for f, e in zip(followers, idx):
        so = f.o.state(); eo = f.o.env_state()
        for k in ("bodies", "joints", "wheels", "limit", "on_road", "sleep"):
            assert np.array_equal(st[k][e], so[k]), f"{what} env {e}: {k} differs"
        assert np.array_equal(es["reward"][e], eo["reward"]) and np.array_equal(es["tile_visited_count"][e], eo["tile_visited_count"]), f"{what} env {e}: reward/tvc"
        T = f.o.T
        assert np.array_equal(es["tile_flags"][e, :T] & 0xff, eo["visited"]) and np.array_equal((es["tile_flags"][e, :T] >> 8) & 1, eo["touched"]), f"{what} env {e}: tile flags"
        assert es["num_tiles"][e] == T


def _actions(torch, g, B, N, car1_floors):
    a = torch.rand((B, N, 3), device="cuda", generator=g); a[..., 0] = a[..., 0] * 2 - 1
    if car1_floors:
        a[:, 1:, 1] = 1.0; a[:, 0, 2] *= 0.3
    return a


def _contact_envs(torch, B, N, seed, steps, max_steps, car1_floors):
    """GPU-only pre-pass of the SAME deterministic rollout: which envs hold a touching car<->car pair at some step?
    (the sampled comparison then follows those envs from their reset, so the contact path is certain to be checked)"""
    from multi_car_racing_amd.vec_env import VecMultiCarRacing
    from multi_car_racing_amd import _lib
    env = VecMultiCarRacing(B, N, seed=seed, use_random_direction=True, auto_reset=True, max_episode_steps=max_steps,
                            car_contacts=True, async_refill=True, streams=2)
    env.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    cnt = np.zeros(B, np.int32); seen = np.zeros(B, bool)
    for k in range(steps):
        env.step(_actions(torch, g, B, N, car1_floors))
        if k % 3 == 2:
            _lib.check(env.L.mcr_debug_read_contact_counts(env.h, _lib.ptr(cnt))); seen |= cnt > 0
    assert env.verdict_mismatches() == 0, "the touch verdict of the main launches disagreed with the contact pass"
    env.close()
    return np.nonzero(seen)[0]


def _run_sampled(torch, O, B, N, seed, steps, n_sample, max_steps, masked_reset_at=(), state_every=50, car1_floors=False,
                 prefer=None, with_obs=True):
    from multi_car_racing_amd.vec_env import VecMultiCarRacing
    env = VecMultiCarRacing(B, N, seed=seed, use_random_direction=True, auto_reset=True, max_episode_steps=max_steps,
                            car_contacts=True, async_refill=True, streams=2, obs=with_obs)
    obs = env.reset()
    rs = np.random.RandomState(seed + 99)
    if prefer is not None and len(prefer):                 # half the sample from the preferred envs, the rest at random
        pick = rs.choice(prefer, min(len(prefer), n_sample // 2), replace=False)
        rest = np.setdiff1d(np.arange(B), pick)
        idx = np.sort(np.concatenate([pick, rs.choice(rest, n_sample - len(pick), replace=False)]))
    else:
        idx = np.sort(rs.choice(B, n_sample, replace=False))
    idx_t = torch.from_numpy(idx).cuda()
    fol = [_Follower(O, N, seed, int(g), max_steps) for g in idx]
    if with_obs:
        o0 = obs[idx_t].cpu().numpy()
        for j, f in enumerate(fol):
            _cmp_pixels(o0[j], f.first_obs, f.first_amb, f"reset env {f.g}")
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    threads = os.cpu_count() or 1
    n_resets = n_contacts = 0
    for k in range(steps):
        a = _actions(torch, g, B, N, car1_floors)
        obs, rew, done, info = env.step(a)
        check_px = (k % state_every == state_every - 1)
        check_st = check_px
        check_px = check_px and with_obs
        a_s = a[idx_t].cpu().numpy()
        rw = rew[idx_t].cpu().numpy(); dn = done[idx_t].cpu().numpy().astype(bool); tr = info["TimeLimit.truncated"][idx_t].cpu().numpy().astype(bool)
        o_obs, o_amb, o_rew, o_done = O.step_batch([f.o for f in fol], a_s, np.full(n_sample, int(check_px), np.uint8), threads=threads)
        resets = []
        for j, f in enumerate(fol):
            d, t = f.after_step(bool(o_done[j]))
            assert np.array_equal(o_rew[j], rw[j]), f"step {k} env {f.g}: reward {rw[j]} vs oracle {o_rew[j]}"
            assert d == dn[j] and t == tr[j], f"step {k} env {f.g}: done/trunc {dn[j]}/{tr[j]} vs oracle {d}/{t}"
            n_contacts += f.o.num_car_contacts() > 0
            if d:
                resets.append(j)
        if check_px or resets:
            got = obs[idx_t].cpu().numpy() if with_obs else None
            for j, f in enumerate(fol):
                if j in resets:
                    f.new_episode(); n_resets += 1
                    if with_obs:
                        _cmp_pixels(got[j], f.first_obs, f.first_amb, f"step {k} env {f.g} first frame after auto-reset")
                elif check_px:
                    _cmp_pixels(got[j], o_obs[j], o_amb[j], f"step {k} env {f.g}")
        if check_st:
            _cmp_state(env, fol, idx, f"step {k}")
        if k in masked_reset_at:                       # masked reset of a random quarter of the batch incl. some followers
            m = (rs.uniform(size=B) < 0.25); m[idx[::3]] = True
            obs = env.reset_envs(torch.from_numpy(m.astype(np.uint8)).cuda())
            got = obs[idx_t].cpu().numpy() if with_obs else None
            for j, f in enumerate(fol):
                if m[f.g]:
                    f.new_episode(); n_resets += 1
                    if with_obs:
                        _cmp_pixels(got[j], f.first_obs, f.first_amb, f"masked reset at step {k} env {f.g}")
            _cmp_state(env, fol, idx, f"after masked reset at step {k}")
    frozen = int(env.debug_counters()[3])
    assert env.verdict_mismatches() == 0, "the touch verdict of the main launches disagreed with the contact pass"
    env.close()
    return n_resets, n_contacts, frozen


def test_benched_config_b4096_sampled_oracles(torch_cuda, oracle):
    """bench.py's default workload (BASELINE configs[1]) for 1,100 steps: crosses TimeLimit(1000) -> 4096 device-side
    auto-resets in one step + host refill, plus two masked resets; 48 sampled envs == their oracles throughout."""
    n_resets, n_contacts, frozen = _run_sampled(torch_cuda, oracle, B=4096, N=2, seed=20, steps=1100, n_sample=48,
                                                max_steps=1000, masked_reset_at=(333, 720))
    assert n_resets >= 48, n_resets
    assert frozen == 0, f"{frozen} env-steps frozen waiting for the host"


def test_benched_config_b4096_contacts_sample(torch_cuda, oracle):
    """Same batch with car 1 flooring it (rear-end collisions): the sampled envs must exercise car<->car contacts and
    the contact side stream at full batch size."""
    hot = _contact_envs(torch_cuda, 4096, 2, 21, 260, 120, True)
    assert len(hot) > 0, "rollout produced no car<->car contact in 4096 envs"
    n_resets, n_contacts, frozen = _run_sampled(torch_cuda, oracle, B=4096, N=2, seed=21, steps=260, n_sample=96,
                                                max_steps=120, state_every=40, car1_floors=True, prefer=hot)
    assert n_resets >= 96 and n_contacts > 0, (n_resets, n_contacts)
    assert frozen == 0


def test_physics_only_b4096_sampled_oracles(torch_cuda, oracle):
    """BASELINE configs[4] at full size: num_agents=2, batch=4096, obs=none (the physics-only step bench.py --obs 0 times) — 64 sampled envs
    == their oracles (rewards, done, truncation every step; full rigid-body and bookkeeping state every 50) across a TimeLimit round with its
    4096 device-side auto-resets and a masked reset; half of the sample are envs known to hold touching car<->car pairs."""
    prefer = _contact_envs(torch_cuda, 4096, 2, seed=23, steps=240, max_steps=200, car1_floors=True)
    n_resets, n_contacts, frozen = _run_sampled(torch_cuda, oracle, B=4096, N=2, seed=23, steps=260, n_sample=64, max_steps=200,
                                                masked_reset_at=(120,), car1_floors=True, prefer=prefer, with_obs=False)
    assert n_resets >= 64 and frozen == 0
    assert n_contacts > 0, "no sampled env ever held a car<->car contact"


@pytest.mark.parametrize("knobs, ordering", [({}, 1), ({"MCR_UNFUSED_COLLIDE": "1", "MCR_POST_DYN": "0"}, 1), ({"MCR_SOFT_SYNC": "0"}, 2), ({"MCR_SOFT_SYNC": "0", "MCR_STOP_EVENTS": "0"}, 0)])
def test_every_stream_ordering_of_the_step_matches_the_oracle(torch_cuda, oracle, monkeypatch, knobs, ordering):
    """The three-chain step orders its streams through phase words in device memory (default where kernels overlap) or through
    events (profilers that serialise kernels, a wait that gave up, graph capture) — completed by the launches they mark or recorded
    behind them; with phase words the contact chain runs its envs' contact pass itself (default) or waits for the all-env pass
    (MCR_UNFUSED_COLLIDE=1): every variant is the same computation — rear-end collisions, TimeLimit resets and refills included."""
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    import gc
    from multi_car_racing_amd.vec_env import VecMultiCarRacing
    gc.collect()                                             # (one phase-word handle per device at a time: no stale ones from earlier tests)
    probe = VecMultiCarRacing(64, 2, seed=1, auto_reset=True, car_contacts=True, streams=2)
    mode, overlap = probe.L.mcr_step_ordering(probe.h), probe.L.mcr_concurrent_collide(probe.h)
    probe.close()
    if ordering == 1 and not overlap:
        pytest.skip("kernels of different streams do not overlap here: the phase-word path is off")
    assert (mode & 1) == (ordering & 1) and (ordering == 1 or (mode & 2) == (ordering & 2)), (mode, ordering)
    hot = _contact_envs(torch_cuda, 1024, 2, 33, 200, 90, True)
    n_resets, n_contacts, frozen = _run_sampled(torch_cuda, oracle, B=1024, N=2, seed=33, steps=200, n_sample=48,
                                                max_steps=90, state_every=30, car1_floors=True, prefer=hot, masked_reset_at=(77,))
    assert n_resets >= 96 and n_contacts > 0, (n_resets, n_contacts)
    assert frozen == 0


def test_n8_b4096_properties_and_sampled_oracles(torch_cuda, oracle):
    """BASELINE configs[3]: num_agents=8, batch=4096 — dense car<->car rasterization.  Sampled oracles (state, reward,
    pixels of all 8 views) + size-independent properties + batch independence (env g in B=4096 == env g in B=4)."""
    torch = torch_cuda
    from multi_car_racing_amd.vec_env import VecMultiCarRacing
    hot = _contact_envs(torch, 4096, 8, 23, 130, 100, True)
    n_resets, n_contacts, frozen = _run_sampled(torch, oracle, B=4096, N=8, seed=23, steps=130, n_sample=24, max_steps=100,
                                                state_every=26, car1_floors=True, prefer=hot)
    assert n_resets >= 24 and n_contacts > 0 and frozen == 0, (n_resets, n_contacts, frozen)
    big = VecMultiCarRacing(4096, 8, seed=5, use_random_direction=True, auto_reset=True, async_refill=True, streams=2)
    small = VecMultiCarRacing(4, 8, seed=5, use_random_direction=True, auto_reset=True, async_refill=False, streams=1)
    ob = big.reset(); osm = small.reset()
    assert torch.equal(ob[:4], osm)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    total = torch.zeros((4096, 8), dtype=torch.float64, device="cuda")
    for k in range(70):
        a = torch.rand((4096, 8, 3), device="cuda", generator=g); a[..., 0] = a[..., 0] * 2 - 1; a[..., 2] *= 0.2
        ob, rew, done, _ = big.step(a); total += rew
        osm, rs, ds, _ = small.step(a[:4].contiguous())
        assert torch.equal(rew[:4], rs) and torch.equal(done[:4], ds), k
    assert torch.equal(ob[:4], osm)
    es = big.get_env_state()
    assert (es["tile_visited_count"] <= es["num_tiles"][:, None]).all()
    assert np.allclose(total.cpu().numpy(), es["reward"], atol=1e-9)           # sum of step rewards == self.reward
    o = ob.cpu().numpy()
    assert (o[:, :, 84:, 0, :] == 0).all()                                      # HUD bar: left column black in all 32,768 views
    big.close(); small.close()


@pytest.mark.parametrize("streams", [1, 2])
def test_freeze_and_thaw_when_host_withholds_staging(torch_cuda, oracle, streams):
    """an env that finishes while no episode is staged freezes (zero reward, done 0) and must THAW
    as soon as the host stages one: its next step returns the first observation of the next episode of its own RNG
    streams and it keeps stepping bit-exact afterwards."""
    torch = torch_cuda
    from multi_car_racing_amd.vec_env import VecMultiCarRacing
    B, N, seed, L = 6, 2, 90, 12
    env = VecMultiCarRacing(B, N, seed=seed, use_random_direction=True, auto_reset=True, max_episode_steps=L,
                            car_contacts=True, async_refill=False, streams=streams)
    env.reset()                                          # consumes episode 1 and stages episode 2
    env.hold_refills = True                              # from now on nothing new gets staged
    fol = [_Follower(oracle, N, seed, g, L) for g in range(B)]
    rs = np.random.RandomState(4)

    def step_both(expect_done=None):
        a = np.stack([rs.uniform(-1, 1, (B, N)), rs.uniform(0, 1, (B, N)), rs.uniform(0, 0.2, (B, N))], -1).astype(np.float32)
        obs, rew, done, _ = env.step(torch.from_numpy(a).cuda())
        return a, obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy().astype(bool)

    for k in range(L):                                   # episode 1 -> auto-reset consumes the staged episode 2
        a, obs, rw, dn = step_both()
        _, _, orw, od = oracle.step_batch([f.o for f in fol], a, None, threads=2)
        for j, f in enumerate(fol):
            d, _ = f.after_step(bool(od[j])); assert np.array_equal(orw[j], rw[j]) and d == dn[j]
            if d:
                f.new_episode()
    assert dn.all()
    for k in range(L):                                   # episode 2; when it ends there is no staged episode 3
        a, obs, rw, dn = step_both()
        _, _, orw, od = oracle.step_batch([f.o for f in fol], a, None, threads=2)
        for j, f in enumerate(fol):
            d, _ = f.after_step(bool(od[j])); assert np.array_equal(orw[j], rw[j]) and d == dn[j], (k, j)
    assert dn.all(), "episode 2 should end by TimeLimit"
    c0 = int(env.debug_counters()[3])
    stale = obs.copy()
    for k in range(3):                                   # frozen: zero reward, done 0, obs rows untouched
        a, obs, rw, dn = step_both()
        assert (rw == 0).all() and not dn.any() and np.array_equal(obs, stale)
    assert int(env.debug_counters()[3]) - c0 == 3 * B, "frozen env-steps are counted every step"
    env.hold_refills = False
    env._settle_staging(torch.cuda.current_stream())      # host catches up: next episodes staged
    a, obs, rw, dn = step_both()                         # thaw step: first observation of episode 3, no reward, not done
    assert (rw == 0).all() and not dn.any()
    for j, f in enumerate(fol):
        f.new_episode()
        _cmp_pixels(obs[j], f.first_obs, f.first_amb, f"thaw env {j}")
    _cmp_state(env, fol, range(B), "after thaw")
    for k in range(L - 1):                               # and the episode continues bit-exact
        a, obs, rw, dn = step_both()
        _, _, orw, od = oracle.step_batch([f.o for f in fol], a, None, threads=2)
        for j, f in enumerate(fol):
            d, _ = f.after_step(bool(od[j])); assert np.array_equal(orw[j], rw[j]) and d == dn[j], (k, j)
    assert env.verdict_mismatches() == 0
    env.close()


def test_freeze_with_touching_cars_contact_pass_in_front(torch_cuda, oracle, monkeypatch):
    """with the contact pass in FRONT of the dynamics (MCR_SEQUENTIAL_COLLIDE=1; also N > 4, serialised kernels) it
    is k_collide that marks the contact chain's envs.  An env that freezes (episode over, nothing staged) while its cars
    are in car<->car contact must not keep that mark — it would be skipped by every main launch and never thaw."""
    torch = torch_cuda
    from multi_car_racing_amd.vec_env import VecMultiCarRacing
    monkeypatch.setenv("MCR_SEQUENTIAL_COLLIDE", "1")
    B, N, seed, L = 6, 2, 91, 14
    env = VecMultiCarRacing(B, N, seed=seed, use_random_direction=True, auto_reset=True, max_episode_steps=L,
                            car_contacts=True, async_refill=False, streams=2)
    assert not env.L.mcr_concurrent_collide(env.h)
    env.reset()
    env.hold_refills = True
    fol = [_Follower(oracle, N, seed, g, L) for g in range(B)]
    rs = np.random.RandomState(5)

    def step_both(touching_drive):
        a = np.stack([rs.uniform(-1, 1, (B, N)), rs.uniform(0, 1, (B, N)), rs.uniform(0, 0.2, (B, N))], -1).astype(np.float32)
        if touching_drive:
            a[:, 0, 1] = 0.0; a[:, 0, 2] = 0.8; a[:, 1, 0] *= 0.1; a[:, 1, 1] = 1.0       # car 0 brakes, car 1 pushes into it
        obs, rew, done, _ = env.step(torch.from_numpy(a).cuda())
        return a, obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy().astype(bool)

    for k in range(L):                                   # episode 1
        a, obs, rw, dn = step_both(False)
        _, _, orw, od = oracle.step_batch([f.o for f in fol], a, None, threads=2)
        for j, f in enumerate(fol):
            d, _ = f.after_step(bool(od[j])); assert np.array_equal(orw[j], rw[j]) and d == dn[j]
            if d:
                f.new_episode()
    # episode 2: car 1 is put right behind car 0 (overlapping it a little), so the pair touches until the TimeLimit ends the episode
    st = env.get_state()["bodies"].copy()
    for e in range(B):
        ang = st[e, 0, 0, 2]
        fwd = np.array([-np.sin(ang), np.cos(ang)], np.float32)
        delta = (st[e, 0, 0, :2] - fwd * np.float32(4.9)) - st[e, 1, 0, :2]
        st[e, 1, :, 0] += delta[0]; st[e, 1, :, 1] += delta[1]; st[e, 1, :, 2] = ang
    env.set_bodies(st)
    for e, f in enumerate(fol):
        for k in range(5):
            f.o.set_body(1, k, st[e, 1, k])
    touching_at_end = 0
    for k in range(L):
        a, obs, rw, dn = step_both(True)
        _, _, orw, od = oracle.step_batch([f.o for f in fol], a, None, threads=2)
        for j, f in enumerate(fol):
            d, _ = f.after_step(bool(od[j])); assert np.array_equal(orw[j], rw[j]) and d == dn[j], (k, j)
    touching_at_end = sum(f.o.num_car_contacts() > 0 for f in fol)
    assert dn.all() and touching_at_end >= B // 2, f"only {touching_at_end} envs ended their episode in car<->car contact"
    for k in range(3):                                   # frozen
        a, obs, rw, dn = step_both(False)
        assert (rw == 0).all() and not dn.any()
    env.hold_refills = False
    env._settle_staging(torch.cuda.current_stream())
    a, obs, rw, dn = step_both(False)                    # thaw step: every env, also those that froze in contact
    assert (rw == 0).all() and not dn.any()
    for j, f in enumerate(fol):
        f.new_episode()
        _cmp_pixels(obs[j], f.first_obs, f.first_amb, f"thaw env {j}")
    _cmp_state(env, fol, range(B), "after thaw")
    for k in range(L - 1):
        a, obs, rw, dn = step_both(False)
        _, _, orw, od = oracle.step_batch([f.o for f in fol], a, None, threads=2)
        for j, f in enumerate(fol):
            d, _ = f.after_step(bool(od[j])); assert np.array_equal(orw[j], rw[j]) and d == dn[j], (k, j)
    env.close()
