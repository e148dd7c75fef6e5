"""MI355X: terminal observations of auto-reset episodes (include/mcr.h: mcr_set_terminal_obs; VERDICT r04 "missing #1").

The reference renders the state after the last solve of an episode and returns it with done = True (multi_car_racing.py:431, :509;
TimeLimit: gym_multi_car_racing/__init__.py:8).  VecMultiCarRacing re-spawns a finished env inside the same step — its row of `obs` is
the first frame of the next episode — and hands the last frame out as info["terminal_observation"].  Here: per-env oracles follow the
batch; at every step that ends an episode the oracle's frame of THAT step (o.step(..., render=True)) must equal the entry's frames outside
the oracle's ambiguity mask — under TimeLimit endings (all envs in one step), out-of-playfield endings (scattered steps), in the
three-chain and the single-stream step, at N = 2 and N = 8, and at B = 4096 on a sample."""
import os

import numpy as np
import pytest

from tests.test_gpu_benched_config import _Follower, _cmp_pixels

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def _run(torch, O, B, N, seed, steps, max_steps, make_actions, follow, render_every_step, streams, cap=None, before_step=None):
    from multi_car_racing_amd.vec_env import VecMultiCarRacing
    env = VecMultiCarRacing(B, N, seed=seed, use_random_direction=True, auto_reset=True, max_episode_steps=max_steps,
                            car_contacts=True, async_refill=True, streams=streams, terminal_obs=True, terminal_cap=cap)
    obs = env.reset()
    idx = np.asarray(follow); idx_t = torch.from_numpy(idx).cuda()
    fol = [_Follower(O, N, seed, int(g), max_steps) for g in idx]
    pos = {int(g): j for j, g in enumerate(idx)}
    threads = os.cpu_count() or 1
    n_term = n_checked = 0
    for k in range(steps):
        if before_step is not None:
            before_step(k, env, fol)
        a = make_actions(k)
        obs, rew, done, info = env.step(a)
        ids_t, frames_t = env.terminal_observations()
        ids = ids_t.cpu().numpy(); dn = done.cpu().numpy().astype(bool)
        # exactly one entry per finished env (no freeze in a healthy rollout), none otherwise
        assert len(ids) == len(set(ids.tolist())), f"step {k}: an env is listed twice"
        assert sorted(ids.tolist()) == np.nonzero(dn)[0].tolist(), f"step {k}: entries {sorted(ids.tolist())[:8]}.. vs done rows {np.nonzero(dn)[0][:8]}.."
        assert int(info["terminal_count"].item()) == len(ids)
        n_term += len(ids)
        # the oracles: render where an ending is possible (every step, or the TimeLimit step)
        rm = np.array([1 if (render_every_step or (max_steps > 0 and f.steps + 1 >= max_steps)) else 0 for f in fol], np.uint8)
        o_obs, o_amb, o_rew, o_done = O.step_batch([f.o for f in fol], a[idx_t].cpu().numpy(), rm, threads=threads)
        frames = frames_t.cpu().numpy() if len(ids) else None
        got_first = obs[idx_t].cpu().numpy() if dn[idx].any() else None
        where = {int(e): i for i, e in enumerate(ids)}
        for j, f in enumerate(fol):
            d, t = f.after_step(bool(o_done[j]))
            assert d == dn[f.g], f"step {k} env {f.g}: done {dn[f.g]} vs oracle {d}"
            if d:
                assert rm[j], f"step {k} env {f.g}: ended without a rendered oracle frame (test set-up)"
                _cmp_pixels(frames[where[f.g]], o_obs[j], o_amb[j], f"step {k} env {f.g} terminal frame")
                f.new_episode()
                _cmp_pixels(got_first[j], f.first_obs, f.first_amb, f"step {k} env {f.g} first frame after auto-reset")
                n_checked += 1
    frozen = int(env.debug_counters()[3])
    env.close()
    assert frozen == 0
    return n_term, n_checked


@pytest.mark.parametrize("streams,N", [(2, 2), (1, 2), (2, 8)])
def test_terminal_frames_under_time_limit(torch_cuda, oracle, streams, N):
    """every env of the batch ends by TimeLimit in the same step, three times over: B entries per ending step, each the oracle's last frame"""
    torch = torch_cuda
    B, max_steps = (96, 30) if N == 2 else (24, 25)
    g = torch.Generator(device="cuda"); g.manual_seed(5)

    def acts(k):
        a = torch.rand((B, N, 3), device="cuda", generator=g); a[..., 0] = a[..., 0] * 2 - 1; a[..., 2] *= 0.2
        return a
    n_term, n_checked = _run(torch, oracle, B, N, seed=31 + N, steps=3 * max_steps + 5, max_steps=max_steps, make_actions=acts,
                             follow=np.arange(B), render_every_step=False, streams=streams)
    assert n_term == 3 * B and n_checked == 3 * B


def test_terminal_frames_when_cars_leave_the_playfield(torch_cuda, oracle):
    """no TimeLimit: episodes end when a car is off the playfield (multi_car_racing.py:503-506) — here: put there, in a few envs at a time, at
    steps of their own; the last frame then looks at the black beyond the playfield quad, and the reward row says -100"""
    torch = torch_cuda
    B, N = 24, 2
    g = torch.Generator(device="cuda"); g.manual_seed(6)

    def acts(k):
        a = torch.rand((B, N, 3), device="cuda", generator=g); a[..., 0] = a[..., 0] * 2 - 1; a[..., 2] *= 0.2
        return a
    moved = []

    def teleport(k, env, fol):
        if k in (15, 16, 40, 75):
            st = env.get_state()["bodies"].copy()
            for e in range(k % 5, B, 5):
                car = (e + k) % N
                st[e, car, :, 0] += 800.0 if e % 2 else -800.0; st[e, car, :, 1] += 120.0
                for b in range(5):
                    fol[e].o.set_body(car, b, st[e, car, b])
                moved.append((k, e))
            env.set_bodies(st)
    n_term, n_checked = _run(torch, oracle, B, N, seed=77, steps=90, max_steps=0, make_actions=acts, follow=np.arange(B),
                             render_every_step=True, streams=2, before_step=teleport)
    assert n_checked == n_term == len(moved) and n_term >= 16


def test_terminal_frames_b4096_sampled(torch_cuda, oracle):
    """the benched batch: 4096 envs, TimeLimit 120, two rounds of 4096 endings in one step each; 48 sampled envs against their oracles, all
    envs listed exactly once per round; a capacity smaller than the number of endings keeps the first `cap` entries"""
    torch = torch_cuda
    B, N, max_steps = 4096, 2, 120
    g = torch.Generator(device="cuda"); g.manual_seed(7)

    def acts(k):
        a = torch.rand((B, N, 3), device="cuda", generator=g); a[..., 0] = a[..., 0] * 2 - 1
        return a
    rs = np.random.RandomState(3)
    n_term, n_checked = _run(torch, oracle, B, N, seed=40, steps=2 * max_steps + 3, max_steps=max_steps, make_actions=acts,
                             follow=np.sort(rs.choice(B, 48, replace=False)), render_every_step=False, streams=2)
    assert n_term >= 2 * B and n_checked >= 96
    # capacity: 4096 endings, room for 100
    from multi_car_racing_amd.vec_env import VecMultiCarRacing
    env = VecMultiCarRacing(256, 2, seed=1, auto_reset=True, max_episode_steps=10, terminal_obs=True, terminal_cap=100)
    env.reset()
    for k in range(10):
        obs, rew, done, info = env.step(acts(k)[:256])
    ids, frames = env.terminal_observations()
    assert int(done.sum().item()) == 256 and len(ids) == 100 and len(set(ids.cpu().numpy().tolist())) == 100
    for k in range(3):                                          # the envs beyond the capacity were re-spawned all the same: the rollout goes on
        obs, rew, done, info = env.step(acts(k)[:256])
    assert int(info["terminal_count"].item()) == 0 and int(env.debug_counters()[3]) == 0
    env.close()


def test_steps_without_terminal_frames_between_terminal_steps(torch_cuda, oracle):
    """the terminal entries' counters are double-buffered by step parity and zeroed a step ahead by the step that draws; a step
    that draws none (step(None): no actions) flips the parity without zeroing, and the next terminal step of that parity started from the
    counts of two steps back — stale entries, stale terminal_count after the step(None) itself.  An action-less step after EVERY ending step
    (and elsewhere): the ending steps list exactly the envs that ended, each frame the oracle's last one; after step(None) the count reads 0."""
    torch = torch_cuda
    from multi_car_racing_amd.vec_env import VecMultiCarRacing
    B, N, seed, max_steps = 48, 2, 77, 12
    env = VecMultiCarRacing(B, N, seed=seed, use_random_direction=True, auto_reset=True, max_episode_steps=max_steps, car_contacts=True,
                            async_refill=True, streams=2, terminal_obs=True)
    env.reset()
    fol = [_Follower(oracle, N, seed, g, max_steps) for g in range(B)]
    g = torch.Generator(device="cuda"); g.manual_seed(9)
    endings = 0
    for k in range(5 * max_steps + 3):
        a = torch.rand((B, N, 3), device="cuda", generator=g); a[..., 0] = a[..., 0] * 2 - 1; a[..., 2] *= 0.2
        obs, rew, done, info = env.step(a)
        ids_t, frames_t = env.terminal_observations()
        ids = ids_t.cpu().numpy(); dn = done.cpu().numpy().astype(bool)
        assert sorted(ids.tolist()) == np.nonzero(dn)[0].tolist(), f"step {k}: entries {sorted(ids.tolist())[:8]} vs done rows {np.nonzero(dn)[0][:8]}"
        rm = np.array([1 if f.steps + 1 >= max_steps else 0 for f in fol], np.uint8)
        o_obs, o_amb, _, o_done = oracle.step_batch([f.o for f in fol], a.cpu().numpy(), rm, threads=os.cpu_count() or 1)
        frames = frames_t.cpu().numpy() if len(ids) else None
        where = {int(e): i for i, e in enumerate(ids)}
        ended = False
        for j, f in enumerate(fol):
            d, _ = f.after_step(bool(o_done[j]))
            assert d == dn[j], (k, j)
            if d:
                _cmp_pixels(frames[where[j]], o_obs[j], o_amb[j], f"step {k} env {j} terminal frame")
                f.new_episode(); ended = True; endings += 1
        if ended or k % 5 == 2:
            # the reference's step(None) (mcr.py:410-431 with action None: no controls, no reward decrement, the world still steps)
            _, rew0, done0, info0 = env.step(None)
            assert int(info0["terminal_count"].item()) == 0, f"step {k}: an action-less step reports {int(info0['terminal_count'].item())} terminal entries"
            assert not bool(done0.any())
            oracle.step_batch([f.o for f in fol], None, None, threads=os.cpu_count() or 1)      # (not a TimeLimit step: k_dynamics counts steps with actions, as gym's wrapper counts step() calls of the agent)
    assert endings == 5 * B
    assert int(env.debug_counters()[3]) == 0
    env.close()
