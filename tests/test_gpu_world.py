"""MI355X: VecMultiCarRacing keeps ONE b2World per env across its auto-resets, as the reference does (multi_car_racing.py:138; _destroy
:173-181, reset :341) — csrc/k_world.h re-issues the fixtures' broadphase proxy ids inside the reset pass.  The HIP path through the C-ABI
against the oracle's literal b2DynamicTree (world mode 1): ids, every step's rewards, done, the whole state, over several consecutive
episodes with cars that touch (fixtureA of a car<->car contact is the lower proxy id) — and NOT equal to the fresh-world oracle (rounds
1-5's definition) on the same rollouts, so the tests can see the difference."""
import ctypes
import os

import numpy as np
import pytest

from tests.test_gpu_benched_config import _Follower, _cmp_state

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def _device_ids(env, lib, g):
    out = np.zeros(512 + 64, np.int32)
    n = lib.load().mcr_debug_read_proxy_ids(env.h, int(g), lib.ptr(out), len(out))
    assert n > 0
    return out[:n]


def _drive(torch, gen, B, N, k, L=84):
    """a policy that drives INTO its neighbours (bench.py --actions drive plus a bias): gas 1, steering noise +-0.05; in the first 40 steps of
    an episode of L steps the even cars steer one way and the odd cars the other (the grid's pairs converge in half of the envs), then the
    even cars brake for 35 steps (whoever is behind runs into them)"""
    a = torch.zeros((B, N, 3), device="cuda")
    a[..., 0] = torch.rand((B, N), device="cuda", generator=gen) * 0.1 - 0.05
    a[..., 1] = 1.0
    ke = k % L
    if ke < 40: a[:, ::2, 0] += 0.12; a[:, 1::2, 0] -= 0.12
    if 40 <= ke < 75: a[:, ::2, 1] = 0.0; a[:, ::2, 2] = 0.9
    return a


@pytest.mark.parametrize("B,N,n_sample,max_steps", [(4096, 2, 14, 84), (512, 8, 7, 84)])
def test_auto_reset_episodes_on_one_world(torch_cuda, oracle, lib, B, N, n_sample, max_steps):
    """B = 4096 (N = 2), sampled: 4 consecutive episodes per env (TimeLimit), driving policy with pile-ups; rewards and done every step, the
    whole state every 42 steps and the proxy ids of every episode equal to the one-world oracle; a fresh-world oracle beside it disagrees."""
    torch = torch_cuda
    from multi_car_racing_amd.vec_env import VecMultiCarRacing
    seed, episodes = 600 + N, 4
    env = VecMultiCarRacing(B, N, seed=seed, use_random_direction=True, auto_reset=True, max_episode_steps=max_steps, car_contacts=True,
                            async_refill=True, streams=2, obs=False)
    env.reset()
    idx = np.sort(np.random.RandomState(seed).choice(B, n_sample, replace=False)); idx_t = torch.from_numpy(idx).cuda()
    fol = [_Follower(oracle, N, seed, int(g), max_steps) for g in idx]                        # one world per oracle: the reference
    control = N == 2                                                                          # (the negative control doubles the oracle work: one configuration carries it)
    fresh = [_Follower(oracle, N, seed, int(g), max_steps) for g in idx] if control else []   # rounds 1-5: a fresh world per episode
    for f in fresh: f.o.set_world_mode(0)
    gen = torch.Generator(device="cuda"); gen.manual_seed(seed)
    thr = os.cpu_count() or 1
    fresh_reward_differs = fresh_state_differs = contacts = ids_not_ascending = 0
    for k in range(episodes * max_steps):
        a = _drive(torch, gen, B, N, k)
        _, rew, done, _ = env.step(a)
        a_s = a[idx_t].cpu().numpy(); rw = rew[idx_t].cpu().numpy(); dn = done[idx_t].cpu().numpy().astype(bool)
        _, _, o_rew, o_done = oracle.step_batch([f.o for f in fol], a_s, None, threads=thr)
        if control:
            _, _, f_rew, f_done = oracle.step_batch([f.o for f in fresh], a_s, None, threads=thr)
        ended = []
        for j, f in enumerate(fol):
            d, _ = f.after_step(bool(o_done[j]))
            assert np.array_equal(o_rew[j], rw[j]), f"step {k} env {f.g}: reward {rw[j]} vs the one-world oracle {o_rew[j]}"
            assert d == dn[j], f"step {k} env {f.g}: done"
            if control:
                fresh[j].after_step(bool(f_done[j]))
                fresh_reward_differs += int(not np.array_equal(f_rew[j], rw[j]))
            contacts += f.o.num_car_contacts() > 0
            if d: ended.append(j)
        if k % 42 == 41 and not ended:
            _cmp_state(env, fol, idx, f"step {k}")
            st = env.get_state()["bodies"]
            fresh_state_differs += sum(int(not np.array_equal(st[z.g], z.o.state()["bodies"])) for z in fresh)
        for j in ended:
            fol[j].new_episode()
            if control: fresh[j].new_episode()
            tid, fid = fol[j].o.proxy_ids()
            want = np.concatenate([tid, fid.ravel()])
            got = _device_ids(env, lib, fol[j].g)
            assert np.array_equal(got, want), f"step {k} env {fol[j].g}: proxy ids of the new episode: {got[:6]}.. vs the oracle's tree {want[:6]}.."
            ids_not_ascending += int(not np.all(np.diff(want) > 0))
    assert int(env.debug_counters()[3]) == 0 and env.verdict_mismatches() == 0 and env.status_words()[:2].tolist() == [0, 0]
    env.close()
    assert ids_not_ascending >= n_sample, "second and later episodes must draw their ids off the free list"
    assert contacts > 20, "the driving policy produced no car<->car contacts"
    assert not control or (fresh_reward_differs > 0 and fresh_state_differs > 0), "a fresh-world oracle agreed throughout: the test did not exercise the world's ids"


def test_masked_reset_and_snapshot_keep_the_world(torch_cuda, oracle, lib):
    """reset_envs(mask) is reset() on the env's world as well (the reference has one reset path); a state blob carries the world (ids, free
    stack) so that a restored env's NEXT episode draws the same ids."""
    torch = torch_cuda
    from multi_car_racing_amd.vec_env import VecMultiCarRacing
    B, N, seed = 6, 2, 911
    env = VecMultiCarRacing(B, N, seed=seed, use_random_direction=True, auto_reset=True, max_episode_steps=0, car_contacts=True,
                            async_refill=False, streams=2, obs=False)
    env.reset()
    fol = [_Follower(oracle, N, seed, g, 0) for g in range(B)]
    gen = torch.Generator(device="cuda"); gen.manual_seed(1)
    L = lib.load()

    def steps(n, k0):
        for k in range(k0, k0 + n):
            a = _drive(torch, gen, B, N, k)
            _, rew, _, _ = env.step(a)
            _, _, o_rew, _ = oracle.step_batch([f.o for f in fol], a.cpu().numpy(), None, threads=4)
            assert np.array_equal(o_rew, rew.cpu().numpy()), f"step {k}"
    steps(40, 0)
    for rnd in range(3):
        mask = np.zeros(B, np.uint8); mask[rnd::2] = 1
        env.reset_envs(torch.from_numpy(mask).cuda()); env.wait_refills()
        for g in np.nonzero(mask)[0]:
            fol[g].new_episode()
            tid, fid = fol[g].o.proxy_ids()
            assert np.array_equal(_device_ids(env, lib, g), np.concatenate([tid, fid.ravel()])), f"round {rnd} env {g}"
        steps(40, 40 * (rnd + 1))
    _cmp_state(env, fol, np.arange(B), "after three masked resets")
    # snapshot env 0, run it through another reset, restore: the world comes back with the blob
    nb = L.mcr_state_blob_bytes(env.h)
    blob = np.zeros(nb, np.uint8)
    assert L.mcr_get_state_blob(env.h, 0, lib.ptr(blob)) == 0
    ids_before = _device_ids(env, lib, 0).copy()
    m0 = np.zeros(B, np.uint8); m0[0] = 1
    env.reset_envs(torch.from_numpy(m0).cuda()); env.wait_refills()
    assert not np.array_equal(_device_ids(env, lib, 0)[:8], ids_before[:8])
    assert L.mcr_set_state_blob(env.h, 0, lib.ptr(blob)) == 0
    assert np.array_equal(_device_ids(env, lib, 0), ids_before)
    env.close()


def test_fresh_world_switch_is_rounds_1_to_5_definition(torch_cuda, oracle):
    """mcr_config::fresh_world = 1 (VecMultiCarRacing(fresh_world=True)): every episode the first episode of a fresh world — the oracle's world
    mode 0 — over three auto-reset episodes with touching cars; the default handle beside it (one world) follows the mode-1 oracle, and the two
    handles differ from each other from the second episode on."""
    torch = torch_cuda
    from multi_car_racing_amd.vec_env import VecMultiCarRacing
    B, N, seed, L = 12, 2, 1234, 60
    kw = dict(seed=seed, use_random_direction=True, auto_reset=True, max_episode_steps=L, car_contacts=True, async_refill=False, streams=2, obs=False)
    fresh, one = VecMultiCarRacing(B, N, fresh_world=True, **kw), VecMultiCarRacing(B, N, **kw)
    fresh.reset(); one.reset()
    f0 = [_Follower(oracle, N, seed, g, L) for g in range(B)]
    for f in f0: f.o.set_world_mode(0)
    f1 = [_Follower(oracle, N, seed, g, L) for g in range(B)]
    gen = torch.Generator(device="cuda"); gen.manual_seed(3)
    handles_differ = 0
    for k in range(3 * L):
        a = _drive(torch, gen, B, N, k, L)
        _, r0, d0, _ = fresh.step(a); _, r1, d1, _ = one.step(a)
        an = a.cpu().numpy()
        _, _, o0, od0 = oracle.step_batch([f.o for f in f0], an, None, threads=4)
        _, _, o1, od1 = oracle.step_batch([f.o for f in f1], an, None, threads=4)
        assert np.array_equal(o0, r0.cpu().numpy()), f"step {k}: fresh-world handle vs the mode-0 oracle"
        assert np.array_equal(o1, r1.cpu().numpy()), f"step {k}: one-world handle vs the mode-1 oracle"
        handles_differ += int(not torch.equal(r0, r1))
        for j in range(B):
            e0, _ = f0[j].after_step(bool(od0[j])); e1, _ = f1[j].after_step(bool(od1[j]))
            assert e0 == bool(d0[j].item()) and e1 == bool(d1[j].item())
            if e0: f0[j].new_episode()
            if e1: f1[j].new_episode()
    _cmp_state(fresh, f0, np.arange(B), "fresh-world handle, end"); _cmp_state(one, f1, np.arange(B), "one-world handle, end")
    assert handles_differ > 0, "the two definitions never disagreed: the rollout did not exercise the ids"
    fresh.close(); one.close()
