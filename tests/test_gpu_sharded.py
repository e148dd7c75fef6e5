"""MI355X, world_size 2 over gloo with BOTH ranks on cuda:0: two ShardedVecEnv halves step their env slices and must
reproduce one full batch bit for bit (obs, reward, done) — the claim that makes the 8-GPU config (BASELINE configs[2])
exact by construction: env g behaves the same whatever rank owns it."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _actions(k, total, N):
    rs = np.random.RandomState(1000 + k)                   # same global action table on every rank
    a = np.stack([rs.uniform(-1, 1, (total, N)), rs.uniform(0, 1, (total, N)), rs.uniform(0, 0.3, (total, N))], -1).astype(np.float32)
    a[:, 1, 1] = 1.0                                       # car 1 floors it: car<->car contacts -> side stream
    return a


def _worker(rank, world, port, total, N, seed, steps, L, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multi_car_racing_amd.sharded import ShardedVecEnv, reduce_metrics
    torch.cuda.set_device(0)
    env = ShardedVecEnv(total, N, seed=seed, rank=rank, world_size=world, device="cuda:0", use_random_direction=True,
                        auto_reset=True, max_episode_steps=L, car_contacts=True, async_refill=True, streams=2)
    lo, hi = env.lo, env.hi
    rec_obs, rec_rew, rec_done = [env.reset().cpu().numpy().copy()], [], []
    for k in range(steps):
        a = torch.from_numpy(_actions(k, total, N)[lo:hi]).cuda()
        obs, rew, done, _ = env.step(a)
        rec_rew.append(rew.cpu().numpy().copy()); rec_done.append(done.cpu().numpy().copy())
        if k % 10 == 9 or bool(done.any()):
            rec_obs.append(obs.cpu().numpy().copy())
    episodes, ret = env.env.rollout_stats()
    m = reduce_metrics(env_steps=(hi - lo) * steps, elapsed_s=1.0 + rank, episodes=episodes, return_sum=ret)
    q.put((rank, lo, hi, np.stack(rec_obs), np.stack(rec_rew), np.stack(rec_done), m))
    dist.barrier()
    env.close()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_equal_one_full_batch():
    import torch
    import torch.multiprocessing as mp
    assert torch.cuda.is_available()
    world, total, N, seed, steps, L = 2, 48, 2, 17, 90, 40
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, N, seed, steps, L, q)) for r in range(world)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    for p in procs: p.join(timeout=120)
    assert all(p.exitcode == 0 for p in procs)
    # the full batch in this process
    from multi_car_racing_amd.vec_env import VecMultiCarRacing
    env = VecMultiCarRacing(total, N, seed=seed, use_random_direction=True, auto_reset=True, max_episode_steps=L,
                            car_contacts=True, async_refill=True, streams=2)
    f_obs, f_rew, f_done = [env.reset().cpu().numpy().copy()], [], []
    snap_steps = []
    for k in range(steps):
        obs, rew, done, _ = env.step(torch.from_numpy(_actions(k, total, N)).cuda())
        f_rew.append(rew.cpu().numpy().copy()); f_done.append(done.cpu().numpy().copy())
        f_obs.append(obs.cpu().numpy().copy()); snap_steps.append(k)
    episodes, ret = env.rollout_stats()
    env.close()
    f_rew, f_done = np.stack(f_rew), np.stack(f_done)
    assert f_done.any(), "rollout never crossed the TimeLimit"
    for (rank, lo, hi, obs, rew, done, m) in res:
        assert np.array_equal(rew, f_rew[:, lo:hi]), f"rank {rank}: rewards differ from the full batch"
        assert np.array_equal(done, f_done[:, lo:hi]), f"rank {rank}: done flags differ"
        # the rank recorded obs at reset, every 10th step and every step in which one of ITS envs finished
        want = [f_obs[0][lo:hi]] + [f_obs[1 + k][lo:hi] for k in range(steps) if k % 10 == 9 or f_done[k, lo:hi].any()]
        assert obs.shape[0] == len(want) and np.array_equal(obs, np.stack(want)), f"rank {rank}: observations differ"
        assert m["world_size"] == 2 and m["env_steps"] == total * steps and m["elapsed_s"] == 2.0
        assert m["episodes"] == episodes and abs(m["return_sum"] - ret) < 1e-6
