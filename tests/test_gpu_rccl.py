"""MI355X, the `nccl` (= RCCL) branch of sharded.reduce_metrics and the three-chain step's phase-word ordering in a process
that holds a live RCCL communicator (BASELINE configs[2] uses both; with one GPU the group has one rank).  A child process
initialises a 1-rank RCCL group on cuda:0, all-reduces once (so the communicator and its streams exist), steps a ShardedVecEnv
200 steps with contacts and auto-resets, reduces the rollout metrics over RCCL, and hands everything back; the parent steps the
same envs without any process group: results must be bit-identical, the step must still order its streams with phase words and
no status word may be set."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _actions(k, total, N):
    rs = np.random.RandomState(5000 + k)
    a = np.stack([rs.uniform(-1, 1, (total, N)), rs.uniform(0, 1, (total, N)), rs.uniform(0, 0.3, (total, N))], -1).astype(np.float32)
    a[:, 1, 1] = 1.0                                       # car 1 floors it: car<->car contacts -> the contact chain runs
    return a


def _rollout(env, total, N, steps):
    import torch
    rec_obs, rec_rew, rec_done = [env.reset().cpu().numpy().copy()], [], []
    for k in range(steps):
        obs, rew, done, _ = env.step(torch.from_numpy(_actions(k, total, N)).cuda())
        rec_rew.append(rew.cpu().numpy().copy()); rec_done.append(done.cpu().numpy().copy())
        if k % 20 == 19:
            rec_obs.append(obs.cpu().numpy().copy())
    return np.stack(rec_obs), np.stack(rec_rew), np.stack(rec_done)


_KW = dict(use_random_direction=True, auto_reset=True, max_episode_steps=60, car_contacts=True, async_refill=True, streams=2)


def _worker(port, total, N, seed, steps, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    warm = torch.ones(4, device="cuda:0"); dist.all_reduce(warm); torch.cuda.synchronize()      # the communicator and its streams exist from here on
    from multi_car_racing_amd.sharded import ShardedVecEnv, reduce_metrics
    env = ShardedVecEnv(total, N, seed=seed, rank=0, world_size=1, device="cuda:0", **_KW)
    obs, rew, done = _rollout(env, total, N, steps)
    episodes, ret = env.env.rollout_stats()
    m = reduce_metrics(env_steps=total * steps, elapsed_s=1.25, episodes=episodes, return_sum=ret)
    ordering = int(env.env.L.mcr_step_ordering(env.env.h))
    status = [int(x) for x in env.env.status_words()]
    backend = dist.get_backend()
    env.close()
    dist.barrier(); torch.cuda.synchronize()
    dist.destroy_process_group()
    q.put((obs, rew, done, m, ordering, status, backend, episodes, ret))


def test_step_and_metrics_under_a_live_rccl_group():
    import torch
    import torch.multiprocessing as mp
    assert torch.cuda.is_available()
    total, N, seed, steps = 256, 2, 23, 200
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(_free_port(), total, N, seed, steps, q))
    p.start()
    obs, rew, done, m, ordering, status, backend, episodes, ret = q.get(timeout=900)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert backend == "nccl"
    # the RCCL all-reduce of the metrics (sharded.py: the `nccl` branch) returned this rank's values
    assert m["world_size"] == 1 and m["env_steps"] == total * steps and m["elapsed_s"] == 1.25
    assert m["episodes"] == episodes and m["return_sum"] == ret and episodes > 0
    # phase words (bit 0) still order the step's streams beside RCCL's, and nothing gave up or overflowed
    assert ordering & 1, f"the step fell back to events under RCCL (ordering {ordering})"
    assert not any(status), f"status words set: {status}"
    # the same rollout without any process group in this process
    from multi_car_racing_amd.vec_env import VecMultiCarRacing
    env = VecMultiCarRacing(total, N, seed=seed, **_KW)
    f_obs, f_rew, f_done = _rollout(env, total, N, steps)
    assert int(env.L.mcr_step_ordering(env.h)) & 1
    env.close()
    assert f_done.any(), "rollout never crossed the TimeLimit"
    assert np.array_equal(rew, f_rew) and np.array_equal(done, f_done), "rewards / done flags differ under RCCL"
    assert np.array_equal(obs, f_obs), "observations differ under RCCL"
