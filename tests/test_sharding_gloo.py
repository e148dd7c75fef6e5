"""CPU, world_size 2 over gloo: env slices + the scalar metric all-reduce."""
import ctypes
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, total, N, seed, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multi_car_racing_amd import _lib
    from multi_car_racing_amd.sharded import shard_range, reduce_metrics
    L = _lib.load()
    lo, hi = shard_range(total, rank, world)
    n = hi - lo
    mt_t = np.zeros((n, _lib.MT_WORDS), np.uint32); mt_g = np.zeros((n, _lib.MT_WORDS), np.uint32)
    for e in range(n):
        g = seed + lo + e
        L.mcr_mt_seed(_lib.ptr(mt_t[e]), ctypes.c_uint32(g)); L.mcr_mt_seed(_lib.ptr(mt_g[e]), ctypes.c_uint32((g + 2 ** 31) % 2 ** 32))
    blobs = np.zeros((n, _lib.episode_bytes()), np.uint8); info = np.zeros((n, 12), np.int32)
    L.mcr_episodes_generate(_lib.ptr(mt_t), _lib.ptr(mt_g), n, N, 2, _lib.ptr(blobs), _lib.ptr(info), 2)
    digest = [int(np.frombuffer(b.tobytes(), np.uint32).astype(np.uint64).sum()) for b in blobs]
    m = reduce_metrics(env_steps=n * 10, elapsed_s=1.0 + rank, episodes=rank + 1, return_sum=float(info[:, 0].sum()))
    q.put((rank, lo, hi, digest, m, info[:, 0].tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_metric_allreduce(lib):
    world, total, N, seed = 2, 9, 2, 31
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, N, seed, q)) for r in range(world)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)])
    for p in procs: p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    (r0, lo0, hi0, d0, m0, t0), (r1, lo1, hi1, d1, m1, t1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 5, 5, 9)                       # contiguous, sizes differ by <= 1
    assert m0 == m1                                                    # identical on every rank
    assert m0["env_steps"] == 90 and m0["elapsed_s"] == 2.0 and m0["episodes"] == 3 and m0["world_size"] == 2
    assert m0["return_sum"] == float(sum(t0) + sum(t1))
    # the same global env produces the same episode whatever the world size
    L = lib.load()
    mt_t = np.zeros((total, lib.MT_WORDS), np.uint32); mt_g = np.zeros((total, lib.MT_WORDS), np.uint32)
    for e in range(total):
        L.mcr_mt_seed(lib.ptr(mt_t[e]), ctypes.c_uint32(seed + e)); L.mcr_mt_seed(lib.ptr(mt_g[e]), ctypes.c_uint32((seed + e + 2 ** 31) % 2 ** 32))
    blobs = np.zeros((total, lib.episode_bytes()), np.uint8); info = np.zeros((total, 12), np.int32)
    L.mcr_episodes_generate(lib.ptr(mt_t), lib.ptr(mt_g), total, N, 2, lib.ptr(blobs), lib.ptr(info), 3)
    digest = [int(np.frombuffer(b.tobytes(), np.uint32).astype(np.uint64).sum()) for b in blobs]
    assert digest == d0 + d1


def test_shard_range_properties():
    from multi_car_racing_amd.sharded import shard_range
    for total in (1, 7, 8, 4096, 32768, 1000):
        for w in (1, 2, 3, 4, 8):
            spans = [shard_range(total, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
