"""Data-parallel env sharding: one process per GPU, contiguous env slices, no data-path
collective.  The only communication is a tiny SUM/MAX all-reduce of rollout metrics (RCCL over xGMI when the
process group backend is "nccl"; gloo on CPU in the tests)."""
import numpy as np


def shard_range(total_envs, rank, world_size):
    """Contiguous slice [lo, hi) of the global env index space owned by `rank` (sizes differ by at most 1)."""
    base, rem = divmod(int(total_envs), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def reduce_metrics(env_steps, elapsed_s, episodes=0.0, return_sum=0.0, group=None):
    """All-reduce rollout metrics: SUM of env_steps/episodes/return_sum, MAX of elapsed seconds.
    Returns dict with the job-wide values (identical on every rank).  Works without torch.distributed
    (single process) by returning the local values."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return dict(env_steps=float(env_steps), elapsed_s=float(elapsed_s), episodes=float(episodes),
                    return_sum=float(return_sum), world_size=1)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    sums = torch.tensor([env_steps, episodes, return_sum], dtype=torch.float64, device=dev)
    mx = torch.tensor([elapsed_s], dtype=torch.float64, device=dev)
    dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
    s = sums.cpu().numpy(); m = mx.cpu().numpy()
    return dict(env_steps=float(s[0]), elapsed_s=float(m[0]), episodes=float(s[1]), return_sum=float(s[2]),
                world_size=dist.get_world_size(group))


class ShardedVecEnv:
    """This rank's slice of a job-wide batch of `total_envs` environments.  Env g (global index) behaves
    identically whatever the world size because its RNG streams are keyed by g (vec_env.py)."""

    def __init__(self, total_envs, num_agents=2, seed=0, rank=0, world_size=1, device=None, **kw):
        from .vec_env import VecMultiCarRacing
        self.rank, self.world_size = rank, world_size
        self.lo, self.hi = shard_range(total_envs, rank, world_size)
        self.env = VecMultiCarRacing(self.hi - self.lo, num_agents, device=device, seed=seed, env_offset=self.lo,
                                     world_size=world_size, **kw)

    def __getattr__(self, name):
        return getattr(self.env, name)
