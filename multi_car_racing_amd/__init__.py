"""multi_car_racing_amd — MI355X-native batched MultiCarRacing-v0 step.

Host mirror of the reference's interface for the hot path: `make("MultiCarRacing-v0", ...)`,
`MultiCarRacing` (single env, gym surface) and `VecMultiCarRacing` (B envs per GPU on device tensors).
"""
from ._lib import McrError, load as load_library  # noqa: F401
from .registry import make, register, TimeLimit, ENV_ID  # noqa: F401
from .env import MultiCarRacing  # noqa: F401


def __getattr__(name):
    if name == "VecMultiCarRacing":      # needs torch: imported lazily
        from .vec_env import VecMultiCarRacing
        return VecMultiCarRacing
    if name == "ShardedVecEnv":
        from .sharded import ShardedVecEnv
        return ShardedVecEnv
    raise AttributeError(name)
