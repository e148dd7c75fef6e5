"""Shared helpers for the parity tests: build the oracle's episode for the same seeds the product uses."""
import numpy as np


def oracle_episode(O, N, seed, g, direction="CCW", use_random_direction=False):
    """Episode of global env index g exactly as VecMultiCarRacing seeds it (vec_env.py docstring)."""
    s = (seed + g) % 2 ** 32
    return O.new_episode(N, np.random.RandomState(s), np.random.RandomState((s + 2 ** 31) % 2 ** 32),
                         direction=direction, use_random_direction=use_random_direction)


def random_actions(rng, B, N, brake_scale=1.0):
    a = np.empty((B, N, 3), np.float32)
    a[..., 0] = rng.uniform(-1, 1, (B, N))
    a[..., 1] = rng.uniform(0, 1, (B, N))
    a[..., 2] = rng.uniform(0, 1, (B, N)) * brake_scale
    return a
