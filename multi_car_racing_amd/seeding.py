"""gym 0.17.2 `gym.utils.seeding.np_random` restated ([3P-recalled]): the env stream is a numpy
RandomState seeded with the little-endian uint32 words of the first 8 bytes of sha512(str(seed)).
(gym is MIT-licensed, (c) 2016 OpenAI; this file restates the published behaviour of three of its helper functions —
`np_random`, `hash_seed`, `_bigint_from_bytes` / `_int_list_from_bigint` — so that `env.seed(s)` draws the same tracks as the
reference; it is not a copy of gym's source and the reference repository does not contain that file.)"""
import hashlib
import os
import struct

import numpy as np


def _bigint_from_bytes(b):
    sizeof_int = 4
    padding = sizeof_int - len(b) % sizeof_int
    b += b"\0" * padding
    n = len(b) // sizeof_int
    acc = 0
    for i, val in enumerate(struct.unpack("{}I".format(n), b)):
        acc += 2 ** (sizeof_int * 8 * i) * val
    return acc


def create_seed(a=None, max_bytes=8):
    if a is None:
        a = _bigint_from_bytes(os.urandom(max_bytes))
    elif isinstance(a, int):
        a = a % 2 ** (8 * max_bytes)
    else:
        raise ValueError("Invalid type for seed: {} ({})".format(type(a), a))
    return a


def hash_seed(seed=None, max_bytes=8):
    if seed is None:
        seed = create_seed(max_bytes=max_bytes)
    h = hashlib.sha512(str(seed).encode("utf8")).digest()
    return _bigint_from_bytes(h[:max_bytes])


def _int_list_from_bigint(bigint):
    if bigint < 0:
        raise ValueError("Seed must be non-negative, not {}".format(bigint))
    if bigint == 0:
        return [0]
    ints = []
    while bigint > 0:
        bigint, mod = divmod(bigint, 2 ** 32)
        ints.append(mod)
    return ints


def np_random(seed=None):
    if seed is not None and not (isinstance(seed, int) and 0 <= seed):
        raise ValueError("Seed must be a non-negative integer or omitted, not {}".format(seed))
    seed = create_seed(seed)
    rng = np.random.RandomState()
    rng.seed(_int_list_from_bigint(hash_seed(seed)))
    return rng, seed
