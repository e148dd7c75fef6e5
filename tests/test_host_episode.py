"""CPU: the PRODUCT's host code (multi_car_racing_amd/csrc/mcr_host.cpp through the C-ABI) against the
reference-generated goldens and against numpy's RandomState.  No GPU, no compute kernels."""
import ctypes
import json
import os

import numpy as np

G = os.path.join(os.path.dirname(__file__), "golden")


def _mt(L, lib, seed):
    mt = np.zeros(lib.MT_WORDS, np.uint32)
    L.mcr_mt_seed(lib.ptr(mt), ctypes.c_uint32(seed))
    return mt


def test_mt19937_matches_numpy(lib):
    L = lib.load()
    for s in [0, 1, 12345, 2 ** 32 - 1]:
        mt = _mt(L, lib, s)
        st = np.random.RandomState(s).get_state()
        assert np.array_equal(mt[:624], st[1]) and mt[624] == st[2]
        rs = np.random.RandomState(s)
        for _ in range(1500):
            assert L.mcr_mt_random_sample(lib.ptr(mt)) == rs.random_sample()
    key = np.array([123, 456, 7], np.uint32); mt = np.zeros(lib.MT_WORDS, np.uint32)
    L.mcr_mt_seed_by_array(lib.ptr(mt), lib.ptr(key), 3)
    rs = np.random.RandomState(); rs.seed([123, 456, 7])
    assert np.array_equal(mt[:624], rs.get_state()[1])


def test_direction_and_car_order_draws_match_numpy(lib):
    L = lib.load()
    for s in range(40):
        for N in (1, 2, 3, 4, 5, 8):
            mt = _mt(L, lib, s); rs = np.random.RandomState(s)
            for _ in range(3):
                assert bool(L.mcr_mt_choice_cw(lib.ptr(mt))) == (rs.choice(["CW", "CCW"]) == "CW")
                o = np.zeros(N, np.int32); L.mcr_mt_car_order(lib.ptr(mt), N, lib.ptr(o))
                assert list(o) == list(rs.choice(list(range(N)), size=N, replace=False))


def test_tracks_bit_exact_vs_reference_goldens(lib):
    L = lib.load()
    g = np.load(os.path.join(G, "tracks.npz"))
    blob = np.zeros(lib.episode_bytes(), np.uint8); info = np.zeros(4, np.int32); order = np.array([0, 1], np.int32)
    for s in g["seeds"]:
        mt = _mt(L, lib, int(s))
        assert L.mcr_episode_generate(lib.ptr(mt), 2, 0, lib.ptr(order), lib.ptr(blob), lib.ptr(info)) == 0
        ep = lib.unpack_episode(blob); tr = g[f"s{s}_track"]
        assert info[2] == int(g[f"s{s}_retries"]) and ep["T"] == len(tr)
        assert np.array_equal(ep["track"], tr[:, [2, 3, 1]]) and np.array_equal(ep["alpha"], tr[:, 0])
        assert np.array_equal(ep["quads"], g[f"s{s}_poly"].astype(np.float32))       # glVertex3f / b2Vec2 see f32
        assert np.array_equal(((ep["quad_meta"] >> 8) & 0x3ff) > 0, g[f"s{s}_is_tile"].astype(bool))
        # colour ids: tile i -> shade i%3; kerb -> white if i%2==0 else red
        col = g[f"s{s}_color"]; ids = ep["quad_meta"] & 0xff; ti = -1
        for q in range(ep["P"]):
            if (ep["quad_meta"][q] >> 8) & 0x3ff:
                ti = int((ep["quad_meta"][q] >> 8) & 0x3ff) - 1
                assert ids[q] == ti % 3 and abs(col[q, 0] - (0.4 + 0.01 * (ti % 3))) < 1e-12
            else:
                assert ids[q] == (3 if ti % 2 == 0 else 4) and tuple(col[q]) == ((1, 1, 1) if ti % 2 == 0 else (1, 0, 0))
                assert int(ep["quad_meta"][q] >> 18) - 1 == ti          # kerb quads name their owner tile
        # the numpy stream advanced exactly as the reference's did
        rs = np.random.RandomState(int(s))
        for _ in range(24 * (int(info[2]) + 1)):
            rs.random_sample()
        assert L.mcr_mt_random_sample(lib.ptr(mt)) == rs.random_sample()


def test_spawn_poses_vs_reference_goldens(lib):
    L = lib.load()
    sp = json.load(open(os.path.join(G, "spawn.json")))
    blob = np.zeros(lib.episode_bytes(), np.uint8); info = np.zeros(4, np.int32)
    n = 0
    for c in sp["cases"]:
        if c.get("random_direction"):
            continue
        mt = _mt(L, lib, c["track_seed"]); order = np.array(c["car_order"], np.int32)
        L.mcr_episode_generate(lib.ptr(mt), c["N"], int(c["direction"] == "CW"), lib.ptr(order), lib.ptr(blob), lib.ptr(info))
        ep = lib.unpack_episode(blob)
        assert ep["cw"] == (c["direction"] == "CW") and ep["T"] == c["T"]
        assert np.array_equal(ep["spawn"][:c["N"]], np.array(c["poses"]))
        n += 1
    assert n == 40


def test_batched_generator_is_deterministic_and_matches_numpy_streams(lib, oracle):
    L = lib.load()
    B, N, seed = 12, 3, 77
    def run(threads):
        mt_t = np.zeros((B, lib.MT_WORDS), np.uint32); mt_g = np.zeros((B, lib.MT_WORDS), np.uint32)
        for e in range(B):
            L.mcr_mt_seed(lib.ptr(mt_t[e]), ctypes.c_uint32(seed + e)); L.mcr_mt_seed(lib.ptr(mt_g[e]), ctypes.c_uint32((seed + e + 2 ** 31) % 2 ** 32))
        blobs = np.zeros((B, lib.episode_bytes()), np.uint8); info = np.zeros((B, 12), np.int32)
        assert L.mcr_episodes_generate(lib.ptr(mt_t), lib.ptr(mt_g), B, N, 2, lib.ptr(blobs), lib.ptr(info), threads) == 0
        return blobs, info, mt_t
    b1, i1, m1 = run(1); b4, i4, m4 = run(4)
    assert np.array_equal(b1, b4) and np.array_equal(i1, i4) and np.array_equal(m1, m4)
    from tests.util import oracle_episode
    for e in range(B):
        ep = lib.unpack_episode(b1[e]); oe = oracle_episode(oracle, N, seed, e, use_random_direction=True)
        assert ep["cw"] == (oe["direction"] == "CW") and list(i1[e, 4:4 + N]) == oe["car_order"]
        assert np.array_equal(ep["track"], oe["track"][:, [2, 3, 1]])
        assert np.array_equal(ep["spawn"][:N], oe["poses"])


def test_mass_kats_product(lib, oracle):
    L = lib.load()
    m = np.zeros(6, np.float32); L.mcr_mass_props(lib.ptr(m))
    assert abs(1 / m[0] - 7.06) < 1e-5 and abs(1 / m[1] - 18.2122788) < 2e-5 and abs(m[3] + 0.0825307) < 1e-6
    assert abs(1 / m[4] - 0.06048) < 1e-7 and abs(1 / m[5] - 0.0074592) < 1e-8
    assert np.array_equal(m, oracle.mass_props())          # two independent restatements agree bitwise


def test_host_sincos_equals_oracle_spec(lib, oracle):
    L = lib.load()
    rng = np.random.RandomState(1)
    for a in rng.uniform(-30, 30, 2000).astype(np.float32):
        s, c = ctypes.c_float(), ctypes.c_float()
        L.mcr_sincos_host(ctypes.c_float(a), ctypes.byref(s), ctypes.byref(c))
        assert (s.value, c.value) == oracle.sincos(float(a), 0)


def test_gym_seeding_restatement():
    import hashlib, struct
    from multi_car_racing_amd import seeding
    rng, s = seeding.np_random(42)
    assert s == 42
    h = hashlib.sha512(b"42").digest()[:8]
    lo, hi = struct.unpack("<II", h)
    ref = np.random.RandomState(); ref.seed([lo, hi])
    assert rng.random_sample() == ref.random_sample()
    import pytest
    with pytest.raises(ValueError):
        seeding.np_random(-1)


def test_generated_episodes_are_well_formed_over_many_seeds(lib):
    """Size-independent properties of the host episode generator (:183-338, :366-406) over 300 seeds: capacities hold,
    the track is a closed loop of ~TRACK_DETAIL_STEP-spaced points inside the playfield, every tile owns one road quad,
    kerb quads point at a tile, and the cars spawn on the track."""
    L = lib.load()
    n, N = 300, 4
    mt_t = np.zeros((n, lib.MT_WORDS), np.uint32); mt_d = np.zeros((n, lib.MT_WORDS), np.uint32)
    for e in range(n):
        L.mcr_mt_seed(lib.ptr(mt_t[e]), ctypes.c_uint32(1000 + e)); L.mcr_mt_seed(lib.ptr(mt_d[e]), ctypes.c_uint32(5000 + e))
    blobs = np.empty((n, lib.episode_bytes()), np.uint8); info = np.zeros((n, 12), np.int32)
    lib.check(L.mcr_episodes_generate(lib.ptr(mt_t), lib.ptr(mt_d), n, N, 2, lib.ptr(blobs), lib.ptr(info), 4))
    dirs = set()
    for e in range(n):
        ep = lib.unpack_episode(blobs[e])
        T, P = ep["T"], ep["P"]
        assert 150 <= T <= 512 and T <= P <= 768 and info[e, 0] == T and info[e, 1] == P
        xy = ep["track"][:, :2]
        assert np.isfinite(ep["track"]).all() and np.isfinite(ep["quads"]).all()
        assert np.abs(xy).max() < 2000 / 6.0                                       # inside PLAYFIELD
        d = np.linalg.norm(np.diff(np.vstack([xy, xy[:1]]), axis=0), axis=1)       # closed loop incl. the seam
        assert np.allclose(d[:-1], 3.5, atol=1e-6) and d[-1] < 30.0               # TRACK_DETAIL_STEP apart; the seam is looser (:262-268)
        tiles = (ep["quad_meta"] >> 8) & 0x3ff
        owners = ep["quad_meta"] >> 18
        assert np.array_equal(np.sort(tiles[tiles > 0]), np.arange(1, T + 1))       # every tile exactly one road quad
        assert ((tiles > 0) | ((owners >= 1) & (owners <= T))).all()               # kerbs reference their tile
        sp = ep["spawn"][:N]
        dist = np.linalg.norm(sp[:, None, 1:3] - xy[None], axis=2).min(1)
        assert (dist < 8.0).all()                                                  # spawned within the road's half width + lateral offset
        dirs.add(ep["cw"])
    assert dirs == {True, False}                                                   # use_random_direction draws both


def test_generator_pool_survives_fork_and_concurrent_callers(lib):
    """the generator's helper threads live for the life of the process.  (a) a fork()ed child inherits the pool object but not
    its threads: a generate call there must still return (it starts a pool of its own); (b) two callers at once — reset() of a batch while the
    refill thread holds the pool — both get helper threads and the same blobs as a serial run."""
    import ctypes
    import os
    import threading
    L = lib.load()
    N, n = 2, 24

    def gen(seed0, threads):
        mt_t = np.zeros((n, lib.MT_WORDS), np.uint32); mt_g = np.zeros((n, lib.MT_WORDS), np.uint32)
        for e in range(n):
            L.mcr_mt_seed(lib.ptr(mt_t[e]), ctypes.c_uint32(seed0 + e)); L.mcr_mt_seed(lib.ptr(mt_g[e]), ctypes.c_uint32(seed0 + e + 77777))
        blobs = np.zeros((n, lib.episode_bytes()), np.uint8); info = np.zeros((n, 12), np.int32)
        assert L.mcr_episodes_generate(lib.ptr(mt_t), lib.ptr(mt_g), n, N, 2, lib.ptr(blobs), lib.ptr(info), threads) == 0
        return blobs
    ref_a, ref_b = gen(300, 1), gen(900, 1)
    assert np.array_equal(gen(300, 4), ref_a)                       # the pool exists now (helper threads started)
    # (b) concurrent callers
    out = {}
    ts = [threading.Thread(target=lambda k=k, s=s: out.__setitem__(k, gen(s, 4))) for k, s in (("a", 300), ("b", 900))]
    for t in ts: t.start()
    for t in ts: t.join(60)
    assert not any(t.is_alive() for t in ts)
    assert np.array_equal(out["a"], ref_a) and np.array_equal(out["b"], ref_b)
    # (a) fork: the child must not wait for threads it does not have
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:
        ok = b"0"
        try:
            ok = b"1" if np.array_equal(gen(300, 4), ref_a) else b"0"
        finally:
            os.write(w, ok); os._exit(0)
    os.close(w)
    import select
    ready, _, _ = select.select([r], [], [], 60)
    if not ready:
        os.kill(pid, 9); os.waitpid(pid, 0)
        raise AssertionError("mcr_episodes_generate hung in a forked child")
    assert os.read(r, 1) == b"1"
    os.waitpid(pid, 0)


def _product_blob(L, lib, N, seed, g=0):
    """the product's episode of global env index g (vec_env.py: track stream seed + g, draw stream seed + g + 2**31) — what tests.util.oracle_episode
    builds for the oracle"""
    s = (seed + g) % 2 ** 32
    mt_t = _mt(L, lib, s).reshape(1, -1); mt_g = _mt(L, lib, (s + 2 ** 31) % 2 ** 32).reshape(1, -1)
    blob = np.zeros((1, lib.episode_bytes()), np.uint8); info = np.zeros((1, 12), np.int32)
    assert L.mcr_episodes_generate(lib.ptr(mt_t), lib.ptr(mt_g), 1, N, 2, lib.ptr(blob), lib.ptr(info), 1) == 0
    return blob[0]


def test_world_reuse_tree_ids_of_the_product_host_tree(lib, oracle):
    """The product's own b2DynamicTree (csrc/mcr_world.cpp, include/mcr.h: mcr_world_*) — what the facade carries across reset() the way the
    reference carries its b2World (multi_car_racing.py:138, 341) — against the oracle's literal tree (world mode 1): fed the same episodes and
    the same body transforms step by step, the two hand out the same proxy ids in the first episode (ascending, 2k - 1), in the second (off the
    free list: no longer ascending) and in the third — which they only do if every MoveProxy in between restructured both trees alike."""
    from tests.util import oracle_episode
    L = lib.load()
    for N in (2, 4):
        w = ctypes.c_void_p(L.mcr_world_create(N))
        o = oracle.OracleEnv(N); o.set_world_mode(1)
        rng = np.random.RandomState(3)
        not_ascending = 0
        for ei, (seed, steps) in enumerate([(31, 160), (77, 120), (5, 200), (12, 10)]):
            blob = _product_blob(L, lib, N, seed)
            ep = oracle_episode(oracle, N, seed, 0, use_random_direction=True)
            o.reset(ep, render=False)
            assert L.mcr_world_reset(w, lib.ptr(blob)) == 0
            assert L.mcr_world_step(w, lib.ptr(np.ascontiguousarray(o.state()["bodies"], np.float32))) == 0      # the reset's own step (:408)
            tid, fid = o.proxy_ids()
            want = np.concatenate([tid, fid.ravel()]).astype(np.int32)
            got = np.zeros(len(want) + 8, np.int32)
            n = L.mcr_world_proxy_ids(w, lib.ptr(got), len(got))
            assert n == len(want) and np.array_equal(got[:n], want), f"N={N} episode {ei}: ids differ at {np.nonzero(got[:n] != want)[0][:8]}"
            hdr = blob[:16].view(np.int32)
            assert hdr[3] == 1, "the blob says that it carries proxy-id tables"
            not_ascending += int(not np.all(np.diff(want) > 0))
            for k in range(steps):
                act = np.stack([rng.uniform(-0.4, 0.4, N), np.ones(N), np.zeros(N)], -1).astype(np.float32)
                o.step(act, render=False)
                assert L.mcr_world_step(w, lib.ptr(np.ascontiguousarray(o.state()["bodies"], np.float32))) == 0
        assert not_ascending >= 2, "the later episodes' ids come off the free list"
        o.close(); L.mcr_world_destroy(w)
