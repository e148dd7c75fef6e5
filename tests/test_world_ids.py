"""CPU: the rule csrc/k_world.h implements on the device — an env's ONE b2World (multi_car_racing.py:138, 173-181, 341) is a stack of free
LEAF ids, no tree — against two literal b2DynamicTrees by the same author but different code: the oracle's (mcr_oracle.cpp: DynTree, world
mode 1, advanced by real MoveProxy calls of driving cars) and the product library's host twin (csrc/mcr_world.cpp).  Consecutive episodes
on tracks that shrink and grow (the free list's leftovers come into play), N = 2 / 4 / 8."""
import ctypes

import numpy as np
import pytest

from tests.util import oracle_episode


class LeafStack:
    """k_world.h: mcr_world_reissue_ids, restated in Python (the same three copies, one proxy at a time)"""

    def __init__(self, N):
        self.N, self.stk, self.fresh, self.tile, self.fix = N, [], 0, [], None

    def reset(self, T):
        # _destroy (:173-181): tiles in road order; per car b2World::DestroyBody walks the hull's fixture list from its head (polygons 3..0),
        # then the wheels in order (gym Car.destroy)
        self.stk.extend(self.tile)
        if self.fix is not None:
            for c in range(self.N):
                self.stk.extend(self.fix[c][f] for f in (3, 2, 1, 0, 4, 5, 6, 7))

        def pop():
            if self.stk:
                return self.stk.pop()
            j = self.fresh; self.fresh += 1
            return 0 if j == 0 else 2 * j - 1
        self.tile = [pop() for _ in range(T)]                    # _create_track (:318-327), then the cars by id (:366-406): hull polygons, wheels
        self.fix = [[pop() for _ in range(8)] for _ in range(self.N)]


@pytest.mark.parametrize("N", [2, 4, 8])
def test_leaf_stack_rule_equals_two_literal_trees(oracle, lib, N):
    L = lib.load()
    grew = shrank = 0
    for e in range(2):
        o = oracle.OracleEnv(N, world_mode=1)
        w = ctypes.c_void_p(L.mcr_world_create(N))
        m = LeafStack(N)
        blob = np.zeros(lib.episode_bytes(), np.uint8)
        r = np.random.RandomState(10 * N + e)
        lastT = None
        for epi in range(6):
            s = (4000 + 977 * epi + 31 * N + e) % 2 ** 32
            ep = oracle.new_episode(N, np.random.RandomState(s), np.random.RandomState(s + 1), use_random_direction=True)
            T = len(ep["track"])
            if lastT is not None:
                grew += T > lastT; shrank += T < lastT
            lastT = T
            o.reset(ep, render=False)
            m.reset(T)
            tid, fid = o.proxy_ids()
            assert tid.tolist() == m.tile and fid.tolist() == m.fix, f"N={N} env {e} episode {epi}: the stack rule vs the oracle's literal tree"
            # the product's host twin on the same episode (its own generator: same seeds -> the same track, checked by the golden tests)
            mt = np.zeros(lib.MT_WORDS, np.uint32); L.mcr_mt_seed(lib.ptr(mt), ctypes.c_uint32(s))
            order = np.array([ep["car_order"][i] for i in range(N)], np.int32) if "car_order" in ep else np.arange(N, dtype=np.int32)
            info = np.zeros(4, np.int32)
            assert L.mcr_episode_generate(lib.ptr(mt), N, int(ep["direction"] == "CW"), lib.ptr(order), lib.ptr(blob), lib.ptr(info)) == 0
            assert int(info[0]) == T
            assert L.mcr_world_reset(w, lib.ptr(blob)) == 0
            ids = np.zeros(T + 8 * N, np.int32)
            assert L.mcr_world_proxy_ids(w, lib.ptr(ids), len(ids)) == len(ids)
            assert ids[:T].tolist() == m.tile and ids[T:].reshape(N, 8).tolist() == m.fix, f"N={N} env {e} episode {epi}: the stack rule vs mcr_world.cpp's tree"
            # drive: every MoveProxy of the episode goes through both trees (RemoveLeaf + InsertLeaf with rotations) — and changes no leaf id
            for k in range(int(r.randint(20, 120))):
                a = np.stack([r.uniform(-0.4, 0.4, N), np.ones(N), np.zeros(N)], -1).astype(np.float32)
                o.step(a, render=False)
                bodies = np.ascontiguousarray(o.state()["bodies"], np.float32)
                assert L.mcr_world_step(w, lib.ptr(bodies)) == 0
            tid2, fid2 = o.proxy_ids()
            assert tid2.tolist() == m.tile and fid2.tolist() == m.fix
        o.close(); L.mcr_world_destroy(w)
    assert grew > 1 and shrank > 1, "the episodes never both shrank and grew: the free list's leftovers were not exercised"


def test_first_episode_ids_are_creation_order():
    m = LeafStack(2); m.reset(300)
    ids = m.tile + m.fix[0] + m.fix[1]
    assert ids == [0] + [2 * k - 1 for k in range(1, 316)]
