"""CPU: what can be pinned about the oracle's third-party restatements without Box2D / gym / pyglet.

* b2TestOverlap: Box2D's actual algorithm (GJK b2Distance, restated in the oracle — and, since round 3, what the step of
  both the oracle and the kernels uses) against an independent exact-distance predicate (SAT + vertex-edge), >= 1e7 random
  wheel/tile poses at the 0.02 threshold -> the width of the zone in which the two can differ, which sizes the SAT
  far-field filter in front of the device GJK (k_collide.h: 1e-3, i.e. 20x that width);
* closed-form solver KATs: momentum conservation of the joint/contact solver, a revolute joint driven into its limit,
  a point-symmetric head-on collision;
* the raster: an independent scanline rasteriser (f64 edge intersections, no half-plane tests) re-draws the oracle's
  polygon list; the two may differ only inside the oracle's ambiguity mask."""
import numpy as np
import pytest

from tests.util import oracle_episode


# ----------------------------------------------------------------------------- b2TestOverlap: GJK vs SAT
def test_gjk_vs_sat_overlap_at_the_threshold(oracle):
    """1e7 poses with the exact (f64) core separation drawn from 0.02 +- 1e-5, i.e. inside the f32 noise of 250-unit world
    coordinates (1 ulp = 1.5e-5).  GJK (what Box2D and the step use) and an exact-distance predicate in f32 decide
    differently in ~19 % of those poses — both are at the mercy of the same rounding noise there — and never beyond
    |delta| = 5e-5 (0 of 4e6 at band 1e-3): the zone in which ANY sound f32 predicate can differ from GJK has an effective
    half-width of 2.4e-6 units.  That is what makes the kernel's SAT far-field filter (apart beyond 0.02 + 1e-3, touching
    when the cores intersect, GJK in between) exact.  GJK itself never needed more than 5 of its 20 iterations."""
    r = oracle.overlap_sweep(10_000_000, seed=1, band=1e-5, far=5e-5)
    assert r["samples"] == 10_000_000
    rate = r["disagree"] / r["samples"]
    assert 0.10 < rate < 0.30, r
    assert r["max_gjk_iters"] <= 20 and r["disagree_far"] == 0
    assert abs(r["gjk_touching"] - r["sat_touching"]) < 0.02 * r["samples"]      # neither is biased towards touching
    wide = oracle.overlap_sweep(4_000_000, seed=2, band=1e-3, far=5e-5)
    assert wide["disagree_far"] == 0, wide                                        # agree whenever |separation - 0.02| > 5e-5
    half_width = wide["disagree"] / wide["samples"] * 1e-3
    assert half_width < 5e-6, half_width


# ----------------------------------------------------------------------------- solver KATs
def _env(oracle, N=2, seed=3, contacts=True):
    o = oracle.OracleEnv(N, car_contacts=contacts)
    o.reset(oracle_episode(oracle, N, seed, 0), render=False)
    return o


def _momentum(oracle, st):
    m = oracle.mass_props()
    mass = np.array([1 / m[0]] + [1 / m[4]] * 4); inertia = np.array([1 / m[1]] + [1 / m[5]] * 4)
    b = st["bodies"].astype(np.float64)                                  # [N,5,6] c.x c.y a v.x v.y w
    p = (mass[None, :, None] * b[:, :, 3:5]).sum((0, 1))
    L = (mass[None, :] * (b[:, :, 0] * b[:, :, 4] - b[:, :, 1] * b[:, :, 3]) + inertia[None, :] * b[:, :, 5]).sum()
    return p, L


def test_joint_solver_conserves_momentum(oracle):
    """Closed system (b2World::Step without Car.step: revolute joints only act between hull and wheels): linear and
    angular momentum of every car are invariants of the exact solver; in f32 they drift by rounding only."""
    o = _env(oracle, N=1, contacts=False)
    st = o.state()["bodies"].copy()
    st[0, :, 3] = 7.5; st[0, :, 4] = -3.25                                # common translation ...
    st[0, 0, 5] = 1.7                                                     # ... hull spinning, wheels not: the joints have to work
    st[0, 1:, 3] += np.array([0.4, -0.3, 0.2, -0.1])                      # and the wheels drifting apart a little
    for k in range(5):
        o.set_body(0, k, st[0, k])
    p0, L0 = _momentum(oracle, o.state())
    for _ in range(100):
        o.solve_only(1)
    p1, L1 = _momentum(oracle, o.state())
    assert np.abs(p1 - p0).max() < 2e-4 * np.abs(p0).max(), (p0, p1)
    assert abs(L1 - L0) < 2e-4 * abs(L0), (L0, L1)
    b = o.state()["bodies"][0].astype(np.float64)
    # the joints made one rigid assembly of it: every wheel centre moves with the hull's velocity field v + w x r
    r = b[1:, :2] - b[0, :2]
    v_rigid = b[0, 3:5][None, :] + b[0, 5] * np.stack([-r[:, 1], r[:, 0]], 1)
    assert np.abs(b[1:, 3:5] - v_rigid).max() < 0.2, np.abs(b[1:, 3:5] - v_rigid).max()   # initial spread 0.7; residual of the soft position correction
    o.close()


def test_revolute_joint_limit(oracle):
    """Full steering lock: the front-wheel joints run into their +-0.4 rad limits (b2RevoluteJoint limit branch, Solve33):
    limitState at-lower / at-upper, the joint angle stays within the limit + angular slop, and the accumulated limit
    impulse has the sign Box2D clamps it to (>= 0 at the lower limit, <= 0 at the upper)."""
    slop = 2.0 / 180.0 * np.pi
    for steer, want in ((1.0, None), (-1.0, None)):
        o = _env(oracle, N=1, contacts=False)
        a = np.array([[steer, 0.0, 0.0]], np.float32)
        seen = set()
        for k in range(80):
            o.step(a, render=False)
            st = o.state()
            ang = st["bodies"][0, 1:3, 2] - st["bodies"][0, 0, 2]          # front wheels' joint angles
            assert np.all(np.abs(ang) <= 0.4 + slop + 1e-4), (k, ang)
            for w in range(2):
                ls = int(st["limit"][0, w]); seen.add(ls)
                if ls == 1: assert st["joints"][0, w, 2] >= 0.0
                if ls == 2: assert st["joints"][0, w, 2] <= 0.0
        assert (1 in seen) or (2 in seen), "steering lock never reached a joint limit"
        ang = o.state()["bodies"][0, 1:3, 2] - o.state()["bodies"][0, 0, 2]
        assert np.all(np.abs(np.abs(ang) - 0.4) < slop + 1e-3), ang         # parked at the limit
        assert np.sign(ang[0]) == -np.sign(steer)                          # Car.steer(-action[0]) (:422): positive action turns the wheels clockwise
        o.close()


def test_point_symmetric_head_on_collision(oracle):
    """Two cars that are each other's image under a rotation by pi about a point, driving at each other on grass with no
    input: the set-up, the tyre model and the exact contact solver are symmetric, so the cars stay mirror images through
    the collision (b2CollidePolygons + block solver with equal impulses) up to f32 rounding, and the total momentum,
    zero by symmetry, stays zero."""
    o = _env(oracle, N=2, contacts=True)
    st = o.state()["bodies"].copy()
    mid = np.array([250.0, 250.0], np.float32)                            # inside the playfield, away from any track tile
    a0 = np.float32(0.3)
    fwd = np.array([-np.sin(a0), np.cos(a0)], np.float32)
    base = st[0, 0, :2].copy()
    for k in range(5):                                                    # car 0: 4.5 units before the midpoint (noses 3.8 apart), driving at it
        rel = st[0, k, :2] - base
        ca, sa = np.cos(a0 - st[0, 0, 2]), np.sin(a0 - st[0, 0, 2])
        rel = np.array([ca * rel[0] - sa * rel[1], sa * rel[0] + ca * rel[1]], np.float32)
        st[0, k, :2] = mid - 4.5 * fwd + rel
        st[0, k, 2] = st[0, k, 2] + (a0 - o.state()["bodies"][0, 0, 2]) if k else a0
        st[0, k, 3:5] = 15.0 * fwd; st[0, k, 5] = 0.0
    for k in range(1, 5):
        st[0, k, 2] = a0
    for k in range(5):                                                    # car 1: the image of car 0 under rotation by pi about mid
        st[1, k, :2] = 2 * mid - st[0, k, :2]
        st[1, k, 2] = st[0, k, 2] + np.float32(np.pi)
        st[1, k, 3:5] = -st[0, k, 3:5]; st[1, k, 5] = st[0, k, 5]
    for c in range(2):
        for k in range(5):
            o.set_body(c, k, st[c, k])
    act = np.zeros((2, 3), np.float32)
    touched = 0
    for k in range(45):
        o.step(act, render=False)
        touched += o.num_car_contacts()
        b = o.state()["bodies"].astype(np.float64)
        tol = 2e-3 if touched == 0 else 5e-2                              # rounding is amplified by the stiff contact
        assert np.abs(b[1, :, :2] - (2 * mid - b[0, :, :2])).max() < tol, (k, touched)
        assert np.abs(b[1, :, 3:5] + b[0, :, 3:5]).max() < 10 * tol, (k, touched)
        p, _ = _momentum(oracle, o.state())
        assert np.abs(p).max() < 0.05, (k, p)                             # of ~86 per car before the impact
    assert touched > 0, "the cars never touched"
    v0 = o.state()["bodies"][0, 0, 3:5]
    assert float(np.dot(v0, fwd)) < 3.0                                   # and the collision took the approach speed away
    o.close()


# ----------------------------------------------------------------------------- raster: independent scanline evaluation
def _scanline(polys, W, H):
    """Painter's algorithm by scanline intersection: for every pixel row, the x-interval in which the row's centre line
    cuts the (convex) polygon; pixels whose centres lie inside the interval are painted.  No half-plane evaluation."""
    img = np.zeros((H, W, 3), np.uint8)
    for xy, rgb in polys:
        n = len(xy)
        if n < 3:
            continue
        ys = xy[:, 1]
        y0 = max(int(np.floor(ys.min() - 0.5)), 0); y1 = min(int(np.ceil(ys.max() - 0.5)), H - 1)
        for y in range(y0, y1 + 1):
            cy = y + 0.5
            xs = []
            for i in range(n):
                (xa, ya), (xb, yb) = xy[i], xy[(i + 1) % n]
                if ya == yb:
                    continue
                if (ya <= cy < yb) or (yb <= cy < ya):
                    xs.append(xa + (cy - ya) * (xb - xa) / (yb - ya))
            if len(xs) < 2:
                continue
            xl, xr = min(xs), max(xs)
            i0 = max(int(np.ceil(xl - 0.5)), 0); i1 = min(int(np.floor(xr - 0.5)), W - 1)
            if i0 <= i1:
                img[H - 1 - y, i0:i1 + 1] = rgb                         # rows top-down like the observation (arr[::-1], :602)
    return img


@pytest.mark.parametrize("N,seed", [(2, 11), (3, 5)])
def test_oracle_raster_vs_independent_scanline(oracle, N, seed):
    """The oracle's raster decides coverage by signed distances to every edge; the scanline rasteriser above decides it
    by where each pixel row cuts the polygon.  Same polygon list, same sampling rule -> the two frames may differ only
    where the oracle itself says a pixel centre is within 0.02 px of an edge (its ambiguity mask).  The score label
    (a bitmap, not a polygon) is outside this comparison."""
    o = _env(oracle, N=N, seed=seed)
    rng = np.random.RandomState(seed)
    label = np.zeros((96, 96), bool); label[86:94, 0:18] = True
    checked = 0
    for k in range(75):
        a = np.stack([rng.uniform(-1, 1, N), rng.uniform(0, 1, N), rng.uniform(0, 0.2, N)], 1).astype(np.float32)
        if k > 50:
            a[:, 0] = 1.0                                                 # spin: backwards flag comes up
        o.step(a, render=False)
        if k in (0, 3, 12, 30, 49, 74):                                   # zoomed out ... zoomed in, flag
            obs, amb = o.render_with_mask()
            for ag in range(N):
                img = _scanline(o.draw_list(ag), 96, 96)
                d = (img != obs[ag]).any(-1) & ~label
                assert (d & (amb[ag] == 0)).sum() == 0, f"step {k} agent {ag}: {(d & (amb[ag] == 0)).sum()} unambiguous pixels differ"
                assert d.sum() <= 40
                checked += 1
    assert checked == 6 * N
    o.close()


# ----------------------------------------------------------------------------- Box2D's callback order (broadphase model)
def test_spawn_step_serves_the_car_created_last(oracle):
    """reset() -> step(None): every tile<->wheel contact is made by ONE FindNewContacts (e_newFixture), so Collide serves them
    in descending (tile, car, wheel) proxy order: on the tiles the cars of a start row share, the car created LAST is the first
    visitor (full 1000/T), the others get the damped shares in descending car order (multi_car_racing.py:113-120)."""
    for N, seed in ((2, 7), (4, 8), (3, 9)):
        ep = oracle_episode(oracle, N, seed, 0)
        o = oracle.OracleEnv(N); o.reset(ep, render=False)
        es = o.env_state()
        T = len(ep["track"])
        order = ep["car_order"]                                            # car_id -> grid slot; slots 2r, 2r+1 share row r
        for row in range((N + 1) // 2):
            cars = sorted(c for c in range(N) if order[c] // 2 == row)
            if len(cars) < 2:
                continue
            lo, hi = cars
            assert es["tile_visited_count"][lo] == es["tile_visited_count"][hi] > 0
            n = es["tile_visited_count"][hi]
            # rows are 5 tiles apart: no tile is shared across rows at the spawn step, so the row's two cars split each tile
            # (first visitor 1, second 1 - 1/N)
            assert abs(es["reward"][hi] - n * 1000.0 / T) < 1e-9 and abs(es["reward"][lo] - n * (1 - 1 / N) * 1000.0 / T) < 1e-9, (N, row, es["reward"])
        o.close()


def test_event_order_is_a_function_of_the_fat_aabbs(oracle):
    """The legacy order (rounds 1-2: tile^, car^, wheel^) and Box2D's differ in who gets the first-visitor share, never in
    WHICH tiles are visited or when: counts, visited sets and done flags agree step by step."""
    N = 2
    ep = oracle_episode(oracle, N, 21, 0)
    a = oracle.OracleEnv(N); b = oracle.OracleEnv(N)
    b.L.orc_set_event_order(b.h, 1)
    a.reset(ep, render=False); b.reset(ep, render=False)
    ra, rb = a.env_state()["reward"], b.env_state()["reward"]
    assert ra[1] > ra[0] and rb[0] > rb[1] and abs(ra.sum() - rb.sum()) < 1e-9
    rng = np.random.RandomState(0)
    for k in range(150):
        act = np.zeros((N, 3), np.float32); act[:, 1] = 0.5; act[:, 0] = rng.uniform(-0.1, 0.1)
        _, r1, d1, _ = a.step(act, render=False); _, r2, d2, _ = b.step(act, render=False)
        assert d1 == d2 and abs(r1.sum() - r2.sum()) < 1e-9
        ea, eb = a.env_state(), b.env_state()
        assert np.array_equal(ea["tile_visited_count"], eb["tile_visited_count"]) and np.array_equal(ea["visited"], eb["visited"])
        assert np.array_equal(a.state()["bodies"], b.state()["bodies"])
    a.close(); b.close()


# ----------------------------------------------------------------------------- the raster's hard-wired backwards flag
def test_hud_flag_masks_follow_from_the_triangle():
    """k_view.h composes the HUD rows analytically; the backwards flag (mcr.py:669-674: triangle (W-100,30) (W-75,70) (W-50,30) in
    window units) is a table of byte masks there.  Re-derive the table from the vertices (pixel centres, closed edges) and check
    that no centre is near an edge — the table is then what any correct rasteriser draws."""
    import json
    import os
    import re
    kx, ky = 96 / 1000.0, 96 / 800.0
    # the triangle as the REFERENCE draws it: pyglet.graphics.draw's 'v2i' data recorded from render_indicators (render_stream.json)
    cases = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "render_stream.json")))
    flags = [v["flag"] for c in cases for v in c["views"] if v["flag"] is not None]
    assert flags and all(f == flags[0] for f in flags) and flags[0]["data"][1] == ["c3B", [0, 0, 255] * 3]
    V = np.array(flags[0]["data"][0][1], dtype=np.float64).reshape(3, 2) * [kx, ky]
    assert V.tolist() == [[900 * kx, 30 * ky], [925 * kx, 70 * ky], [950 * kx, 30 * ky]]
    cover, margin = {}, 1e9
    for row in range(12):
        for x in range(96):
            c = np.array([x + 0.5, row + 0.5])
            d = []
            for i in range(3):
                a, b = V[i], V[(i + 1) % 3]
                e = b - a
                d.append(-(e[0] * (c[1] - a[1]) - e[1] * (c[0] - a[0])) / np.hypot(*e))      # + inside (the triangle is clockwise)
            margin = min(margin, min(abs(t) for t in d))
            if min(d) >= 0:
                cover.setdefault(row, []).append(x)
    assert cover == {4: [87, 88, 89, 90], 5: [87, 88, 89], 6: [88, 89], 7: [88]} and margin > 0.08, (cover, margin)
    # as byte masks of the 4-pixel groups 21 (x 84..87) and 22 (x 88..91), rows 4 + r4
    def mask(row, g):
        return sum(0xff << (8 * (x - 4 * g)) for x in cover.get(row, []) if x // 4 == g)
    g21 = [mask(4 + r, 21) for r in range(4)]; g22 = [mask(4 + r, 22) for r in range(4)]
    assert g21 == [0xff000000, 0xff000000, 0, 0] and g22 == [0x00ffffff, 0x0000ffff, 0x0000ffff, 0x000000ff]
    src = open(os.path.join(os.path.dirname(__file__), "..", "multi_car_racing_amd", "csrc", "k_view.h")).read()
    m = re.search(r"hud_flag_mask\(int cg, int r4\) \{(.*?)\n\}", src, re.S)
    assert m and "r4 < 2 ? 0xff000000u : 0u" in m.group(1) and "r4 == 0 ? 0x00ffffffu : r4 <= 2 ? 0x0000ffffu : 0x000000ffu" in m.group(1)


# ----------------------------------------------------------------------------- b2DynamicTree: proxy ids across reset()
def test_world_reuse_tree_ids(oracle):
    """The oracle carries a literal b2DynamicTree (world mode 1) to answer what the reference's reuse of ONE b2World across reset()
    (multi_car_racing.py:138, 341) does to proxy ids.  Pins: (a) DESIGN 4's claim for the first episode of a world — the k-th fixture
    created gets leaf id 2k-1 (the first one 0) —, so mode 1 and mode 0 (what the kernels implement) are the same computation there;
    (b) in a second episode the ids are the free list's: still one distinct id per proxy, no longer ascending in creation order;
    (c) visits (which tile, which car, when) do not depend on ids — the order of same-step events does, and (through fixtureA of a car<->car
    contact = the lower id) the manifold of two cars that touch."""
    N = 2
    a, b = oracle.OracleEnv(N, world_mode=0), oracle.OracleEnv(N)
    b.set_world_mode(1)
    ep = oracle_episode(oracle, N, 31, 0, use_random_direction=True)
    a.reset(ep, render=False); b.reset(ep, render=False)
    tid, fid = b.proxy_ids()
    want = np.arange(b.T + N * 8) * 2 - 1; want[0] = 0
    assert np.array_equal(np.concatenate([tid, fid.ravel()]), want)
    assert np.array_equal(a.env_state()["reward"], b.env_state()["reward"])
    rng = np.random.RandomState(3)
    for k in range(160):
        act = np.stack([rng.uniform(-0.4, 0.4, N), np.ones(N), np.zeros(N)], -1).astype(np.float32)
        _, r1, d1, _ = a.step(act, render=False); _, r2, d2, _ = b.step(act, render=False)
        assert np.array_equal(r1, r2) and d1 == d2, k
    assert np.array_equal(a.state()["bodies"], b.state()["bodies"])
    # second episode on the same world
    ep2 = oracle_episode(oracle, N, 77, 0, use_random_direction=True)
    a.reset(ep2, render=False); b.reset(ep2, render=False)
    tid2, fid2 = b.proxy_ids()
    ids = np.concatenate([tid2, fid2.ravel()])
    assert len(set(ids.tolist())) == len(ids) and ids.min() >= 0
    assert not np.all(np.diff(ids) > 0), "ids of a reused world come off the free list: not ascending in creation order"
    touched = False
    same_until_touch = True
    for k in range(120):
        act = np.stack([rng.uniform(-0.4, 0.4, N), np.ones(N), np.zeros(N)], -1).astype(np.float32)
        a.step(act, render=False); b.step(act, render=False)
        touched = touched or b.num_car_contacts() > 0 or a.num_car_contacts() > 0     # (the contacts the step just solved)
        if not touched:
            same_until_touch = same_until_touch and np.array_equal(a.state()["bodies"], b.state()["bodies"])
    sa, sb = a.env_state(), b.env_state()
    assert np.array_equal(sa["visited"], sb["visited"]) and np.array_equal(sa["tile_visited_count"], sb["tile_visited_count"])
    # ids reach the solver in ONE way: fixtureA of a car<->car contact is the fixture with the lower proxy id (b2Contact::Create), and
    # b2CollidePolygons is not symmetric in its arguments (reference-face tie-break) — until two cars touch, the poses are the fresh world's
    assert same_until_touch
    if not touched:
        assert np.array_equal(a.state()["bodies"], b.state()["bodies"])
    a.close(); b.close()


# ----------------------------------------------------------------------------- b2World::Solve's island order
def test_island_dfs_order_in_the_oracle(oracle):
    """The oracle can solve an island's joints and contacts in the order Box2D's depth-first island construction appends them (island
    order 1) instead of the order the build DEFINES (0: contacts ascending, joints 3,2,1,0 — what the kernels implement).  Pins: without
    car<->car contacts the two are the same computation (every car is a seed entered through its wheel 3); with contacts the DFS order
    does differ — contact order and, when a car is entered through another wheel, that car's joint order — and so do the last bits."""
    N = 2
    ep = oracle_episode(oracle, N, 4002, 3, use_random_direction=True)
    a, b = oracle.OracleEnv(N, car_contacts=False), oracle.OracleEnv(N, car_contacts=False)
    a.set_island_order(0); b.set_island_order(1)
    a.reset(ep, render=False); b.reset(ep, render=False)
    rng = np.random.RandomState(0)
    for k in range(80):
        act = np.stack([rng.uniform(-1, 1, N), rng.uniform(0, 1, N), rng.uniform(0, 1, N)], -1).astype(np.float32)
        a.step(act, render=False); b.step(act, render=False)
        assert b.island_diff() == 0
    assert np.array_equal(a.state()["bodies"], b.state()["bodies"])
    a.close(); b.close()
    # rear-end collisions: the car behind floors it
    seen = 0
    for e in range(12):
        ep = oracle_episode(oracle, N, 4002, e, use_random_direction=True)
        a, b = oracle.OracleEnv(N), oracle.OracleEnv(N)
        a.set_island_order(0); b.set_island_order(1)
        a.reset(ep, render=False); b.reset(ep, render=False)
        rng = np.random.RandomState(e)
        for k in range(160):
            act = np.stack([rng.uniform(-0.3, 0.3, N), rng.uniform(0.2, 1.0, N), np.zeros(N)], -1).astype(np.float32)
            act[N // 2:, 1] = 1.0
            a.step(act, render=False); b.step(act, render=False)
            if b.num_car_contacts() > 0:
                seen |= b.island_diff()
            else:
                assert b.island_diff() == 0
        a.close(); b.close()
    assert seen == 3, f"expected both a permuted joint order and a permuted contact order among the contact steps (got {seen})"


def test_squared_slop_threshold_of_the_joint_position_verdict():
    """k_dynamics.h: joint_position compares the SQUARED position error with T = 0x37d1b718 instead of its square root with b2_linearSlop
    (b2RevoluteJoint::SolvePositionConstraints returns positionError <= b2_linearSlop): equivalent because sqrtf is monotonic and correctly
    rounded and T is the largest float whose root does not exceed 0.005f."""
    s = np.float32(0.005)
    T = np.array([0x37d1b718], np.uint32).view(np.float32)[0]
    assert np.sqrt(T, dtype=np.float32) <= s and np.sqrt(np.nextafter(T, np.float32(np.inf)), dtype=np.float32) > s
    x = np.float32(2.5e-05)
    lo = x
    for _ in range(2000): lo = np.nextafter(lo, np.float32(0))
    xs = [lo]
    for _ in range(4000): xs.append(np.nextafter(xs[-1], np.float32(np.inf)))
    xs = np.array(xs, np.float32)
    assert np.array_equal(np.sqrt(xs, dtype=np.float32) <= s, xs <= T)
    rng = np.random.RandomState(0)
    xs = np.abs(rng.standard_normal(200000).astype(np.float32)) * np.float32(1e-4)
    assert np.array_equal(np.sqrt(xs, dtype=np.float32) <= s, xs <= T)
