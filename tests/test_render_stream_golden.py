"""The oracle's frame against the REFERENCE'S OWN render code.

`tests/golden/render_stream.{json,npz}` is the call stream of the reference's `render` / `_render_window` / `render_road` /
`render_indicators` (multi_car_racing.py:511-604, 613-674) run for real under recording stubs (oracle/make_goldens.py:
gen_render_stream) on scripted car states: the `Transform` setters' arguments, the viewport of each mode, the draw order,
every glColor4f / glVertex3f of the two glBegin/glEnd pairs, the label's text and the flag triangle.  This test replays the same
script on the oracle (contact events + hull poses as in bookkeeping.json) and compares what `render_view` hands to its rasteriser
BEFORE any transform with that stream: same primitives, same order, colours, and coordinates to the bit in the precision
each side holds (f64 for the camera, f32 for everything GL receives through a gl*f entry point).

What stays unpinned (third party, absent here): what a gl call DOES — gym's `Transform.enable` (translate·rotate·scale), GL's
sampling rule and float->u8 colour conversion, `Car.draw`'s polygons, pyglet's glyphs.
"""
import json, os
import numpy as np
import pytest

G = os.path.join(os.path.dirname(__file__), "golden")
VIEWPORT = {"state_pixels": (96, 96), "rgb_array": (600, 400), "human": (1000, 800)}       # :573-586
CAR_COLORS = [(0.8, 0.0, 0.0), (0.0, 0.0, 0.8), (0.0, 0.8, 0.0), (0.0, 0.8, 0.8), (0.8, 0.8, 0.8), (0.0, 0.0, 0.0), (0.8, 0.0, 0.8), (0.8, 0.8, 0.0)]


def c8(c):
    """the build's GL colour rule (unpinned): round-to-nearest of the f32 colour * 255"""
    return int(np.floor(np.float64(np.float32(c)) * 255.0 + 0.5))


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="module")
def stream():
    return json.load(open(os.path.join(G, "render_stream.json"))), np.load(os.path.join(G, "render_stream.npz"))


def replay(O, case):
    N = case["N"]
    np.random.seed(case["global_seed"])
    ep = O.new_episode(N, np.random.RandomState(case["track_seed"]), np.random, direction=case["direction"], use_random_direction=False)
    assert len(ep["track"]) == case["T"]
    env = O.OracleEnv(N, h_ratio=case["h_ratio"], backwards_flag=case["backwards_flag"], use_ego_color=case["use_ego_color"])
    env.reset_nostep(ep)
    for st in case["script"]:
        for begin, c, w, tidx in st["events"]:
            env.contact_event(begin, c, w, tidx)
        for c, (px, py, vx, vy, ang) in enumerate(st["poses"]):
            env.set_hull_pose(c, px, py, vx, vy, ang)
        env.bookkeeping(True)
    for c, cs in enumerate(case["car_state"]):
        env.set_render_state(c, cs["hull"][5], cs["wheel_angle"], cs["omega"])
    env.set_time(case["t"])
    return env, ep


def test_fixture_covers_what_the_verdict_asked(stream):
    cases, _ = stream
    assert len(cases) >= 12
    assert {c["t"] for c in cases} >= {0.02, 0.5, 1.0, 3.0}
    assert {c["mode"] for c in cases} == {"state_pixels", "rgb_array", "human"}
    assert {c["h_ratio"] for c in cases} >= {0.25, 0.4}
    assert any(c["use_ego_color"] for c in cases) and any(not c["backwards_flag"] for c in cases)
    speeds = [np.hypot(cs["hull"][2], cs["hull"][3]) for c in cases for cs in c["car_state"]]
    assert min(speeds) < 0.5 < max(speeds) and any(s == 0.5 for s in speeds)
    assert any(q["v"][0][1] < 20.0 for c in cases for v in c["views"] for q in v["hud"][1:6])       # a negative vertical gauge
    assert any(q["v"][1][0] < q["v"][0][0] for c in cases for v in c["views"] for q in v["hud"][6:8])  # a negative horizontal gauge
    assert any(len(c["touched"]) > c["T"] // 3 + 5 for c in cases)


def test_bookkeeping_state_behind_the_frames(oracle, stream):
    """rewards, flags and touched tiles the frames show are the reference's own (the replay is the bookkeeping golden's mechanism)"""
    cases, _ = stream
    for ci, case in enumerate(cases):
        env, _ = replay(oracle, case)
        es = env.env_state()
        assert es["reward"].tolist() == case["reward"], ci
        assert es["driving_backward"].astype(bool).tolist() == case["driving_backward"], ci
        shown = np.flatnonzero(es["touched"].astype(bool) | (np.arange(case["T"]) % 3 == 0)).tolist()
        assert shown == case["touched"], ci
        env.close()


def test_camera_viewport_and_order(oracle, stream):
    """:540-556 zoom / rotation / translation arguments bit for bit (f64); :573-586 viewport per mode; draw order of :559-593"""
    cases, _ = stream
    for ci, case in enumerate(cases):
        env, _ = replay(oracle, case)
        W, H = VIEWPORT[case["mode"]]
        N = case["N"]
        for a, v in enumerate(case["views"]):
            cam, prims = env.render_stream(a, W, H, particles=case["mode"] != "state_pixels")
            assert v["viewport"] == [0, 0, W, H], (ci, a)
            assert v["scale"] == [cam["zoom"], cam["zoom"]], (ci, a, v["scale"], cam["zoom"])
            assert v["rotation"] == cam["angle"], (ci, a, v["rotation"], cam["angle"])
            assert v["translation"] == [cam["tx"], cam["ty"]], (ci, a, v["translation"], cam)
            # order: the Transform setters, Car.draw per car in self.cars order (queues geoms), clear, viewport, enable, road,
            # the queued car geoms, disable, indicators, label, [flag], read-back / flip
            tail = ["flip"] if case["mode"] == "human" else ["readback"]
            flag = ["graphics_draw"] if v["flag"] is not None else []
            assert v["ops"] == (["set_scale", "set_translation", "set_rotation"] + ["car_draw"] * N + ["switch_to", "dispatch_events", "clear", "viewport", "enable",
                                "quads:%d" % (1 + 400 + case["n_road_poly"])] + ["car_geoms:%d" % c for c in range(N)] + ["disable", "quads:8", "label_draw"] + flag + tail), (ci, a)
            # the oracle's order of the same: road block, cars ascending, indicators, flag
            tags = [p[1] for p in prims]
            blocks = [t for i, t in enumerate(tags) if i == 0 or t != tags[i - 1]]
            assert blocks == [0] + [1 + c for c in range(N)] + [-1] + ([-2] if v["flag"] is not None else []), (ci, a, blocks)
            assert all(p[0] == (0 if p[1] >= 0 else 1) for p in prims)          # world space under the transform, window space after disable()
            assert [d["draw_particles"] for d in v["car_draws"]] == [case["mode"] != "state_pixels"] * N        # :564
            assert [d["viewer"] for d in v["car_draws"]] == [a] * N
            if case["mode"] != "human":
                assert v["readback"] == [W, H] and case["frames_shape"] == [N, H, W, 3]                            # :599-604
        env.close()


def test_render_road_stream(oracle, stream):
    """:613-632 — playfield quad, the 400 grass squares, road_poly in creation order with the touched tiles' colour: every vertex
    as the f32 glVertex3f receives, every colour through the build's u8 rule"""
    cases, arrays = stream
    for ci, case in enumerate(cases):
        env, _ = replay(oracle, case)
        W, H = VIEWPORT[case["mode"]]
        gold = arrays["c%d_road" % ci]                          # [n, 4 + 12] f32: rgba, 4 x (x, y, z)
        assert gold.shape == (1 + 400 + case["n_road_poly"], 16)
        assert (gold[:, 3] == 1.0).all() and (gold[:, 6::3] == 0.0).all()        # alpha 1, z 0
        gxy = gold[:, 4:].reshape(-1, 4, 3)[:, :, :2]
        gcol = np.array([[c8(c) for c in row[:3]] for row in gold])
        for a in range(case["N"]):
            _, prims = env.render_stream(a, W, H)
            road = [p for p in prims if p[1] == 0]
            assert len(road) == len(gold), (ci, a)
            oxy = np.array([p[2] for p in road])
            assert oxy.shape == gxy.shape
            assert (oxy.astype(np.float32) == oxy).all()                          # the oracle's vertices ARE f32 values
            assert (oxy.astype(np.float32) == gxy).all(), (ci, a, np.argwhere(oxy.astype(np.float32) != gxy)[:4])
            ocol = np.array([p[3] for p in road])
            assert (ocol == gcol).all(), (ci, a, np.argwhere(ocol != gcol)[:4])
        env.close()


def test_hull_colours_and_ego_mutation(oracle, stream):
    """:559-564 — hull colours at each Car.draw call: CAR_COLORS[car_id] (:402), or ego red / others blue rewritten per view; the
    mutation persists after render() (hull_colors_after = what the last view left)"""
    cases, _ = stream
    for ci, case in enumerate(cases):
        env, _ = replay(oracle, case)
        W, H = VIEWPORT[case["mode"]]
        N = case["N"]
        for a, v in enumerate(case["views"]):
            _, prims = env.render_stream(a, W, H)
            for d in v["car_draws"]:
                want = [c8(x) for x in d["hull_color"]]
                hull = [p for p in prims if p[1] == 1 + d["car"]][-4:]            # Car.draw: wheels first, the 4 hull fixtures last
                assert all(p[3].tolist() == want for p in hull), (ci, a, d)
                if not case["use_ego_color"]:
                    assert tuple(d["hull_color"]) == CAR_COLORS[d["car"] % 8]
                else:
                    assert tuple(d["hull_color"]) == ((0.8, 0.0, 0.0) if d["car"] == a else (0.0, 0.0, 0.8))
        if case["use_ego_color"]:
            assert case["hull_colors_after"] == [d["hull_color"] for d in case["views"][-1]["car_draws"]]
        env.close()


def test_render_indicators_stream(oracle, stream):
    """:634-674 — the black bar, five vertical and two horizontal gauges (f32 of the window coordinates the reference computes in
    f64), the label's text and constructor arguments, the flag triangle"""
    cases, _ = stream
    for ci, case in enumerate(cases):
        env, _ = replay(oracle, case)
        W, H = VIEWPORT[case["mode"]]
        for a, v in enumerate(case["views"]):
            cam, prims = env.render_stream(a, W, H)
            hud = [p for p in prims if p[1] == -1]
            assert len(hud) == len(v["hud"]) == 8, (ci, a)
            for k, (p, q) in enumerate(zip(hud, v["hud"])):
                g = np.array(q["v"])
                assert (p[2] == g).all(), (ci, a, k, p[2], g)                    # f64 bit for bit (stronger than the f32 GL holds)
                assert p[3].tolist() == [c8(c) for c in q["color"][:3]] and q["color"][3] == 1.0, (ci, a, k)
            lab = v["label"]
            assert lab["text"] == cam["label"], (ci, a, lab["text"], cam["label"])
            assert (lab["x"], lab["y"], lab["font_size"], lab["anchor_x"], lab["anchor_y"], lab["color"]) == (20, 50.0, 36, "left", "center", [255, 255, 255, 255])
            assert (v["flag"] is not None) == cam["flag"] == (case["driving_backward"][a] and case["backwards_flag"]), (ci, a)
            if v["flag"] is not None:
                assert v["flag"]["count"] == 3 and v["flag"]["mode"] == 4          # GL_TRIANGLES
                assert v["flag"]["data"] == [["v2i", [900, 30, 925, 70, 950, 30]], ["c3B", [0, 0, 255] * 3]]
                tri = [p for p in prims if p[1] == -2]
                assert len(tri) == 1 and tri[0][2].tolist() == [[900, 30], [925, 70], [950, 30]] and tri[0][3].tolist() == [0, 0, 255]
        env.close()
