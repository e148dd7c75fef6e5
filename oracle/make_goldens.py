#!/usr/bin/env python3
"""Generate golden fixtures from the *reference module itself* (test infrastructure).

Runs ONLY in the build container (needs /root/reference, which never travels to
the GPU box).  The reference's third-party imports (Box2D, gym, pyglet, shapely)
are absent from this image, so `sys.modules` is pre-seeded with inert stubs; the
reference code that then executes for real is exactly:

  * `MultiCarRacing._create_track`      multi_car_racing.py:183-338
  * `MultiCarRacing.reset` spawn logic  multi_car_racing.py:340-406
  * `FrictionDetector._contact`         multi_car_racing.py:88-123
  * `MultiCarRacing.step` bookkeeping   multi_car_racing.py:433-507
  * `render` / `_render_window`         multi_car_racing.py:511-604   (under RECORDING stubs, see gen_render_stream)
  * `render_road`, `render_indicators`  multi_car_racing.py:613-674

What the stubs replace (NOT pinned by these goldens): Box2D world/bodies, the gym
`Car` class (incl. `Car.draw`), gym's `rendering.Viewer/Transform`, pyglet's GL
(what a gl* call DOES: model transform, sampling rule, colour conversion, glyphs), shapely.  `shapely.geometry.Point.within(Polygon)` is
stubbed by a strict-interior even-odd point-in-polygon test.

Outputs (data only — inputs and expected outputs, never reference source):
  tests/golden/tracks.npz        per-seed track / road_poly / colours / retries
  tests/golden/spawn.json        car order + spawn poses for N x direction x seeds
  tests/golden/bookkeeping.json  scripted contact + pose traces -> rewards/flags
  tests/golden/render_stream.json + .npz   the call stream of render(mode) on scripted car states: Transform setters,
                                 viewport, draw order, every glColor4f/glVertex3f, label text, flag triangle
"""
import sys, os, json, math, types, importlib

sys.dont_write_bytecode = True
import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


# --------------------------------------------------------------------------- stubs
def _install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Shape:
        def __init__(self, vertices=None, **kw):
            self.vertices = vertices

    class _FixtureDef:
        def __init__(self, shape=None, **kw):
            self.shape = shape

    class _Fixture:
        sensor = False

    class _Body:
        def __init__(self):
            self.userData = None
            self.fixtures = [_Fixture()]

    class contactListener:
        def __init__(self):
            pass

    class b2World:
        """Inert world: records static bodies, runs a scripted hook inside Step
        (emulating Box2D firing contact callbacks from Collide)."""

        def __init__(self, gravity=None, contactListener=None):
            self.listener = contactListener
            self.step_hook = None
            self.n_created = 0

        def CreateStaticBody(self, fixtures=None):
            b = _Body()
            b.vertices = [tuple(v) for v in fixtures.shape.vertices]
            self.n_created += 1
            return b

        def DestroyBody(self, b):
            pass

        def Step(self, dt, vi, pi):
            if self.step_hook is not None:
                self.step_hook(self)

    b2 = mod("Box2D.b2", edgeShape=_Shape, circleShape=_Shape, fixtureDef=_FixtureDef,
             polygonShape=_Shape, revoluteJointDef=object, contactListener=contactListener)
    mod("Box2D", b2World=b2World, b2=b2)

    class Env:
        pass

    class EzPickle:
        def __init__(self, *a, **k):
            pass

    class Box:
        def __init__(self, low=None, high=None, shape=None, dtype=None):
            self.low, self.high, self.shape, self.dtype = low, high, shape, dtype

    class _Seeding:
        @staticmethod
        def np_random(seed=None):
            return np.random.RandomState(0), seed

    class _Hull:
        def __init__(self, angle, x, y):
            self.position = (x, y)
            self.angle = angle
            self.linearVelocity = (0.0, 0.0)
            self.angularVelocity = 0.0
            self.color = None
            self.userData = None

    class _Joint:
        angle = 0.0

    class _Wheel:
        def __init__(self):
            self.tiles = set()
            self.userData = self
            self.omega = 0.0
            self.joint = _Joint()

    class Car:
        created = []

        def __init__(self, world, init_angle, init_x, init_y):
            self.init = (float(init_angle), float(init_x), float(init_y))
            self.hull = _Hull(init_angle, init_x, init_y)
            self.wheels = [_Wheel() for _ in range(4)]
            Car.created.append(self)

        def steer(self, s): pass
        def gas(self, g): pass
        def brake(self, b): pass
        def step(self, dt): pass
        def destroy(self): pass
        draw_hook = None                      # gen_render_stream: records (viewer, draw_particles, hull.color) and queues a marker geom

        def draw(self, viewer, draw_particles=True):
            if Car.draw_hook is not None:
                Car.draw_hook(self, viewer, draw_particles)

    cd = mod("gym.envs.box2d.car_dynamics", SIZE=0.02, WHEEL_W=14,
             WHEELPOS=[(-55, +80), (+55, +80), (-55, -82), (+55, -82)], Car=Car)
    box2d = mod("gym.envs.box2d", car_dynamics=cd)
    reg = mod("gym.envs.registration", register=lambda **k: None)
    envs = mod("gym.envs", box2d=box2d, registration=reg)
    spaces = mod("gym.spaces", Box=Box)
    utils = mod("gym.utils", colorize=lambda s, *a, **k: s, seeding=_Seeding, EzPickle=EzPickle)
    mod("gym", Env=Env, spaces=spaces, utils=utils, envs=envs)
    gl = mod("pyglet.gl")
    mod("pyglet", gl=gl)

    class Polygon:
        def __init__(self, pts):
            self.pts = [(float(x), float(y)) for x, y in pts]

    class Point:
        def __init__(self, xy):
            self.x, self.y = float(xy[0]), float(xy[1])

        def within(self, poly):
            # strict interior, even-odd rule; boundary -> False
            x, y = self.x, self.y
            inside = False
            pts = poly.pts
            n = len(pts)
            for i in range(n):
                x1, y1 = pts[i]
                x2, y2 = pts[(i + 1) % n]
                # on-segment => boundary
                cross = (x2 - x1) * (y - y1) - (y2 - y1) * (x - x1)
                if cross == 0 and min(x1, x2) <= x <= max(x1, x2) and min(y1, y2) <= y <= max(y1, y2):
                    return False
                if (y1 > y) != (y2 > y):
                    xin = x1 + (y - y1) * (x2 - x1) / (y2 - y1)
                    if xin > x:
                        inside = not inside
            return inside

    mod("shapely.geometry", Point=Point, Polygon=Polygon)
    mod("shapely")
    return Car


def load_reference():
    Car = _install_stubs()
    sys.path.insert(0, REF)
    m = importlib.import_module("gym_multi_car_racing.multi_car_racing")
    return m, Car


# --------------------------------------------------------------------------- goldens
def gen_tracks(mcr, seeds):
    out = {}
    for s in seeds:
        env = mcr.MultiCarRacing(num_agents=2, verbose=0, use_random_direction=False)
        env.np_random = np.random.RandomState(s)
        env.road_poly = []
        retries = 0
        while True:
            env.road_poly = []      # reset() does this only once; failed attempts never append (returns before tile loop)
            if env._create_track():
                break
            retries += 1
        track = np.array(env.track, dtype=np.float64)
        verts = np.array([p for p, c in env.road_poly], dtype=np.float64)     # (P,4,2)
        cols = np.array([list(c) for p, c in env.road_poly], dtype=np.float64)  # (P,3)
        # which road_poly entries are tiles (share the colour list object with a body)
        tile_colors = {id(t.color) for t in env.road}
        is_tile = np.array([id(c) in tile_colors for p, c in env.road_poly], dtype=np.uint8)
        out[f"s{s}_track"] = track
        out[f"s{s}_poly"] = verts
        out[f"s{s}_color"] = cols
        out[f"s{s}_is_tile"] = is_tile
        out[f"s{s}_retries"] = np.array(retries)
        out[f"s{s}_start_alpha"] = np.array(env.start_alpha)
        print(f"seed {s}: T={len(track)} P={len(verts)} retries={retries}")
    out["seeds"] = np.array(list(seeds))
    return out


def gen_spawn(mcr, Car):
    cases = []
    for N in (1, 2, 3, 4, 8):
        for direction in ("CCW", "CW"):
            for gseed in (0, 123):
                for tseed in (1, 7):
                    np.random.seed(gseed)
                    env = mcr.MultiCarRacing(num_agents=N, verbose=0, direction=direction,
                                             use_random_direction=False)
                    env.np_random = np.random.RandomState(tseed)
                    env.render = lambda mode="human": np.zeros((N, 96, 96, 3), np.uint8)
                    Car.created.clear()
                    env.reset()
                    cases.append(dict(
                        N=N, direction=direction, global_seed=gseed, track_seed=tseed,
                        car_order=[int(env.car_order[i]) for i in range(N)],
                        T=len(env.track),
                        poses=[list(c.init) for c in env.cars],
                    ))
    # random direction: draws from the global stream before the car order
    for gseed in (0, 1, 2, 3, 4, 5):
        np.random.seed(gseed)
        env = mcr.MultiCarRacing(num_agents=2, verbose=0, use_random_direction=True)
        ctor_dir = str(env.episode_direction)
        env.np_random = np.random.RandomState(5)
        env.render = lambda mode="human": np.zeros((2, 96, 96, 3), np.uint8)
        dirs = []
        orders = []
        for ep in range(3):
            Car.created.clear()
            env.reset()
            dirs.append(str(env.episode_direction))
            orders.append([int(env.car_order[i]) for i in range(2)])
        cases.append(dict(random_direction=True, global_seed=gseed, track_seed=5,
                          ctor_direction=ctor_dir, episode_directions=dirs, car_orders=orders,
                          poses=[list(c.init) for c in env.cars], T=len(env.track)))
    return cases


class _FakeContact:
    class _F:
        def __init__(self, body): self.body = body
    def __init__(self, a, b):
        self.fixtureA = self._F(a); self.fixtureB = self._F(b)


def gen_bookkeeping(mcr, Car):
    """Scripted episodes: per step, a list of contact events fired inside world.Step
    and a list of hull (x,y,vx,vy,angle) per car set before the step."""
    rng = np.random.RandomState(2024)
    episodes = []
    for N, direction, tseed in ((1, "CCW", 3), (2, "CCW", 0), (2, "CW", 1), (4, "CCW", 2), (3, "CW", 4)):
        np.random.seed(11)
        env = mcr.MultiCarRacing(num_agents=N, verbose=0, direction=direction, use_random_direction=False)
        env.np_random = np.random.RandomState(tseed)
        env.render = lambda mode="human": np.zeros((N, 96, 96, 3), np.uint8)
        Car.created.clear()
        env.reset()
        T = len(env.track)
        track = np.array(env.track)
        script = []
        # build a script: cars advance along the track at different rates, sometimes
        # reversed / off-road / out of the playfield at the end.
        steps = 40
        prog = [0.0] * N
        for k in range(steps):
            events = []
            poses = []
            for c in range(N):
                rate = 1.0 + 0.7 * c
                old = int(prog[c])
                prog[c] += rate
                new = int(prog[c])
                sign = -1 if direction == "CW" else 1
                for ti in range(old, new):
                    tidx = (sign * ti) % T
                    w = int(rng.randint(0, 4))
                    events.append([1, c, w, tidx])        # begin
                    if rng.rand() < 0.7:
                        events.append([0, c, w, tidx])    # end
                if rng.rand() < 0.15:                      # hull (userData None) touches a tile
                    events.append([1, c, -1, int(rng.randint(0, T))])
                idx = (sign * new) % T
                a, b, x, y = track[idx]
                mode = rng.randint(0, 5)
                speed = [0.0, 0.3, 5.0, 20.0, 20.0][mode]
                head = b + (math.pi if direction == "CW" else 0.0)
                if mode == 4:
                    head += math.pi  # going backwards, fast
                # velocity along heading: hull forward axis is (-sin a, cos a)
                vx, vy = -math.sin(head) * speed, math.cos(head) * speed
                ang = head if rng.rand() < 0.8 else head + 2.5
                off = [0.0, 2.0, 9.5, -30.0][rng.randint(0, 4)]
                px, py = x + off * math.cos(b), y + off * math.sin(b)
                if k == steps - 1 and c == N - 1:
                    px = 400.0  # out of playfield
                poses.append([float(np.float32(px)), float(np.float32(py)),
                              float(np.float32(vx)), float(np.float32(vy)), float(np.float32(ang))])
            script.append(dict(events=events, poses=poses))
        # run it through the reference
        trace = []
        for k, st in enumerate(script):
            def hook(world, st=st):
                for begin, c, w, tidx in st["events"]:
                    tile = env.road[tidx]
                    obj = env.cars[c].wheels[w] if w >= 0 else env.cars[c].hull
                    ct = _FakeContact(tile, obj) if (k + c) % 2 == 0 else _FakeContact(obj, tile)
                    if begin:
                        env.contactListener_keepref.BeginContact(ct)
                    else:
                        env.contactListener_keepref.EndContact(ct)
            env.world.step_hook = hook
            for c, (px, py, vx, vy, ang) in enumerate(st["poses"]):
                h = env.cars[c].hull
                h.position = (px, py); h.linearVelocity = (vx, vy); h.angle = ang
            for c in range(N):
                for w in range(4):
                    env.cars[c].wheels[w].car_id = c
            obs, r, done, info = env.step(np.zeros((N, 3)))
            trace.append(dict(
                step_reward=[float(v) for v in r], done=bool(done),
                reward=[float(v) for v in env.reward],
                tile_visited_count=[int(v) for v in env.tile_visited_count],
                driving_backward=[bool(v) for v in env.driving_backward],
                driving_on_grass=[bool(v) for v in env.driving_on_grass],
                n_wheel_tiles=[[len(wh.tiles) for wh in car.wheels] for car in env.cars],
                touched=[i for i, t in enumerate(env.road) if t.color == mcr.ROAD_COLOR],
            ))
        episodes.append(dict(N=N, direction=direction, track_seed=tseed, T=T, script=script, trace=trace))
    return episodes


# --------------------------------------------------------------------------- render stream
class _RenderRecorder:
    """RECORDING stubs for what `_render_window` / `render_road` / `render_indicators` call (multi_car_racing.py:520-604, 613-674):
    `pyglet.gl` (glViewport/glBegin/glColor4f/glVertex3f/glEnd), gym's `rendering.Viewer` (window methods, onetime_geoms) and
    `rendering.Transform` (setters, enable/disable), `pyglet.text.Label`, `pyglet.graphics.draw`, `pyglet.image` read-back (a zero
    colour buffer of the viewport's size), `Car.draw`.  Every call lands in `self.log` in call order; nothing is drawn."""
    GL_QUADS, GL_TRIANGLES = 7, 4             # the GL enum values

    def __init__(self, mcr, Car):
        self.log = []
        self.viewport = (0, 0, 0, 0)
        rec = self
        f32 = lambda v: float(np.float32(v))  # gl*f entry points take C floats (ctypes c_float): what GL receives

        gl = mcr.gl                            # the module object the reference bound at import
        gl.GL_QUADS, gl.GL_TRIANGLES = self.GL_QUADS, self.GL_TRIANGLES
        gl.glViewport = lambda x, y, w, h: (setattr(rec, "viewport", (x, y, w, h)), rec.log.append(("viewport", x, y, w, h)))[1]
        gl.glBegin = lambda mode: rec.log.append(("begin", mode))
        gl.glEnd = lambda: rec.log.append(("end",))
        gl.glColor4f = lambda r, g, b, a: rec.log.append(("color", f32(r), f32(g), f32(b), f32(a)))
        gl.glVertex3f = lambda x, y, z: rec.log.append(("vertex", f32(x), f32(y), f32(z), float(x), float(y)))

        class _Context:                        # no `_nscontext` attribute: pixel_scale stays 1 (:580-583)
            pass

        class _Window:
            def __init__(self):
                self.context = _Context()
            def set_caption(self, c): rec.log.append(("set_caption", c))
            def switch_to(self): rec.log.append(("switch_to",))
            def dispatch_events(self): rec.log.append(("dispatch_events",))
            def clear(self): rec.log.append(("clear",))
            def flip(self): rec.log.append(("flip",))

        class Viewer:
            count = 0
            def __init__(self, w, h):
                self.size = (w, h); self.window = _Window(); self.onetime_geoms = []; self.isopen = True
                self.index = Viewer.count; Viewer.count += 1
                rec.log.append(("viewer", w, h))
            def close(self): pass

        class Transform:
            def set_scale(self, x, y): rec.log.append(("set_scale", float(x), float(y)))
            def set_translation(self, x, y): rec.log.append(("set_translation", float(x), float(y)))
            def set_rotation(self, a): rec.log.append(("set_rotation", float(a)))
            def enable(self): rec.log.append(("enable",))
            def disable(self): rec.log.append(("disable",))

        self.Viewer = Viewer
        rendering = types.ModuleType("gym.envs.classic_control.rendering")
        rendering.Viewer, rendering.Transform = Viewer, Transform
        cc = types.ModuleType("gym.envs.classic_control"); cc.rendering = rendering
        sys.modules["gym.envs.classic_control"] = cc
        sys.modules["gym.envs.classic_control.rendering"] = rendering
        sys.modules["gym.envs"].classic_control = cc

        class Label:
            def __init__(self, text, **kw):
                self.text = text; self.kw = kw
                rec.log.append(("label_new", text, dict(kw)))
            def draw(self): rec.log.append(("label_draw", self.text, dict(self.kw)))

        class _ImageData:
            def get_data(self_inner):
                rec.log.append(("readback", rec.viewport[2], rec.viewport[3]))
                return bytes(rec.viewport[2] * rec.viewport[3] * 4)     # pyglet sizes the colour buffer from the viewport
        class _ColorBuffer:
            def get_image_data(self_inner): return _ImageData()
        class _BufferManager:
            def get_color_buffer(self_inner): return _ColorBuffer()

        pg = mcr.pyglet
        pg.text = types.SimpleNamespace(Label=Label)
        pg.graphics = types.SimpleNamespace(draw=lambda n, mode, *data: rec.log.append(("graphics_draw", n, mode, [(d[0], list(d[1])) for d in data])))
        pg.image = types.SimpleNamespace(get_buffer_manager=lambda: _BufferManager())

        class _Marker:                         # what gym's Car.draw would queue: viewer.draw_polygon(...) appends FilledPolygon geoms
            def __init__(self, car): self.car = car
            def render(self): rec.log.append(("car_geoms", self.car))

        def draw_hook(car, viewer, draw_particles):
            idx = rec.env.cars.index(car)
            rec.log.append(("car_draw", idx, viewer.index, bool(draw_particles), [float(c) for c in car.hull.color]))
            viewer.onetime_geoms.append(_Marker(idx))
        Car.draw_hook = staticmethod(draw_hook)
        self.Car = Car

    def uninstall(self):
        self.Car.draw_hook = None


def _parse_view(log):
    """One `_render_window` call's log -> the fixture's record of the view."""
    ops, quads, groups = [], [], []
    cur_col, verts, mode = None, [], None
    out = dict(car_draws=[], hud=[], label=None, flag=None)
    for e in log:
        k = e[0]
        if k == "begin":
            mode = e[1]; verts = []; cur_col = None; quads = []
        elif k == "color":
            cur_col = list(e[1:5])
        elif k == "vertex":
            verts.append((cur_col, e[1:4], e[4:6]))
            if len(verts) == 4:
                assert all(v[0] == verts[0][0] for v in verts), "one colour per quad"
                quads.append(dict(color=verts[0][0], v32=[list(v[1]) for v in verts], v64=[list(v[2]) for v in verts]))
                verts = []
        elif k == "end":
            assert not verts and mode == _RenderRecorder.GL_QUADS
            groups.append(quads); ops.append("quads:%d" % len(quads))
        elif k == "car_draw":
            out["car_draws"].append(dict(car=e[1], viewer=e[2], draw_particles=e[3], hull_color=e[4])); ops.append("car_draw")
        elif k == "car_geoms":
            ops.append("car_geoms:%d" % e[1])
        elif k == "viewport":
            out["viewport"] = list(e[1:5]); ops.append("viewport")
        elif k == "set_scale":
            out["scale"] = list(e[1:3]); ops.append(k)
        elif k == "set_translation":
            out["translation"] = list(e[1:3]); ops.append(k)
        elif k == "set_rotation":
            out["rotation"] = e[1]; ops.append(k)
        elif k == "label_draw":
            out["label"] = dict(text=e[1], **{kk: (list(vv) if isinstance(vv, tuple) else vv) for kk, vv in e[2].items()}); ops.append("label_draw")
        elif k == "graphics_draw":
            out["flag"] = dict(count=e[1], mode=e[2], data=e[3]); ops.append("graphics_draw")
        elif k == "readback":
            out["readback"] = list(e[1:3]); ops.append("readback")
        elif k in ("viewer", "label_new", "set_caption"):
            ops.append(k)                      # first use of a viewer (:527-537)
        else:
            ops.append(k)
    assert len(groups) == 2, "render_road's and render_indicators' glBegin/glEnd pairs"
    out["ops"] = ops
    road, hud = groups
    out["hud"] = [dict(color=q["color"], v=q["v64"]) for q in hud]          # window coordinates as computed (f64); GL holds their f32
    road_arr = np.array([q["color"] + sum(q["v32"], []) for q in road], dtype=np.float32)   # [n, 4 + 12] what GL receives
    return out, road_arr


def gen_render_stream(mcr, Car):
    """render(mode) of the reference on scripted car states.  Each case: the reference's own reset() (track + spawn), a few scripted
    steps (contact events + hull poses, as gen_bookkeeping: rewards, touched tiles and the backward flag are the reference's own),
    then t / wheel omegas / joint angle / hull angular velocity are set and render(mode) runs under the recording stubs."""
    rec = _RenderRecorder(mcr, Car)
    rng = np.random.RandomState(77)
    f32 = lambda v: float(np.float32(v))
    cases_def = [
        # N, dir, track_seed, h_ratio, ego, flag, mode, t, steps, speed
        (2, "CCW", 0, 0.25, False, True, "state_pixels", 0.02, 1, 0.0),
        (2, "CCW", 0, 0.25, False, True, "state_pixels", 0.5, 6, 0.3),      # below the 0.5 camera threshold
        (2, "CW", 1, 0.25, False, True, "state_pixels", 1.0, 10, 12.0),
        (2, "CCW", 0, 0.25, False, True, "state_pixels", 3.0, 14, 30.0),
        (1, "CCW", 3, 0.25, False, False, "state_pixels", 3.0, 8, 25.0),     # CarRacing-v0 special case: N=1, flag off
        (2, "CCW", 0, 0.4, False, True, "state_pixels", 1.0, 8, 8.0),
        (3, "CW", 4, 0.4, True, True, "state_pixels", 2.0, 8, 0.45),
        (4, "CCW", 2, 0.25, True, False, "state_pixels", 0.5, 8, 5.0),
        (2, "CCW", 0, 0.25, False, True, "rgb_array", 3.0, 6, 20.0),
        (2, "CW", 1, 0.3, True, True, "rgb_array", 0.02, 1, 0.0),
        (2, "CCW", 0, 0.25, False, True, "human", 3.0, 6, 20.0),
        (8, "CCW", 2, 0.25, False, True, "state_pixels", 1.5, 5, 15.0),
        (2, "CW", 1, 0.25, False, False, "state_pixels", 0.98, 5, 0.5),      # exactly 0.5: not above the threshold; t just below 1
        (2, "CCW", 0, 0.25, True, True, "human", 0.5, 4, 3.0),
    ]
    cases, arrays = [], {}
    for ci, (N, direction, tseed, h_ratio, ego, flag, mode, t_final, steps, speed) in enumerate(cases_def):
        np.random.seed(5)
        env = mcr.MultiCarRacing(num_agents=N, verbose=0, direction=direction, use_random_direction=False,
                                 backwards_flag=flag, h_ratio=h_ratio, use_ego_color=ego)
        rec.env = env
        rec.Viewer.count = 0
        env.np_random = np.random.RandomState(tseed)
        Car.created.clear()
        env.reset()                            # runs step(None) -> render("state_pixels") for real: the viewers are created here
        T = len(env.track)
        track = np.array(env.track)
        sign = -1 if direction == "CW" else 1
        script = []
        prog = [3.0 * c for c in range(N)]
        for k in range(steps):
            events, poses = [], []
            for c in range(N):
                old = int(prog[c]); prog[c] += 1.0 + 0.5 * c; new = int(prog[c])
                for ti in range(old, new):
                    tidx = (sign * ti) % T
                    w = int(rng.randint(0, 4))
                    events.append([1, c, w, tidx])
                    if rng.rand() < 0.5:
                        events.append([0, c, w, tidx])
                idx = (sign * new) % T
                a, b, x, y = track[idx]
                head = b + (math.pi if direction == "CW" else 0.0)
                backwards = (c % 2 == 1)       # odd cars drive the wrong way: the flag
                if backwards:
                    head += math.pi
                sp = speed * (1.0 + 0.1 * c)
                if speed == 0.5:
                    vx, vy = 0.5, 0.0          # |v| == 0.5 exactly
                else:
                    vx, vy = -math.sin(head) * sp, math.cos(head) * sp
                ang = head + (0.3 if c % 3 == 2 else -0.05)
                off = [0.0, 1.5, -2.5][c % 3]
                poses.append([f32(x + off * math.cos(b)), f32(y + off * math.sin(b)), f32(vx), f32(vy), f32(ang)])
            script.append(dict(events=events, poses=poses))
        for k, st in enumerate(script):
            def hook(world, st=st):
                for begin, c, w, tidx in st["events"]:
                    ct = _FakeContact(env.road[tidx], env.cars[c].wheels[w])
                    (env.contactListener_keepref.BeginContact if begin else env.contactListener_keepref.EndContact)(ct)
            env.world.step_hook = hook
            for c, (px, py, vx, vy, ang) in enumerate(st["poses"]):
                h = env.cars[c].hull
                h.position = (px, py); h.linearVelocity = (vx, vy); h.angle = ang
                for w in range(4):
                    env.cars[c].wheels[w].car_id = c
            env.step(np.zeros((N, 3)))
        # the state render_indicators reads: gauges incl. negative values
        car_state = []
        for c in range(N):
            hull = env.cars[c].hull
            hull.angularVelocity = f32([-1.7, 2.3, 0.0, 0.6][c % 4])
            omegas = [float(v) for v in ([35.5, -12.25, 80.0, 0.0] if c % 2 == 0 else [-40.0, 7.5, 130.0, -3.0])]
            wheel_angles = [f32(hull.angle + d) for d in ([0.31, -0.2, 0.0, 0.0] if c % 2 == 0 else [-0.4, 0.4, 0.01, -0.01])]
            for w in range(4):
                env.cars[c].wheels[w].omega = omegas[w]
                env.cars[c].wheels[w].joint.angle = float(np.float32(wheel_angles[w]) - np.float32(hull.angle))   # b2RevoluteJoint::GetJointAngle, f32
            car_state.append(dict(hull=[hull.position[0], hull.position[1], hull.linearVelocity[0], hull.linearVelocity[1], hull.angle,
                                        hull.angularVelocity], omega=omegas, wheel_angle=wheel_angles,
                                  joint0_angle=env.cars[c].wheels[0].joint.angle))
        env.t = t_final
        views, road_ref = [], None
        del rec.log[:]
        if mode == "human":
            ret = []
            for a in range(N):
                ret.append(env._render_window(a, mode))
            shape = None
        else:
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")          # np.fromstring's binary mode (:600) is deprecated in numpy 2
                frames = env.render(mode)
            shape = list(frames.shape); ret = None
        # split the log per view: a view starts at its first set_scale
        starts = [i for i, e in enumerate(rec.log) if e[0] == "set_scale"] + [len(rec.log)]
        assert len(starts) == N + 1 and starts[0] == 0
        for a in range(N):
            v, road = _parse_view(rec.log[starts[a]:starts[a + 1]])
            if road_ref is None:
                road_ref = road
            v["road_same_as_view0"] = bool(road.shape == road_ref.shape and (road == road_ref).all())
            assert v["road_same_as_view0"]
            views.append(v)
        arrays["c%d_road" % ci] = road_ref
        cases.append(dict(
            N=N, direction=direction, track_seed=tseed, global_seed=5, h_ratio=h_ratio, use_ego_color=ego, backwards_flag=flag,
            mode=mode, t=t_final, T=T, script=script, car_state=car_state,
            reward=[float(v) for v in env.reward], driving_backward=[bool(v) for v in env.driving_backward],
            touched=[i for i, tl in enumerate(env.road) if tl.color == mcr.ROAD_COLOR],
            n_road_poly=len(env.road_poly), frames_shape=shape, human_returns=ret,
            hull_colors_after=[[float(x) for x in car.hull.color] for car in env.cars],
            views=views))
        print(f"render case {ci}: N={N} mode={mode} t={t_final} quads={len(road_ref)} flag={[v['flag'] is not None for v in views]}")
    rec.uninstall()
    return cases, arrays


def main():
    os.makedirs(OUT, exist_ok=True)
    mcr, Car = load_reference()
    consts = {k: getattr(mcr, k) for k in (
        "STATE_W", "STATE_H", "VIDEO_W", "VIDEO_H", "WINDOW_W", "WINDOW_H", "SCALE", "TRACK_RAD",
        "PLAYFIELD", "FPS", "ZOOM", "TRACK_DETAIL_STEP", "TRACK_TURN_RATE", "TRACK_WIDTH", "BORDER",
        "BORDER_MIN_COUNT", "ROAD_COLOR", "LINE_SPACING", "LATERAL_SPACING", "K_BACKWARD")}
    consts["CAR_COLORS"] = [list(c) for c in mcr.CAR_COLORS]
    consts["BACKWARD_THRESHOLD"] = float(mcr.BACKWARD_THRESHOLD)
    tracks = gen_tracks(mcr, range(12))
    np.savez_compressed(os.path.join(OUT, "tracks.npz"), **tracks)
    with open(os.path.join(OUT, "spawn.json"), "w") as f:
        json.dump(dict(constants=consts, cases=gen_spawn(mcr, Car)), f)
    with open(os.path.join(OUT, "bookkeeping.json"), "w") as f:
        json.dump(gen_bookkeeping(mcr, Car), f)
    cases, arrays = gen_render_stream(mcr, Car)
    with open(os.path.join(OUT, "render_stream.json"), "w") as f:
        json.dump(cases, f)
    np.savez_compressed(os.path.join(OUT, "render_stream.npz"), **arrays)
    print("wrote goldens to", os.path.abspath(OUT))


if __name__ == "__main__":
    main()
