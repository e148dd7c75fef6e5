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

What the stubs replace (NOT pinned by these goldens): Box2D world/bodies, the gym
`Car` class, rendering, shapely.  `shapely.geometry.Point.within(Polygon)` is
stubbed by a strict-interior even-odd point-in-polygon test.

Outputs (data only — inputs and expected outputs, never reference source):
  tests/golden/tracks.npz        per-seed track / road_poly / colours / retries
  tests/golden/spawn.json        car order + spawn poses for N x direction x seeds
  tests/golden/bookkeeping.json  scripted contact + pose traces -> rewards/flags
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

    class _Wheel:
        def __init__(self):
            self.tiles = set()
            self.userData = self

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
        def draw(self, *a): pass

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
    print("wrote goldens to", os.path.abspath(OUT))


if __name__ == "__main__":
    main()
