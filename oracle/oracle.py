"""oracle/oracle.py — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Python side of the CPU oracle:
  * `make_track` / `spawn_poses` / `new_episode`: numpy restatement of the reference's episode
    setup (track walk multi_car_racing.py:183-338, spawn :355-406).  Pinned by
    tests/golden/{tracks.npz,spawn.json} which were produced by running the reference module
    itself (oracle/make_goldens.py).
  * `OracleEnv`: ctypes binding of oracle/_build/libmcr_oracle.so (physics + tile contacts +
    bookkeeping + raster restatement; see mcr_oracle.cpp header for what is / is not pinned).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes, math, os, subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libmcr_oracle.so")

# constants — multi_car_racing.py:43-78
SCALE = 6.0
TRACK_RAD = 900 / SCALE
PLAYFIELD = 2000 / SCALE
TRACK_DETAIL_STEP = 21 / SCALE
TRACK_TURN_RATE = 0.31
TRACK_WIDTH = 40 / SCALE
BORDER = 8 / SCALE
BORDER_MIN_COUNT = 4
ROAD_COLOR = (0.4, 0.4, 0.4)
LINE_SPACING = 5
LATERAL_SPACING = 3
N_CHECKPOINTS = 12


def build(force=False):
    if force or not os.path.exists(_LIB) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB)
            for f in ("mcr_oracle.cpp", "mcr_oracle_contacts.inc")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


# ----------------------------------------------------------------------------- track
def _walk(rng):
    """One attempt of the checkpoint walk (:186-259). Returns (points, start_alpha)."""
    two_pi = 2 * math.pi
    cps = []
    for c in range(N_CHECKPOINTS):
        jitter = rng.uniform(0, two_pi * 1 / N_CHECKPOINTS)
        rad = rng.uniform(TRACK_RAD / 3, TRACK_RAD)
        ang = two_pi * c / N_CHECKPOINTS + jitter
        if c == 0:
            ang, rad = 0, 1.5 * TRACK_RAD
        if c == N_CHECKPOINTS - 1:
            ang, rad = two_pi * c / N_CHECKPOINTS, 1.5 * TRACK_RAD
        cps.append((ang, rad * math.cos(ang), rad * math.sin(ang)))
    start_alpha = two_pi * (-0.5) / N_CHECKPOINTS

    x, y, beta = 1.5 * TRACK_RAD, 0, 0
    target, laps, budget = 0, 0, 2500
    crossed = False
    pts = []
    while True:
        alpha = math.atan2(y, x)
        if crossed and alpha > 0:
            laps += 1
            crossed = False
        if alpha < 0:
            crossed = True
            alpha += two_pi
        # next checkpoint at or ahead of the current polar angle (wraps by -2pi when exhausted)
        while True:
            found = False
            while True:
                ca, cx, cy = cps[target % N_CHECKPOINTS]
                if alpha <= ca:
                    found = True
                    break
                target += 1
                if target % N_CHECKPOINTS == 0:
                    break
            if found:
                break
            alpha -= two_pi
        hx, hy = math.cos(beta), math.sin(beta)          # "radial" axis of the walker
        fx, fy = -hy, hx                                 # forward axis
        along = hx * (cx - x) + hy * (cy - y)
        while beta - alpha > 1.5 * math.pi:
            beta -= two_pi
        while beta - alpha < -1.5 * math.pi:
            beta += two_pi
        beta0 = beta
        along *= SCALE
        if along > 0.3:
            beta -= min(TRACK_TURN_RATE, abs(0.001 * along))
        if along < -0.3:
            beta += min(TRACK_TURN_RATE, abs(0.001 * along))
        x += fx * TRACK_DETAIL_STEP
        y += fy * TRACK_DETAIL_STEP
        pts.append((alpha, beta0 * 0.5 + beta * 0.5, x, y))
        if laps > 4:
            break
        budget -= 1
        if budget == 0:
            break
    return pts, start_alpha


def make_track(rng):
    """One attempt of `_create_track`. Returns None on failure, else a dict with
    track (T,4) f64, poly (P,4,2) f64, color (P,3) f64, tile_of_quad (P,) int32."""
    pts, start_alpha = _walk(rng)
    # last full lap between two start-line crossings (:262-281)
    i1 = i2 = -1
    i = len(pts)
    while True:
        i -= 1
        if i == 0:
            return None
        crossing = pts[i][0] > start_alpha and pts[i - 1][0] <= start_alpha
        if crossing and i2 == -1:
            i2 = i
        elif crossing and i1 == -1:
            i1 = i
            break
    pts = pts[i1:i2 - 1]
    b0 = pts[0][1]
    gap = np.sqrt(np.square(math.cos(b0) * (pts[0][2] - pts[-1][2])) +
                  np.square(math.sin(b0) * (pts[0][3] - pts[-1][3])))
    if gap > TRACK_DETAIL_STEP:
        return None
    T = len(pts)
    # kerbs on sustained turns (:293-307)
    kerb = [False] * T
    for i in range(T):
        ok, side = True, 0
        for k in range(BORDER_MIN_COUNT):
            d = pts[i - k][1] - pts[i - k - 1][1]
            ok &= abs(d) > TRACK_TURN_RATE * 0.2
            side += np.sign(d)
        kerb[i] = bool(ok and abs(side) == BORDER_MIN_COUNT)
    for i in range(T):
        for k in range(BORDER_MIN_COUNT):
            kerb[i - k] |= kerb[i]
    polys, cols, tile_of = [], [], []
    for i in range(T):
        _, b1, x1, y1 = pts[i]
        _, b2, x2, y2 = pts[i - 1]
        c1, s1, c2, s2 = math.cos(b1), math.sin(b1), math.cos(b2), math.sin(b2)
        quad = [(x1 - TRACK_WIDTH * c1, y1 - TRACK_WIDTH * s1), (x1 + TRACK_WIDTH * c1, y1 + TRACK_WIDTH * s1),
                (x2 + TRACK_WIDTH * c2, y2 + TRACK_WIDTH * s2), (x2 - TRACK_WIDTH * c2, y2 - TRACK_WIDTH * s2)]
        shade = 0.01 * (i % 3)
        polys.append(quad)
        cols.append((ROAD_COLOR[0] + shade, ROAD_COLOR[1] + shade, ROAD_COLOR[2] + shade))
        tile_of.append(i)
        if kerb[i]:
            sd = np.sign(b2 - b1)
            w0, w1 = sd * TRACK_WIDTH, sd * (TRACK_WIDTH + BORDER)
            polys.append([(x1 + w0 * c1, y1 + w0 * s1), (x1 + w1 * c1, y1 + w1 * s1),
                          (x2 + w1 * c2, y2 + w1 * s2), (x2 + w0 * c2, y2 + w0 * s2)])
            cols.append((1, 1, 1) if i % 2 == 0 else (1, 0, 0))
            tile_of.append(-1)
    return dict(track=np.array(pts, dtype=np.float64), poly=np.array(polys, dtype=np.float64),
                color=np.array(cols, dtype=np.float64), tile_of_quad=np.array(tile_of, dtype=np.int32),
                start_alpha=start_alpha)


def spawn_poses(track, car_order, cw):
    """(:366-406) -> (N,3) f64 rows (angle, x, y) handed to the Car constructor."""
    N = len(car_order)
    out = np.zeros((N, 3))
    _, x0, y0 = track[0][1:4]
    for cid in range(N):
        row = math.floor(car_order[cid] / 2)
        side = 2 * (car_order[cid] % 2) - 1
        ref = track[-row * LINE_SPACING]
        dx, dy = ref[2] - x0, ref[3] - y0
        ang = ref[1]
        if cw:
            ang -= np.pi
        nt = ang - np.pi / 2
        out[cid] = (ang, x0 + dx + LATERAL_SPACING * np.sin(nt) * side, y0 + dy + LATERAL_SPACING * np.cos(nt) * side)
    return out


def new_episode(N, track_rng, global_rng, direction="CCW", use_random_direction=True):
    """Everything `reset()` decides before bodies exist (:349-364): direction draw, car order draw
    (both from the *global* stream in the reference), track attempts from the env stream."""
    if use_random_direction:
        direction = str(global_rng.choice(["CW", "CCW"]))
    order = global_rng.choice(list(range(N)), size=N, replace=False)
    retries = 0
    while True:
        tr = make_track(track_rng)
        if tr is not None:
            break
        retries += 1
    tr["direction"] = direction
    tr["car_order"] = [int(v) for v in order]
    tr["retries"] = retries
    tr["poses"] = spawn_poses(tr["track"], tr["car_order"], direction == "CW")
    return tr


# ----------------------------------------------------------------------------- C++ oracle binding
_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB)
        L.orc_create.restype = ctypes.c_void_p
        L.orc_create.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        for name in ("orc_destroy", "orc_set_track", "orc_reset", "orc_step", "orc_render", "orc_get_state",
                     "orc_set_body", "orc_get_env", "orc_positions", "orc_contact_event", "orc_set_hull_pose",
                     "orc_bookkeeping", "orc_wheel_tile_counts", "orc_reset_nostep", "orc_step_masked", "orc_reset_masked", "orc_solve_only"):
            getattr(L, name).restype = None
        L.orc_render_size.restype = None
        L.orc_render_size.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.orc_num_car_contacts.restype = ctypes.c_int
        L.orc_num_car_contacts.argtypes = [ctypes.c_void_p]
        L.orc_set_island_order.restype = None
        L.orc_set_island_order.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.orc_debug_island_diff.restype = ctypes.c_int
        L.orc_debug_island_diff.argtypes = [ctypes.c_void_p]
        L.orc_set_world_mode.restype = None
        L.orc_set_world_mode.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.orc_debug_proxy_ids.restype = ctypes.c_int
        L.orc_debug_proxy_ids.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def rollout(envs, actions, render=True, threads=None):
    """CPU baseline helper: advance independent OracleEnv objects with actions [steps, n, N, 3] on `threads`
    OpenMP threads (one C call, no Python in the loop)."""
    L = lib()
    n = len(envs); steps = actions.shape[0]; N = envs[0].N
    hs = (ctypes.c_void_p * n)(*[e.h for e in envs])
    a = np.ascontiguousarray(actions, np.float32)
    obs = np.zeros((n, N, 96, 96, 3), np.uint8) if render else None
    L.orc_rollout.restype = None
    L.orc_rollout(hs, ctypes.c_int(n), _p(a), ctypes.c_int(steps), ctypes.c_int(int(render)), _p(obs) if render else None,
                  ctypes.c_int(threads or os.cpu_count() or 1))
    return obs


def step_batch(envs, actions, render_mask=None, threads=None):
    """Advance independent OracleEnv objects by ONE step on `threads` OpenMP threads.  actions [n,N,3] f32.
    Returns (obs [n,N,96,96,3], amb [n,N,96,96], rewards [n,N], done [n] bool); obs/amb rows are valid where
    render_mask is set."""
    L = lib()
    n = len(envs); N = envs[0].N
    hs = (ctypes.c_void_p * n)(*[e.h for e in envs])
    a = None if actions is None else np.ascontiguousarray(actions, np.float32)      # None: the reference's step(None) for every env
    rm = None if render_mask is None else np.ascontiguousarray(render_mask, np.uint8)
    obs = np.zeros((n, N, 96, 96, 3), np.uint8); amb = np.zeros((n, N, 96, 96), np.uint8)
    rew = np.zeros((n, N)); done = np.zeros(n, np.uint8)
    L.orc_step_batch.restype = None
    L.orc_step_batch(hs, ctypes.c_int(n), _p(a) if a is not None else None, _p(rm) if rm is not None else None, _p(obs), _p(amb), _p(rew), _p(done),
                     ctypes.c_int(threads or os.cpu_count() or 1))
    return obs, amb, rew, done.astype(bool)


def overlap_sweep(n, seed=1, band=1e-5, far=5e-5, threads=None):
    """b2TestOverlap as Box2D computes it (GJK b2Distance, oracle restatement) vs the SAT + vertex-edge predicate the
    oracle and the kernels use, on n random wheel/tile pairs at core separation 0.02 + U(-band, band).
    Returns dict(samples, disagree, gjk_touching, sat_touching, max_gjk_iters, disagree_far) — `far`: |delta| > far."""
    out = np.zeros(6, np.int64)
    L = lib(); L.orc_overlap_sweep.restype = None
    L.orc_overlap_sweep(ctypes.c_longlong(int(n)), ctypes.c_uint(int(seed)), ctypes.c_double(band), ctypes.c_double(far), _p(out),
                        ctypes.c_int(threads or os.cpu_count() or 1))
    return dict(samples=int(out[0]), disagree=int(out[1]), gjk_touching=int(out[2]), sat_touching=int(out[3]), max_gjk_iters=int(out[4]), disagree_far=int(out[5]))


def overlap_cases(n, seed=1, band=1e-5, threads=None, wheel_first=False):
    """The sweep's cases themselves: (quads [k,4,2] f32 — a tile's 4 input points, poses [k,3] f32 — wheel body x, y, angle,
    gjk [k] bool — Box2D's b2TestOverlap verdict with the tile as proxy A, or (wheel_first) with the wheel as proxy A)."""
    out = np.zeros(7, np.int64)
    lib().orc_set_overlap_order.restype = None
    lib().orc_set_overlap_order(ctypes.c_int(int(bool(wheel_first))))
    quads = np.zeros((int(n), 8), np.float32); poses = np.zeros((int(n), 3), np.float32); g = np.zeros(int(n), np.uint8)
    L = lib(); L.orc_overlap_cases.restype = None
    L.orc_overlap_cases(ctypes.c_longlong(int(n)), ctypes.c_uint(int(seed)), ctypes.c_double(band), ctypes.c_double(5e-5), _p(out),
                        ctypes.c_int(threads or os.cpu_count() or 1), _p(quads), _p(poses), _p(g), ctypes.c_longlong(int(n)))
    k = int(out[6])
    L.orc_set_overlap_order(ctypes.c_int(0))
    return quads[:k].reshape(k, 4, 2), poses[:k], g[:k].astype(bool)


def sincos(a, mode=0):
    L = lib()
    L.orc_set_trig_mode(mode)
    s, c = ctypes.c_float(), ctypes.c_float()
    L.orc_sincos(ctypes.c_float(a), ctypes.byref(s), ctypes.byref(c))
    L.orc_set_trig_mode(0)
    return s.value, c.value


def mass_props():
    out = np.zeros(6, np.float32)
    lib().orc_mass_props(_p(out))
    return out


class OracleEnv:
    """Single-env CPU oracle with the reference's reset/step surface (obs (N,96,96,3) u8,
    reward (N,) f64, done bool)."""

    def __init__(self, num_agents=2, h_ratio=0.25, backwards_flag=True, use_ego_color=False,
                 car_contacts=True, trig_mode=0, world_mode=1):
        self.N = num_agents
        self.L = lib()
        self.trig_mode = trig_mode
        self.h = ctypes.c_void_p(self.L.orc_create(num_agents, float(h_ratio), int(backwards_flag),
                                                   int(use_ego_color), int(car_contacts)))
        self.T = 0
        self.L.orc_set_world_mode(self.h, int(world_mode))     # the reference's semantics by default: ONE b2World for the life of the env

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None

    def set_world_mode(self, mode):
        """1 (default, the reference, what the kernels implement since round 6): one world for the life of this env, as the reference keeps
        it across reset() (multi_car_racing.py:138, 341) — proxy ids come off the b2DynamicTree's free list (mcr_oracle.cpp: DynTree, a
        literal tree); 0: every episode is the first episode of a fresh b2World (mcr_config::fresh_world = 1).  Call before the first reset."""
        self.L.orc_set_world_mode(self.h, int(mode))

    def set_island_order(self, mode):
        """1 (default, what the kernels implement): the order of b2World::Solve's island DFS for the car<->car contacts and each car's
        joints (mcr_oracle_contacts.inc: island_dfs); 0: rounds 1-3's DEFINED order (contacts ascending by (carA, fixA, carB, fixB),
        joints 3,2,1,0 per car); 2: the DFS order of the contacts, joints 3,2,1,0."""
        self.L.orc_set_island_order(self.h, int(mode))

    def island_diff(self):
        """last step: bit 0 — some car's joints in another order than 3,2,1,0; bit 1 — contacts in another order than ascending"""
        return int(self.L.orc_debug_island_diff(self.h))

    def proxy_ids(self):
        """world mode 1: (tile ids [T], car fixture ids [N, 8]) of the current episode"""
        buf = np.zeros(self.T + self.N * 8, np.int32)
        n = self.L.orc_debug_proxy_ids(self.h, _p(buf), len(buf))
        assert n == len(buf), (n, len(buf))
        return buf[:self.T].copy(), buf[self.T:].reshape(self.N, 8).copy()

    def set_episode(self, ep):
        tr = ep["track"]
        self.T = len(tr)
        xyb = np.ascontiguousarray(tr[:, [2, 3, 1]], dtype=np.float64)
        poly = np.ascontiguousarray(ep["poly"], dtype=np.float64)
        col = np.ascontiguousarray(ep["color"], dtype=np.float64)
        tq = np.ascontiguousarray(ep["tile_of_quad"], dtype=np.int32)
        self.L.orc_set_track(self.h, ctypes.c_int(self.T), _p(xyb), ctypes.c_int(len(poly)), _p(poly), _p(col), _p(tq),
                             ctypes.c_int(int(ep["direction"] == "CW")))
        self._poses = np.ascontiguousarray(ep["poses"], dtype=np.float64)

    def reset(self, ep=None, render=True):
        if ep is not None:
            self.set_episode(ep)
        self.L.orc_set_trig_mode(self.trig_mode)
        obs = np.zeros((self.N, 96, 96, 3), np.uint8)
        self.last_amb = np.zeros((self.N, 96, 96), np.uint8)
        self.L.orc_reset_masked(self.h, _p(self._poses), _p(obs) if render else None, _p(self.last_amb))
        self.last_obs = obs
        return obs

    def step(self, action, render=True):
        self.L.orc_set_trig_mode(self.trig_mode)
        obs = np.zeros((self.N, 96, 96, 3), np.uint8) if render else None
        rew = np.zeros(self.N, np.float64)
        done = np.zeros(1, np.uint8)
        a = None if action is None else np.ascontiguousarray(np.reshape(action, (self.N, -1))[:, :3], dtype=np.float32)
        self.last_amb = np.zeros((self.N, 96, 96), np.uint8) if render else None
        self.L.orc_step_masked(self.h, _p(a) if a is not None else None, _p(obs) if render else None,
                               _p(self.last_amb) if render else None, _p(rew), _p(done))
        self.last_obs = obs
        return obs, rew, bool(done[0]), {}

    def render_with_mask(self):
        obs = np.zeros((self.N, 96, 96, 3), np.uint8)
        amb = np.zeros((self.N, 96, 96), np.uint8)
        self.L.orc_render(self.h, _p(obs), _p(amb))
        return obs, amb

    def render_size(self, width, height):
        """render('rgb_array')-style frame of the CURRENT state at width x height (with skid particles and score label);
        returns (frames [N,H,W,3], ambiguity mask [N,H,W])."""
        obs = np.zeros((self.N, height, width, 3), np.uint8)
        amb = np.zeros((self.N, height, width), np.uint8)
        self.L.orc_render_size(self.h, int(width), int(height), _p(obs), _p(amb))
        return obs, amb

    def state(self):
        N = self.N
        bodies = np.zeros((N, 5, 6), np.float32)
        joints = np.zeros((N, 4, 4), np.float32)
        wheels = np.zeros((N, 4, 5), np.float64)
        limit = np.zeros((N, 4), np.int32)
        on_road = np.zeros((N, 4), np.uint8)
        sleep = np.zeros((N, 5), np.float32)
        self.L.orc_get_state(self.h, _p(bodies), _p(joints), _p(wheels), _p(limit), _p(on_road), _p(sleep))
        return dict(bodies=bodies, joints=joints, wheels=wheels, limit=limit, on_road=on_road, sleep=sleep)

    def env_state(self):
        N, T = self.N, self.T
        reward = np.zeros(N); tvc = np.zeros(N, np.int32)
        bw = np.zeros(N, np.uint8); og = np.zeros(N, np.uint8)
        visited = np.zeros(T, np.uint8); touched = np.zeros(T, np.uint8)
        t = np.zeros(1)
        self.L.orc_get_env(self.h, _p(reward), _p(tvc), _p(bw), _p(og), _p(visited), _p(touched), _p(t))
        return dict(reward=reward, tile_visited_count=tvc, driving_backward=bw, driving_on_grass=og,
                    visited=visited, touched=touched, t=float(t[0]))

    def set_body(self, car, body, s6):
        s = np.ascontiguousarray(s6, dtype=np.float32)
        self.L.orc_set_body(self.h, ctypes.c_int(car), ctypes.c_int(body), _p(s))

    def positions(self):
        out = np.zeros((self.N, 2), np.float32)
        self.L.orc_positions(self.h, _p(out))
        return out

    # ---- scripted hooks (tests/golden/bookkeeping.json)
    def reset_nostep(self, ep):
        self.set_episode(ep)
        self.L.orc_reset_nostep(self.h, _p(self._poses))

    def contact_event(self, begin, car, wheel, tile):
        self.L.orc_contact_event(self.h, ctypes.c_int(int(begin)), ctypes.c_int(car), ctypes.c_int(wheel), ctypes.c_int(tile))

    def set_hull_pose(self, car, px, py, vx, vy, angle):
        f = ctypes.c_float
        self.L.orc_set_hull_pose(self.h, ctypes.c_int(car), f(px), f(py), f(vx), f(vy), f(angle))

    def bookkeeping(self, has_action=True):
        rew = np.zeros(self.N, np.float64); done = np.zeros(1, np.uint8)
        self.L.orc_bookkeeping(self.h, ctypes.c_int(int(has_action)), _p(rew), _p(done))
        return rew, bool(done[0])

    def wheel_tile_counts(self):
        out = np.zeros((self.N, 4), np.int32)
        self.L.orc_wheel_tile_counts(self.h, _p(out))
        return out

    def draw_list(self, agent=0, width=96, height=96, cap=2048):
        """Polygons of the frame render_view() would draw now, in draw order: list of (xy [n,2] f64 pixel space, rgb)."""
        buf = np.zeros((cap, 20))
        self.L.orc_draw_list.restype = ctypes.c_int
        n = self.L.orc_draw_list(self.h, ctypes.c_int(agent), ctypes.c_int(width), ctypes.c_int(height), _p(buf), ctypes.c_int(cap))
        return [(buf[i, 4:4 + 2 * int(buf[i, 0])].reshape(-1, 2).copy(), buf[i, 1:4].astype(np.uint8)) for i in range(n)]

    def render_stream(self, agent=0, width=96, height=96, particles=False, cap=4096):
        """The frame of `agent` as the reference hands it to GL, before any transform (mcr_oracle.cpp: orc_render_stream):
        (camera dict, list of (space, tag, xy [n,2] f64, rgb u8))."""
        cam = np.zeros(21)
        buf = np.zeros((cap, 24))
        f = self.L.orc_render_stream
        f.restype = ctypes.c_int
        n = f(self.h, ctypes.c_int(agent), ctypes.c_int(width), ctypes.c_int(height), ctypes.c_int(int(particles)), _p(cam), _p(buf), ctypes.c_int(cap))
        assert n < cap
        chars = [int(c) for c in cam[5:]]
        label = bytes(chars[:chars.index(0)]).decode()
        camera = dict(zoom=float(cam[0]), tx=float(cam[1]), ty=float(cam[2]), angle=float(cam[3]), flag=bool(cam[4]), label=label)
        prims = [(int(buf[i, 0]), int(buf[i, 1]), buf[i, 6:6 + 2 * int(buf[i, 2])].reshape(-1, 2).copy(), buf[i, 3:6].astype(np.uint8)) for i in range(n)]
        return camera, prims

    def set_render_state(self, car, hull_w, wheel_angles, omegas, phases=(0, 0, 0, 0)):
        f = self.L.orc_set_render_state
        f.restype = None
        wa = np.ascontiguousarray(wheel_angles, dtype=np.float32); om = np.ascontiguousarray(omegas, dtype=np.float64)
        ph = np.ascontiguousarray(phases, dtype=np.float64)
        f(self.h, ctypes.c_int(car), ctypes.c_float(hull_w), _p(wa), _p(om), _p(ph))

    def set_time(self, t):
        f = self.L.orc_set_time
        f.restype = None
        f(self.h, ctypes.c_double(t))

    def solve_only(self, steps=1):
        self.L.orc_set_trig_mode(self.trig_mode)
        self.L.orc_solve_only(self.h, ctypes.c_int(steps))

    def num_car_contacts(self):
        return self.L.orc_num_car_contacts(self.h)
