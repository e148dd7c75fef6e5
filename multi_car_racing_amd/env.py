"""Single-env facade with the reference's gym surface (multi_car_racing.py:125-674): same constructor
kwargs and defaults, `seed/reset/step/render/close`, obs (num_agents,96,96,3) uint8, reward float64
(num_agents,), done bool, info {}.  It is a B=1 slice of the batched HIP engine — every step goes through
the C-ABI; nothing is simulated on the host.

Like the reference, the env keeps ONE b2World for its life (multi_car_racing.py:138; _destroy :173-181, reset :341): from the second
episode on Box2D's broadphase proxy ids — the tie-break between cars that reach a tile in the same step, :113-120, and fixtureA of a
car<->car contact — come off the world's free list.  The handle carries the world on the device (include/mcr.h: mcr_config::fresh_world = 0,
csrc/k_world.h: the ids follow from a stack of free leaf ids, no tree), exactly as the batched VecMultiCarRacing does.  (Rounds 4-5 kept a
literal b2DynamicTree on the host for this and synchronised every step to advance it: include/mcr.h mcr_world_*, still there as a CPU twin the
tests hold against the stack rule.)

RNG parity with the reference: the track comes from `self.np_random` (a numpy RandomState, gym seeding), the
direction and the car order from the *global* `np.random` stream, drawn in the reference's order
(:351-357) — so `np.random.seed(s); env.seed(s)` reproduces the reference's episode setup.
"""
import ctypes

import numpy as np

from . import _lib, seeding

STATE_W = STATE_H = 96
VIDEO_W, VIDEO_H = 600, 400          # multi_car_racing.py:45-46
WINDOW_W, WINDOW_H = 1000, 800       # multi_car_racing.py:47-48
FPS = 50


class _Box:
    """Minimal stand-in for gym.spaces.Box (gym itself is optional)."""

    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        if shape is None:
            self.low = np.asarray(low, dtype=dtype); self.high = np.asarray(high, dtype=dtype)
            self.shape = self.low.shape
        else:
            self.shape = tuple(shape)
            self.low = np.full(self.shape, low, dtype=dtype); self.high = np.full(self.shape, high, dtype=dtype)
        self._rng = np.random.RandomState()

    def sample(self):
        if self.dtype.kind == "f":
            return self._rng.uniform(self.low, self.high).astype(self.dtype)
        return self._rng.randint(self.low, self.high + 1).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and np.all(x >= self.low) and np.all(x <= self.high)


def _box(low, high, shape=None, dtype=np.float32):
    try:
        from gym import spaces
        return spaces.Box(low, high, dtype=dtype) if shape is None else spaces.Box(low=low, high=high, shape=shape, dtype=dtype)
    except Exception:
        return _Box(low, high, shape, dtype)


class MultiCarRacing:
    metadata = {"render.modes": ["human", "rgb_array", "state_pixels"], "video.frames_per_second": FPS}
    reward_range = (-float("inf"), float("inf"))
    spec = None

    def __init__(self, num_agents=2, verbose=1, direction="CCW", use_random_direction=True, backwards_flag=True,
                 h_ratio=0.25, use_ego_color=False, device=0, car_contacts=True):
        import torch
        if not torch.cuda.is_available():
            raise _lib.McrError("MultiCarRacing needs a HIP device: the step path has no CPU fallback")
        self._torch = torch
        self.L = _lib.load()
        self.seed()
        self.num_agents = int(num_agents)
        self.verbose = verbose
        self.use_random_direction = use_random_direction
        self.episode_direction = direction
        if self.use_random_direction:
            self.episode_direction = str(np.random.choice(["CW", "CCW"]))      # :157
        self.backwards_flag = backwards_flag
        self.h_ratio = h_ratio
        self.use_ego_color = use_ego_color
        self.action_lb = np.tile(np.array([-1, +0, +0]), 1)
        self.action_ub = np.tile(np.array([+1, +1, +1]), 1)
        self.action_space = _box(self.action_lb.astype(np.float32), self.action_ub.astype(np.float32))
        self.observation_space = _box(0, 255, (STATE_H, STATE_W, 3), np.uint8)
        self._dev = torch.device("cuda", device)
        torch.cuda.set_device(self._dev)
        cfg = _lib.Config(1, self.num_agents, device, 1, 0, int(bool(backwards_flag)), int(bool(use_ego_color)),
                          int(bool(car_contacts)), 0, 0, float(h_ratio), 1, 0)   # skid particles on: render("rgb_array") draws them
        self._h = ctypes.c_void_p()
        _lib.check(self.L.mcr_create(ctypes.byref(cfg), ctypes.byref(self._h)), "mcr_create")
        N = self.num_agents
        self._obs = torch.zeros((1, N, 96, 96, 3), dtype=torch.uint8, device=self._dev)
        self._rew = torch.zeros((1, N), dtype=torch.float64, device=self._dev)
        self._done = torch.zeros((1,), dtype=torch.uint8, device=self._dev)
        self._act = torch.zeros((1, N, 3), dtype=torch.float32, device=self._dev)
        self._blob = np.zeros(_lib.episode_bytes(), np.uint8)
        self._was_reset = False
        self.state = None
        self.track = None
        self.car_order = None
        self.t = None

    # ------------------------------------------------------------------ gym API
    def seed(self, seed=None):
        self.np_random, seed = seeding.np_random(seed)
        return [seed]

    def _mt_from(self, rs):
        st = rs.get_state()
        mt = np.zeros(_lib.MT_WORDS, np.uint32); mt[:624] = st[1]; mt[624] = st[2]
        return mt

    def _mt_to(self, rs, mt):
        rs.set_state(("MT19937", mt[:624].copy(), int(mt[624]), 0, 0.0))

    def reset(self):
        N = self.num_agents
        if self.use_random_direction:
            self.episode_direction = str(np.random.choice(["CW", "CCW"]))      # :351-352
        ids = [i for i in range(N)]
        shuffle_ids = np.random.choice(ids, size=N, replace=False)               # :355-357
        self.car_order = {i: int(shuffle_ids[i]) for i in range(N)}
        order = np.array([self.car_order[i] for i in range(N)], np.int32)
        mt = self._mt_from(self.np_random)
        info = np.zeros(4, np.int32)
        _lib.check(self.L.mcr_episode_generate(_lib.ptr(mt), N, int(self.episode_direction == "CW"), _lib.ptr(order),
                                               _lib.ptr(self._blob), _lib.ptr(info)), "mcr_episode_generate")
        self._mt_to(self.np_random, mt)
        if self.verbose == 1:
            for _ in range(int(info[2])):
                print("retry to generate track (normal if there are not many of this messages)")
            print("Track generation: -> %i-tiles track" % int(info[0]))
        ep = _lib.unpack_episode(self._blob)
        self.track = [(float(a), float(b), float(x), float(y)) for a, (x, y, b) in zip(ep["alpha"], ep["track"])]
        # (_destroy() + _create_track() + the cars on the env's ONE world: the reset pass re-issues the proxy ids on the device, k_world.h)
        st = self._torch.cuda.current_stream(self._dev)
        one = np.zeros(1, np.int32)
        _lib.check(self.L.mcr_stage_episodes(self._h, _lib.ptr(one), 1, _lib.ptr(self._blob), ctypes.c_void_p(st.cuda_stream)), "stage")
        _lib.check(self.L.mcr_reset(self._h, None, ctypes.c_void_p(self._obs.data_ptr()), ctypes.c_void_p(st.cuda_stream)), "mcr_reset")
        self._was_reset = True
        self.state = self._obs[0].cpu().numpy()
        self.t = 1.0 / FPS
        return self.state

    def step(self, action):
        if not self._was_reset:
            raise AttributeError("step() called before reset()")        # reference: NoneType car has no attribute
        N = self.num_agents
        st = self._torch.cuda.current_stream(self._dev)
        a_ptr = None
        if action is not None:
            a = np.reshape(action, (N, -1))                                # ValueError on a bad shape, like :420
            self._act.copy_(self._torch.from_numpy(np.ascontiguousarray(a[:, :3], dtype=np.float32)).view(1, N, 3))
            a_ptr = ctypes.c_void_p(self._act.data_ptr())
        _lib.check(self.L.mcr_step(self._h, a_ptr, ctypes.c_void_p(self._obs.data_ptr()), ctypes.c_void_p(self._rew.data_ptr()),
                                   ctypes.c_void_p(self._done.data_ptr()), None, ctypes.c_void_p(st.cuda_stream)), "mcr_step")
        self.state = self._obs[0].cpu().numpy()
        step_reward = self._rew[0].cpu().numpy().copy()
        done = bool(self._done[0].item())
        self.t += 1.0 / FPS
        return self.state, step_reward, done, {}

    def render(self, mode="human"):
        assert mode in ["human", "state_pixels", "rgb_array"]
        if mode == "state_pixels":
            return self._obs[0].cpu().numpy()
        if mode == "rgb_array":                                            # VIDEO_W x VIDEO_H viewport (:573-575)
            if not self._was_reset:
                return None                                                # reference: "reset() not called yet" (:538)
            out = self._torch.empty((self.num_agents, VIDEO_H, VIDEO_W, 3), dtype=self._torch.uint8, device=self._dev)
            st = self._torch.cuda.current_stream(self._dev)
            _lib.check(self.L.mcr_render(self._h, 0, VIDEO_W, VIDEO_H, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(st.cuda_stream)), "mcr_render")
            return out.cpu().numpy()
        # 'human' (:577-583, 595-597): the reference draws the WINDOW_W x WINDOW_H viewport into one window per agent, flips it and
        # returns the windows' `isopen` flags.  There is no display here: the same frames are drawn off-screen and kept in
        # `self.human_frames` (uint8 [N, WINDOW_H, WINDOW_W, 3], device) for whoever wants to show them; every "window" is open.
        if not self._was_reset:
            return np.array([None] * self.num_agents, dtype=object)
        self.human_frames = self._torch.empty((self.num_agents, WINDOW_H, WINDOW_W, 3), dtype=self._torch.uint8, device=self._dev)
        st = self._torch.cuda.current_stream(self._dev)
        _lib.check(self.L.mcr_render(self._h, 0, WINDOW_W, WINDOW_H, ctypes.c_void_p(self.human_frames.data_ptr()), ctypes.c_void_p(st.cuda_stream)), "mcr_render")
        return np.ones(self.num_agents, dtype=bool)

    def close(self):
        if getattr(self, "_h", None):
            self.L.mcr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ attributes users of the reference read
    def _env_state(self):
        N = self.num_agents
        reward = np.zeros((1, N)); tvc = np.zeros((1, N), np.int32); bw = np.zeros((1, N), np.uint8); og = np.zeros((1, N), np.uint8)
        t = np.zeros(1)
        _lib.check(self.L.mcr_get_env_state(self._h, _lib.ptr(reward), _lib.ptr(tvc), _lib.ptr(bw), _lib.ptr(og), _lib.ptr(t), None, None))
        return reward[0], tvc[0], bw[0].astype(bool), og[0].astype(bool)

    @property
    def reward(self):
        return self._env_state()[0]

    @property
    def tile_visited_count(self):
        return [int(v) for v in self._env_state()[1]]

    @property
    def driving_backward(self):
        return self._env_state()[2]

    @property
    def driving_on_grass(self):
        return self._env_state()[3]

    @property
    def car_positions(self):
        pos = np.zeros((1, self.num_agents, 2), np.float32)
        _lib.check(self.L.mcr_get_positions(self._h, _lib.ptr(pos)))
        return pos[0]

    @property
    def unwrapped(self):
        return self
