"""Batched MultiCarRacing-v0 on one MI355X: B independent envs advanced by the HIP kernels behind
include/mcr.h.  torch is plumbing only (device memory, streams); every step is the C-ABI `mcr_step`.

Semantics per env follow the reference's `reset()`/`step()` (multi_car_racing.py:340-509) with the
TimeLimit(1000) of gym_multi_car_racing/__init__.py:8.  With `auto_reset=True` a finished env is re-spawned
on the device inside the same `step` call (its `obs` row is the first observation of the next episode,
its `done` row is 1) — the convention of baselines-style VecEnvs; with `terminal_obs=True` the LAST frame of the finished
episode (what the reference returns with done = True, multi_car_racing.py:431, :509) is handed out as well, as
info["terminal_observation"][i] for env info["terminal_env_ids"][i], i < info["terminal_count"].

Like the reference, every env keeps ONE b2World for its life (multi_car_racing.py:138; _destroy :173-181, reset :341): from an env's second
episode on the fixtures' broadphase proxy ids come off the world's free list — they order same-step tile events, i.e. which of two cars that
reach a tile in the same step is its first visitor (`1000/T` vs `(1 - 1/N)·1000/T`, :113-120), and name fixtureA of a car<->car contact (the
manifold's reference face: poses once cars touch).  The ids need no tree on the device: leaf ids never depend on the tree's shape, a per-env
stack of free leaf ids reproduces them (csrc/k_world.h), advanced by the env's reset pass inside `step`.  `fresh_world=True` is rounds 1-5's
definition instead — every episode the first episode of a fresh world —; measured against the reference's semantics on the oracle
(profiles/r06_world_reuse_effect.txt, 5 episodes back to back): 50-64 % of the (env, car) pairs of an even episode see a different reward in
some step, and with a policy that drives the tile-visit counts of 4-17 % of the envs differ at the end of episodes 2-5.

Determinism: env with global index g uses two numpy-compatible MT19937 streams,
  track stream  RandomState(seed + g)                (the reference's `env.np_random`)
  draw stream   RandomState((seed + g + 2**31) % 2**32)   (stands in for the reference's *global* np.random:
                direction choice then car order, multi_car_racing.py:351-357)
so results do not depend on B, on the GPU count, or on scheduling.
"""
import atexit
import collections
import ctypes
import weakref
import os
import queue
import threading
import time

import numpy as np
import torch

from . import _lib

_DIRECTION_MODE = {"CCW": 0, "CW": 1}


_LIVE = weakref.WeakSet()


def _close_all():
    """Interpreter exit with envs still open (e.g. after an exception): stop the refill threads before the runtime is
    torn down — a worker inside the native generator at that moment would abort the process."""
    for env in list(_LIVE):
        try:
            env.close()
        except Exception:
            pass


atexit.register(_close_all)


class VecMultiCarRacing:
    def __init__(self, num_envs, num_agents=2, device=None, seed=0, env_offset=0, direction="CCW",
                 use_random_direction=True, backwards_flag=True, h_ratio=0.25, use_ego_color=False,
                 obs=True, auto_reset=True, max_episode_steps=1000, car_contacts=True,
                 gen_threads=None, async_refill=True, streams=None, refill_lag=64, world_size=1, graph=None,
                 skid_particles=False, terminal_obs=False, terminal_cap=None, fresh_world=False):
        if not torch.cuda.is_available():
            raise _lib.McrError("VecMultiCarRacing needs a HIP device: the step path has no CPU fallback")
        self.L = _lib.load()
        self.B, self.N = int(num_envs), int(num_agents)
        self.env_offset = int(env_offset)
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        torch.cuda.set_device(self.device)
        self.obs_enabled = bool(obs)
        if streams is None:           # contact side stream (include/mcr.h: num_streams) pays as soon as there is a batch
            streams = 2 if int(num_envs) >= 64 and int(num_agents) > 1 and car_contacts else 1
        self.auto_reset = bool(auto_reset)
        self.direction_mode = 2 if use_random_direction else _DIRECTION_MODE[direction]
        # host threads of the track generator: the ranks of one node share the cores the cgroup allows.  (No core is set aside for
        # the stepping thread: it sleeps at its fences — blocking events — and one generator thread per rank runs at ~85 % duty at
        # 12 M env-steps/s, too close to the edge: bench.py --emulate-world 8 measured 10.9 M with one thread on a 2-core share)
        self.gen_threads = gen_threads or max(1, _lib.effective_cpus() // max(1, int(world_size)))
        # A re-spawned env consumes its staged episode; the refill thread generates + stages the next one.  An env could
        # only FREEZE (k_dynamics: inactive, zero outputs until the episode arrives) if it finished a whole episode before
        # that refill landed, so step() waits for any refill batch queued more than `refill_lag` steps ago — fewer steps
        # than any episode can last (a car needs a few hundred steps to leave the playfield from the track; a TimeLimit
        # shorter than the lag shortens it) — and the freeze path stays a safety net (debug_counters()[3] counts its
        # env-steps).  (8 steps until round 3: at 0.28 ms per step that is 2 ms for generate + stage + a blocking event
        # on a 2-core host share, and step() sat in wait_refills() 80 % of the time with the cores 2/3 busy.)
        self.refill_lag = max(1, min(int(refill_lag), int(max_episode_steps) - 1)) if int(max_episode_steps) > 0 else max(1, int(refill_lag))
        self._step_idx = 0
        self.blocked_s = 0.0          # wall time step() spent waiting for the refill thread (host behind the device)
        self._pending = collections.deque()      # step index at which each queued refill batch was queued
        self._pending_lock = threading.Lock()
        self._worker_exc = None
        self._hold_refills = False    # tests: withhold staging to exercise the freeze/thaw path
        cfg = _lib.Config(self.B, self.N, self.device.index or 0, int(self.obs_enabled), int(self.auto_reset),
                          int(backwards_flag), int(use_ego_color), int(car_contacts), int(max_episode_steps), int(streams),
                          float(h_ratio), int(bool(skid_particles)), int(bool(fresh_world)))
        self.fresh_world = bool(fresh_world)
        self.h = ctypes.c_void_p()
        _lib.check(self.L.mcr_create(ctypes.byref(cfg), ctypes.byref(self.h)), "mcr_create")
        self._status_now = np.zeros(8, np.uint32); self._status_seen = np.zeros(8, np.uint32)
        self._bound_streams = set()   # caller streams already handed to mcr_bind_stream (the check synchronises the device: once per stream)
        if int(self.L.mcr_step_ordering(self.h)) & 4:
            import warnings
            warnings.warn("another VecMultiCarRacing of this process holds this device's phase-word ordering: this one orders the streams "
                          "of its step with events (same results, about 0.02 ms more per step)", _lib.McrWarning, stacklevel=2)
        if graph is None:             # hipGraph replay of the step: measured r02 at B=4096 — 0.433 vs 0.435 ms per step, i.e. the gaps
            graph = False             # between the step's dependent kernels are drain/start-up on the GPU, not host launch cost: off
        _lib.check(self.L.mcr_set_step_graph(self.h, int(bool(graph))), "mcr_set_step_graph")
        # persistent outputs (overwritten by every step)
        self.obs = torch.zeros((self.B, self.N, 96, 96, 3), dtype=torch.uint8, device=self.device) if self.obs_enabled else None
        self.reward = torch.zeros((self.B, self.N), dtype=torch.float64, device=self.device)
        self.done = torch.zeros((self.B,), dtype=torch.uint8, device=self.device)
        self.truncated = torch.zeros((self.B,), dtype=torch.uint8, device=self.device)
        # episode statistics, valid in the rows where `done` is set (overwritten when that env's next episode ends)
        self.episode_return = torch.zeros((self.B, self.N), dtype=torch.float64, device=self.device)
        self.episode_length = torch.zeros((self.B,), dtype=torch.int32, device=self.device)
        _lib.check(self.L.mcr_set_episode_stats(self.h, ctypes.c_void_p(self.episode_return.data_ptr()), ctypes.c_void_p(self.episode_length.data_ptr())), "mcr_set_episode_stats")
        # terminal observations (include/mcr.h: mcr_set_terminal_obs): the last frame of every episode that ends in a step, next to the first
        # frame of the next episode that the env's row of `obs` shows.  Compact: entry i belongs to env terminal_env_ids[i], i < terminal_count.
        self.terminal_obs = self.terminal_env_ids = self.terminal_count = None
        if terminal_obs:
            if not (self.obs_enabled and self.auto_reset):
                raise ValueError("terminal_obs needs obs=True and auto_reset=True")
            cap = self.B if terminal_cap is None else max(1, min(int(terminal_cap), self.B))
            self.terminal_obs = torch.zeros((cap, self.N, 96, 96, 3), dtype=torch.uint8, device=self.device)
            self.terminal_env_ids = torch.zeros((cap,), dtype=torch.int32, device=self.device)
            self.terminal_count = torch.zeros((1,), dtype=torch.int32, device=self.device)
            _lib.check(self.L.mcr_set_terminal_obs(self.h, ctypes.c_void_p(self.terminal_obs.data_ptr()), ctypes.c_void_p(self.terminal_env_ids.data_ptr()),
                                                   ctypes.c_void_p(self.terminal_count.data_ptr()), cap), "mcr_set_terminal_obs")
        # RNG streams
        self.mt_track = np.zeros((self.B, _lib.MT_WORDS), np.uint32)
        self.mt_draw = np.zeros((self.B, _lib.MT_WORDS), np.uint32)
        for e in range(self.B):
            g = (int(seed) + int(env_offset) + e) % 2 ** 32
            self.L.mcr_mt_seed(_lib.ptr(self.mt_track[e]), ctypes.c_uint32(g))
            self.L.mcr_mt_seed(_lib.ptr(self.mt_draw[e]), ctypes.c_uint32((g + 2 ** 31) % 2 ** 32))
        self.slot_bytes = _lib.episode_bytes()
        self._blobs = torch.empty((self.B, self.slot_bytes), dtype=torch.uint8, pin_memory=True)
        self._blobs_np = self._blobs.numpy()
        self._refill_pin = None       # pinned bounce buffer of the refill thread (grown on demand)
        self.episode_info = np.zeros((self.B, 12), np.int32)      # T, P, retries, cw, car_order[8] of the newest generated episode
        self._ids = np.zeros(self.B, np.int32)
        self._episodes_generated = 0
        self._copy_stream = torch.cuda.Stream(device=self.device)
        self._copy_done = torch.cuda.Event(blocking=True)
        # async_refill: True (default) — the handle's own native thread polls, generates and stages (include/mcr.h: mcr_refill_start; no
        # interpreter in the loop: 2.1 -> 1.x host cores per rank, bench.py --emulate-world); "python" — rounds 2-5's worker thread in this
        # module (kept for comparison); False — synchronously inside step() (tests that need the staging at a known point)
        self._async = bool(async_refill)
        self._native = async_refill is True or async_refill == "native"
        self._svc = False             # the native service is running
        self._q = None
        self._worker = None
        self._closed = False
        self._has_reset = False
        _LIVE.add(self)

    # ------------------------------------------------------------------ episode generation / staging
    def _generate(self, ids):
        """Generate each listed env's next episode (advances its RNG streams).  Returns the pinned, contiguous array
        [len(ids), slot_bytes] holding the blobs in the order of `ids` — what `_stage` uploads; it stays valid until the
        next `_generate` call of the same thread."""
        ids = np.ascontiguousarray(ids, np.int32)
        n = len(ids)
        if n == 0:
            return None
        mt_t = np.ascontiguousarray(self.mt_track[ids]); mt_d = np.ascontiguousarray(self.mt_draw[ids])
        if n != self.B:               # subset: generate into a contiguous pinned buffer so that staging is ONE call
            if self._refill_pin is None or self._refill_pin.shape[0] < n:
                self._refill_pin = torch.empty((max(n, 64), self.slot_bytes), dtype=torch.uint8, pin_memory=True)
            blobs = self._refill_pin.numpy()[:n]
        else:
            blobs = self._blobs_np
        info = np.zeros((n, 12), np.int32)
        _lib.check(self.L.mcr_episodes_generate(_lib.ptr(mt_t), _lib.ptr(mt_d), n, self.N, self.direction_mode,
                                                _lib.ptr(blobs), _lib.ptr(info), min(self.gen_threads, n)), "mcr_episodes_generate")
        self.mt_track[ids] = mt_t; self.mt_draw[ids] = mt_d
        self.episode_info[ids] = info
        if n != self.B:
            self._blobs_np[ids] = blobs              # per-env copy for introspection (current_episode / facade env.track)
        self._episodes_generated += n
        return blobs

    def _stage(self, ids, rows, stream):
        """Upload `rows` (from `_generate(ids)`) into the staged device slots of envs `ids` (async on `stream`)."""
        ids = np.ascontiguousarray(ids, np.int32)
        if len(ids) == 0:
            return
        _lib.check(self.L.mcr_stage_episodes(self.h, _lib.ptr(ids), len(ids), _lib.ptr(rows), ctypes.c_void_p(stream.cuda_stream)), "mcr_stage_episodes")

    def _refill(self, ids):
        rows = self._generate(ids)
        self._stage(ids, rows, self._copy_stream)
        # the bounce buffer may be overwritten as soon as the copies are done.  Polled with a sleep in between: hipStreamSynchronize
        # spins, and so does hipEventSynchronize on a "blocking" event (measured: 0.8 of a core in bench.py's look-ahead fence); a
        # thread that spins takes a core from the track generator — with the ranks of a node sharing few cores
        # (bench.py --emulate-world) that is what the host side runs out of first
        self._copy_done.record(self._copy_stream)
        while not self._copy_done.query():          # (not .synchronize(): see above — it spins on this runtime, blocking event or not)
            time.sleep(5e-5)

    def _worker_main(self):
        try:                                                   # name the thread for top / bench.py's per-thread CPU report (PR_SET_NAME)
            ctypes.CDLL(None).prctl(15, b"mcr-refill", 0, 0, 0)
        except Exception:
            pass
        torch.cuda.set_device(self.device)
        while True:
            ids = self._q.get()
            if ids is None:
                self._q.task_done()
                return
            batch, taken, stop = [ids], 1, False
            while True:               # drain: everything queued meanwhile goes into the same generate + stage batch
                try:
                    more = self._q.get_nowait()
                except queue.Empty:
                    break
                taken += 1
                if more is None:
                    stop = True
                    break
                batch.append(more)
            try:
                if self._worker_exc is None:
                    self._refill(np.concatenate(batch))
            except BaseException as exc:   # surfaced by wait_refills()/step(); the queue must still drain or join() hangs
                self._worker_exc = exc
            finally:
                with self._pending_lock:
                    for _ in range(len(batch)):
                        if self._pending:
                            self._pending.popleft()
                for _ in range(taken):
                    self._q.task_done()
            if stop:
                return

    def _raise_worker_error(self):
        if self._worker_exc is not None:
            exc, self._worker_exc = self._worker_exc, None
            raise _lib.McrError(f"episode refill thread failed: {exc!r}") from exc

    @property
    def episodes_generated(self):
        return self._episodes_generated + (int(self.L.mcr_refill_generated(self.h)) if self._svc else 0)

    @property
    def hold_refills(self):
        return self._hold_refills

    @hold_refills.setter
    def hold_refills(self, v):
        self._hold_refills = bool(v)
        if self._svc:
            _lib.check(self.L.mcr_refill_hold(self.h, int(self._hold_refills)), "mcr_refill_hold")

    def _poll_and_refill(self):
        if self._native:
            if not self._svc:         # (started after the first reset()'s own staging: from here on the RNG states and the blob rows are the service's)
                _lib.check(self.L.mcr_refill_start(self.h, _lib.ptr(self.mt_track), _lib.ptr(self.mt_draw), self.direction_mode, self.gen_threads,
                                                   ctypes.c_void_p(self._blobs.data_ptr()), _lib.ptr(self.episode_info)), "mcr_refill_start")
                self._svc = True
                if self._hold_refills:
                    _lib.check(self.L.mcr_refill_hold(self.h, 1), "mcr_refill_hold")
            return 0
        if self.hold_refills:
            return 0
        n = self.L.mcr_poll_consumed(self.h, _lib.ptr(self._ids), self.B, None)
        if n <= 0:
            return 0
        ids = self._ids[:n].copy()
        if self._async:
            if self._worker is None:
                self._q = queue.Queue()
                self._worker = threading.Thread(target=self._worker_main, daemon=True)
                self._worker.start()
            with self._pending_lock:
                self._pending.append(self._step_idx)
            self._q.put(ids)
        else:
            self._refill(ids)
        return n

    def wait_refills(self):
        if self._svc:
            _lib.check(self.L.mcr_refill_wait(self.h), "mcr_refill_wait")
            return
        if self._q is not None:
            self._q.join()
        self._raise_worker_error()

    def _settle_staging(self, st):
        """Every env that consumed its staged episode gets the next one staged NOW (stream drained, consumption polled,
        refills finished): afterwards k_install finds `staged_ready` set for every env it is asked to reset."""
        st.synchronize()
        self._poll_and_refill()
        self.wait_refills()

    # ------------------------------------------------------------------ API
    def reset(self):
        """Reset every env; returns obs [B,N,96,96,3] uint8 (device tensor, overwritten by later steps)."""
        st = torch.cuda.current_stream(self.device)
        self.wait_refills()
        if not self._has_reset:
            every = np.arange(self.B, dtype=np.int32)
            self._stage(every, self._generate(every), st)
        else:
            # envs re-spawned by steps whose consumption has not been polled yet would find their staged slot empty
            # (k_install skips those): drain, poll and refill first, then every env installs a fresh episode
            self._settle_staging(st)
        _lib.check(self.L.mcr_reset(self.h, None, ctypes.c_void_p(self.obs.data_ptr()) if self.obs_enabled else None,
                                    ctypes.c_void_p(st.cuda_stream)), "mcr_reset")
        st.synchronize()
        self._has_reset = True
        self._poll_and_refill()
        return self.obs

    def reset_envs(self, mask):
        """Reset the envs whose byte in `mask` (uint8 device tensor [B]) is non-zero: each installs its staged
        episode (reference reset(), :340-408) and gets its first observation written into `self.obs`."""
        if not self._has_reset:
            raise AttributeError("reset_envs() before reset()")
        st = torch.cuda.current_stream(self.device)
        if mask.dtype != torch.uint8 or mask.device != self.device or mask.numel() != self.B or not mask.is_contiguous():
            mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
            if mask.numel() != self.B:
                raise ValueError(f"mask must have {self.B} elements")
        self._settle_staging(st)              # a masked env must find its staged slot filled
        _lib.check(self.L.mcr_reset(self.h, ctypes.c_void_p(mask.data_ptr()),
                                    ctypes.c_void_p(self.obs.data_ptr()) if self.obs_enabled else None,
                                    ctypes.c_void_p(st.cuda_stream)), "mcr_reset")
        st.synchronize()
        self._poll_and_refill()
        return self.obs

    def step(self, actions):
        """actions: float32 device tensor [B,N,3] (steer, gas, brake) or None. Returns (obs, reward, done, info)."""
        st = torch.cuda.current_stream(self.device)
        self._raise_worker_error()
        if self._svc:
            lag = int(self.L.mcr_refill_lag(self.h))
            if lag < 0:
                _lib.check(lag, "mcr_refill_lag")
            behind = lag >= max(1, self.refill_lag - 4)      # (the service notices a consumption up to a few steps after it happened)
        else:
            with self._pending_lock:          # (the refill worker pops entries under the same lock)
                behind = bool(self._pending) and self._step_idx - self._pending[0] >= self.refill_lag
        if behind:
            t0 = time.perf_counter()
            self.wait_refills()               # the host fell behind: block instead of letting an env freeze
            self.blocked_s += time.perf_counter() - t0
        if st.cuda_stream not in self._bound_streams:      # first step on this stream: may it use the phase-word ordering? (synchronises, once)
            _lib.check(self.L.mcr_bind_stream(self.h, ctypes.c_void_p(st.cuda_stream)), "mcr_bind_stream")
            self._bound_streams.add(st.cuda_stream)
        a_ptr = None
        if actions is not None:
            if actions.dtype != torch.float32 or not actions.is_contiguous() or actions.device != self.device:
                actions = actions.to(device=self.device, dtype=torch.float32).contiguous()
            if actions.numel() != self.B * self.N * 3:
                raise ValueError(f"actions must have {self.B * self.N * 3} elements, got {actions.numel()}")
            a_ptr = ctypes.c_void_p(actions.data_ptr())
        _lib.check(self.L.mcr_step(self.h, a_ptr, ctypes.c_void_p(self.obs.data_ptr()) if self.obs_enabled else None,
                                   ctypes.c_void_p(self.reward.data_ptr()), ctypes.c_void_p(self.done.data_ptr()),
                                   ctypes.c_void_p(self.truncated.data_ptr()), ctypes.c_void_p(st.cuda_stream)), "mcr_step")
        self._step_idx += 1                   # (only a step that was launched counts: a reported McrError leaves the accounting alone)
        self._warn_degraded()
        if self.auto_reset:
            self._poll_and_refill()
        info = {"TimeLimit.truncated": self.truncated, "episode_return": self.episode_return, "episode_length": self.episode_length}
        if self.terminal_obs is not None:     # (device tensors, like everything else here: entry i < terminal_count is env terminal_env_ids[i])
            info["terminal_observation"] = self.terminal_obs
            info["terminal_env_ids"] = self.terminal_env_ids
            info["terminal_count"] = self.terminal_count
        return self.obs, self.reward, self.done, info

    def terminal_observations(self):
        """(env ids [k], frames [k, N, 96, 96, 3]) of the episodes that ended in the last step — what the reference returns as its observation
        with done = True (multi_car_racing.py:431, :509) — as device tensors; synchronises (reads the count)."""
        if self.terminal_obs is None:
            raise _lib.McrError("created without terminal_obs=True")
        k = int(self.terminal_count.item())
        return self.terminal_env_ids[:k], self.terminal_obs[:k]

    _DEGRADED = {2: "more touching car<->car fixture pairs in an env than the manifold store holds (MCR_CC_MAX): the excess contacts were dropped — that env's "
                    "physics deviates from the reference from here on",
                 3: "more tile begin events in one env-step than the replay buffer holds: the excess events (tile visits, rewards) were dropped",
                 4: "an env ended its episode before the host had staged its next one and froze (zero reward, done = 0, stale frame) until the episode "
                    "arrived: the track generator is behind the device (more gen_threads, async_refill=True, a longer TimeLimit, or fence the stepping loop)"}

    def _warn_degraded(self):
        """The capacity / starvation conditions the kernels count without failing (include/mcr.h: mcr_status words 2..4): say so, once per change.
        Reads mapped host memory — no synchronisation; a condition raised by a step still in flight surfaces a call later."""
        _lib.check(self.L.mcr_status(self.h, _lib.ptr(self._status_now), 8), "mcr_status")
        for w, what in self._DEGRADED.items():
            if self._status_now[w] != self._status_seen[w]:
                import warnings
                warnings.warn(f"{what} ({int(self._status_now[w])} so far)", _lib.McrWarning, stacklevel=3)
                self._status_seen[w] = self._status_now[w]

    def debug_counters(self):
        """cumulative [envs deferred, envs resumed, contact envs routed to the side stream, env-steps spent frozen
        waiting for a staged episode] (synchronises)"""
        out = np.zeros(4, np.uint64)
        _lib.check(self.L.mcr_debug_read_counters(self.h, _lib.ptr(out)), "mcr_debug_read_counters")
        return out


    def verdict_mismatches(self):
        """envs in which the contact pass disagreed with the touch verdict the main launches went by (must be 0; synchronises)"""
        out = np.zeros(1, np.uint64)
        _lib.check(self.L.mcr_debug_read_verdict_mismatches(self.h, _lib.ptr(out)), "mcr_debug_read_verdict_mismatches")
        return int(out[0])

    def status_words(self):
        """cumulative status words of include/mcr.h `mcr_status`: [0] in-kernel waits given up, [1] touch-verdict mismatches,
        [2] car<->car manifold overflows, [3] begin-event queue overflows, [4] envs that froze waiting for a staged episode (all 0 in a healthy
        rollout; does not synchronise)"""
        out = np.zeros(8, np.uint32)
        _lib.check(self.L.mcr_status(self.h, _lib.ptr(out), 8), "mcr_status")
        return out

    def rollout_stats(self, reset=False):
        """(episodes finished, sum of their returns over all agents) accumulated on the device; synchronises."""
        out = np.zeros(2)
        _lib.check(self.L.mcr_read_rollout_stats(self.h, _lib.ptr(out), int(bool(reset))), "mcr_read_rollout_stats")
        return float(out[0]), float(out[1])

    def render_rgb(self, e=0, width=600, height=400):
        """render('rgb_array') of env e: uint8 device tensor [N, height, width, 3] of its CURRENT state (reference
        :511-604 with the VIDEO_W x VIDEO_H viewport), score label included; skid particles (Car.draw(viewer, True), :564) are
        drawn when the env was created with skid_particles=True (the facade does; batched envs leave them off)."""
        if not self._has_reset:
            raise AttributeError("render before reset()")
        out = torch.empty((self.N, int(height), int(width), 3), dtype=torch.uint8, device=self.device)
        st = torch.cuda.current_stream(self.device)
        _lib.check(self.L.mcr_render(self.h, int(e), int(width), int(height), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(st.cuda_stream)), "mcr_render")
        return out

    # ------------------------------------------------------------------ introspection (synchronous; tests / debugging)
    def get_state(self):
        B, N = self.B, self.N
        bodies = np.zeros((B, N, 5, 6), np.float32); joints = np.zeros((B, N, 4, 4), np.float32)
        wheels = np.zeros((B, N, 4, 5), np.float64); limit = np.zeros((B, N, 4), np.int32)
        on_road = np.zeros((B, N, 4), np.uint8); sleep = np.zeros((B, N, 5), np.float32)
        _lib.check(self.L.mcr_get_state(self.h, _lib.ptr(bodies), _lib.ptr(joints), _lib.ptr(wheels), _lib.ptr(limit),
                                        _lib.ptr(on_road), _lib.ptr(sleep)), "mcr_get_state")
        return dict(bodies=bodies, joints=joints, wheels=wheels, limit=limit, on_road=on_road, sleep=sleep)

    def set_bodies(self, bodies):
        b = np.ascontiguousarray(bodies, np.float32)
        assert b.shape == (self.B, self.N, 5, 6)
        _lib.check(self.L.mcr_set_bodies(self.h, _lib.ptr(b)), "mcr_set_bodies")

    def get_env_state(self):
        B, N = self.B, self.N
        reward = np.zeros((B, N)); tvc = np.zeros((B, N), np.int32)
        bw = np.zeros((B, N), np.uint8); og = np.zeros((B, N), np.uint8); t = np.zeros(B)
        flags = np.zeros((B, _lib.TILE_CAP), np.uint16); nt = np.zeros(B, np.int32)
        _lib.check(self.L.mcr_get_env_state(self.h, _lib.ptr(reward), _lib.ptr(tvc), _lib.ptr(bw), _lib.ptr(og), _lib.ptr(t),
                                            _lib.ptr(flags), _lib.ptr(nt)), "mcr_get_env_state")
        return dict(reward=reward, tile_visited_count=tvc, driving_backward=bw, driving_on_grass=og, t=t,
                    tile_flags=flags, num_tiles=nt)

    def get_state_blob(self, e):
        """Full snapshot of env e (uint8 host array): everything `step` reads — see include/mcr.h mcr_get_state_blob."""
        blob = np.zeros(int(self.L.mcr_state_blob_bytes(self.h)), np.uint8)
        _lib.check(self.L.mcr_get_state_blob(self.h, int(e), _lib.ptr(blob)), "mcr_get_state_blob")
        return blob

    def set_state_blob(self, e, blob):
        """Restore a snapshot into env e (any env index, any handle with the same num_agents); stepping continues
        bit-identically from it."""
        blob = np.ascontiguousarray(blob, np.uint8)
        _lib.check(self.L.mcr_set_state_blob(self.h, int(e), _lib.ptr(blob)), "mcr_set_state_blob")
        self._has_reset = self._has_reset or True

    def synth_actions(self, t, seed=0, out=None, steps=None):
        """Counter-based synthetic actions (bench/tests): a pure function of (seed, global env index, agent, t).  Step t as a
        device tensor [B,N,3] f32, or — with `steps` — the steps t .. t+steps-1 in one launch as [steps,B,N,3]."""
        n = 1 if steps is None else int(steps)
        if out is None:
            out = torch.empty(((self.B, self.N, 3) if steps is None else (n, self.B, self.N, 3)), dtype=torch.float32, device=self.device)
        st = torch.cuda.current_stream(self.device)
        _lib.check(self.L.mcr_synth_actions_block(self.h, ctypes.c_void_p(out.data_ptr()), ctypes.c_uint64(int(seed)), ctypes.c_uint32(int(t) & 0xffffffff),
                                                  n, ctypes.c_uint32(self.env_offset), ctypes.c_void_p(st.cuda_stream)), "mcr_synth_actions_block")
        return out

    def positions(self):
        pos = np.zeros((self.B, self.N, 2), np.float32)
        _lib.check(self.L.mcr_get_positions(self.h, _lib.ptr(pos)), "mcr_get_positions")
        return pos

    def current_episode(self, e):
        """Host copy of the NEWEST generated episode of env e (the staged one once the env has reset)."""
        return _lib.unpack_episode(np.ascontiguousarray(self._blobs_np[e]))

    def timing(self, mask):
        """HIP-event kernel timing; mask bit 0 collide, 1 dynamics, 2 view, 3/4 reset-pass collide/dynamics,
        5/6 dynamics/view of the contact side stream, 7 its reset pass."""
        _lib.check(self.L.mcr_timing_enable(self.h, int(mask)))

    def timing_read(self):
        ms = np.zeros(8); n = np.zeros(8, np.int64)
        _lib.check(self.L.mcr_timing_read(self.h, _lib.ptr(ms), _lib.ptr(n)))
        return ms, n

    def close(self):
        if self._closed:
            return
        self._closed = True
        if self._worker is not None:
            self._q.put(None)
            self._worker.join()
        if self._svc and self.h:
            self.L.mcr_refill_stop(self.h)
            self._svc = False
        if self.h:
            self.L.mcr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
