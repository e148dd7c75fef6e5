/* include/mcr.h — C ABI of the MI355X-native MultiCarRacing-v0 batched step.
 *
 * The reference has no FFI of its own for this path: its native boundary is SWIG'd Box2D
 * (`Box2D.b2World.Step`, multi_car_racing.py:428) plus ctypes OpenGL (`gl.glVertex3f`, :613-674).
 * This header is the boundary a maintainer would bind instead (INTEGRATION.md shows the ctypes stub);
 * each entry point names the reference interface it replaces.
 *
 * Conventions: plain pointers and sizes only, `int` status returns (0 = MCR_OK, <0 = error enum),
 * nothing throws across the boundary, no exit().  `d_*` pointers are DEVICE pointers on the handle's
 * HIP device; `stream` is a `hipStream_t` passed as `void*` (NULL = the null stream).  Kernels are only
 * enqueued — no hidden device synchronisation in mcr_step / mcr_reset (the one check that needs one is mcr_bind_stream).
 * One handle = one device = one env slice; handles are independent (thread-compatible).
 */
#ifndef MCR_H
#define MCR_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MCR_OK 0
#define MCR_ERR_ARG (-1)      /* bad argument (NULL, range) */
#define MCR_ERR_HIP (-2)      /* a HIP runtime call failed; see mcr_last_error() */
#define MCR_ERR_STATE (-3)    /* call order violated (e.g. step before reset: AttributeError in the reference) */
#define MCR_ERR_CAPACITY (-4) /* track larger than MCR_TILE_CAP / MCR_QUAD_CAP */

#define MCR_MAX_AGENTS 8      /* CAR_COLORS has 8 entries (multi_car_racing.py:67-70) */
#define MCR_TILE_CAP 512      /* tiles per track: observed 239..379 */
#define MCR_QUAD_CAP 768      /* road_poly entries: observed 279..460 */
#define MCR_OBS_H 96
#define MCR_OBS_W 96
#define MCR_MT_WORDS 625      /* MT19937 key[624] + pos — numpy RandomState layout */

typedef struct mcr_env mcr_env; /* opaque */

/* ctor kwargs of MultiCarRacing.__init__ (multi_car_racing.py:131-133) + batch/device knobs */
typedef struct mcr_config {
  int32_t num_envs;          /* B: environments in this slice */
  int32_t num_agents;        /* N: 1..8 */
  int32_t device;            /* HIP device ordinal */
  int32_t obs_enabled;       /* 0: physics only ("obs=none"); 1: 96x96x3 state_pixels per agent */
  int32_t auto_reset;        /* 1: finished envs are re-spawned inside mcr_step from their staged episode */
  int32_t backwards_flag;    /* :158 */
  int32_t use_ego_color;     /* :160 */
  int32_t car_contacts;      /* 1: car<->car rigid contacts (Box2D default); 0: ghost cars (debug) */
  int32_t max_episode_steps; /* gym TimeLimit from __init__.py:8 (1000); 0 disables */
  int32_t num_streams;       /* 0/1: every kernel on the caller's stream; 2: envs whose dynamics chain is long (a touching car<->car pair,
                              * or a position loop still iterating after 2 sweeps) run it + their raster on internal streams, concurrently
                              * with the others; results are bit-identical in both modes */
  double h_ratio;            /* :159 */
  int32_t skid_particles;    /* 1: keep the skid particles of gym car_dynamics.Car (step(): "Skid trace", _create_particle) so that
                              * mcr_render can draw them (Car.draw(viewer, True), :564); 0: not tracked (observations never show them) */
  int32_t fresh_world;       /* 0 (default): ONE b2World per env for the env's life, as the reference keeps it (:138; _destroy :173-181, reset :341): from an
                              * env's second episode on the fixtures' proxy ids come off the world's free list, which orders same-step tile events
                              * (:113-120: who is a tile's first visitor) and names fixtureA of a car<->car contact.  The ids' rule needs no tree
                              * (csrc/k_world.h): a per-env stack of free leaf ids, advanced by the env's reset pass on the device.
                              * 1: every episode is the first episode of a fresh world (rounds 1-5; the oracle's world mode 0) */
} mcr_config;

const char* mcr_last_error(void);
const char* mcr_version(void);

/* ---- lifetime.  Replaces MultiCarRacing.__init__ (:131-166) / close (:606-611). */
int mcr_create(const mcr_config* cfg, mcr_env** out);
int mcr_destroy(mcr_env* h);

/* ---- host-side episode setup (NO GPU needed).  Replaces reset()'s RNG draws + _create_track + spawn
 * maths (:349-406, :183-338).  An "episode blob" is the host image of one env's device track slot. */
size_t mcr_episode_bytes(void);
/* numpy-compatible MT19937 helpers (RandomState.seed(int) / .seed(array), legacy stream). */
void mcr_mt_seed(uint32_t* mt, uint32_t seed);
void mcr_mt_seed_by_array(uint32_t* mt, const uint32_t* key, int key_len);
double mcr_mt_random_sample(uint32_t* mt);
/* np.random.choice(['CW','CCW']) (:352) -> 1 if 'CW';  np.random.choice(ids,size=N,replace=False) (:356) */
int mcr_mt_choice_cw(uint32_t* mt);
void mcr_mt_car_order(uint32_t* mt, int n, int32_t* order_out);
/* One episode: retries _create_track until success (:359-364), builds tiles/quads/spawn poses.
 * info_out[0..3] = T, P, retries, cw.  `mt_track` advances exactly as env.np_random does. */
int mcr_episode_generate(uint32_t* mt_track, int num_agents, int cw, const int32_t* car_order,
                         void* blob_out, int32_t* info_out);
/* Batched + threaded: per env i draws direction (direction_mode 2) and car order from mt_global[i],
 * then the track from mt_track[i].  direction_mode: 0 'CCW', 1 'CW', 2 random per episode. */
int mcr_episodes_generate(uint32_t* mt_track, uint32_t* mt_global, int n, int num_agents,
                          int direction_mode, void* blobs_out, int32_t* info_out, int num_threads);
/* ... for rows ids[0 .. n) of PER-ENV arrays (mt_track_all / mt_global_all [num_envs][MCR_MT_WORDS], blobs_all [num_envs][mcr_episode_bytes()],
 * info_all [num_envs][12] or NULL), in place. */
int mcr_episodes_generate_rows(uint32_t* mt_track_all, uint32_t* mt_global_all, const int32_t* ids, int n, int num_agents,
                               int direction_mode, void* blobs_all, int32_t* info_all, int num_threads);
/* Read-only views into a blob (tests / facade attributes such as env.track). */
int mcr_episode_unpack(const void* blob, int32_t* T, int32_t* P, int32_t* cw, double* track_xyb /*[T*3]*/,
                       float* quads /*[P*8]*/, uint32_t* quad_meta /*[P]*/, double* spawn /*[8*3]*/,
                       double* track_alpha /*[T]*/);

/* ---- the env's b2World across reset() as a LITERAL host-side tree (NO GPU needed).  The reference keeps ONE world for the life of an env
 * (multi_car_racing.py:138; _destroy :173-181 and reset :341 destroy and re-create its bodies): from the second episode on, the fixtures'
 * broadphase proxy ids come off the dynamic tree's free list, and the ids order the contact callbacks of a step — which of two cars that reach a
 * tile in the same step is its first visitor (:113-120) — and name fixtureA of a car<->car contact.  Since round 6 every handle carries that world
 * itself (mcr_config::fresh_world = 0: a per-env stack of free leaf ids on the device, csrc/k_world.h — a proxy's leaf id never depends on the
 * tree's shape); this host-side dynamic AABB tree with Box2D's insertion / balance / free-list rules (rounds 4-5's facade used it) stays as the
 * independent twin the tests hold the stack rule against (tests/test_world_ids.py), and for callers of a fresh_world = 1 handle who want to supply
 * the ids themselves: mcr_world_reset(w, blob) before staging an episode blob (destroys the old episode's proxies and creates the new one's in the
 * reference's order; writes the ids into the blob, header word pad0 = 1, where the contact pass of a fresh_world = 1 handle finds them),
 * mcr_world_step(w, bodies) after EVERY step and after the reset's own step, with the bodies of mcr_get_state (b2Body::SynchronizeFixtures ->
 * b2DynamicTree::MoveProxy in b2World::Solve's order). */
typedef struct mcr_world mcr_world;
mcr_world* mcr_world_create(int num_agents);
void mcr_world_destroy(mcr_world* w);
int mcr_world_reset(mcr_world* w, void* blob_io);
int mcr_world_step(mcr_world* w, const float* bodies /*[N,5,6]*/);
int mcr_world_proxy_ids(const mcr_world* w, int32_t* out, int cap);

/* ---- device side */
/* Copy n host blobs into the STAGED slot of envs env_ids[0..n) (async on stream; blobs must stay valid
 * until the stream reaches this point — use pinned memory for real overlap). */
int mcr_stage_episodes(mcr_env* h, const int32_t* env_ids, int n, const void* blobs, void* stream);
/* reset() (:340-408): installs the staged episode for every env whose d_env_mask byte != 0 (NULL = all),
 * spawns the cars, runs the no-action step of :408 and writes the first observation.
 * d_obs: [B,N,96,96,3] u8 or NULL. */
int mcr_reset(mcr_env* h, const uint8_t* d_env_mask, uint8_t* d_obs, void* stream);
/* step() (:410-509) for all B envs.
 *   d_actions  [B,N,3] f32 (steer,gas,brake), NULL = the action-less step of :408
 *   d_obs      [B,N,96,96,3] u8 or NULL (ignored when obs_enabled == 0)
 *   d_reward   [B,N] f64 step_reward (:443,:507)
 *   d_done     [B] u8   (:498-499, :503-506, TimeLimit)
 *   d_trunc    [B] u8 or NULL: info['TimeLimit.truncated']
 * With auto_reset, a finished env's obs row is the first observation of its next episode (its last frame: mcr_set_terminal_obs). */
int mcr_step(mcr_env* h, const float* d_actions, uint8_t* d_obs, double* d_reward, uint8_t* d_done,
             uint8_t* d_trunc, void* stream);
/* render(mode) at another viewport (multi_car_racing.py:511-604 with VP_W x VP_H of :573-586; 'rgb_array' = 600 x 400):
 * the CURRENT state of env `env` as seen by each of its agents, d_out [N, height, width, 3] u8 (device), rows top-down.
 * Needs obs_enabled. */
int mcr_render(mcr_env* h, int env, int width, int height, uint8_t* d_out, void* stream);
/* Episode statistics (what gym's RecordEpisodeStatistics would add): in the step that ends an env's
 * episode (done), mcr_step writes the sum of the step rewards of that episode per agent into d_ep_return[B,N] and
 * its length in steps into d_ep_len[B]; other rows are left untouched.  NULL disables either. */
int mcr_set_episode_stats(mcr_env* h, double* d_ep_return, int32_t* d_ep_len);
/* Terminal observations (what a baselines / SB3-style VecEnv hands out as info["terminal_observation"]).  The reference renders
 * the state AFTER the last solve of an episode and returns it with done = True (multi_car_racing.py:431, :509; TimeLimit: __init__.py:8); with
 * auto_reset the env's row of d_obs already shows the first observation of the next episode.  With this set, every step that ends an env's
 * episode and re-spawns it also draws that last frame: entry i = env d_term_ids[i], frames d_term_obs[i] [N,96,96,3]; *d_term_count = the
 * number of entries of the last completed step (0 when no episode ended), at most `cap` — episodes that end beyond `cap` in one step get no
 * entry (cap = num_envs never drops one).  The three buffers are the caller's device memory, overwritten by every mcr_step with actions and
 * valid once that step is complete on its stream.  An env that ends while the host has not staged its next episode FREEZES instead of being
 * re-spawned (mcr_debug_read_counters [3]) and gets no entry.  All NULL: off.  Synchronises the device (allocates the entries' state). */
int mcr_set_terminal_obs(mcr_env* h, uint8_t* d_term_obs, int32_t* d_term_ids, int32_t* d_term_count, int cap);
/* Rollout statistics accumulated on the device since creation / the last reset of the counters (synchronises):
 * out2[0] = episodes finished, out2[1] = sum of their returns over all agents.  These are the per-rank inputs of the
 * job-wide metric all-reduce. */
int mcr_read_rollout_stats(mcr_env* h, double* out2, int reset);
/* Replay mcr_step as a hipGraph (1) or as plain launches (0, default).  A step is a fixed sequence of launches whose
 * arguments change only with an internal parity; the graph is (re)captured whenever an argument of mcr_step differs from
 * the captured call (buffers, stream) and bypassed while kernel timing is enabled.  Results are identical; the measured
 * gain at B=4096 is 0.4 % (the gaps between dependent kernels are not host launch cost), so it is off by default. */
int mcr_set_step_graph(mcr_env* h, int enable);
/* Envs whose staged episode was consumed (installed by a reset or an auto-reset) since the last poll: the host must
 * stage a fresh one for each.  Reads per-env install counters the kernels write to mapped host memory — no device
 * synchronisation, `stream` is unused; an install whose kernel has not finished yet shows up in a later poll.
 * Writes up to `cap` env ids; returns the count (>=0) or an error. */
int mcr_poll_consumed(mcr_env* h, int32_t* env_ids_out, int cap, void* stream);
/* The refill service: host threads owned by the handle (one service thread + up to min(gen_threads, 6) generator threads that stay awake while
 * there is work) do what a stepping loop would do with the three calls above after every step —
 * poll the consumed-episode counters, generate the next episode of every env that re-spawned from ITS rows of the caller's RNG-state arrays
 * (mt_track / mt_draw [num_envs][MCR_MT_WORDS], advanced in place), write the blob into ITS row of blobs_pinned [num_envs][mcr_episode_bytes()]
 * (page-locked host memory: the staging source) and episode_info [num_envs][12] (T, P, retries, cw, car_order[8]; may be NULL), stage it.
 * The reference generates a track inside reset() on the stepping thread (:359-364); here that work is off the stepping loop, in native
 * code (no interpreter: bench.py --emulate-world).  The arrays belong to the caller and must outlive the service; while it runs,
 * mcr_poll_consumed is refused and the caller must not call mcr_stage_episodes.  mcr_destroy stops it.
 * mcr_refill_wait: every consumption visible now is staged on return (synchronise the stepping stream first to mean "all").
 * mcr_refill_lag: steps launched since the oldest not-yet-staged consumption was noticed (0: none) — a stepping loop that finds it near
 * the shortest possible episode should wait instead of letting an env freeze.  mcr_refill_hold(1): the service ignores consumptions (tests). */
int mcr_refill_start(mcr_env* h, uint32_t* mt_track, uint32_t* mt_draw, int direction_mode, int gen_threads, void* blobs_pinned, int32_t* episode_info);
int mcr_refill_stop(mcr_env* h);
int mcr_refill_wait(mcr_env* h);
int mcr_refill_lag(mcr_env* h);
int mcr_refill_hold(mcr_env* h, int hold);
long long mcr_refill_generated(mcr_env* h);
/* diagnostics of the last mcr_refill_wait: tracks queued / being generated / finished-but-unstaged at entry, service cycles run, tracks the waiting
 * thread generated itself, microseconds inside the cycles (polling + staging), microseconds in total, generator threads */
int mcr_refill_debug(mcr_env* h, long long* out8);

/* ---- state access for differential tests / facade attributes (synchronous) */
/* bodies [B,N,5,6] f32 (c.x c.y angle v.x v.y w; body 0 hull, 1..4 wheels FL FR RL RR)
 * joints [B,N,4,4] f32 (impulse x y z, motorImpulse); wheels [B,N,4,5] f64 (gas brake steer phase omega)
 * limit [B,N,4] i32; on_road [B,N,4] u8; sleep [B,N,5] f32.  Any pointer may be NULL. */
int mcr_get_state(mcr_env* h, float* bodies, float* joints, double* wheels, int32_t* limit, uint8_t* on_road,
                  float* sleep);
int mcr_set_bodies(mcr_env* h, const float* bodies /*[B,N,5,6]*/);
/* reward [B,N] f64 (self.reward), tile_visited_count [B,N] i32, backward/on_grass [B,N] u8,
 * t [B] f64, tile_flags [B,MCR_TILE_CAP] u16 (bits 0..7 road_visited per agent, bit 8 recoloured) */
int mcr_get_env_state(mcr_env* h, double* reward, int32_t* tile_visited_count, uint8_t* backward,
                      uint8_t* on_grass, double* t, uint16_t* tile_flags, int32_t* num_tiles);
/* hull.position per car [B,N,2] f32 */
int mcr_get_positions(mcr_env* h, float* pos);
/* mass KATs: hull invMass, invI, localCenter.x, .y, wheel invMass, invI (host computation) */
void mcr_mass_props(float* out6);
/* the build's sinf/cosf spec evaluated on the host (tests compare with the device kernel's) */
void mcr_sincos_host(float a, float* s, float* c);
int mcr_sincos_device(mcr_env* h, const float* d_in, float* d_sin, float* d_cos, int n, void* stream);

/* ---- full state snapshot / restore of one env (differential tests, checkpoint/resume; SURVEY 8b "mcr_get_state /
 * mcr_set_state").  The blob holds everything the step path reads for env `env`: per-car f32/f64/u32 state (bodies, joint
 * impulses, sleep timers, wheel omega/phase, controls, rewards, episode return, limit states, on-road bits, tile-visit
 * counts, flags), the env record (t, TimeLimit counter, flags), per-tile touch/visit state, the car<->car manifold store
 * (warm-start impulses) and the CURRENT episode slot (track, quads, tile hulls, spawn poses).  Restoring a blob into any
 * env index of any handle with the same num_agents continues bit-identically.  Both calls synchronise the device. */
size_t mcr_state_blob_bytes(const mcr_env* h);
int mcr_get_state_blob(mcr_env* h, int env, void* blob_out);
int mcr_set_state_blob(mcr_env* h, int env, const void* blob);

/* ---- synthetic workload (bench.py, tests): counter-based action stream, action of (global env, agent) at step t is a
 * pure function of (seed, env_offset + env, agent, t): steer ~ U(-1,1), gas ~ U(0,1), brake ~ U(0,1) (the action_space
 * bounds, multi_car_racing.py:162-165).  Device version writes d_actions [B,N,3] f32 on `stream`; the host twin produces
 * the same values for the CPU baseline. */
int mcr_synth_actions(mcr_env* h, float* d_actions, uint64_t seed, uint32_t t, uint32_t env_offset, void* stream);
/* the same stream for steps t0 .. t0 + nsteps - 1 in one launch: d_actions [nsteps][num_envs][num_agents][3] */
int mcr_synth_actions_block(mcr_env* h, float* d_actions, uint64_t seed, uint32_t t0, int nsteps, uint32_t env_offset, void* stream);
void mcr_synth_actions_host(float* out, int num_envs, int num_agents, uint64_t seed, uint32_t t, uint32_t env_offset);

/* ---- instrumentation for bench.py: HIP-event timing of the kernels enqueued by mcr_step, recorded on the
 * launch stream.  `mask` bit k enables timing slot k: 0 collide, 1 dynamics, 2 view, 3/4 = collide/dynamics of
 * the auto-reset pass, 5/6 = dynamics/view of the contact side stream, 7 = its reset pass (255 = all, 0 = off).
 * mcr_timing_read synchronises the device and drains accumulated milliseconds + launch counts. */
int mcr_timing_enable(mcr_env* h, int mask);
/* Profiling switches, 0 in production.  Bits 0-4, 6, 7, 9, 10 are ABLATIONS (results are WRONG when set):
 *   raster: 0 skip flags block, 1 skip road shading, 2 skip cars, 3 skip write-out, 4 skip binning/cull;
 *   dynamics: 6 cap the position loops at 2 sweeps, 7 cap the velocity sweeps of contact waves at 2,
 *             9 skip the contact velocity solve, 10 skip the LDS body exchange of contact waves.
 *             14 solve car<->car contacts and joints in rounds 1-3's defined order instead of b2World::Solve's island order.
 * Bits 5, 8 and 15 only add clock stamps (raster / dynamics / contact-pass phases; 16 with 8: the velocity sweeps in a build with
 * MCR_POSLOOP_PROFILE) and leave the results untouched.  11-13: memory-ordering / starvation experiments of the tests (mcr_kernels.h). */
int mcr_debug_set(mcr_env* h, int value);
/* debug bit 5 (32): the raster kernel stamps s_memtime per phase; read the 64-float tail of a view's scratch */
int mcr_debug_read_view_scratch(mcr_env* h, int view, void* out, int nbytes);
/* debug bit 8 (256): k_dynamics stamps the clock per phase, [2 roles][blocks][8]; read n_u64 words of it */
int mcr_debug_read_dynamics_stamps(mcr_env* h, uint64_t* out, int n_u64);
/* cumulative diagnostics of the multi-stream step: [0] envs deferred by the main dynamics launch, [1] envs resumed,
 * [2] contact envs routed to the side stream, [3] env-steps frozen because the host had not staged the next episode yet
 * (a healthy rollout keeps this at 0) */
int mcr_debug_read_counters(mcr_env* h, uint64_t* out4);
/* all eight: [4] touch-verdict mismatches, [5] waits given up, [6] the last mismatch (env | manifolds << 20 | verdict << 28 | role << 32 | episode step << 36), [7] its step counter */
int mcr_debug_read_counters8(mcr_env* h, uint64_t* out8);
/* the three-chain step decides one step ahead which envs hold a touching car<->car pair (the main dynamics launch runs
 * beside the contact pass); the contact pass counts the envs where it disagrees: must stay 0 */
int mcr_debug_read_verdict_mismatches(mcr_env* h, uint64_t* out1);
/* 1: the three-chain step runs the contact pass beside the main dynamics (mcr_create found that kernels of different streams
 * overlap in this process); 0: it runs first (single stream, profilers that serialise kernels, MCR_SEQUENTIAL_COLLIDE=1) */
int mcr_concurrent_collide(const mcr_env* h);
/* How the streams of the three-chain step are ordered (bit mask; 0 for a single-stream handle):
 *   1  phase words in device memory, posted and awaited by kernels (no marker / barrier packets; needs overlapping kernels, like
 *      the concurrent contact pass; MCR_SOFT_SYNC=0 turns it off) — for steps launched on a caller stream that mcr_bind_stream
 *      accepted; on any other stream, and without this bit: events;
 *   2  on the event path, events are completed by the launches they mark (hipExtLaunchKernelGGL) rather than recorded behind them
 *      (MCR_STOP_EVENTS=0 turns it off);
 *   4  kernels do overlap here, but the internal streams of this handle share a hardware queue with those of another live phase-word
 *      handle of this process (probed pairwise at mcr_create; HIP spreads the streams of a priority class over GPU_MAX_HW_QUEUES queues,
 *      4 by default): this handle runs on events (same results, ~0.02 ms more per step).  Handles whose streams have queues of their own
 *      all keep bit 1.
 * A wait that gave up (status word 0) puts the handle on the event path for the rest of its life.
 * gfx950-specific: a phase word is posted with a RELAXED agent-scope store behind the end-of-kernel write-back of the kernels it
 * follows and polled with RELAXED agent-scope loads (sc1 accesses, served by the memory side — the device's coherence point); the
 * release / acquire pair the HIP memory model asks for between kernels of different streams costs +12 us per step (measured) and
 * is not used.  The event path makes no such assumption. */
int mcr_step_ordering(const mcr_env* h);
/* The same mask for steps launched on caller stream `stream`: bit 0 only if mcr_bind_stream accepted that stream (a stream that was never
 * bound, or that the check rejected, steps on events whatever the handle could do). */
int mcr_step_ordering_for(const mcr_env* h, void* stream);
/* Check ONCE whether steps launched on caller stream `stream` may use the phase-word ordering: two probe kernels and device
 * synchronisations (~1 ms).  Call it when a stream is first used with the handle (VecMultiCarRacing.step does); mcr_step itself never
 * synchronises: a step on a stream that was not bound, or that the check rejected, orders the internal streams with events. */
int mcr_bind_stream(mcr_env* h, void* stream);
/* the per-env records (mcr_common.h: McrEnvState, 48 bytes each: t f64, then steps, slot, staged_ready, consumed, active, resetting, just_reset,
 * frozen as i32, touch_blocks, bp_step as u32) of the first n_bytes / 48 envs; synchronises.  Diagnostics of the auto-reset's staging protocol. */
int mcr_debug_read_env_records(mcr_env* h, void* out, int n_bytes);
/* the last step's contact partition (three-chain step): part_out[B] = the touch verdicts it went by, clist_out[1 + B] = its contact list (count, env ids); synchronises */
int mcr_debug_read_partition(mcr_env* h, uint8_t* part_out, int32_t* clist_out);
/* out NULL: fill the verdict buffer the next step WRITES with fill_value; out != NULL (after that step): read it back [B].  Synchronises. */
int mcr_debug_next_verdicts(mcr_env* h, int fill_value, uint8_t* out_or_null);
/* number of touching car<->car fixture pairs (stored manifolds) per env after the last collide pass */
int mcr_debug_read_contact_counts(mcr_env* h, int32_t* out /*[num_envs]*/);
/* fresh_world = 0: the broadphase proxy ids of env `env`'s live episode on its one world (csrc/k_world.h) — out[0 .. T) tiles in track order,
 * then num_agents * 8 car fixtures (car * 8 + fixture; 0..3 hull polygons, 4..7 wheels); returns the count written; synchronises.  Tests
 * hold it against the oracle's literal b2DynamicTree (what mcr_world_proxy_ids is for the host-side twin). */
int mcr_debug_read_proxy_ids(mcr_env* h, int env, int32_t* out, int cap);
/* Conditions reported by the kernels in mapped host memory (counted on the device, stored with system scope: no PCIe atomics needed),
 * read by mcr_step without synchronising (a condition raised by a step still in flight surfaces one call later).  Words:
 * [0] a bounded in-kernel wait gave up (the main dynamics for the contact pass of an env, any kernel for a phase word),
 * [1] contact pass vs one-step-ahead touch verdict mismatches — FATAL: the next mcr_step returns MCR_ERR_STATE once per change (the
 *     handle then runs the contact pass in front of the dynamics and orders its streams with events);
 * [2] car<->car manifold store / LDS pool overflows, [3] tile begin-event queue overflows — DEGRADED: a documented capacity was
 *     exceeded and the excess dropped (that env's contacts / tile events of that step are incomplete); [4] envs that FROZE: their episode
 *     ended before the host had staged the next one (zero outputs until it arrives; stage earlier).  The rollout goes on, the counts are
 *     visible here; VecMultiCarRacing.step turns every change of [2..4] into an McrWarning.
 * mcr_status copies the cumulative counts (n_words <= 8). */
int mcr_status(mcr_env* h, uint32_t* out, int n_words);
/* The sensor predicate of the contact pass (Box2D's b2TestOverlap: GJK b2Distance behind mcr.py:428 -> b2Contact::Update) on
 * caller-supplied cases, for differential tests: case i = a tile given by its 4 points quads[i][8] (host, f32; the hull is
 * built as the episode generator builds it) against car fixture `fixture` (0..3 hull polygons, 4 the wheel box) of a body
 * whose origin and angle are poses[i][3]; out[i] = touching (host).  fixture + 8: the car fixture is fixtureA (b2TestOverlap(fixture, tile):
 * what a world that lives across reset() can ask for, mcr_world above).  Synchronous. */
int mcr_debug_overlap(mcr_env* h, int n, const float* quads, const float* poses, int fixture, uint8_t* out);
#define MCR_TIMING_SLOTS 8
int mcr_timing_read(mcr_env* h, double* ms_out /*[MCR_TIMING_SLOTS]*/, int64_t* launches_out /*[MCR_TIMING_SLOTS]*/);

#ifdef __cplusplus
}
#endif
#endif /* MCR_H */
