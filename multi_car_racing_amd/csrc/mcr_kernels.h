// mcr_kernels.h — kernel parameter block shared by the three step kernels (gfx950 only).
#pragma once
#include "mcr_common.h"

// Terminal observations (include/mcr.h: mcr_set_terminal_obs).  An env whose episode ends in a step that re-spawns it (auto-reset) gets an
// ENTRY: the state its cars ended the episode with (the per-car SoA fields k_viewprep reads, at stride term_cap * N), the env's clock and the
// episode slot it played in; the reset pass of the env turns that into view records / car polygons and saves the tiles' recolour flags before
// it clears them; the list raster launch of the env's chain draws the entry's N frames into the caller's buffer.
struct McrTermEnv { double t; int32_t slot, env, consumed, pad; };

struct McrParams {
  int32_t B, N, G;              // envs, agents, lanes per env in the dynamics kernel (pow2 >= N)
  int32_t BN;                   // B*N: stride of every per-car SoA field
  int32_t env0, nenv;           // env sub-range [env0, env0+nenv) covered by this launch (stream-level pipelining)
  float* carf;                  // [CF_COUNT][BN]
  double* card;                 // [CD_COUNT][BN]
  uint32_t* caru;               // [CU_COUNT][BN]
  McrEnvState* env;             // [B]
  uint8_t* slots;               // [B][2][MCR_SLOT_BYTES]
  uint32_t* tile_touch;         // [B][TILE_CAP]  bit (car*4+wheel): wheel currently in contact with the tile
  uint16_t* tile_flags;         // [B][TILE_CAP]  bits 0..7 road_visited[car], bit 8 recoloured
  uint32_t* cc_store;           // [B][...] car<->car manifold store (warm starting)
  float4* bpf;                  // [BP_COUNT][8 * BN] broadphase proxies of the car fixtures (fat AABBs; what the last contact pass saw)
  uint32_t* cc_stamp;           // [B][mcr_cc_stamp_words(N)] car<->car broadphase contacts: which FindNewContacts batch made the contact of a fixture pair (k_collide.h)
  uint32_t* bp_stamp;           // [B][TILE_CAP][4 * N] batch label of the tile<->wheel contact: the contact pass that first saw the two fat AABBs overlap
  int32_t bp_fresh;             // the car proxies are re-created at the current poses by this step's contact pass (after mcr_set_bodies)
  uint32_t* status;             // [MCR_STATUS_WORDS] mapped host memory: conditions that make results wrong or degraded (mcr_step checks them without synchronising)
  uint32_t* host_counts;        // [MCR_HOST_COUNTS] mapped host memory (HC_*)
  uint32_t* status_dev;         // [MCR_STATUS_WORDS] the same counts in device memory: where the kernels count (mcr_raise)
  const McrShapes* shapes;
  float* viewp;                 // [BN][MCR_VIEWP_FLOATS] per-car camera + HUD geometry, written by k_dynamics, read by k_view
  float* carpoly;               // [BN][MCR_CARPOLY_FLOATS] world-space vertices of the car's 12 draw polygons (Car.draw)
  int32_t* consumed_host;       // [B] mapped host memory: episode counter of the last install
  // contact side stream (mcr_config.num_streams == 2): envs holding a touching car<->car pair run their (much
  // longer) dynamics chain, reset pass and raster on a second stream, concurrently with everyone else's.
  uint8_t* part;                // [B] 1: a car<->car fixture pair of the env touches at this step's entry poses: the env is the contact chain's.
                                // The verdict (k_touch.h) is evaluated one step ahead by the bookkeeping kernels: two buffers, by step parity
  uint8_t* part_next;           // ... the next step's, which this step's bookkeeping writes
  int32_t* clist;               // [1+B] count, then the env ids of the side-stream envs (any order); this step's buffer
  int32_t* clist_next;          // the other buffer (steps alternate): its count is zeroed by this step's main k_dynamics
  int32_t* next_counts[4];      // the counts of the OTHER parity's deferred / re-spawn / raster-order lists (the lists of a step live in the
                                // buffers of its parity): zeroed by this step's main k_dynamics — every reader of them finished last step
  int32_t fuse_collide;         // (cc_mode) the contact chain's workgroups run the contact pass of their env themselves, in front of its dynamics: the chain
                                // starts with the step instead of behind k_collide, which skips those envs.  The contact LIST then has to exist before the
                                // step: whoever settles an env's touch verdict for the next step (mcr_set_verdict) appends it to clist_next
  int32_t split;                // k_collide pass 0 fills part/clist
  int32_t* dlist;               // [1+B] count + env ids deferred by the main k_dynamics (zeroed by k_collide pass 0)
  int32_t* rlist;               // [1+B] count + env ids the main k_dynamics re-spawned in this step (zeroed by k_collide pass 0); filled when respawn_list
  int32_t* collide_epoch;       // [B] k_collide pass 0 stores `epoch` here when it is through with the env (release); see cc_mode
  int32_t epoch;                // the handle's step counter (unique per step) ...
  const int32_t* epoch_ptr;     // ... or, when the step is replayed as a hipGraph (constant arguments), where a device-side counter holds it
  int32_t* sync_words;          // [MCR_SYNC_WORDS * 16] step-phase words (one per 64-byte line), each holding the epoch of the last step that reached the
                                // phase: how the step's three streams order their kernels when soft_sync is set (see mcr_post / mcr_await below)
  int32_t soft_sync;            // 1: the streams of the three-chain step meet through sync_words (tiny kernels / kernel prologues that poll) instead
                                // of events (marker and barrier packets that the command processor evaluates: 4-17 us each on the critical path)
  int32_t flags_blocks;         // list raster launches: the last `flags_blocks` workgroups of the grid do the bookkeeping (k_flags.h) of the list's cars, a
                                // wavefront per car, beside the workgroups that draw them — a launch of its own for them sat between a chain and its raster
  int32_t post_dyn;             // soft_sync, main dynamics: the LAST workgroup to finish posts W_DYN itself (a release fence per workgroup + one counter) instead of the
                                // first thread of the kernel behind it in the stream (3-4 us later, after the dynamics' drain)
  int32_t await_tail;           // soft_sync, list raster at the tail of the caller's stream: its first workgroup ends by awaiting W_SIDE and W_MAIN — the
                                // launch completes when the whole step has (a kernel of its own for that costs 5-7 us beside the main raster)
  int32_t cc_mode;              // 1: the main k_dynamics runs CONCURRENTLY with k_collide pass 0 (three-chain step): it finds the envs whose
                                // car boxes overlap by itself (they are the contact chain's), and waits for collide_epoch[env] before it reads
                                // what the collide pass produces for the step's bookkeeping (reward, tile count, wheel/tile bits)
  uint8_t* dpart;               // [B] 1: the main k_dynamics deferred this env in this step (written for every env it handles)
  uint32_t* particles;          // [B*N][MCR_PART_WORDS] skid particles of gym Car.step / _create_particle (drawn by render('rgb_array') only); null: not tracked
  int32_t respawn_list;         // the host runs the main envs' reset pass as a list launch (role 4)
  int32_t list_envs_per_block;  // list launches: envs a workgroup (= a wavefront) takes at a time, 1 .. MCR_SIDE_ENVS_PER_WAVE
  uint8_t* defer_state;         // [BN] per car: 0 keep iterating, 1 position loop solved, 2 failed at a fixed point
  double* stats;                // [2] rollout statistics accumulated on the device: episodes finished, sum of their returns over all agents
  unsigned long long* counters; // [4] diagnostics: 0 envs deferred, 1 envs resumed, 2 contact envs routed to the side stream
  int32_t defer_after;          // position sweeps the main launch grants before it defers an env (0: never)
  int32_t* vorder;              // [B] raster order of the main launch: heavy envs from the front, the others from the back, -1 in between.
                                // An entry is env | episode slot << 20 | road_poly entries of that slot << 21 (everything a raster
                                // workgroup needs to start loading); the workgroup that draws it resets it to -1
  int32_t* vcount;              // [2] number of heavy / other envs in vorder (zeroed by the previous step's main k_dynamics)
  int32_t viewprep_in_flags;    // the main envs' view records / car polygons are produced by k_viewprep (beside the bookkeeping kernel), not by the main k_dynamics' epilogue
  int32_t split_views;          // list raster launches: one workgroup per VIEW of a listed env instead of one per env
  int32_t use_vorder;           // k_view maps workgroups to envs through vorder (step path, roles 0/1)
  int32_t role;                 // 0: every env; 1: main stream (skips part envs); 2: contact envs (clist); 3: deferred envs (dlist); 4: both lists
  // step I/O
  const float* actions;         // [B,N,3] or null
  uint8_t* obs;                 // [B,N,96,96,3] or null
  double* reward_out;           // [B,N]
  uint8_t* done_out;            // [B]
  uint8_t* trunc_out;           // [B] or null
  double* ep_return_out;        // [B,N] or null: episode return per agent, written in the step that ends an episode
  int32_t* ep_len_out;          // [B] or null: episode length in steps, written in the step that ends an episode
  // terminal observations (null / 0: off)
  uint8_t* term_obs;            // [term_cap][N][96][96][3] the caller's buffer
  int32_t* term_ids;            // [term_cap] the caller's: env of entry i
  int32_t* term_count_out;      // [1] the caller's: entries of the step (<= term_cap), written when the step is complete
  int32_t term_cap;
  int32_t* term_cnt;            // [4] this step's counters (buffers by step parity): [0] episodes that ended with a re-spawn, [1] entries of the caller-side chains, [2] of the contact chain
  int32_t* term_cnt_next;       // the other parity's: zeroed by this step's main k_dynamics
  int32_t* term_list;           // [2][term_cap] this parity: entry indices per chain (0 caller side, 1 contact chain)
  int32_t* term_idx;            // [B] env -> entry of the step that last re-spawned it (-1: none: a thaw, or more endings than term_cap)
  float* term_carf; double* term_card;   // [CF_COUNT][term_cap * N], [CD_COUNT][term_cap * N]
  float* term_viewp; float* term_carpoly; uint16_t* term_tflags;   // [term_cap * N][..], [term_cap * N][..], [term_cap][MCR_TILE_CAP]
  McrTermEnv* term_env;         // [term_cap]
  const uint8_t* reset_mask;    // [B] or null (k_install)
  // one b2World per env across its episodes (k_world.h); null: every episode is the first of a fresh world (mcr_config::fresh_world)
  uint16_t* pid_tab;            // [B][MCR_PID_TAB] proxy ids of the live episode's fixtures: tile t, then car * 8 + fixture
  uint16_t* pid_stack;          // [B][MCR_PID_STACK] free leaf ids of the world's tree, last freed on top
  int32_t* pid_meta;            // [B][4] stack height, fresh leaves issued, tiles of the live episode, spare
  int32_t auto_reset, max_steps, car_contacts, backwards_flag, use_ego_color;
  int32_t debug;                // ablation switches for profiling (0 in production)
  unsigned long long* dbg_stamps; // [2][dyn_blocks][8] phase clocks of k_dynamics (debug bit 8)
  double h_ratio;
};

// status words (mapped host memory).  FATAL ones (give-up, verdict) make the next mcr_step return MCR_ERR_STATE; an OVERFLOW truncated a
// capacity-bound list (documented deviation): the step goes on, mcr_status shows the count
enum { ST_SPIN_GIVEUP = 0,     // a kernel gave up waiting for another stream's kernels (three-chain step: the contact pass of an env, a phase word)
       ST_VERDICT = 1,         // the contact pass disagreed with the one-step-ahead touch verdict
       ST_CC_OVERFLOW = 2,     // more touching car<->car fixture pairs than the manifold store / the LDS pool holds: the excess was dropped
       ST_EVENT_OVERFLOW = 3,  // more tile begin events in one env-step than the replay buffer holds
       ST_FROZEN = 4,          // an env ended its episode before the host had staged the next one: it froze (zero outputs) until the episode arrived
       MCR_STATUS_WORDS = 8,
       // behind the status words, in the same mapped allocation: counts the host sizes the NEXT step's list launches by (read without synchronising)
       HC_CONTACT_ENVS = 0,    // the length of the last step's contact list
       MCR_HOST_COUNTS = 8 };
__device__ __forceinline__ int mcr_epoch(const McrParams& p) { return p.epoch_ptr ? *p.epoch_ptr : p.epoch; }
// Report condition `w` (ST_*): counted in device memory, the new count then STORED (system scope) into the mapped host word mcr_step polls —
// no atomic on host memory, which needs PCIe atomics and is silently dropped where the platform lacks them.  (Two reports racing may land
// out of order: the host word then holds the smaller of two non-zero counts until the next report; what mcr_step acts on is "it changed".)
__device__ __forceinline__ void mcr_raise(const McrParams& p, int w) {
  const uint32_t n = atomicAdd(&p.status_dev[w], 1u) + 1u;
  __hip_atomic_store(&p.status[w], n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// ---- soft_sync: ordering between the step's streams without command-processor packets (tools/ubench/event_gap.hip: a
// hipEventRecord costs the next kernel of its stream 3 us; a hipStreamWaitEvent 3 us when its event completed long ago, 7.5-11 us
// when it completes last, and 17 us when it completed a few microseconds before the packet's turn — the queue had been parked on it).
// A phase word holds the epoch (the handle's step counter) of the last step that reached the phase.  It is POSTED either by a
// one-thread kernel behind the kernels of the phase, in their stream, or by the first thread of the kernel that follows them in
// their stream: both run after the end-of-kernel release of everything before them.  It is AWAITED by a one-wavefront kernel in
// front of the dependent kernels, in their stream (the kernels behind it start with the usual acquire), or — k_list_chain — by a
// kernel's prologue (poll, then one agent-scope acquire).  Waits are bounded and a give-up is reported like the contact
// pass's (ST_SPIN_GIVEUP: mcr_step fails, the handle goes back to events).
// gfx950, not the HIP memory model: a post is a RELAXED agent-scope store (sc1: written through to the memory side, the device's
// coherence point) behind the end-of-kernel write-back of the kernels it follows, a poll a RELAXED agent-scope load (sc1: served by the
// memory side, not by an L2 of another XCD); the release / acquire pair the model asks for writes the XCD's dirty L2 back per post
// (+12 us per step, measured).  include/mcr.h says so.
enum { W_BEGIN = 0,    // the caller's stream reached this step's main dynamics (everything it held before is complete)
       W_COL = 1,      // the contact pass (k_collide pass 0, side stream) is complete
       W_DYN = 2,      // the main dynamics is complete
       W_SIDE = 3,     // the side stream's part of the step is complete
       W_MAIN = 4,     // the third stream's part of the step is complete
       W_DYN_COUNT = 5,// (not a phase word: workgroups of the main dynamics that are through — post_dyn)
       MCR_SYNC_WORDS = 8 };
__device__ __forceinline__ void mcr_post(const McrParams& p, int w) {
  __hip_atomic_store(&p.sync_words[w * 16], mcr_epoch(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one lane polls; the caller adds the acquire if it reads the phase's data in the same kernel.  Epochs are compared as a wrapping 32-bit
// distance ("word - epoch >= 0" in unsigned arithmetic, reinterpreted: no signed overflow at the 2^31-st step, 6.5 days of stepping); it
// also lets a waiter through that is late by a step, which cannot happen while every step awaits the one before.
// The bound: 2^15 polls 0.2 us apart, then polls 3.4 us apart — 2^24 of them (about a minute) for the words whose wait is enqueued AHEAD
// of the caller's stream (begin, dynamics done: whatever the caller's stream still holds in front of the step — a long inference kernel,
// an event wait — is waited out there and must not be mistaken for a stalled stream), 2^20 (3.6 s) for the words that only the step's
// own internal streams stand between (contact pass, side stream, third stream).  debug bit 12 shortens all of them to a few milliseconds.
__device__ __forceinline__ bool mcr_behind(const McrParams& p, int w, uint32_t epoch) {
  return (int32_t)((uint32_t)__hip_atomic_load(&p.sync_words[w * 16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) < 0;
}
__device__ __forceinline__ bool mcr_await(const McrParams& p, int w) {
  const uint32_t epoch = (uint32_t)mcr_epoch(p);
  const int fast = (p.debug & 4096) ? (1 << 14) : (1 << 15), slow = (p.debug & 4096) ? 0 : (w == W_BEGIN || w == W_DYN) ? (1 << 24) : (1 << 20);
  int spin = 0;
  for (; spin < fast && mcr_behind(p, w, epoch); ++spin) __builtin_amdgcn_s_sleep(8);
  if (spin < fast) return true;
  for (spin = 0; spin < slow && mcr_behind(p, w, epoch); ++spin) __builtin_amdgcn_s_sleep(127);
  if (spin == slow) { atomicAdd(&p.counters[5], 1ull); mcr_raise(p, ST_SPIN_GIVEUP); return false; }
  return true;
}
// Terminal observations, the end of the step: the entry count for the caller, and — only now, the entries' frames are drawn — the news that
// the envs' staged episodes were consumed (the frames read the episode slots the envs left, which the host refills when it hears).  One workgroup.
__device__ __forceinline__ void term_finish(const McrParams& p) {
  if (p.term_cnt == nullptr) return;
  const int n = min(p.term_cnt[0], p.term_cap);
  for (int i = threadIdx.x; i < n; i += blockDim.x) { const McrTermEnv te = p.term_env[i]; p.consumed_host[te.env] = te.consumed; }
  if (threadIdx.x == 0) *p.term_count_out = n;
}
// debug bit 17 (diagnostics of the step's ordering): who touched env `env` when — dbg_stamps[(dyn blocks + env) * 8 + slot] = wall clock | tag << 56
#define MCR_TRACE(p, env, slot, tag) do { if ((p).debug & 131072) (p).dbg_stamps[((size_t)(((p).B * (p).G + 63) / 64) + (size_t)(env)) * 8 + (slot)] = (unsigned long long)__builtin_amdgcn_s_memrealtime() | ((unsigned long long)(tag) << 56); } while (0)
// the touch verdict of env `env` for the NEXT step (one writer per env and step), and — fuse_collide — its place in the next step's contact list
__device__ __forceinline__ void mcr_set_verdict(const McrParams& p, const int env, const bool v) {
  p.part_next[env] = v ? 1 : 0;
  MCR_TRACE(p, env, 2, 0x10 | (v ? 1 : 0) | (p.role << 1));
  if (v && p.fuse_collide) p.clist_next[1 + atomicAdd(&p.clist_next[0], 1)] = env;
}
#define MCR_VORDER_ENV_MASK 0xfffff
#define MCR_VORDER_SLOT_SHIFT 20
#define MCR_VORDER_P_SHIFT 21
// per-car view parameters (f32): camera (:540-556) and HUD rectangles (:634-674) in pixel units
#define MCR_VIEWP_FLOATS 48
enum { VP_CAM = 0 /*m00 m01 m10 m11 tx ty*/, VP_INV = 6 /*ax bx cx0 ay by cy0*/, VP_IND = 12 /*7 x (x0 x1 y0 y1)*/, VP_HUDTOP = 40,
       VP_SCORE = 41 /*int bits: the integer the score label shows, "%04i" % reward[agent] at render time (:431 precedes :437)*/,
       VP_OLDFLAGS = 42 /*u32 bits: the car's backward / on-grass flags as of the previous step — what this step's HUD flag shows*/,
       VP_GRASS = 43 /*5 x int bits: light grass squares the viewport can see (first column, columns, first row, rows in the
                       20 x 20 lattice of :620-627) and 1 if the whole viewport lies inside the playfield*/ };
// "%04i" % reward: Python formats a float with %i by truncating towards zero
__device__ __forceinline__ int mcr_label_value(double reward) { return (reward > -2.0e9 && reward < 2.0e9) ? (int)reward : 0; }

// Car.draw polygons per car in draw order: 4 x (wheel box, white stripe) then the 4 hull polygons; each slot holds
// 8 vertices (x0 y0 .. x7 y7) and slot header words live in carpoly_n: vertex count (0 = not drawn)
#define MCR_CARPOLY_FLOATS (12 * 16 + 16)
#define MCR_CARPOLY_NOFF (12 * 16)
// Env handled by work slot `s` of a launch (slot = env index for roles 0/1; position in the contact / deferred / re-spawn
// list for roles 2 / 3 / 4); returns
// p.env0 + p.nenv ("no env") for slots that are not this launch's business.
__device__ __forceinline__ int mcr_env_of_slot(const McrParams& p, int s, bool main_dynamics = false) {
  const int end = p.env0 + p.nenv;
  if (s < 0) return end;
  if (p.role == 2) return s < p.clist[0] ? p.clist[1 + s] : end;
  if (p.role == 3) return s < p.dlist[0] ? p.dlist[1 + s] : end;
  if (p.role == 4) return s < p.rlist[0] ? p.rlist[1 + s] : end;
  const int env = p.env0 + s;
  if (env >= end) return end;
  // role 1 = the main launches: not the contact envs (p.part: the touch verdict of this step's entry poses, written by last
  // step's bookkeeping) and — for everybody but the main dynamics, which sets that mark itself — not the envs the main
  // dynamics deferred
  return (p.role == 1 && (p.part[env] || (!main_dynamics && p.dpart[env]))) ? end : env;
}
// Work slot of a k_dynamics lane.  Roles 0/1: 64/G consecutive envs per wavefront.  Role 2 (side stream, every env
// holds car<->car contacts): MCR_SIDE_ENVS_PER_WAVE envs per wavefront, so that the wavefront's LDS pool of contact
// constraints (DYN_VC_POOL = MCR_SIDE_ENVS_PER_WAVE * MCR_CC_MAX) can never overflow, however the envs are packed.
#define MCR_SIDE_ENVS_PER_WAVE 2
#define MCR_DEFER_AFTER 3          // position sweeps the main dynamics launch grants an env before deferring it (99.86 % need 1; 1 / 2 / 3 / 4 / 6 sweeps: 14.99 / 15.01 / 15.10 / 15.10 / 15.09 M env-steps/s)
__device__ __forceinline__ int mcr_dyn_slot(const McrParams& p, int blk) {
  const int grp = (int)threadIdx.x / p.G;
  if (p.role >= 2) return grp < p.list_envs_per_block ? blk * p.list_envs_per_block + grp : -1;
  return (blk * 64 + (int)threadIdx.x) / p.G;
}
// List launches (role >= 2: the contact / deferred envs, a few per step, how many only the device knows) run a small
// grid whose workgroups walk the list: the (virtual) block indices blockIdx, blockIdx + gridDim, .. below this bound.
// A grid sized for the worst case would queue thousands of empty workgroups behind the raster that saturates the LDS
// of every CU.  Main launches have exactly one block index per workgroup.
__device__ __forceinline__ int mcr_list_len(const McrParams& p) { return p.role == 2 ? p.clist[0] : p.role == 3 ? p.dlist[0] : p.role == 4 ? p.rlist[0] : 0; }
__device__ __forceinline__ int mcr_virtual_blocks(const McrParams& p, int slots_per_block) { return (mcr_list_len(p) + slots_per_block - 1) / slots_per_block; }
// Skid particles (gym car_dynamics.Car: step() item "Skid trace", _create_particle, draw(viewer, True); call site
// multi_car_racing.py:564).  Per car: a ring of the last MCR_PART_MAX particles, each a polyline of up to MCR_PART_PTS
// wheel positions; per wheel: skid_start and the particle it is extending.  Words of a car's block:
//   [0] particles created so far (particle id = creation index, ring slot = id % MCR_PART_MAX, alive: id >= created - MAX)
//   [1..4] wheel k: id of w.skid_particle (-1: None)   [5..8] wheel k: len(w.skid_particle.poly) | grass << 8 | (skid_start is not None) << 9
//   [9..16] wheel k: skid_start x, y (f32 bits)          [17..46] ring slot: points | grass << 8
//   [48 ..] ring slot s, point i: x, y (f32 bits) at 48 + (s * MCR_PART_PTS + i) * 2
#define MCR_PART_MAX 30
#define MCR_PART_PTS 30
#define MCR_PART_HDR 48
#define MCR_PART_WORDS (MCR_PART_HDR + MCR_PART_MAX * MCR_PART_PTS * 2)
__device__ __forceinline__ void mcr_particles_clear(uint32_t* pc) {          // Car.__init__: particles = [], wheels without skid state
  pc[0] = 0u;
  for (int k = 0; k < 4; ++k) { pc[1 + k] = 0xffffffffu; pc[5 + k] = 0u; }
  for (int s = 0; s < MCR_PART_MAX; ++s) pc[17 + s] = 0u;
}
// one wheel's "Skid trace" block of Car.step: `skid` = abs(force) > 2 * friction_limit, `grass` = the wheel touches no tile
__device__ __noinline__ void mcr_particle_step(uint32_t* pc, int k, bool skid, bool grass, float x, float y) {
  if (!skid) { pc[1 + k] = 0xffffffffu; pc[5 + k] = 0u; return; }           // w.skid_start = None; w.skid_particle = None
  const int cur = (int)pc[1 + k];
  const uint32_t st = pc[5 + k];
  const int len = (int)(st & 255u);
  if (cur >= 0 && (((st >> 8) & 1u) != 0u) == grass && len < MCR_PART_PTS) {  // extend the wheel's particle (drawn only while it is among the last 30)
    const int created = (int)pc[0];
    if (cur >= created - MCR_PART_MAX) {
      const int sl = cur % MCR_PART_MAX;
      pc[MCR_PART_HDR + (sl * MCR_PART_PTS + len) * 2] = __float_as_uint(x); pc[MCR_PART_HDR + (sl * MCR_PART_PTS + len) * 2 + 1] = __float_as_uint(y);
      pc[17 + sl] = (uint32_t)(len + 1) | (grass ? 256u : 0u);
    }
    pc[5 + k] = (st & ~255u) | (uint32_t)(len + 1);
  } else if (!((st >> 9) & 1u)) {                                            // w.skid_start = w.position
    pc[9 + 2 * k] = __float_as_uint(x); pc[10 + 2 * k] = __float_as_uint(y);
    pc[5 + k] = st | 512u;
  } else {                                                                   // _create_particle(w.skid_start, w.position, grass)
    const int id = (int)pc[0];
    const int sl = id % MCR_PART_MAX;
    pc[MCR_PART_HDR + sl * MCR_PART_PTS * 2] = pc[9 + 2 * k]; pc[MCR_PART_HDR + sl * MCR_PART_PTS * 2 + 1] = pc[10 + 2 * k];
    pc[MCR_PART_HDR + sl * MCR_PART_PTS * 2 + 2] = __float_as_uint(x); pc[MCR_PART_HDR + sl * MCR_PART_PTS * 2 + 3] = __float_as_uint(y);
    pc[17 + sl] = 2u | (grass ? 256u : 0u);
    pc[0] = (uint32_t)(id + 1);
    pc[1 + k] = (uint32_t)id;
    pc[5 + k] = 2u | (grass ? 256u : 0u);                                    // skid_start = None
  }
}
#define MCR_CC_MAX 24           // touching car<->car fixture pairs kept per env (warm start)
#define MCR_CC_WORDS 20         // u32 words per stored manifold
