// mcr_hip.hip — C-ABI implementation (include/mcr.h) over the gfx950 kernels.
// One handle owns one env slice on one device: SoA car state, per-env state, two episode slots per env.
#include "../../include/mcr.h"
#include "mcr_kernels.h"
#include "k_dynamics.h"
#include "k_collide.h"
#include "k_raster_common.h"
#include "k_flags.h"
#include "k_viewprep.h"
#include "k_list_chain.h"
#include "k_render.h"
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <string>
#include <vector>
#include <cstring>
#include <cstdio>
#include <cstdlib>

void mcr_build_shapes(McrShapes* S);   // mcr_host.cpp

static thread_local std::string g_err;
extern "C" const char* mcr_last_error(void) { return g_err.c_str(); }
extern "C" const char* mcr_version(void) { return "mcr-hip 0.1 (gfx950)"; }

#define HIPCHK(x)                                                                                         \
  do {                                                                                                    \
    hipError_t e_ = (x);                                                                                  \
    if (e_ != hipSuccess) {                                                                               \
      g_err = std::string(#x) + ": " + hipGetErrorString(e_);                                             \
      return MCR_ERR_HIP;                                                                                 \
    }                                                                                                     \
  } while (0)

#define MCR_LIST_GRID 128        // workgroups of a list launch (they walk the device-side list)
struct TimedLaunch { int id; hipEvent_t a, b; };

struct mcr_env {
  mcr_config cfg;
  McrParams P;
  void* slab;
  size_t slab_bytes;
  int32_t* consumed_host;     // mapped host memory
  int32_t* consumed_seen;     // host copy of the last polled counters
  int timing;                 // bit mask of kernel ids to time with HIP events
  std::vector<TimedLaunch> pending;
  std::vector<hipEvent_t> free_events;
  double t_ms[MCR_TIMING_SLOTS]; int64_t t_n[MCR_TIMING_SLOTS];
  bool any_reset;
  bool split;                 // contact side stream enabled (cfg.num_streams == 2)
  int step_parity;            // which contact-list buffer the next step fills
  int32_t* stage_ids;         // [B] device scratch of mcr_stage_episodes
  hipStream_t s_side, s_defer; // internal streams: the contact envs' chain, the deferred envs' chain
  hipEvent_t ev_fork, ev_join, ev_fork2, ev_join2, ev_col;
  unsigned long long* view_stamps;   // [BN][16] phase clocks of the rasteriser (debug bit 5)
  // hipGraph of one step (mcr_set_step_graph): one per contact-list parity, re-captured when any argument changes
  struct StepGraph { bool valid; McrParams P; hipStream_t st; int view_flags; hipGraph_t graph; hipGraphExec_t exec; };
  StepGraph sg[2];
  int use_graph;              // 0 off, 1 on, -1 capture failed once: stay off
  bool unfused_collide;       // MCR_UNFUSED_COLLIDE=1 (read at create): the contact chain waits for the all-env contact pass instead of running its envs' own
  bool concurrent_collide;    // the contact pass may run beside the main dynamics (kernels of different streams do overlap here: probed at create)
  bool verdict_fresh;         // the touch verdicts (k_touch.h) of the next step's entry poses are in place (last step's bookkeeping wrote them)
  bool last_fused = false;    // ... and so is the next step's contact list (the last step ran with McrParams::fuse_collide)
  uint32_t* status_host;      // [MCR_STATUS_WORDS] mapped host memory the kernels report trouble in (mcr_kernels.h: ST_*)
  uint32_t status_seen[MCR_STATUS_WORDS];   // what mcr_step has already reported
  int32_t step_count;         // steps launched: the epoch of the three-chain step's per-env "contact pass done" words
  bool bp_fresh;              // mcr_set_bodies teleported cars: the next contact pass re-creates their broadphase proxies
  int simd_count;             // SIMDs of the device (4 per CU)
  int32_t* dev_step_ctr;      // device-side step counter (the epoch of a replayed step graph)
  bool viewprep_in_flags;     // three-chain step: k_viewprep (side stream, beside the bookkeeping) produces the main envs' view records / car polygons
  int list_view_grid;         // workgroups of a list raster launch
  std::vector<std::pair<hipStream_t, bool>> bound;   // caller streams checked by mcr_bind_stream: may the step order its streams with phase words when launched on this one?
  bool soft_denied = false;   // kernels overlap here, but another handle of this process holds the device's one phase-word token (mcr_create)
  bool soft_token = false;    // this handle is its device's one phase-word handle (mcr_create)
  bool soft_sync = false;     // the step's streams meet through phase words in device memory (mcr_kernels.h: mcr_post / mcr_await) instead of events
  bool stop_events = true;    // events completed by the launches they mark (hipExtLaunchKernelGGL) instead of marker packets behind them
  int chain_grid;             // workgroups of a list chain launch (each walks the list, 2 envs at a time)
  bool post_dyn = true;       // the main dynamics posts its own completion (McrParams::post_dyn; MCR_POST_DYN=0, read at create: the kernel behind it does, as in rounds 3-5)
  bool vorder_dirty[2];       // the raster order list of that step parity was filled by a step that did not draw
  void* term_slab = nullptr;  // terminal observations (mcr_set_terminal_obs): entry state, view records, per-parity counters and lists
  int32_t* term_cnt2 = nullptr;   // [2][4] counters by step parity
  int32_t* term_list2 = nullptr;  // [2][2][cap] entry lists by step parity and chain
  struct RefillSvc* svc = nullptr;  // mcr_refill_start: the handle's own host thread that generates and stages consumed episodes
};

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
// Does the three-chain step run the contact pass BESIDE the main dynamics?  Only where kernels of different streams overlap
// (probed at create), up to 7 cars per env (measured), and only for batches whose main dynamics launch puts at most one of its
// 256-VGPR wavefronts on a SIMD: that launch waits INSIDE the kernel for words of the contact pass, which must be able to get onto
// the machine beside it (one such wavefront per SIMD leaves half the register file and nearly all LDS free; two on every SIMD
// could starve a contact pass dispatched second).  After a reported give-up the handle stays with the contact pass in front.
static bool cc_active(const mcr_env* h) { return h->split && h->concurrent_collide; }

#include <mutex>
#include <algorithm>
// live handles of this process that order their streams with phase words.  Several per device are fine AS LONG AS their internal streams sit
// on hardware queues of their own (probed pairwise in mcr_create): an await at the head of a queue that two handles share could hold back the
// very kernel another handle's await waits for (host threads stepping two handles interleave their enqueues freely) — a cycle that only the
// wait bound would break.  HIP multiplexes the streams of a priority class over a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default):
// where the probe finds two handles' streams on one queue, the later handle orders its streams with events.
static std::mutex g_soft_mu;
static std::vector<mcr_env*> g_soft_list;

// Do kernels of two streams really run side by side in this process?  Under a counter-collecting profiler, a debugger or
// AMD_SERIALIZE_KERNEL they do not — and the cc_mode step (the main dynamics waits inside the kernel for words the
// contact pass, launched on another stream, writes) must not be used then.  Probe: kernel A spins (bounded, ~2 ms) until
// kernel B, launched AFTER it on another stream, has set a flag.
__global__ void k_probe_wait(int* flag, int* result) {
  int seen = 0;
  for (int i = 0; i < 20000 && !seen; ++i) { seen = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __builtin_amdgcn_s_sleep(32); }
  *result = seen;
}
__global__ void k_step_begin(int32_t* step_ctr) { *step_ctr = (int32_t)((uint32_t)*step_ctr + 1u); }      // (wraps: epochs are compared as 32-bit distances)
__global__ void k_probe_set(int* flag) { __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
static bool kernels_overlap(hipStream_t sa, hipStream_t sb) {
  int* d = nullptr; int r = 0;
  if (hipMalloc(&d, 2 * sizeof(int)) != hipSuccess) return false;
  bool ok = hipMemset(d, 0, 2 * sizeof(int)) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
  if (ok) {
    hipLaunchKernelGGL(k_probe_wait, dim3(1), dim3(1), 0, sa, d, d + 1);
    hipLaunchKernelGGL(k_probe_set, dim3(1), dim3(1), 0, sb, d);
    ok = hipDeviceSynchronize() == hipSuccess && hipMemcpy(&r, d + 1, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess && r == 1;
  }
  (void)hipFree(d);
  return ok;
}

extern "C" int mcr_create(const mcr_config* cfg, mcr_env** out) {
  if (!cfg || !out) { g_err = "null argument"; return MCR_ERR_ARG; }
  if (cfg->num_envs < 1 || cfg->num_envs > MCR_VORDER_ENV_MASK || cfg->num_agents < 1 || cfg->num_agents > MCR_MAX_AGENTS) { g_err = "num_envs/num_agents out of range"; return MCR_ERR_ARG; }
  HIPCHK(hipSetDevice(cfg->device));
  mcr_env* h = new mcr_env();
  h->cfg = *cfg; h->timing = 0; h->any_reset = false; h->use_graph = 0; h->verdict_fresh = false; h->concurrent_collide = false; h->unfused_collide = getenv("MCR_UNFUSED_COLLIDE") != nullptr; h->sg[0].valid = h->sg[1].valid = false;
  h->vorder_dirty[0] = h->vorder_dirty[1] = false;
  // A list chain is a serial solver chain per wavefront (2 envs each).  With i.i.d. random actions and two cars per env the
  // contact list holds ~15 envs of 4096, but a policy that actually drives (or N = 8: ~340 envs) fills it with hundreds, and
  // every further round of the walk adds a whole chain (~300 us) to the side stream: the grid covers 1024 envs in one round;
  // surplus workgroups exit on their first load.
  h->chain_grid = 4 * MCR_LIST_GRID;
  h->viewprep_in_flags = cfg->num_agents <= 3;     // beyond three cars per env the bookkeeping + raster chain is the step's critical path: the epilogue stays in the dynamics (round 3, phase-word path: N = 3 13.4 -> 13.7 M with it, N = 4 10.8 -> 10.6, N = 8 no difference)
  h->list_view_grid = cfg->num_agents <= 2 ? MCR_LIST_GRID : 8 * MCR_LIST_GRID;
  h->status_host = nullptr; h->step_count = 0; h->bp_fresh = false; memset(h->status_seen, 0, sizeof(h->status_seen));
  { hipDeviceProp_t prop; h->simd_count = (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess ? prop.multiProcessorCount : 256) * 4; }
  for (int i = 0; i < MCR_TIMING_SLOTS; ++i) { h->t_ms[i] = 0; h->t_n[i] = 0; }
  const int B = cfg->num_envs, N = cfg->num_agents;
  int G = 1; while (G < N) G <<= 1;
  const size_t BN = (size_t)B * N;
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
  const size_t o_carf = carve(sizeof(float) * CF_COUNT * BN);
  const size_t o_card = carve(sizeof(double) * CD_COUNT * BN);
  const size_t o_caru = carve(sizeof(uint32_t) * CU_COUNT * BN);
  const size_t o_env = carve(sizeof(McrEnvState) * B);
  const size_t o_touch = carve(sizeof(uint32_t) * MCR_TILE_CAP * (size_t)B);
  const size_t o_tflags = carve(sizeof(uint16_t) * MCR_TILE_CAP * (size_t)B);
  const size_t o_cc = carve(sizeof(uint32_t) * (size_t)B * (MCR_CC_MAX * MCR_CC_WORDS + 4));
  const size_t o_bpf = carve(sizeof(float4) * BP_COUNT * MCR_BP_FIX * BN);
  const size_t o_ccstamp = carve(sizeof(uint32_t) * (size_t)B * mcr_cc_stamp_words(N));
  const size_t o_bpstamp = carve(sizeof(uint32_t) * (size_t)B * MCR_TILE_CAP * 4 * N);
  const size_t o_part = carve(2 * (size_t)B);                 // x2: the touch verdicts of a step live in the buffer of its parity
  const size_t o_dpart = carve(B);
  const size_t o_epoch = carve(sizeof(int32_t) * ((size_t)B + 1));
  const size_t o_sync = carve(sizeof(int32_t) * 16 * MCR_SYNC_WORDS);
  const size_t o_dlist = carve(sizeof(int32_t) * 2 * ((size_t)B + 1));      // x2: the lists of a step live in the buffers of its parity
  const size_t o_rlist = carve(sizeof(int32_t) * 2 * ((size_t)B + 1));
  const size_t o_dstate = carve(BN);
  const size_t o_counters = carve(sizeof(unsigned long long) * 8);
  const size_t o_statusdev = carve(sizeof(uint32_t) * MCR_STATUS_WORDS);
  const size_t o_stage_ids = carve(sizeof(int32_t) * (size_t)B);
  const size_t o_stats = carve(sizeof(double) * 2);
  const size_t o_vorder = carve(sizeof(int32_t) * 2 * ((size_t)B + 2));
  const size_t o_stamps = carve(sizeof(unsigned long long) * 8 * ((((size_t)B * G + 63) / 64) + 2 * (size_t)B));
  const size_t o_clist = carve(sizeof(int32_t) * 2 * ((size_t)B + 1));
  const size_t o_shapes = carve(sizeof(McrShapes));
  const size_t o_viewp = carve(sizeof(float) * MCR_VIEWP_FLOATS * BN);
  const size_t o_carpoly = carve(sizeof(float) * MCR_CARPOLY_FLOATS * BN);
  const size_t o_vscratch = carve(sizeof(unsigned long long) * 16 * BN);
  const size_t o_particles = carve(cfg->skid_particles ? sizeof(uint32_t) * MCR_PART_WORDS * BN : 0);
  const bool one_world = cfg->fresh_world == 0;      // (k_world.h: the env's b2World across its episodes, the reference's semantics; zero-initialised = an empty world)
  const size_t o_pidtab = carve(one_world ? sizeof(uint16_t) * MCR_PID_TAB * (size_t)B : 0);
  const size_t o_pidstk = carve(one_world ? sizeof(uint16_t) * MCR_PID_STACK * (size_t)B : 0);
  const size_t o_pidmeta = carve(one_world ? sizeof(int32_t) * 4 * (size_t)B : 0);
  const size_t o_slots = carve((size_t)B * 2 * MCR_SLOT_BYTES);
  h->slab_bytes = off;
  if (hipMalloc(&h->slab, off) != hipSuccess) { g_err = "hipMalloc failed"; delete h; return MCR_ERR_HIP; }
  (void)hipMemset(h->slab, 0, o_slots);
  for (int par = 0; par < 2; ++par)      // raster order lists (per step parity: 2 counts, then B entries): -1 = unused entry
    (void)hipMemset((uint8_t*)h->slab + o_vorder + sizeof(int32_t) * ((size_t)par * (B + 2) + 2), 0xff, sizeof(int32_t) * (size_t)B);
  uint8_t* base = (uint8_t*)h->slab;
  McrParams& P = h->P;
  memset(&P, 0, sizeof(P));
  P.B = B; P.N = N; P.G = G; P.BN = (int)BN;
  P.carf = (float*)(base + o_carf); P.card = (double*)(base + o_card); P.caru = (uint32_t*)(base + o_caru);
  P.env = (McrEnvState*)(base + o_env); P.tile_touch = (uint32_t*)(base + o_touch); P.tile_flags = (uint16_t*)(base + o_tflags);
  P.cc_store = (uint32_t*)(base + o_cc); P.bpf = (float4*)(base + o_bpf); P.cc_stamp = (uint32_t*)(base + o_ccstamp); P.bp_stamp = (uint32_t*)(base + o_bpstamp); P.shapes = (const McrShapes*)(base + o_shapes); P.slots = base + o_slots;
  P.pid_tab = one_world ? (uint16_t*)(base + o_pidtab) : nullptr; P.pid_stack = one_world ? (uint16_t*)(base + o_pidstk) : nullptr; P.pid_meta = one_world ? (int32_t*)(base + o_pidmeta) : nullptr;
  h->view_stamps = (unsigned long long*)(base + o_vscratch);
  P.viewp = (float*)(base + o_viewp);
  P.part = base + o_part; P.dpart = base + o_dpart; P.collide_epoch = (int32_t*)(base + o_epoch); h->dev_step_ctr = P.collide_epoch + B; P.sync_words = (int32_t*)(base + o_sync); P.dlist = (int32_t*)(base + o_dlist); P.rlist = (int32_t*)(base + o_rlist); P.defer_state = base + o_dstate; P.counters = (unsigned long long*)(base + o_counters); P.status_dev = (uint32_t*)(base + o_statusdev); P.stats = (double*)(base + o_stats);
  h->stage_ids = (int32_t*)(base + o_stage_ids); P.vcount = (int32_t*)(base + o_vorder); P.vorder = P.vcount + 2; P.dbg_stamps = (unsigned long long*)(base + o_stamps); P.clist = (int32_t*)(base + o_clist);
  P.carpoly = (float*)(base + o_carpoly);
  P.particles = cfg->skid_particles ? (uint32_t*)(base + o_particles) : nullptr;
  P.auto_reset = cfg->auto_reset; P.max_steps = cfg->max_episode_steps; P.car_contacts = cfg->car_contacts;
  P.backwards_flag = cfg->backwards_flag; P.use_ego_color = cfg->use_ego_color; P.h_ratio = cfg->h_ratio;
  McrShapes S; mcr_build_shapes(&S);
  // the raster gives two record slots to hull polygon 2 (HULL_POLY3, 8 vertices) and one to every other car polygon
  if (S.hull[0].n > 4 || S.hull[1].n > 4 || S.hull[2].n > 8 || S.hull[3].n > 4 || S.wheel.n > 4) { g_err = "unexpected car fixture vertex counts"; (void)hipFree(h->slab); delete h; return MCR_ERR_STATE; }
  (void)hipMemcpy((void*)P.shapes, &S, sizeof(S), hipMemcpyHostToDevice);
  if (hipHostMalloc((void**)&h->consumed_host, sizeof(int32_t) * B, hipHostMallocMapped) != hipSuccess) { g_err = "hipHostMalloc failed"; (void)hipFree(h->slab); delete h; return MCR_ERR_HIP; }
  memset(h->consumed_host, 0, sizeof(int32_t) * B);
  void* dptr = nullptr;
  (void)hipHostGetDevicePointer(&dptr, h->consumed_host, 0);
  P.consumed_host = (int32_t*)dptr;
  h->consumed_seen = new int32_t[B]();
  if (hipHostMalloc((void**)&h->status_host, sizeof(uint32_t) * (MCR_STATUS_WORDS + MCR_HOST_COUNTS), hipHostMallocMapped) != hipSuccess) { g_err = "hipHostMalloc failed"; (void)hipHostFree(h->consumed_host); delete[] h->consumed_seen; (void)hipFree(h->slab); delete h; return MCR_ERR_HIP; }
  memset(h->status_host, 0, sizeof(uint32_t) * (MCR_STATUS_WORDS + MCR_HOST_COUNTS));
  (void)hipHostGetDevicePointer(&dptr, h->status_host, 0);
  P.status = (uint32_t*)dptr; P.host_counts = P.status + MCR_STATUS_WORDS;
  // Contact side stream.  k_dynamics is a serial dependency chain whose duration is set by its slowest wavefront,
  // and a wavefront holding a touching car<->car pair takes 2-3x as long as the others (sequential Gauss-Seidel
  // over the contacts).  With num_streams == 2 those envs (a handful per step) run dynamics -> reset pass ->
  // raster on an internal stream, concurrently with the same chain of all the other envs on the caller's stream;
  // the two event hops are off the critical path because the side chain is the shorter one.
  P.env0 = 0; P.nenv = B; P.split = 0; P.role = 0;
  h->split = false; h->step_parity = 0;
  if (cfg->num_streams == 2) {
    // highest priority: their few workgroups must not queue behind the main stream's saturating raster launch.
    // (Tried: giving the chains CUs of their own with hipExtStreamCreateWithCUMask — tools/ubench/cumask_probe.hip shows
    // the mask layout — restores their stand-alone kernel times beside the raster, but every launch and event hop on a
    // masked stream costs ~10 us more: 0.81 ms per step.)
    int prio_lo = 0, prio_hi = 0; (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    if (hipStreamCreateWithPriority(&h->s_side, hipStreamNonBlocking, prio_hi) == hipSuccess) {
      // (the priority of the third stream — it carries the main raster, whose backlog of workgroups competes with the list raster at the
      // tail of the caller's stream for every LDS slot that frees up — makes no difference: high / normal / low 14.73 / 14.72 / 14.72 M)
      if (hipStreamCreateWithPriority(&h->s_defer, hipStreamNonBlocking, prio_hi) == hipSuccess) {
        const unsigned evf = hipEventDisableTiming;     // (hipEventReleaseToDevice changes nothing measurable: tools/ubench/event_gap.hip)
        for (hipEvent_t* e : {&h->ev_fork, &h->ev_join, &h->ev_fork2, &h->ev_join2, &h->ev_col}) (void)hipEventCreateWithFlags(e, evf);
        // (the switches of the environment — MCR_SOFT_SYNC=0, MCR_STOP_EVENTS=0, MCR_SEQUENTIAL_COLLIDE=1 — select paths that exist for
        // environments where the default cannot run: profilers that serialise kernels, graph capture; the parity suite covers each)
        if (const char* g = getenv("MCR_STOP_EVENTS")) h->stop_events = atoi(g) != 0;
        if (const char* g = getenv("MCR_POST_DYN")) h->post_dyn = atoi(g) != 0;
        h->soft_sync = kernels_overlap(h->s_defer, h->s_side);     // (a waiting kernel needs the kernels it waits for to run beside it)
        if (const char* g = getenv("MCR_SOFT_SYNC")) h->soft_sync = h->soft_sync && atoi(g) != 0;
        // One handle per device and process at a time: the side stream's wait for "begin" is enqueued BEFORE the kernel that posts it.
        // With the streams of two handles multiplexed over the same hardware queues and their steps interleaved on one caller's stream,
        // handle A's waiting kernel can sit in front of the post handle B's join waits for, which in turn sits in front of A's dynamics:
        // a cycle that only the wait bound would break.  Further handles order their streams with events (every wait there is for work
        // enqueued earlier).
        if (h->soft_sync) {
          std::lock_guard<std::mutex> lk(g_soft_mu);
          bool own_queues = !getenv("MCR_ONE_SOFT_HANDLE");
          for (mcr_env* o : g_soft_list)
            if (own_queues && o->cfg.device == cfg->device)
              own_queues = kernels_overlap(h->s_side, o->s_side) && kernels_overlap(h->s_side, o->s_defer) && kernels_overlap(h->s_defer, o->s_side) && kernels_overlap(h->s_defer, o->s_defer) &&
                           kernels_overlap(o->s_side, h->s_side) && kernels_overlap(o->s_defer, h->s_defer);
          bool other = false; for (mcr_env* o : g_soft_list) other = other || o->cfg.device == cfg->device;
          if (own_queues || !other) { h->soft_token = true; g_soft_list.push_back(h); }
          else { h->soft_sync = false; h->soft_denied = true; }
        }
        // (at 8 cars per env the one-step-ahead touch verdict — a second pass over up to 28 car pairs — costs more than the
        // contact pass gains by running beside the dynamics: measured at N = 8, round 2: 3.66 vs 4.28 M env-steps/s.  Round 3 tried a
        // CONSERVATIVE verdict there instead — bounding discs + car boxes, no narrowphase, the contact chain taking every env it
        // marks: with ~340 of 4096 envs in contact and as many near misses per step the side stream's chain, bookkeeping and raster
        // grow faster than the 110 us of k_collide that leave the critical path: 1.40 vs 0.975 ms per step.  What bounds N = 8 is the
        // contact chain itself — 585 us for its slowest wavefront, a sequential Gauss-Seidel over the contacts between the joint sweeps)
        // (round 3, phase-word path, contact pass beside / in front of the dynamics: N = 5 8.99 / 8.50 M env-steps/s, N = 6 7.87 / 7.49,
        // N = 7 7.16 / 6.80, N = 8 4.90 / 5.47 — up to seven cars per env it runs beside)
        h->concurrent_collide = N <= 8 && !getenv("MCR_SEQUENTIAL_COLLIDE") && (B * G + 63) / 64 <= h->simd_count && kernels_overlap(h->s_defer, h->s_side);
        h->split = true;
      } else (void)hipStreamDestroy(h->s_side);
    }
    (void)hipGetLastError();
  }
  (void)hipDeviceSynchronize();
  *out = h;
  return MCR_OK;
}

extern "C" int mcr_refill_stop(mcr_env* h);
extern "C" int mcr_destroy(mcr_env* h) {
  if (!h) return MCR_ERR_ARG;
  (void)mcr_refill_stop(h);
  if (h->soft_token) { std::lock_guard<std::mutex> lk(g_soft_mu); g_soft_list.erase(std::remove(g_soft_list.begin(), g_soft_list.end(), h), g_soft_list.end()); }
  (void)hipSetDevice(h->cfg.device);
  (void)hipDeviceSynchronize();
  for (auto& g : h->sg) if (g.valid) { (void)hipGraphExecDestroy(g.exec); (void)hipGraphDestroy(g.graph); g.valid = false; }
  for (auto& t : h->pending) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
  for (auto e : h->free_events) (void)hipEventDestroy(e);
  if (h->split) {
    (void)hipStreamDestroy(h->s_side); (void)hipStreamDestroy(h->s_defer);
    (void)hipEventDestroy(h->ev_fork); (void)hipEventDestroy(h->ev_join); (void)hipEventDestroy(h->ev_fork2); (void)hipEventDestroy(h->ev_join2); (void)hipEventDestroy(h->ev_col);
  }
  (void)hipFree(h->slab);
  if (h->term_slab) (void)hipFree(h->term_slab);
  (void)hipHostFree(h->consumed_host);
  (void)hipHostFree(h->status_host);
  delete[] h->consumed_seen;
  delete h;
  return MCR_OK;
}

extern "C" int mcr_stage_episodes(mcr_env* h, const int32_t* env_ids, int n, const void* blobs, void* stream) {
  if (!h || !blobs || n < 0) { g_err = "bad argument"; return MCR_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  const int B = h->cfg.num_envs;
  // The staged slot of env e is ((installs & 1) ^ 1): slot 0 is "current" before the first install and every
  // install flips it.  `installs` is read from the mapped-host counter the install wrote; the device cannot
  // install again before this copy lands (staged_ready is 0 until then), so the slot chosen here is free.
  for (int i = 0; i < n; ++i) {
    const int e = env_ids ? env_ids[i] : i;
    if (e < 0 || e >= B) { g_err = "env id out of range"; return MCR_ERR_ARG; }
    const int32_t installs = ((volatile int32_t*)h->consumed_host)[e];
    uint8_t* dst = h->P.slots + ((size_t)e * 2 + ((installs & 1) ^ 1)) * MCR_SLOT_BYTES;
    HIPCHK(hipMemcpyAsync(dst, (const uint8_t*)blobs + (size_t)i * MCR_SLOT_BYTES, MCR_SLOT_BYTES, hipMemcpyHostToDevice, st));
  }
  // one id upload + one tiny kernel flip the n staged_ready flags (a 4-byte copy per env from pageable memory made
  // staging a whole batch take longer than generating it)
  if (n > 0) {
    const int32_t* d_ids = nullptr;
    if (env_ids) {
      HIPCHK(hipMemcpyAsync(h->stage_ids, env_ids, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, st));
      d_ids = h->stage_ids;
    }
    hipLaunchKernelGGL(k_mark_staged, dim3((n + 255) / 256), dim3(256), 0, st, h->P, d_ids, n);
    HIPCHK(hipGetLastError());
  }
  return MCR_OK;
}

static hipEvent_t get_event(mcr_env* h) {
  if (!h->free_events.empty()) { hipEvent_t e = h->free_events.back(); h->free_events.pop_back(); return e; }
  hipEvent_t e; (void)hipEventCreate(&e); return e;
}
#define LAUNCH(kid_, kernel, grid, block, st, ...) LAUNCH_LDS(kid_, kernel, grid, block, 0, st, __VA_ARGS__)
// a launch that completes the event `stop_` itself (nullptr: none): see mcr_view_launch
#define LAUNCH_LDS_STOP(kid_, kernel, grid, block, lds_, st, stop_, ...)                               \
  do {                                                                                   \
    TimedLaunch tl_; bool tm_ = (h->timing >> (kid_)) & 1;                                              \
    if (tm_) { tl_.id = (kid_); tl_.a = get_event(h); tl_.b = get_event(h); (void)hipEventRecord(tl_.a, st); } \
    hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(block), lds_, st, nullptr, stop_, 0, __VA_ARGS__);          \
    if (tm_) { (void)hipEventRecord(tl_.b, st); h->pending.push_back(tl_); }                   \
  } while (0)
#define LAUNCH_LDS(kid_, kernel, grid, block, lds_, st, ...)                                         \
  do {                                                                                   \
    TimedLaunch tl_; bool tm_ = (h->timing >> (kid_)) & 1;                                              \
    if (tm_) { tl_.id = (kid_); tl_.a = get_event(h); tl_.b = get_event(h); (void)hipEventRecord(tl_.a, st); } \
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), lds_, st, __VA_ARGS__);          \
    if (tm_) { (void)hipEventRecord(tl_.b, st); h->pending.push_back(tl_); }                   \
  } while (0)

// a launch of the dynamics over all envs: with or without the contact code (k_dynamics.h)
#define LAUNCH_DYN(kid_, cc_, grid, st, ...) do { if (cc_) LAUNCH(kid_, k_dynamics<true>, grid, 64, st, __VA_ARGS__); else LAUNCH(kid_, k_dynamics<false>, grid, 64, st, __VA_ARGS__); } while (0)

void mcr_view_launch(int variant, int grid, hipStream_t st, const McrParams& P, unsigned long long* stamps, int only_just_reset, hipEvent_t stop, hipEvent_t start);   // mcr_view.hip
// raster launch (k_view.h).  Main launches: one workgroup per work slot.  List launches (role >= 2): MCR_LIST_GRID
// persistent workgroups that walk the list (lane k of a wavefront holds a workgroup's k-th env, so never fewer than
// slots / 64 workgroups).
static int list_grid(int slots, int want) { return std::min(slots, std::max(want, (slots + 63) / 64)); }
static void launch_view(mcr_env* h, int kid, int slots, hipStream_t st, const McrParams& P, int only_just_reset, hipEvent_t stop = nullptr, int want_grid = 0) {
  TimedLaunch tl; const bool tm = (h->timing >> kid) & 1;
  // mcr_timing: where the launch has no stop event of the step's own, the two timing events ARE the dispatch's begin / end timestamps
  // (hipExtLaunchKernelGGL's start and stop events: the kernel's duration as a kernel trace reports it, no marker packets around the launch —
  // records in front of and behind the launch add the gap to the kernel before it: 85.2 against the trace's 82.6 us, round 6)
  const bool own_ts = tm && stop == nullptr;
  hipEvent_t start = nullptr;
  if (tm) { tl.id = kid; tl.a = get_event(h); tl.b = get_event(h); if (own_ts) { start = tl.a; stop = tl.b; } else (void)hipEventRecord(tl.a, st); }
  // (list launches: with more than two cars per env the contact list is long — N = 8: ~340 envs x 8 views per step — and 128
  // workgroups would draw ~20 views each, one after the other, at the end of the side stream's chain)
  if (P.role >= 2) { McrParams Q = P; Q.split_views = 1; mcr_view_launch(2, list_grid(slots * (Q.split_views ? P.N : 1), std::max(h->list_view_grid, want_grid)) + Q.flags_blocks, st, Q, h->view_stamps, only_just_reset, stop, start); }
  else mcr_view_launch((P.debug & 32) ? 1 : 0, slots, st, P, h->view_stamps, only_just_reset, stop, start);
  if (tm) { if (!own_ts) (void)hipEventRecord(tl.b, st); h->pending.push_back(tl); }
}

// reset(): install -> collide(1) -> dynamics(1) -> view, in order on one stream
static void launch_reset(mcr_env* h, McrParams P, hipStream_t st) {
  const int B = P.B, N = P.N;
  const int dyn_blocks = (B * P.G + 63) / 64;
  P.split = 0; P.role = 0; P.use_vorder = 0;
  hipLaunchKernelGGL(k_install, dim3(dyn_blocks), dim3(64), 0, st, P);
  LAUNCH_LDS(3, k_collide, B, 64, col::lds_bytes(N), st, P, 1);
  LAUNCH_DYN(4, P.car_contacts && N > 1, dyn_blocks, st, P, 1);
  if (P.obs) launch_view(h, 2, B, st, P, 1);
}

// step(): collide -> dynamics [-> auto-reset pass] -> view on the caller's stream `st`; with num_streams == 2 the step forks
// into three chains that meet again at the end (see the diagram inside; grids are sized for the worst case, surplus workgroups
// exit on their first load).  Contact envs: a wavefront holding a touching car<->car pair takes 2-4x as long as the others.
// Deferred envs: the few whose position loop is still iterating after 3 sweeps (a slow marginal crawl that would hold the whole
// main launch for up to 60).  An env of either list that ended its episode in this step (rare) takes its reset pass inside its
// own chain.
// may a step launched on caller stream `st` order its streams with phase words?  (mcr_bind_stream checked it; an unbound stream: events)
static bool stream_bound(const mcr_env* h, hipStream_t st) {
  for (const auto& b : h->bound) if (b.first == st) return b.second;
  return false;
}
// Do the contact chain's workgroups run their envs' contact pass themselves (McrParams::fuse_collide)?  With the contact pass beside the
// dynamics (the list then comes from the one-step-ahead verdicts) and the phase-word ordering (the side stream's last kernel empties the list)
// — measured +0.9 % (N = 2 default), +1.8 % (N = 4).  MCR_UNFUSED_COLLIDE=1: off (the round-4 topology).  (Round 5's first attempt diverged in 7
// of 10 rollouts with ~1000 contact envs: the chain's wavefronts, filling the machine at the step's begin, held the contact pass of the OTHER
// envs up long enough for a latent race of the contact-pass-beside-the-dynamics mode to show — k_dynamics.h "Parking overwrites", NOTES 11.)
static bool fused_collide(const mcr_env* h, hipStream_t st) {
  if (!cc_active(h) || h->unfused_collide) return false;
  hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(st, &capturing);
  return h->soft_sync && h->use_graph <= 0 && capturing == hipStreamCaptureStatusNone && stream_bound(h, st);
}
static void launch_step(mcr_env* h, McrParams P, hipStream_t st, int view_flags) {
  const int B = P.B, N = P.N;
  const int dyn_blocks = (B * P.G + 63) / 64;
  // list launches (contact / deferred / re-spawned envs): small grids whose workgroups walk the device-side lists
  const int lg_col = std::min(B, MCR_LIST_GRID), lg_dyn = std::min((B + MCR_SIDE_ENVS_PER_WAVE - 1) / MCR_SIDE_ENVS_PER_WAVE, h->chain_grid);
  const int lg_con = std::min(B, 2 * h->chain_grid);             // the contact chain: one env per wavefront
  const int prev_contacts = std::min(B, (int)((volatile uint32_t*)h->status_host)[MCR_STATUS_WORDS + HC_CONTACT_ENVS]);   // (mapped host word: no synchronisation)
  const bool draw = P.obs != nullptr;
  // (the raster workgroups reset the raster order entries they consume; a step that filled the list of its parity without
  // drawing — mcr_step without an observation buffer on a handle that has one — left it unconsumed: wipe it before its next use)
  if (h->vorder_dirty[h->step_parity]) { (void)hipMemsetAsync(h->P.vcount + (size_t)h->step_parity * (B + 2) + 2, 0xff, sizeof(int32_t) * (size_t)B, st); h->vorder_dirty[h->step_parity] = false; }
  if (!draw && h->cfg.obs_enabled) h->vorder_dirty[h->step_parity] = true;
  P.role = 0; P.split = h->split ? 1 : 0; P.defer_after = 0; P.respawn_list = 0;
  {   // the device-side lists of a step are double-buffered by step parity: no memset, and nobody zeroes a list somebody reads
    const size_t par = (size_t)h->step_parity, oth = par ^ 1;
    P.clist = h->P.clist + par * (B + 1); P.clist_next = h->P.clist + oth * (B + 1);
    P.dlist = h->P.dlist + par * (B + 1); P.rlist = h->P.rlist + par * (B + 1);
    P.vcount = h->P.vcount + par * (B + 2); P.vorder = P.vcount + 2;
    P.part = h->P.part + par * B; P.part_next = cc_active(h) ? h->P.part + oth * B : nullptr;
    P.next_counts[0] = h->P.dlist + oth * (B + 1); P.next_counts[1] = h->P.rlist + oth * (B + 1);
    P.next_counts[2] = h->P.vcount + oth * (B + 2); P.next_counts[3] = P.next_counts[2] + 1;
    if (h->term_slab && P.obs && P.actions) { P.term_cnt = h->term_cnt2 + par * 4; P.term_cnt_next = h->term_cnt2 + oth * 4; P.term_list = h->term_list2 + par * 2 * (size_t)P.term_cap; }
    else {
      // a step that draws no terminal frames (no actions: frame skip, step(None); no observation buffer) still flips the parity: it zeroes
      // the counters the NEXT step starts from in the drawing step's stead, and reports "no entries" (the next terminal step of
      // that parity started from the counts of two steps back)
      if (h->term_slab) {
        (void)hipMemsetAsync(h->term_cnt2 + oth * 4, 0, sizeof(int32_t) * 4, st);
        if (P.term_count_out) (void)hipMemsetAsync(P.term_count_out, 0, sizeof(int32_t), st);
      }
      P.term_cnt = P.term_cnt_next = nullptr; P.term_idx = nullptr;
    }
    h->step_parity ^= 1;
  }
  if (!h->split) {
    // single stream: collide -> dynamics -> reset pass of the re-spawned envs (:408) -> raster -> bookkeeping, all envs
    LAUNCH_LDS(0, k_collide, B, 64, col::lds_bytes(N), st, P, 0);
    LAUNCH_DYN(1, P.car_contacts && N > 1, dyn_blocks, st, P, 0);
    if (P.auto_reset) {
      if (P.term_cnt) hipLaunchKernelGGL(k_term_prep, dim3(B), dim3(64), 0, st, P);     // (the terminal entries of the envs that end here, before the pass clears their tile flags)
      LAUNCH_LDS(3, k_collide, B, 64, col::lds_bytes(N), st, P, 1);
      LAUNCH_DYN(4, P.car_contacts && N > 1, dyn_blocks, st, P, 1);
    }
    P.use_vorder = 1;
    if (draw) launch_view(h, 2, B, st, P, 0);
    P.use_vorder = 0;
    if (P.term_cnt) { McrParams Pt = P; Pt.role = 6; launch_view(h, 7, B, st, Pt, 0); hipLaunchKernelGGL(k_term_finish, dim3(1), dim3(256), 0, st, P); }
    if (view_flags) hipLaunchKernelGGL(k_flags, dim3(B * N), dim3(64), 0, st, P);
    return;
  }
  // Three chains that meet again at the end (k_list_chain.h: a list chain is one fused launch + its raster):
  //   st      : -+-> dynamics(main envs, 3 position sweeps) -+-> chain(resume the deferred envs || reset pass of the re-spawned envs) -> their raster -+
  //   s_side  :  +-> collide(all) -> chain(contact envs) -> their raster ---------------------------------------------------------------------------+
  //   s_defer :                                              +-> view records, bookkeeping(main envs) -> raster(main envs) -------------------------+
  // ordered by phase words that kernels post and await (the first branch below) or by events (the second).
  // Contact envs: a wavefront holding a touching car<->car pair takes 2-4x as long as the others.  Deferred envs: the
  // few whose position loop is still iterating after 3 sweeps (a slow marginal crawl that would hold the whole main
  // launch for up to 60).  Re-spawned envs (auto-reset, ~B/1000 per step): their reset pass would hold the raster of
  // everybody else (they are not in the main raster's list, see k_dynamics).
  P.defer_after = MCR_DEFER_AFTER; P.respawn_list = P.auto_reset ? 1 : 0; P.list_envs_per_block = MCR_SIDE_ENVS_PER_WAVE;
  // The contact pass runs on the side stream BESIDE the main dynamics (cc_mode, mcr_kernels.h): nothing the solver needs
  // comes from it (Car.step reads the wheels' tile bits of the PREVIOUS pass; which envs are the contact chain's follows
  // from the entry poses), only the step's bookkeeping at the end of the dynamics kernel does, and that waits for
  // k_collide's per-env "done" word.  24 us + a kernel boundary off the critical path.
  // Where kernels of different streams do not overlap (mcr_create probes it: counter-collecting profilers, debuggers) the
  // contact pass simply runs first, on the caller's stream.
    const bool cc = cc_active(h);
  P.cc_mode = cc ? 1 : 0; P.epoch = h->step_count; P.epoch_ptr = nullptr;
  P.fuse_collide = fused_collide(h, st) ? 1 : 0;
  if (h->use_graph > 0) {
    // a replayed graph has constant arguments: the epoch lives in a device-side counter that the first node of the step advances
    // (before the fork: the contact pass and the dynamics read the same value)
    hipLaunchKernelGGL(k_step_begin, dim3(1), dim3(1), 0, st, h->dev_step_ctr);
    P.epoch = 0; P.epoch_ptr = h->dev_step_ctr;
  }
  hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(st, &capturing);
  if (h->soft_sync && h->use_graph <= 0 && capturing == hipStreamCaptureStatusNone && stream_bound(h, st)) {
    // ---- the three chains ordered by phase words (mcr_kernels.h: soft_sync) — no marker / barrier packets on any stream:
    //   s_side  : await(BEGIN) -> [collide -> post(COL)] -> chain(contact envs) -> raster -> post(SIDE)
    //   st      : [posts BEGIN] dynamics(main envs) -> [posts DYN, awaits COL] chain(resume + re-spawned envs) -> raster -> await(SIDE, MAIN)
    //   s_defer : await(DYN, COL) -> view records -> bookkeeping(main envs) -> raster(main envs) -> post(MAIN)
    // (kernel trace, round 3: the resume chain starts 3 us after the dynamics instead of 11-21, the next step's dynamics 0-1 us after
    // the step's last kernel instead of 20-32; the chains' bookkeeping: list launches behind the chains; N > 3: the main envs'
    // bookkeeping on the caller's stream)
    P.soft_sync = 1;
    const int lg_flags = std::min(B * N, (N <= 2 ? 4 : 32) * MCR_LIST_GRID);
    if (!cc) LAUNCH_LDS(0, k_collide, B, 64, col::lds_bytes(N), st, P, 0);
    // The step's critical kernel goes out FIRST (round 6): everything the other two streams run in this step waits for a word the dynamics posts
    // (BEGIN from its first thread), so the order of enqueueing across the streams is free — and behind a host synchronisation (an RL loop
    // reads its observations every step) the seven launches that used to precede it cost the step their enqueue time, ~45 us.
    const int vif = (view_flags && draw && h->viewprep_in_flags) ? 1 : 0;
    { McrParams Pd = P; Pd.split = 0; Pd.role = 1; Pd.post_dyn = h->post_dyn ? 1 : 0; Pd.viewprep_in_flags = vif; LAUNCH(1, k_dynamics<false>, dyn_blocks, 64, st, Pd, 0); }          // (the main envs: no touching car<->car pair)
    hipLaunchKernelGGL(k_await, dim3(1), dim3(64), 0, h->s_side, P, (int)W_BEGIN, -1);
    if (cc && !P.fuse_collide) LAUNCH_LDS(0, k_collide, B, 64, col::lds_bytes(N), h->s_side, P, 0);     // (W_COL: posted by the chain that follows)
    if (cc && P.fuse_collide) {
      // the contact pass of the main envs on the THIRD stream (idle until the dynamics is through), the contact chain — each workgroup with its
      // env's own contact pass in front — on the side stream from the step's begin: the chain, the step's critical path when cars pile up, no
      // longer waits for the 4096-env launch
      hipLaunchKernelGGL(k_await, dim3(1), dim3(64), 0, h->s_defer, P, (int)W_BEGIN, -1);
      LAUNCH_LDS(0, k_collide, B, 64, col::lds_bytes(N), h->s_defer, P, 0);
      hipLaunchKernelGGL(k_post, dim3(1), dim3(64), 0, h->s_defer, P, (int)W_COL);
    }
    P.split = 0;
    P.role = 2;
    { McrParams Pc = P; Pc.list_envs_per_block = 1; LAUNCH_LDS(5, k_list_chain<true>, lg_con, 64, col::lds_bytes(N), h->s_side, Pc, Pc, 0, lg_con); }   // ONE contact env per wavefront: the uniform contact sweeps (k_dynamics.h)
    // (the chains' bookkeeping: workgroups of its own inside the chain's raster launch when there is one, a list launch otherwise)
    // (beyond four cars per env the lists hold thousands of cars — ~315 contact envs x 8 at N = 8 — and a few 256-thread workgroups
    // would take them in many rounds at the end of the contact chain, the critical path there; measured N = 2 15.37 -> 15.65 M
    // env-steps/s, N = 4 11.00 -> 11.07, N = 8 5.47 -> 5.44)
    const int fiv = (view_flags && draw && N <= 8) ? (N <= 2 ? 8 : 64) : 0;
    // (the contact list's raster and bookkeeping workgroups: sized by the LAST step's list — the lists change slowly — so that a long list,
    // a policy that drives: ~90 envs of 4096, takes one round of workgroups instead of two or three behind the chain, the step's critical path)
    const int fiv_c = fiv ? std::min(512, std::max(fiv, (prev_contacts * (N + 1) + 3) / 4 + 2)) : 0;
    const int vg_c = std::min(2048, prev_contacts * N + prev_contacts * N / 4 + 8);
    if (view_flags && !fiv) hipLaunchKernelGGL(k_flags_list, dim3(lg_flags), dim3(64), 0, h->s_side, P);
    if (draw) { McrParams Pv = P; Pv.flags_blocks = fiv_c; launch_view(h, 6, P.term_cnt ? 2 * B : B, h->s_side, Pv, 0, nullptr, vg_c); }
    hipLaunchKernelGGL(k_post, dim3(1), dim3(64), 0, h->s_side, P, (int)W_SIDE);
    P.role = 1;
    P.viewprep_in_flags = vif;
    const bool flags_on_caller = view_flags && draw && !P.viewprep_in_flags && N <= 7;   // (N = 8: the raster would share the machine with the bookkeeping: 6.70 -> 6.49 M env-steps/s, round 5)
    P.role = 3;
    {
      const int ga = std::min(lg_dyn, MCR_LIST_GRID / 2), gb = P.auto_reset ? lg_col : 0;
      McrParams Pr = P; Pr.role = 4; Pr.list_envs_per_block = 1;
      LAUNCH_LDS(7, k_list_chain<false>, ga + gb, 64, col::lds_bytes(N), st, P, Pr, 0, ga);
      if (view_flags && !fiv) hipLaunchKernelGGL(k_flags_list, dim3(lg_flags), dim3(64), 0, st, P);
      // Beyond three cars per env the third stream's chain (bookkeeping of B*N cars, then B*N views) is the longer one and the caller's
      // has slack: the main envs' bookkeeping — which the raster does not depend on — moves here, between the resume chain and its raster
      if (flags_on_caller) { McrParams Pm = P; Pm.role = 1; hipLaunchKernelGGL(k_flags, dim3(B * N), dim3(64), 0, st, Pm); }
      if (draw) { McrParams Pv = P; Pv.role = P.auto_reset ? 5 : 3; Pv.await_tail = 1; Pv.flags_blocks = fiv; launch_view(h, 7, P.term_cnt ? 2 * B : B, st, Pv, 0); }     // ... and the step's join (+ the terminal entries' frames and count)
    }
    P.role = 1;
    hipLaunchKernelGGL(k_await, dim3(1), dim3(64), 0, h->s_defer, P, (int)W_DYN, (cc && !P.fuse_collide) ? (int)W_COL : -1);
    if (P.viewprep_in_flags) hipLaunchKernelGGL(k_flags_viewprep, dim3(dyn_blocks + B * N), dim3(64), 0, h->s_defer, P, dyn_blocks);      // view records + bookkeeping: one launch
    else if (view_flags && !flags_on_caller) hipLaunchKernelGGL(k_flags, dim3(B * N), dim3(64), 0, h->s_defer, P);
    P.use_vorder = 1;
    if (draw) launch_view(h, 2, B, h->s_defer, P, 0);
    P.use_vorder = 0;
    hipLaunchKernelGGL(k_post, dim3(1), dim3(64), 0, h->s_defer, P, (int)W_MAIN);
    if (!draw) hipLaunchKernelGGL(k_await, dim3(1), dim3(64), 0, st, P, (int)W_SIDE, (int)W_MAIN);
    return;
  }
  // Events: where the kernel an event marks is known, the launch completes the event itself (hipExtLaunchKernelGGL's stop event:
  // the dispatch packet's own completion signal).  A hipEventRecord is a marker packet BEHIND the kernel: +3 us before the next
  // kernel of the same stream, 11 instead of 7.5 us for a dependent kernel on another stream (tools/ubench/event_gap.hip), and the
  // step has four of them on its critical path.  Not inside a graph capture (plain records there).
  const bool sev = h->stop_events && h->use_graph <= 0 && capturing == hipStreamCaptureStatusNone;
#define STOP(ev) (sev ? (ev) : (hipEvent_t) nullptr)
#define RECORD_UNLESS_STOP(ev, stream) do { if (!sev) (void)hipEventRecord(ev, stream); } while (0)
  if (!cc) LAUNCH_LDS_STOP(0, k_collide, B, 64, col::lds_bytes(N), st, STOP(h->ev_col), P, 0);
  (void)hipEventRecord(h->ev_fork, st);
  (void)hipStreamWaitEvent(h->s_side, h->ev_fork, 0);
  if (cc) LAUNCH_LDS_STOP(0, k_collide, B, 64, col::lds_bytes(N), h->s_side, STOP(h->ev_col), P, 0);
  RECORD_UNLESS_STOP(h->ev_col, cc ? h->s_side : st);
  P.split = 0;
  // bookkeeping of a chain's cars: fused into the chain for N <= 2 (2 envs x N cars take their turns on one wavefront),
  // a list launch of its own beyond that
  const int fuse_flags = (view_flags && N <= 2) ? 1 : 0;
  const int lg_flags = std::min(B * N, (N <= 2 ? 4 : 32) * MCR_LIST_GRID);
  const bool flags_list = view_flags && !fuse_flags;
  P.role = 2;
  // (the side stream's last kernel completes ev_join; which one that is depends on the step's shape)
  { McrParams Pc = P; Pc.list_envs_per_block = 1; LAUNCH_LDS_STOP(5, k_list_chain<true>, lg_con, 64, col::lds_bytes(N), h->s_side, STOP((!draw && !flags_list) ? h->ev_join : nullptr), Pc, Pc, fuse_flags, lg_con); }
  if (flags_list) hipExtLaunchKernelGGL(k_flags_list, dim3(lg_flags), dim3(64), 0, h->s_side, nullptr, STOP(!draw ? h->ev_join : nullptr), 0, P);
  if (draw) launch_view(h, 6, P.term_cnt ? 2 * B : B, h->s_side, P, 0, STOP(h->ev_join), std::min(2048, prev_contacts * N + prev_contacts * N / 4 + 8));
  P.role = 1;
  // the main envs' view records and car polygons: by k_viewprep on the side stream, beside the bookkeeping kernel, in a drawn step
  // with actions; otherwise by the dynamics' own epilogue
  P.viewprep_in_flags = (view_flags && draw && h->viewprep_in_flags) ? 1 : 0;
  LAUNCH_LDS_STOP(1, k_dynamics<false>, dyn_blocks, 64, 0, st, STOP(h->ev_fork2), P, 0);
  RECORD_UNLESS_STOP(h->ev_fork2, st);
  // The chain that follows the dynamics IN-STREAM starts ~2 us after it, one that has to hop to another stream ~8 us (kernel trace,
  // round 3).  The longer chain is the resume chain (85-150 us beside the raster + its own raster, 25-70 us) — not bookkeeping + main
  // raster (40 + 100 us): it keeps the caller's stream and gets its wavefronts placed before the raster starts (a chain that starts
  // beside a raster that fills every CU runs 2-3x slower); the main envs' bookkeeping + raster take the hop to the third stream.
  hipStream_t s_resume = st, s_mainview = h->s_defer;
  (void)hipStreamWaitEvent(h->s_defer, h->ev_fork2, 0);
  (void)hipStreamWaitEvent(s_resume, h->ev_col, 0);              // the resume chain reads the contact pass's results of its envs (a deferred env never got to the main dynamics' in-kernel wait)
  if (!draw) (void)hipStreamWaitEvent(h->s_side, h->ev_fork2, 0);
  P.role = 3;
  const hipEvent_t resume_done = nullptr;     // (the caller's stream needs no event for its own chain)
  {
    // the resume chain and — same launch, workgroups of their own — the reset pass (:408, ~50 us of serial solver work) of the
    // envs the main dynamics re-spawned; then, in one list launch, the deferred envs' frames and the re-spawned envs' first observations
    const int ga = std::min(lg_dyn, MCR_LIST_GRID / 2), gb = P.auto_reset ? lg_col : 0;
    McrParams Pr = P; Pr.role = 4; Pr.list_envs_per_block = 1;     // one re-spawned env per workgroup: they run side by side
    LAUNCH_LDS_STOP(7, k_list_chain<false>, ga + gb, 64, col::lds_bytes(N), s_resume, STOP((!draw && !flags_list) ? resume_done : nullptr), P, Pr, fuse_flags, ga);
    if (flags_list) hipExtLaunchKernelGGL(k_flags_list, dim3(lg_flags), dim3(64), 0, s_resume, nullptr, STOP(!draw ? resume_done : nullptr), 0, P);
    if (draw) { McrParams Pv = P; Pv.role = P.auto_reset ? 5 : 3; launch_view(h, 7, P.term_cnt ? 2 * B : B, s_resume, Pv, 0, STOP(resume_done)); }
  }
  P.role = 1;
  RECORD_UNLESS_STOP(h->ev_join, h->s_side);
  // The bookkeeping of the main envs (:446-495; k_flags.h, one wavefront per car) needs the poses only.  It runs right
  // before the raster, and those ~16 us are what the list chains — forked off at the same moment — need to get their
  // wavefronts placed: a chain that starts beside a raster that already fills every CU runs 2-3x slower (measured).
  // (k_viewprep in front of it, in-stream: the side stream may be held by a long contact chain, and a further stream slows every queue)
  const hipEvent_t main_done = h->ev_join2;
  if (P.viewprep_in_flags) hipLaunchKernelGGL(k_viewprep, dim3(dyn_blocks), dim3(64), 0, s_mainview, P);
  if (view_flags) hipExtLaunchKernelGGL(k_flags, dim3(B * N), dim3(64), 0, s_mainview, nullptr, STOP(!draw ? main_done : nullptr), 0, P);
  (void)hipStreamWaitEvent(s_mainview, h->ev_col, 0);              // the raster reads the tiles' recolour flags (long done)
  P.use_vorder = 1;
  if (draw) launch_view(h, 2, B, s_mainview, P, 0, STOP(main_done));
  P.use_vorder = 0;
  // the join: the caller's stream waits for the other two (waiting for them one behind the other — one wait on the caller's
  // stream — costs a second hop: 23 vs 18.5 us after the later of the two, tools/ubench/event_gap.hip)
  if (!sev || (!draw && !view_flags)) (void)hipEventRecord(h->ev_join2, h->s_defer);
  (void)hipStreamWaitEvent(st, h->ev_join, 0);
  (void)hipStreamWaitEvent(st, h->ev_join2, 0);
  if (P.term_cnt) hipLaunchKernelGGL(k_term_finish, dim3(1), dim3(256), 0, st, P);     // (the phase-word path: the join's workgroup does it)
#undef STOP
#undef RECORD_UNLESS_STOP
}

// The kernels report conditions that make results wrong in mapped host memory (mcr_kernels.h: ST_*); read here without
// synchronising, like the install counters — a condition raised by a step that is still running shows up in a later call.
static int check_status(mcr_env* h) {
  static const char* what[MCR_STATUS_WORDS] = {
      "a kernel gave up waiting for the kernels of another stream (three-chain step: the contact pass of an env, a phase word); the handle now runs the contact pass in front and orders its streams with events",
      "the contact pass disagreed with the one-step-ahead touch verdict; the handle now runs the contact pass in front of the dynamics", "", "", "", "", "", ""};
  // FATAL: a wait that gave up, a verdict mismatch — results of that step are wrong.  The overflow words (ST_CC_OVERFLOW, ST_EVENT_OVERFLOW) are
  // documented capacity deviations (the excess was dropped and flagged): the rollout goes on, mcr_status / VecMultiCarRacing.status_words show them.
  for (int i : {(int)ST_SPIN_GIVEUP, (int)ST_VERDICT}) {
    const uint32_t v = ((volatile uint32_t*)h->status_host)[i];
    if (v != h->status_seen[i]) {
      h->status_seen[i] = v;
      if (i == ST_SPIN_GIVEUP) { h->concurrent_collide = false; h->soft_sync = false; h->verdict_fresh = false; }
      if (i == ST_VERDICT) { h->concurrent_collide = false; h->verdict_fresh = false; }     // (the contact pass in front from now on: it marks the contact chain's envs itself)
      g_err = std::string("results of an earlier step are wrong: ") + what[i] + " (" + std::to_string(v) + " so far)";
      return MCR_ERR_STATE;
    }
  }
  return MCR_OK;
}
extern "C" int mcr_status(mcr_env* h, uint32_t* out, int n_words) {
  if (!h || !out || n_words < 0) return MCR_ERR_ARG;
  for (int i = 0; i < n_words && i < MCR_STATUS_WORDS; ++i) out[i] = ((volatile uint32_t*)h->status_host)[i];
  return MCR_OK;
}

extern "C" int mcr_reset(mcr_env* h, const uint8_t* d_env_mask, uint8_t* d_obs, void* stream) {
  if (!h) { g_err = "null handle"; return MCR_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  McrParams P = h->P;
  P.reset_mask = d_env_mask; P.obs = h->cfg.obs_enabled ? d_obs : nullptr; P.actions = nullptr;
  launch_reset(h, P, st);
  HIPCHK(hipGetLastError());
  h->any_reset = true; h->verdict_fresh = false;
  return MCR_OK;
}

extern "C" int mcr_step(mcr_env* h, const float* d_actions, uint8_t* d_obs, double* d_reward, uint8_t* d_done, uint8_t* d_trunc, void* stream) {
  if (!h || !d_reward || !d_done) { g_err = "null argument"; return MCR_ERR_ARG; }
  if (!h->any_reset) { g_err = "step() before reset()"; return MCR_ERR_STATE; }
  if (int rc = check_status(h)) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (h->use_graph <= 0) {
    // a step captured by the CALLER cannot be replayed: its launches carry this step's epoch and list parity as constants.  The supported
    // way to replay a step is mcr_set_step_graph (the handle captures both parities itself and keeps the epoch in device memory).
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) { g_err = "mcr_step inside a stream capture: use mcr_set_step_graph for graph replay"; return MCR_ERR_STATE; }
  }
  McrParams P = h->P;
  P.actions = d_actions; P.obs = h->cfg.obs_enabled ? d_obs : nullptr;
  P.reward_out = d_reward; P.done_out = d_done; P.trunc_out = d_trunc;
  P.bp_fresh = h->bp_fresh ? 1 : 0; h->bp_fresh = false;
  h->step_count = (int32_t)((uint32_t)h->step_count + 1u);      // (wraps: epochs are compared as 32-bit distances, mcr_kernels.h)
  // with auto_reset, finished envs are re-spawned on the device and take the action-less first step of their
  // new episode inside this call; the view kernel always runs (it also owns the backward/on-grass flags)
  const int vf = d_actions ? 1 : 0;
  // (the contact list of a fused step is made by the step before it, if that one was fused too: a change of mode — another caller stream, a
  // give-up — is treated like stale verdicts one way, and empties the half-made list the other way)
  const bool fz = fused_collide(h, st);
  if (h->last_fused && !fz) (void)hipMemsetAsync(h->P.clist + (size_t)h->step_parity * (P.B + 1), 0, sizeof(int32_t), st);
  // ... and into fused mode: this step's verdict writers APPEND to the other parity's list, which a fused dynamics no longer zeroes and
  // which an unfused step's contact pass filled two steps ago and nobody emptied (the next step's chain re-stepped stale envs)
  if (cc_active(h) && fz && !h->last_fused) (void)hipMemsetAsync(h->P.clist + (size_t)(h->step_parity ^ 1) * (P.B + 1), 0, sizeof(int32_t), st);
  if (cc_active(h) && (!h->verdict_fresh || (fz && !h->last_fused))) {   // after reset() / reset_envs() / a state restore / a step without actions: which envs hold a touching car<->car pair?
    McrParams Pt = P; Pt.role = 0; Pt.part = h->P.part + (size_t)h->step_parity * P.B;
    Pt.fuse_collide = fused_collide(h, st) ? 1 : 0; Pt.clist = h->P.clist + (size_t)h->step_parity * (P.B + 1);
    if (Pt.fuse_collide) (void)hipMemsetAsync(Pt.clist, 0, sizeof(int32_t), st);
    hipLaunchKernelGGL(k_touch, dim3(P.B), dim3(64), 0, st, Pt);
  }
  h->verdict_fresh = vf != 0;             // this step's bookkeeping evaluates the next step's
  h->last_fused = fz && vf != 0;
  if (h->use_graph > 0 && !h->timing) {
    // The step is a fixed sequence of ~13 launches on up to three streams whose arguments only change with the
    // contact-list parity: it can be replayed as a hipGraph (measured r02: 0.4 % faster — the gaps between the step's
    // dependent kernels are GPU-side drain/start-up, not host launch cost — so VecMultiCarRacing leaves it off).  Any change of an argument
    // (other buffers, another stream, debug switches) re-captures.
    const int par = h->step_parity;
    mcr_env::StepGraph& G = h->sg[par];
    if (G.valid && G.st == st && G.view_flags == vf && memcmp(&G.P, &P, sizeof(P)) == 0) {
      h->step_parity ^= 1;                              // what launch_step does on the host side
      HIPCHK(hipGraphLaunch(G.exec, st));
      return MCR_OK;
    }
    if (G.valid) { (void)hipGraphExecDestroy(G.exec); (void)hipGraphDestroy(G.graph); G.valid = false; }
    const int parity_before = h->step_parity;
    if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
      launch_step(h, P, st, vf);
      hipGraph_t graph = nullptr;
      if (hipStreamEndCapture(st, &graph) == hipSuccess && graph && hipGraphInstantiate(&G.exec, graph, nullptr, nullptr, 0) == hipSuccess) {
        G.graph = graph; G.P = P; G.st = st; G.view_flags = vf; G.valid = true;
        HIPCHK(hipGraphLaunch(G.exec, st));
        return MCR_OK;
      }
      if (graph) (void)hipGraphDestroy(graph);
    }
    (void)hipGetLastError();
    h->use_graph = -1;                                 // capture is not available here: plain launches from now on
    h->step_parity = parity_before;
  }
  launch_step(h, P, st, vf);
  HIPCHK(hipGetLastError());
  return MCR_OK;
}

// Phase words need the caller's stream and the two internal ones on hardware queues of their own: a waiting kernel at the head of a shared
// queue would hold back the very kernel it waits for (HIP multiplexes streams of one priority over a few queues; the internal streams are
// high-priority ones, so an ordinary caller's stream never shares theirs).  Checked HERE, once per stream, with two probe kernels and
// device synchronisations (~1 ms) — never inside mcr_step, which launches and returns.
extern "C" int mcr_bind_stream(mcr_env* h, void* stream) {
  if (!h) { g_err = "null handle"; return MCR_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  for (const auto& b : h->bound) if (b.first == st) return MCR_OK;
  bool ok = false;
  if (h->split && h->soft_sync) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &cap);
    if (cap != hipStreamCaptureStatusNone) { g_err = "mcr_bind_stream on a capturing stream"; return MCR_ERR_STATE; }
    HIPCHK(hipSetDevice(h->cfg.device));
    ok = kernels_overlap(h->s_side, st) && kernels_overlap(h->s_defer, st);
    // ... and of every other phase-word handle of the device: their awaits must not sit in front of this stream's posts either
    std::lock_guard<std::mutex> lk(g_soft_mu);
    for (mcr_env* o : g_soft_list) if (ok && o != h && o->cfg.device == h->cfg.device) ok = kernels_overlap(o->s_side, st) && kernels_overlap(o->s_defer, st);
  }
  if (h->bound.size() >= 64) h->bound.erase(h->bound.begin());
  h->bound.emplace_back(st, ok);
  return MCR_OK;
}

extern "C" int mcr_set_terminal_obs(mcr_env* h, uint8_t* d_term_obs, int32_t* d_term_ids, int32_t* d_term_count, int cap) {
  if (!h) { g_err = "null handle"; return MCR_ERR_ARG; }
  HIPCHK(hipSetDevice(h->cfg.device));
  HIPCHK(hipDeviceSynchronize());
  for (auto& g : h->sg) if (g.valid) { (void)hipGraphExecDestroy(g.exec); (void)hipGraphDestroy(g.graph); g.valid = false; }
  if (h->term_slab) { (void)hipFree(h->term_slab); h->term_slab = nullptr; }
  McrParams& P = h->P;
  P.term_obs = nullptr; P.term_ids = nullptr; P.term_count_out = nullptr; P.term_cap = 0; P.term_idx = nullptr;
  if (!d_term_obs && !d_term_ids && !d_term_count) return MCR_OK;                      // off
  if (!d_term_obs || !d_term_ids || !d_term_count || cap < 1) { g_err = "mcr_set_terminal_obs: all three buffers and cap >= 1, or all null"; return MCR_ERR_ARG; }
  if (!h->cfg.obs_enabled || !h->cfg.auto_reset) { g_err = "terminal observations need obs_enabled and auto_reset"; return MCR_ERR_STATE; }
  cap = std::min(cap, h->cfg.num_envs);
  const size_t N = h->cfg.num_agents, CN = (size_t)cap * N, B = h->cfg.num_envs;
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
  const size_t o_cnt = carve(sizeof(int32_t) * 8), o_list = carve(sizeof(int32_t) * 4 * cap), o_idx = carve(sizeof(int32_t) * B);
  const size_t o_carf = carve(sizeof(float) * CF_COUNT * CN), o_card = carve(sizeof(double) * CD_COUNT * CN);
  const size_t o_vp = carve(sizeof(float) * MCR_VIEWP_FLOATS * CN), o_cp = carve(sizeof(float) * MCR_CARPOLY_FLOATS * CN);
  const size_t o_tf = carve(sizeof(uint16_t) * MCR_TILE_CAP * (size_t)cap), o_te = carve(sizeof(McrTermEnv) * cap);
  if (hipMalloc(&h->term_slab, off) != hipSuccess) { h->term_slab = nullptr; g_err = "hipMalloc failed"; return MCR_ERR_HIP; }
  uint8_t* base = (uint8_t*)h->term_slab;
  HIPCHK(hipMemset(base, 0, off));
  HIPCHK(hipMemset(base + o_idx, 0xff, sizeof(int32_t) * B));
  HIPCHK(hipMemset(d_term_count, 0, sizeof(int32_t)));
  h->term_cnt2 = (int32_t*)(base + o_cnt); h->term_list2 = (int32_t*)(base + o_list);
  P.term_obs = d_term_obs; P.term_ids = d_term_ids; P.term_count_out = d_term_count; P.term_cap = cap;
  P.term_idx = (int32_t*)(base + o_idx); P.term_carf = (float*)(base + o_carf); P.term_card = (double*)(base + o_card);
  P.term_viewp = (float*)(base + o_vp); P.term_carpoly = (float*)(base + o_cp); P.term_tflags = (uint16_t*)(base + o_tf); P.term_env = (McrTermEnv*)(base + o_te);
  return MCR_OK;
}

extern "C" int mcr_set_step_graph(mcr_env* h, int enable) {
  if (!h) { g_err = "null handle"; return MCR_ERR_ARG; }
  if (enable && h->use_graph <= 0) {     // the device-side epoch counter continues the host's (both advance once per step from here on)
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(h->dev_step_ctr, &h->step_count, sizeof(int32_t), hipMemcpyHostToDevice));
  }
  h->use_graph = enable ? 1 : 0;
  return MCR_OK;
}

// blockIdx.y = step offset: out[nsteps][n_cars][3]
__global__ void k_synth_actions(float* __restrict__ out, int n_cars, int N, unsigned long long seed, unsigned t, unsigned env_offset) {
  const int ci = blockIdx.x * blockDim.x + threadIdx.x;
  if (ci >= n_cars) return;
  float a[3];
  mcr_synth_action(seed, env_offset + (unsigned)(ci / N), (unsigned)(ci % N), t + blockIdx.y, a);
  float* o = out + ((size_t)blockIdx.y * n_cars + ci) * 3;
  o[0] = a[0]; o[1] = a[1]; o[2] = a[2];
}
extern "C" int mcr_synth_actions_block(mcr_env* h, float* d_actions, uint64_t seed, uint32_t t0, int nsteps, uint32_t env_offset, void* stream) {
  if (!h || !d_actions || nsteps < 1 || nsteps > 65535) { g_err = "bad argument"; return MCR_ERR_ARG; }
  hipLaunchKernelGGL(k_synth_actions, dim3((h->P.BN + 255) / 256, nsteps), dim3(256), 0, (hipStream_t)stream, d_actions, h->P.BN, h->P.N,
                     (unsigned long long)seed, t0, env_offset);
  HIPCHK(hipGetLastError());
  return MCR_OK;
}
extern "C" int mcr_synth_actions(mcr_env* h, float* d_actions, uint64_t seed, uint32_t t, uint32_t env_offset, void* stream) {
  return mcr_synth_actions_block(h, d_actions, seed, t, 1, env_offset, stream);
}

extern "C" int mcr_render(mcr_env* h, int env, int width, int height, uint8_t* d_out, void* stream) {
  if (!h || !d_out) { g_err = "null argument"; return MCR_ERR_ARG; }
  if (env < 0 || env >= h->cfg.num_envs || width < 1 || height < 1 || width > 4096 || height > 4096) { g_err = "env / viewport out of range"; return MCR_ERR_ARG; }
  if (!h->any_reset) { g_err = "render() before reset()"; return MCR_ERR_STATE; }
  if (!h->cfg.obs_enabled) { g_err = "render() needs obs_enabled (the camera and car polygons are produced for the observation path)"; return MCR_ERR_STATE; }
  McrParams P = h->P;
  const dim3 grid((width + RENDER_TILE - 1) / RENDER_TILE, (height + RENDER_TILE - 1) / RENDER_TILE, P.N);
  hipLaunchKernelGGL(k_render_frame, grid, dim3(256), 0, (hipStream_t)stream, P, env, width, height, d_out);
  HIPCHK(hipGetLastError());
  return MCR_OK;
}

extern "C" int mcr_read_rollout_stats(mcr_env* h, double* out2, int reset) {
  if (!h || !out2) { g_err = "null argument"; return MCR_ERR_ARG; }
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out2, h->P.stats, sizeof(double) * 2, hipMemcpyDeviceToHost));
  if (reset) HIPCHK(hipMemset(h->P.stats, 0, sizeof(double) * 2));
  return MCR_OK;
}

extern "C" int mcr_set_episode_stats(mcr_env* h, double* d_ep_return, int32_t* d_ep_len) {
  if (!h) { g_err = "null handle"; return MCR_ERR_ARG; }
  h->P.ep_return_out = d_ep_return; h->P.ep_len_out = d_ep_len;
  return MCR_OK;
}

extern "C" int mcr_poll_consumed(mcr_env* h, int32_t* env_ids_out, int cap, void* stream) {
  if (!h) return MCR_ERR_ARG;
  if (h->svc) { g_err = "mcr_poll_consumed: the handle's refill service owns the counters (mcr_refill_start)"; return MCR_ERR_STATE; }
  (void)stream;   // counters live in mapped host memory: no device synchronisation needed
  int n = 0;
  const int B = h->cfg.num_envs;
  for (int e = 0; e < B; ++e) {
    int32_t c = ((volatile int32_t*)h->consumed_host)[e];
    if (c != h->consumed_seen[e]) {
      if (n < cap && env_ids_out) { env_ids_out[n] = e; h->consumed_seen[e] = c; ++n; }
      else if (!env_ids_out) ++n;
    }
  }
  return n;
}

// ---------------------------------------------------------------------------- the refill service
// What VecMultiCarRacing's Python worker did every step — poll the consumed-episode counters, generate the next episode of every env that
// re-spawned (its own RNG streams), stage it — as native threads of the handle: a service thread (polls, queues env ids, stages finished
// tracks) and up to six generator threads (one track at a time off the queue).  No interpreter, no GIL hand-offs, no gather / scatter of RNG
// states, no bounce buffer (the env's row of the caller's pinned blob array is the staging source), no condition variable or per-batch barrier
// on the path.  Host share of a rank: bench.py --emulate-world 8.  The arrays belong to the caller and must outlive the service.
#include <thread>
#include <deque>
#include <condition_variable>
#include <atomic>
#include <unistd.h>
#include <sys/prctl.h>
extern "C" int mcr_episodes_generate_rows(uint32_t*, uint32_t*, const int32_t*, int, int, int, void*, int32_t*, int);   // mcr_host.cpp
struct RefillSvc {
  std::thread th;                      // polls the counters, hands env ids to the generators, stages what they finished
  std::vector<std::thread> gens;       // generator threads: one track at a time each, off a queue — AWAKE while there is work (they poll with
                                       // short sleeps and back off when idle): a batch handed to sleeping helpers through a condition variable
                                       // cost the wake-up latency of the slowest helper per batch, and when the helpers woke late the calling
                                       // thread generated every track itself, 0.1 ms each against 4 consumptions per 0.22 ms step — a 20-step
                                       // bench run then ended with 8 ms of host backlog (round 6, profiles/r06_driver_style_runs.txt)
  std::mutex m;                        // one service cycle at a time (the thread's, or a caller's inside mcr_refill_wait)
  std::mutex qm;                       // todo / done / inflight
  std::deque<int32_t> todo; std::vector<int32_t> done; int inflight = 0;
  std::atomic<bool> stop{false}, hold{false};
  uint32_t* mt_track; uint32_t* mt_draw; int direction_mode, gen_threads;
  uint8_t* blobs; int32_t* info;
  hipStream_t st = nullptr; hipEvent_t ev = nullptr;
  std::vector<int32_t> ids;
  std::atomic<long long> generated{0};
  // the age of the oldest consumption that is not staged yet (mcr_refill_lag): consumptions are served first in, first out — near enough —, so the
  // oldest unserved one is the `staged`-th to arrive; seen_step[k % size] = the step count when the k-th was noticed
  std::vector<int32_t> seen_step; std::atomic<long long> arrived{0}, staged{0};
  std::atomic<int> err{0}; std::string err_msg;
  long long dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // last mcr_refill_wait: queued / in flight / done at entry, cycles, tracks the waiter generated, us staging, us total, generator threads
};
// one track: env e's next episode from ITS rows of the caller's arrays (mcr_host.cpp; num_threads 1: on this thread)
static bool refill_generate_one(mcr_env* h, RefillSvc* s, int32_t e) {
  const int rc = mcr_episodes_generate_rows(s->mt_track, s->mt_draw, &e, 1, h->cfg.num_agents, s->direction_mode, s->blobs, s->info, 1);
  if (rc != MCR_OK) { std::lock_guard<std::mutex> lk(s->qm); s->err_msg = "track generation failed (capacity: MCR_TILE_CAP / MCR_QUAD_CAP)"; s->err.store(rc); return false; }
  return true;
}
// take one env off the queue and generate its track; false: nothing to do
static bool refill_work_one(mcr_env* h, RefillSvc* s) {
  int32_t e = -1;
  { std::lock_guard<std::mutex> lk(s->qm); if (!s->todo.empty()) { e = s->todo.front(); s->todo.pop_front(); ++s->inflight; } }
  if (e < 0) return false;
  const bool ok = refill_generate_one(h, s, e);
  { std::lock_guard<std::mutex> lk(s->qm); --s->inflight; if (ok) s->done.push_back(e); }
  return true;
}
static void refill_gen_main(mcr_env* h, RefillSvc* s) {
  (void)prctl(PR_SET_NAME, "mcr-gen", 0, 0, 0);
  int idle = 0;
  while (!s->stop.load()) {
    if (refill_work_one(h, s)) { idle = 0; continue; }
    ++idle;
    usleep(idle < 400 ? 50 : 500);     // (20 ms without work: an idle env costs its generators next to nothing)
  }
}
// one cycle of the service: new consumptions -> the generators' queue; finished tracks -> their envs' staged slots.  Returns what is still
// pending afterwards (queued + being generated + consumptions not yet seen: 0 = all caught up), < 0: error.  Caller holds s->m.
static int refill_cycle(mcr_env* h, RefillSvc* s, bool urgent = false) {
  const int B = h->cfg.num_envs;
  int n = 0;
  for (int e = 0; e < B; ++e) {
    const int32_t c = ((volatile int32_t*)h->consumed_host)[e];
    if (c != h->consumed_seen[e]) { h->consumed_seen[e] = c; s->ids[n++] = e; }
  }
  int m = 0, left = 0;
  {
    std::lock_guard<std::mutex> lk(s->qm);
    for (int i = 0; i < n; ++i) s->todo.push_back(s->ids[i]);
    m = (int)s->done.size();
    for (int i = 0; i < m; ++i) s->ids[i] = s->done[i];
    s->done.clear();
    left = (int)s->todo.size() + s->inflight;
  }
  { const long long a0 = s->arrived.load(); for (int i = 0; i < n; ++i) s->seen_step[(size_t)((a0 + i) % (long long)s->seen_step.size())] = h->step_count; s->arrived.store(a0 + n); }
  if (s->err.load()) return s->err.load();
  if (m > 0) {
    auto fail = [&](const char* what, hipError_t e) { s->err_msg = std::string(what) + ": " + hipGetErrorString(e); s->err.store(MCR_ERR_HIP); return (int)MCR_ERR_HIP; };
    for (int i = 0; i < m; ++i) {
      const int e = s->ids[i];
      const int32_t installs = ((volatile int32_t*)h->consumed_host)[e];
      uint8_t* dst = h->P.slots + ((size_t)e * 2 + ((installs & 1) ^ 1)) * MCR_SLOT_BYTES;
      const hipError_t er = hipMemcpyAsync(dst, s->blobs + (size_t)e * MCR_SLOT_BYTES, MCR_SLOT_BYTES, hipMemcpyHostToDevice, s->st);
      if (er != hipSuccess) return fail("hipMemcpyAsync (episode)", er);
    }
    hipError_t er = hipMemcpyAsync(h->stage_ids, s->ids.data(), sizeof(int32_t) * (size_t)m, hipMemcpyHostToDevice, s->st);
    if (er != hipSuccess) return fail("hipMemcpyAsync (ids)", er);
    hipLaunchKernelGGL(k_mark_staged, dim3((m + 255) / 256), dim3(256), 0, s->st, h->P, (const int32_t*)h->stage_ids, m);
    if ((er = hipEventRecord(s->ev, s->st)) != hipSuccess) return fail("hipEventRecord", er);
    // (polled with a sleep in between: hipEventSynchronize / hipStreamSynchronize spin on this runtime — 0.8 of a core, measured in round 3 —
    // and a spinning thread takes a core from the track generators where the ranks of a node share few)
    // (urgent — a caller waits in mcr_refill_wait: poll without sleeping for the first 100 us; the copies are 0.1 MB each)
    for (int spins = 0;; ++spins) { er = hipEventQuery(s->ev); if (er == hipSuccess) break; if (er != hipErrorNotReady) return fail("hipEventQuery", er); if (!urgent || spins > 2000) usleep(30); }
    s->generated.fetch_add(m);
    s->staged.fetch_add(m);
  }
  if (left == 0) {                     // nothing queued, nothing in flight — unless a track finished meanwhile (it shows up in `done` next cycle)
    std::lock_guard<std::mutex> lk(s->qm);
    if (!(s->todo.empty() && s->inflight == 0 && s->done.empty())) left = 1;
  }
  return left;
}
static void refill_main(mcr_env* h, RefillSvc* s) {
  (void)prctl(PR_SET_NAME, "mcr-refill", 0, 0, 0);
  (void)hipSetDevice(h->cfg.device);
  while (!s->stop.load()) {
    if (!s->hold.load() && s->err.load() == 0) { std::lock_guard<std::mutex> lk(s->m); (void)refill_cycle(h, s); }
    usleep(100);                       // (a step takes 0.2 ms and an env needs hundreds of steps to end its next episode: nothing is urgent here)
  }
}
extern "C" int mcr_refill_start(mcr_env* h, uint32_t* mt_track, uint32_t* mt_draw, int direction_mode, int gen_threads, void* blobs_pinned, int32_t* episode_info) {
  if (!h || !mt_track || !mt_draw || !blobs_pinned) { g_err = "mcr_refill_start: null argument"; return MCR_ERR_ARG; }
  if (h->svc) { g_err = "mcr_refill_start: already running"; return MCR_ERR_STATE; }
  HIPCHK(hipSetDevice(h->cfg.device));
  RefillSvc* s = new RefillSvc();
  s->mt_track = mt_track; s->mt_draw = mt_draw; s->direction_mode = direction_mode; s->gen_threads = std::max(1, gen_threads);
  s->blobs = (uint8_t*)blobs_pinned; s->info = episode_info; s->ids.resize(h->cfg.num_envs); s->seen_step.assign((size_t)h->cfg.num_envs * 2, 0);
  if (hipStreamCreateWithFlags(&s->st, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&s->ev, hipEventDisableTiming) != hipSuccess) { delete s; g_err = "mcr_refill_start: stream / event"; return MCR_ERR_HIP; }
  h->svc = s;
  s->th = std::thread(refill_main, h, s);
  const int ngen = std::max(1, std::min(s->gen_threads, 6));       // (19 k tracks/s at B = 4096, N = 2 are two of them busy; 30 k at B = 32768 three)
  for (int i = 0; i < ngen; ++i) s->gens.emplace_back(refill_gen_main, h, s);
  return MCR_OK;
}
extern "C" int mcr_refill_stop(mcr_env* h) {
  if (!h) return MCR_ERR_ARG;
  RefillSvc* s = h->svc;
  if (!s) return MCR_OK;
  s->stop.store(true);
  if (s->th.joinable()) s->th.join();
  for (auto& g : s->gens) if (g.joinable()) g.join();
  (void)hipSetDevice(h->cfg.device);
  (void)hipStreamSynchronize(s->st);
  (void)hipStreamDestroy(s->st); (void)hipEventDestroy(s->ev);
  h->svc = nullptr;
  delete s;
  return MCR_OK;
}
// every consumption visible NOW is staged when this returns (the caller synchronised the stepping stream first if it needs "all of them");
// honours the hold switch (tests: an env that finds no staged episode freezes)
extern "C" int mcr_refill_wait(mcr_env* h) {
  if (!h) return MCR_ERR_ARG;
  RefillSvc* s = h->svc;
  if (!s) return MCR_OK;
  if (s->err.load()) { g_err = "the refill service failed: " + s->err_msg; return s->err.load(); }
  if (s->hold.load()) return MCR_OK;
  HIPCHK(hipSetDevice(h->cfg.device));
  timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
  auto us_since = [&](const timespec& a) { timespec b; clock_gettime(CLOCK_MONOTONIC, &b); return (long long)((b.tv_sec - a.tv_sec) * 1000000ll + (b.tv_nsec - a.tv_nsec) / 1000); };
  std::lock_guard<std::mutex> lk(s->m);
  { std::lock_guard<std::mutex> q(s->qm); s->dbg[0] = (long long)s->todo.size(); s->dbg[1] = s->inflight; s->dbg[2] = (long long)s->done.size(); }
  s->dbg[3] = s->dbg[4] = s->dbg[5] = 0; s->dbg[7] = (long long)s->gens.size();
  for (;;) {
    timespec tc; clock_gettime(CLOCK_MONOTONIC, &tc);
    const int left = refill_cycle(h, s, true);
    s->dbg[5] += us_since(tc); ++s->dbg[3];
    if (left < 0) { g_err = "the refill service failed: " + s->err_msg; return left; }
    if (left == 0) break;
    if (refill_work_one(h, s)) ++s->dbg[4]; else usleep(10);       // the waiting thread generates too: progress does not depend on how fast a sleeping generator wakes
  }
  s->dbg[6] = us_since(t0);
  return MCR_OK;
}
// steps launched since the oldest consumption that is not staged yet was noticed (0: nothing pending); < 0: the service failed
extern "C" int mcr_refill_lag(mcr_env* h) {
  if (!h || !h->svc) return 0;
  RefillSvc* s = h->svc;
  if (s->err.load()) { g_err = "the refill service failed: " + s->err_msg; return s->err.load(); }
  const long long st = s->staged.load(), ar = s->arrived.load();
  if (st >= ar) return 0;
  return std::max(0, (int32_t)((uint32_t)h->step_count - (uint32_t)s->seen_step[(size_t)(st % (long long)s->seen_step.size())]));
}
extern "C" int mcr_refill_debug(mcr_env* h, long long* out8) { if (!h || !h->svc || !out8) return MCR_ERR_STATE; for (int i = 0; i < 8; ++i) out8[i] = h->svc->dbg[i]; return MCR_OK; }
extern "C" int mcr_refill_hold(mcr_env* h, int hold) { if (!h || !h->svc) return MCR_ERR_STATE; h->svc->hold.store(hold != 0); return MCR_OK; }
extern "C" long long mcr_refill_generated(mcr_env* h) { return (h && h->svc) ? h->svc->generated.load() : 0; }

// ---------------------------------------------------------------------------- state access (synchronous)
extern "C" int mcr_get_state(mcr_env* h, float* bodies, float* joints, double* wheels, int32_t* limit, uint8_t* on_road, float* sleep) {
  if (!h) return MCR_ERR_ARG;
  HIPCHK(hipDeviceSynchronize());
  const size_t BN = h->P.BN;
  std::vector<float> cf(CF_COUNT * BN); std::vector<double> cd(CD_COUNT * BN); std::vector<uint32_t> cu(CU_COUNT * BN);
  HIPCHK(hipMemcpy(cf.data(), h->P.carf, cf.size() * 4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(cd.data(), h->P.card, cd.size() * 8, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(cu.data(), h->P.caru, cu.size() * 4, hipMemcpyDeviceToHost));
  for (size_t c = 0; c < BN; ++c) {
    if (bodies) for (int k = 0; k < 5; ++k) {
      float* o = bodies + (c * 5 + k) * 6;
      o[0] = cf[(CF_CX + k) * BN + c]; o[1] = cf[(CF_CY + k) * BN + c]; o[2] = cf[(CF_A + k) * BN + c];
      o[3] = cf[(CF_VX + k) * BN + c]; o[4] = cf[(CF_VY + k) * BN + c]; o[5] = cf[(CF_W + k) * BN + c];
    }
    if (sleep) for (int k = 0; k < 5; ++k) sleep[c * 5 + k] = cf[(CF_SLEEP + k) * BN + c];
    for (int k = 0; k < 4; ++k) {
      if (joints) { float* o = joints + (c * 4 + k) * 4; o[0] = cf[(CF_JIX + k) * BN + c]; o[1] = cf[(CF_JIY + k) * BN + c]; o[2] = cf[(CF_JIZ + k) * BN + c]; o[3] = cf[(CF_JM + k) * BN + c]; }
      if (wheels) {
        double* o = wheels + (c * 4 + k) * 5;
        o[0] = k >= 2 ? cd[(CD_GAS + k - 2) * BN + c] : 0.0; o[1] = cd[CD_BRAKE * BN + c]; o[2] = k < 2 ? cd[CD_STEER * BN + c] : 0.0;
        o[3] = cd[(CD_PHASE + k) * BN + c]; o[4] = cd[(CD_OMEGA + k) * BN + c];
      }
      if (limit) limit[c * 4 + k] = (cu[CU_LIMIT * BN + c] >> (2 * k)) & 3;
      if (on_road) on_road[c * 4 + k] = (cu[CU_ONROAD * BN + c] >> k) & 1;
    }
  }
  return MCR_OK;
}

extern "C" int mcr_set_bodies(mcr_env* h, const float* bodies) {
  if (!h || !bodies) return MCR_ERR_ARG;
  HIPCHK(hipDeviceSynchronize());
  const size_t BN = h->P.BN;
  std::vector<float> cf(30 * BN);
  for (size_t c = 0; c < BN; ++c) for (int k = 0; k < 5; ++k) {
    const float* o = bodies + (c * 5 + k) * 6;
    cf[(CF_CX + k) * BN + c] = o[0]; cf[(CF_CY + k) * BN + c] = o[1]; cf[(CF_A + k) * BN + c] = o[2];
    cf[(CF_VX + k) * BN + c] = o[3]; cf[(CF_VY + k) * BN + c] = o[4]; cf[(CF_W + k) * BN + c] = o[5];
  }
  HIPCHK(hipMemcpy(h->P.carf, cf.data(), cf.size() * 4, hipMemcpyHostToDevice));
  h->bp_fresh = true; h->verdict_fresh = false;      // teleported cars: their broadphase proxies are re-created by the next contact pass
  return MCR_OK;
}

extern "C" int mcr_get_env_state(mcr_env* h, double* reward, int32_t* tvc, uint8_t* backward, uint8_t* on_grass, double* t,
                                 uint16_t* tile_flags, int32_t* num_tiles) {
  if (!h) return MCR_ERR_ARG;
  HIPCHK(hipDeviceSynchronize());
  const size_t BN = h->P.BN; const int B = h->P.B;
  std::vector<double> r(BN); std::vector<uint32_t> cu(CU_COUNT * BN); std::vector<McrEnvState> es(B);
  HIPCHK(hipMemcpy(r.data(), h->P.card + CD_REWARD * BN, BN * 8, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(cu.data(), h->P.caru, cu.size() * 4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(es.data(), h->P.env, sizeof(McrEnvState) * B, hipMemcpyDeviceToHost));
  for (size_t c = 0; c < BN; ++c) {
    if (reward) reward[c] = r[c];
    if (tvc) tvc[c] = (int32_t)cu[CU_TVC * BN + c];
    if (backward) backward[c] = cu[CU_FLAGS * BN + c] & 1;
    if (on_grass) on_grass[c] = (cu[CU_FLAGS * BN + c] >> 1) & 1;
  }
  if (t) for (int e = 0; e < B; ++e) t[e] = es[e].t;
  if (tile_flags) HIPCHK(hipMemcpy(tile_flags, h->P.tile_flags, sizeof(uint16_t) * MCR_TILE_CAP * (size_t)B, hipMemcpyDeviceToHost));
  if (num_tiles) for (int e = 0; e < B; ++e) {
    McrSlotHeader H;
    HIPCHK(hipMemcpy(&H, h->P.slots + ((size_t)e * 2 + es[e].slot) * MCR_SLOT_BYTES, sizeof(H), hipMemcpyDeviceToHost));
    num_tiles[e] = H.T;
  }
  return MCR_OK;
}

// ---------------------------------------------------------------------------- full state snapshot / restore
namespace {
struct BlobLayout { size_t carf, card, caru, es, touch, tflags, cc, viewp, carpoly, bpf, stamp, ccstamp, slot, world, particles, total; };
BlobLayout blob_layout(int N, bool with_particles) {
  BlobLayout L; size_t o = 16;                                         // header: magic (carries the layout version), N, flags (bit 0: particles), total bytes
  L.carf = o; o += sizeof(float) * CF_COUNT * N;
  o = (o + 7) & ~(size_t)7; L.card = o; o += sizeof(double) * CD_COUNT * N;
  L.caru = o; o += sizeof(uint32_t) * CU_COUNT * N;
  o = (o + 7) & ~(size_t)7; L.es = o; o += sizeof(McrEnvState);
  L.touch = o; o += sizeof(uint32_t) * MCR_TILE_CAP;
  L.tflags = o; o += sizeof(uint16_t) * MCR_TILE_CAP;
  L.cc = o; o += sizeof(uint32_t) * (MCR_CC_MAX * MCR_CC_WORDS + 4);
  L.viewp = o; o += sizeof(float) * MCR_VIEWP_FLOATS * N;
  L.carpoly = o; o += sizeof(float) * MCR_CARPOLY_FLOATS * N;
  L.bpf = o; o += sizeof(float4) * BP_COUNT * MCR_BP_FIX * N;
  L.stamp = o; o += sizeof(uint32_t) * MCR_TILE_CAP * 4 * N;
  L.ccstamp = o; o += sizeof(uint32_t) * mcr_cc_stamp_words(N);
  o = (o + 15) & ~(size_t)15; L.slot = o; o += MCR_SLOT_BYTES;
  L.world = o; o += sizeof(uint16_t) * (MCR_PID_TAB + MCR_PID_STACK) + sizeof(int32_t) * 4;      // the env's b2World (k_world.h): ids, free leaf stack, meta
  L.particles = o; if (with_particles) o += sizeof(uint32_t) * MCR_PART_WORDS * N;
  L.total = o;
  return L;
}
const uint32_t BLOB_MAGIC = 0x3552434du;   // "MCR5": bumped whenever the layout of a blob (McrEnvState, slot image, field lists) changes
}  // namespace

extern "C" size_t mcr_state_blob_bytes(const mcr_env* h) { return h ? blob_layout(h->P.N, h->P.particles != nullptr).total : 0; }

extern "C" int mcr_get_state_blob(mcr_env* h, int env, void* blob_out) {
  if (!h || !blob_out) { g_err = "null argument"; return MCR_ERR_ARG; }
  if (env < 0 || env >= h->P.B) { g_err = "env out of range"; return MCR_ERR_ARG; }
  if (!h->any_reset) { g_err = "state snapshot before reset()"; return MCR_ERR_STATE; }
  HIPCHK(hipDeviceSynchronize());
  const McrParams& P = h->P; const int N = P.N; const size_t BN = P.BN;
  const BlobLayout L = blob_layout(N, P.particles != nullptr);
  uint8_t* b = (uint8_t*)blob_out;
  memset(b, 0, L.total);
  ((uint32_t*)b)[0] = BLOB_MAGIC; ((uint32_t*)b)[1] = (uint32_t)N; ((uint32_t*)b)[2] = (P.particles ? 1u : 0u) | (P.pid_tab ? 2u : 0u); ((uint32_t*)b)[3] = (uint32_t)L.total;
  HIPCHK(hipMemcpy2D(b + L.carf, sizeof(float) * N, P.carf + (size_t)env * N, sizeof(float) * BN, sizeof(float) * N, CF_COUNT, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy2D(b + L.card, sizeof(double) * N, P.card + (size_t)env * N, sizeof(double) * BN, sizeof(double) * N, CD_COUNT, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy2D(b + L.caru, sizeof(uint32_t) * N, P.caru + (size_t)env * N, sizeof(uint32_t) * BN, sizeof(uint32_t) * N, CU_COUNT, hipMemcpyDeviceToHost));
  McrEnvState es;
  HIPCHK(hipMemcpy(&es, P.env + env, sizeof(es), hipMemcpyDeviceToHost));
  memcpy(b + L.es, &es, sizeof(es));
  HIPCHK(hipMemcpy(b + L.touch, P.tile_touch + (size_t)env * MCR_TILE_CAP, sizeof(uint32_t) * MCR_TILE_CAP, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(b + L.tflags, P.tile_flags + (size_t)env * MCR_TILE_CAP, sizeof(uint16_t) * MCR_TILE_CAP, hipMemcpyDeviceToHost));
  const size_t ccw = MCR_CC_MAX * MCR_CC_WORDS + 4;
  HIPCHK(hipMemcpy(b + L.cc, P.cc_store + (size_t)env * ccw, sizeof(uint32_t) * ccw, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(b + L.viewp, P.viewp + (size_t)env * N * MCR_VIEWP_FLOATS, sizeof(float) * MCR_VIEWP_FLOATS * N, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(b + L.carpoly, P.carpoly + (size_t)env * N * MCR_CARPOLY_FLOATS, sizeof(float) * MCR_CARPOLY_FLOATS * N, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy2D(b + L.bpf, sizeof(float4) * MCR_BP_FIX * N, P.bpf + (size_t)env * MCR_BP_FIX * N, sizeof(float4) * MCR_BP_FIX * BN, sizeof(float4) * MCR_BP_FIX * N, BP_COUNT, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(b + L.ccstamp, P.cc_stamp + (size_t)env * mcr_cc_stamp_words(N), sizeof(uint32_t) * mcr_cc_stamp_words(N), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(b + L.stamp, P.bp_stamp + (size_t)env * MCR_TILE_CAP * 4 * N, sizeof(uint32_t) * MCR_TILE_CAP * 4 * N, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(b + L.slot, P.slots + ((size_t)env * 2 + es.slot) * MCR_SLOT_BYTES, MCR_SLOT_BYTES, hipMemcpyDeviceToHost));
  if (P.particles) HIPCHK(hipMemcpy(b + L.particles, P.particles + (size_t)env * N * MCR_PART_WORDS, sizeof(uint32_t) * MCR_PART_WORDS * N, hipMemcpyDeviceToHost));
  if (P.pid_tab) {
    HIPCHK(hipMemcpy(b + L.world, P.pid_tab + (size_t)env * MCR_PID_TAB, sizeof(uint16_t) * MCR_PID_TAB, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(b + L.world + sizeof(uint16_t) * MCR_PID_TAB, P.pid_stack + (size_t)env * MCR_PID_STACK, sizeof(uint16_t) * MCR_PID_STACK, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(b + L.world + sizeof(uint16_t) * (MCR_PID_TAB + MCR_PID_STACK), P.pid_meta + (size_t)env * 4, sizeof(int32_t) * 4, hipMemcpyDeviceToHost));
  }
  return MCR_OK;
}

extern "C" int mcr_set_state_blob(mcr_env* h, int env, const void* blob) {
  if (!h || !blob) { g_err = "null argument"; return MCR_ERR_ARG; }
  if (env < 0 || env >= h->P.B) { g_err = "env out of range"; return MCR_ERR_ARG; }
  const McrParams& P = h->P; const int N = P.N; const size_t BN = P.BN;
  const uint8_t* b = (const uint8_t*)blob;
  const BlobLayout L = blob_layout(N, P.particles != nullptr);
  if (((const uint32_t*)b)[0] != BLOB_MAGIC || ((const uint32_t*)b)[1] != (uint32_t)N) { g_err = "not a state blob of this build and num_agents"; return MCR_ERR_ARG; }
  if (((const uint32_t*)b)[2] != ((P.particles ? 1u : 0u) | (P.pid_tab ? 2u : 0u)) || ((const uint32_t*)b)[3] != (uint32_t)L.total) { g_err = "state blob was taken from a handle with another skid_particles or fresh_world setting"; return MCR_ERR_ARG; }
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy2D(P.carf + (size_t)env * N, sizeof(float) * BN, b + L.carf, sizeof(float) * N, sizeof(float) * N, CF_COUNT, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy2D(P.card + (size_t)env * N, sizeof(double) * BN, b + L.card, sizeof(double) * N, sizeof(double) * N, CD_COUNT, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy2D(P.caru + (size_t)env * N, sizeof(uint32_t) * BN, b + L.caru, sizeof(uint32_t) * N, sizeof(uint32_t) * N, CU_COUNT, hipMemcpyHostToDevice));
  // the staging protocol (which slot is current, whether a staged episode waits, install counter) belongs to the
  // TARGET handle; everything else of the env record comes from the blob
  McrEnvState cur, in;
  HIPCHK(hipMemcpy(&cur, P.env + env, sizeof(cur), hipMemcpyDeviceToHost));
  memcpy(&in, b + L.es, sizeof(in));
  in.slot = cur.slot; in.staged_ready = cur.staged_ready; in.consumed = cur.consumed;
  HIPCHK(hipMemcpy(P.env + env, &in, sizeof(in), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(P.tile_touch + (size_t)env * MCR_TILE_CAP, b + L.touch, sizeof(uint32_t) * MCR_TILE_CAP, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(P.tile_flags + (size_t)env * MCR_TILE_CAP, b + L.tflags, sizeof(uint16_t) * MCR_TILE_CAP, hipMemcpyHostToDevice));
  const size_t ccw = MCR_CC_MAX * MCR_CC_WORDS + 4;
  HIPCHK(hipMemcpy(P.cc_store + (size_t)env * ccw, b + L.cc, sizeof(uint32_t) * ccw, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(P.viewp + (size_t)env * N * MCR_VIEWP_FLOATS, b + L.viewp, sizeof(float) * MCR_VIEWP_FLOATS * N, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(P.carpoly + (size_t)env * N * MCR_CARPOLY_FLOATS, b + L.carpoly, sizeof(float) * MCR_CARPOLY_FLOATS * N, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy2D(P.bpf + (size_t)env * MCR_BP_FIX * N, sizeof(float4) * MCR_BP_FIX * BN, b + L.bpf, sizeof(float4) * MCR_BP_FIX * N, sizeof(float4) * MCR_BP_FIX * N, BP_COUNT, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(P.cc_stamp + (size_t)env * mcr_cc_stamp_words(N), b + L.ccstamp, sizeof(uint32_t) * mcr_cc_stamp_words(N), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(P.bp_stamp + (size_t)env * MCR_TILE_CAP * 4 * N, b + L.stamp, sizeof(uint32_t) * MCR_TILE_CAP * 4 * N, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(P.slots + ((size_t)env * 2 + cur.slot) * MCR_SLOT_BYTES, b + L.slot, MCR_SLOT_BYTES, hipMemcpyHostToDevice));
  if (P.particles) HIPCHK(hipMemcpy(P.particles + (size_t)env * N * MCR_PART_WORDS, b + L.particles, sizeof(uint32_t) * MCR_PART_WORDS * N, hipMemcpyHostToDevice));
  if (P.pid_tab) {
    HIPCHK(hipMemcpy(P.pid_tab + (size_t)env * MCR_PID_TAB, b + L.world, sizeof(uint16_t) * MCR_PID_TAB, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(P.pid_stack + (size_t)env * MCR_PID_STACK, b + L.world + sizeof(uint16_t) * MCR_PID_TAB, sizeof(uint16_t) * MCR_PID_STACK, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(P.pid_meta + (size_t)env * 4, b + L.world + sizeof(uint16_t) * (MCR_PID_TAB + MCR_PID_STACK), sizeof(int32_t) * 4, hipMemcpyHostToDevice));
  }
  h->any_reset = true; h->verdict_fresh = false;
  return MCR_OK;
}

__global__ void k_positions(McrParams p, float* out) {
  const int ci = blockIdx.x * blockDim.x + threadIdx.x;
  if (ci >= p.BN) return;
  const McrShapes& S = *p.shapes;
  Xf xf = xf_of(v2(p.carf[CF_CX * p.BN + ci], p.carf[CF_CY * p.BN + ci]), p.carf[CF_A * p.BN + ci], v2(S.hull_lcx, S.hull_lcy));
  out[ci * 2] = xf.p.x; out[ci * 2 + 1] = xf.p.y;
}
extern "C" int mcr_get_positions(mcr_env* h, float* pos) {
  if (!h || !pos) return MCR_ERR_ARG;
  float* d = nullptr;
  HIPCHK(hipMalloc(&d, sizeof(float) * 2 * h->P.BN));
  hipLaunchKernelGGL(k_positions, dim3((h->P.BN + 63) / 64), dim3(64), 0, 0, h->P, d);
  HIPCHK(hipMemcpy(pos, d, sizeof(float) * 2 * h->P.BN, hipMemcpyDeviceToHost));
  (void)hipFree(d);
  return MCR_OK;
}

__global__ void k_sincos(const float* in, float* s, float* c, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) mcr_sincosf(in[i], &s[i], &c[i]);
}
extern "C" int mcr_sincos_device(mcr_env* h, const float* d_in, float* d_sin, float* d_cos, int n, void* stream) {
  if (!h || !d_in || !d_sin || !d_cos) return MCR_ERR_ARG;
  hipLaunchKernelGGL(k_sincos, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_in, d_sin, d_cos, n);
  HIPCHK(hipGetLastError());
  return MCR_OK;
}

extern "C" int mcr_debug_read_view_scratch(mcr_env* h, int view, void* out, int nbytes) {
  if (!h || !out || view < 0 || view >= h->P.BN || nbytes < 0 || nbytes > 128) return MCR_ERR_ARG;
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out, h->view_stamps + (size_t)view * 16, nbytes, hipMemcpyDeviceToHost));
  return MCR_OK;
}
extern "C" int mcr_debug_read_counters(mcr_env* h, uint64_t* out4) {
  if (!h || !out4) return MCR_ERR_ARG;
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out4, h->P.counters, sizeof(uint64_t) * 4, hipMemcpyDeviceToHost));
  return MCR_OK;
}
extern "C" int mcr_debug_read_counters8(mcr_env* h, uint64_t* out8) {
  if (!h || !out8) return MCR_ERR_ARG;
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out8, h->P.counters, sizeof(uint64_t) * 8, hipMemcpyDeviceToHost));
  return MCR_OK;
}
extern "C" int mcr_concurrent_collide(const mcr_env* h) { return (h && h->split && h->concurrent_collide) ? 1 : 0; }
extern "C" int mcr_step_ordering(const mcr_env* h) {
  if (!h || !h->split) return 0;
  return ((h->soft_sync && h->use_graph <= 0) ? 1 : 0) | (h->stop_events ? 2 : 0) | (h->soft_denied ? 4 : 0);
}
extern "C" int mcr_step_ordering_for(const mcr_env* h, void* stream) {
  if (!h || !h->split) return 0;
  const int all = mcr_step_ordering(h);
  return stream_bound(h, (hipStream_t)stream) ? all : (all & ~1);
}
extern "C" int mcr_debug_read_verdict_mismatches(mcr_env* h, uint64_t* out) {
  if (!h || !out) return MCR_ERR_ARG;
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out, h->P.counters + 4, sizeof(uint64_t), hipMemcpyDeviceToHost));
  return MCR_OK;
}
// the last step's contact partition: part_out[B] (touch verdicts it went by), clist_out[1 + B] (its contact list: count, env ids); synchronises
extern "C" int mcr_debug_read_partition(mcr_env* h, uint8_t* part_out, int32_t* clist_out) {
  if (!h || !part_out || !clist_out) return MCR_ERR_ARG;
  HIPCHK(hipDeviceSynchronize());
  const size_t B = h->cfg.num_envs, par = (size_t)(h->step_parity ^ 1);
  HIPCHK(hipMemcpy(part_out, h->P.part + par * B, B, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(clist_out, h->P.clist + par * (B + 1), sizeof(int32_t) * (B + 1), hipMemcpyDeviceToHost));
  return MCR_OK;
}
// diagnostics of the verdict protocol: fill the buffer the NEXT step's verdict writers write (its part_next) with `value` / read it back after that step
extern "C" int mcr_debug_next_verdicts(mcr_env* h, int fill_value, uint8_t* out_or_null) {
  if (!h) return MCR_ERR_ARG;
  if (!out_or_null && (fill_value & 0x100)) {     // (no synchronisation: the fill is enqueued on the null stream, in front of a step launched there)
    HIPCHK(hipMemsetAsync(h->P.part + (size_t)(h->step_parity ^ 1) * h->cfg.num_envs, fill_value & 0xff, h->cfg.num_envs, 0));
    return MCR_OK;
  }
  HIPCHK(hipDeviceSynchronize());
  const size_t B = h->cfg.num_envs;
  if (out_or_null) { HIPCHK(hipMemcpy(out_or_null, h->P.part + (size_t)h->step_parity * B, B, hipMemcpyDeviceToHost)); }       // (after a step: the parity has flipped)
  else { HIPCHK(hipMemset(h->P.part + (size_t)(h->step_parity ^ 1) * B, fill_value, B)); }
  return MCR_OK;
}
extern "C" int mcr_debug_read_env_records(mcr_env* h, void* out, int n_bytes) {
  if (!h || !out || n_bytes < 0) return MCR_ERR_ARG;
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out, h->P.env, std::min((size_t)n_bytes, sizeof(McrEnvState) * (size_t)h->cfg.num_envs), hipMemcpyDeviceToHost));
  return MCR_OK;
}
extern "C" int mcr_debug_read_contact_counts(mcr_env* h, int32_t* out) {
  if (!h || !out) return MCR_ERR_ARG;
  HIPCHK(hipDeviceSynchronize());
  const size_t stride = MCR_CC_MAX * MCR_CC_WORDS + 4;
  HIPCHK(hipMemcpy2D(out, sizeof(int32_t), h->P.cc_store, stride * sizeof(uint32_t), sizeof(int32_t), h->cfg.num_envs, hipMemcpyDeviceToHost));
  return MCR_OK;
}
extern "C" int mcr_debug_read_proxy_ids(mcr_env* h, int env, int32_t* out, int cap) {
  if (!h || !out || env < 0 || env >= h->cfg.num_envs) return MCR_ERR_ARG;
  if (!h->P.pid_tab) { g_err = "mcr_debug_read_proxy_ids: the handle was created with fresh_world = 1 (ids ascend in creation order)"; return MCR_ERR_STATE; }
  HIPCHK(hipDeviceSynchronize());
  std::vector<uint16_t> tab(MCR_PID_TAB); int32_t meta[4];
  HIPCHK(hipMemcpy(tab.data(), h->P.pid_tab + (size_t)env * MCR_PID_TAB, sizeof(uint16_t) * MCR_PID_TAB, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(meta, h->P.pid_meta + (size_t)env * 4, sizeof(meta), hipMemcpyDeviceToHost));
  const int T = meta[2], F = 8 * h->cfg.num_agents;
  int n = 0;
  for (int t = 0; t < T && n < cap; ++t) out[n++] = tab[t];
  for (int f = 0; f < F && n < cap; ++f) out[n++] = tab[MCR_TILE_CAP + f];
  return n;
}
extern "C" int mcr_debug_read_dynamics_stamps(mcr_env* h, uint64_t* out, int n_u64) {
  if (!h || !out) return MCR_ERR_ARG;
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out, h->P.dbg_stamps, sizeof(uint64_t) * (size_t)n_u64, hipMemcpyDeviceToHost));
  return MCR_OK;
}
// the sensor predicate of k_collide (col::overlap: SAT far-field filter + Box2D's GJK) on caller-supplied cases
__global__ void k_debug_overlap(const McrShapes* shapes, int fixture_arg, int n, const float4* __restrict__ va, const float4* __restrict__ vb,
                                const float4* __restrict__ na, const float4* __restrict__ nb, const int* __restrict__ cnt,
                                const float* __restrict__ poses, uint8_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const McrShapes& S = *shapes;
  const bool fixture_first = fixture_arg >= 8;                     // b2TestOverlap(fixture, tile): the car fixture holds the lower proxy id
  const int fixture = fixture_arg & 7;
  const McrPoly& P = fixture < 4 ? S.hull[fixture] : S.wheel;
  const V2 lc = fixture < 4 ? v2(S.hull_lcx, S.hull_lcy) : v2(0.0f, 0.0f);
  // the body origin is given (b2BodyDef.position); sweep.c = xf * localCenter, then the transform k_collide derives from (c, a)
  Xf x0; x0.q = rot_of(poses[i * 3 + 2]); x0.p = v2(poses[i * 3], poses[i * 3 + 1]);
  const Xf xf = xf_of(xmul(x0, lc), poses[i * 3 + 2], lc);
  float wx[8], wy[8], nx[8], ny[8];
  for (int k = 0; k < 8; ++k) { wx[k] = wy[k] = nx[k] = ny[k] = 0.0f; }
  for (int k = 0; k < P.n; ++k) { const V2 w = xmul(xf, v2(P.vx[k], P.vy[k])); const V2 nn = rmul(xf.q, v2(P.nx[k], P.ny[k])); wx[k] = w.x; wy[k] = w.y; nx[k] = nn.x; ny[k] = nn.y; }
  col::TilePoly TP; const float4 a = va[i], b = vb[i], c = na[i], d = nb[i];
  TP.n = cnt[i];
  TP.vx[0] = a.x; TP.vy[0] = a.y; TP.vx[1] = a.z; TP.vy[1] = a.w; TP.vx[2] = b.x; TP.vy[2] = b.y; TP.vx[3] = b.z; TP.vy[3] = b.w;
  TP.nx[0] = c.x; TP.ny[0] = c.y; TP.nx[1] = c.z; TP.ny[1] = c.w; TP.nx[2] = d.x; TP.ny[2] = d.y; TP.nx[3] = d.z; TP.ny[3] = d.w;
  out[i] = col::overlap(wx, wy, nx, ny, P.n, TP, a, b, &P, make_float4(xf.p.x, xf.p.y, xf.q.s, xf.q.c), fixture_first) ? 1 : 0;
}
void mcr_tile_hull(const float* fx, const float* fy, float* aabb4, float* va4, float* vb4, float* na4, float* nb4, int* count);   // mcr_host.cpp
extern "C" int mcr_debug_overlap(mcr_env* h, int n, const float* quads, const float* poses, int fixture, uint8_t* out) {
  if (!h || !quads || !poses || !out || n < 0 || fixture < 0 || (fixture & 7) > 4 || fixture > 12) { g_err = "bad argument"; return MCR_ERR_ARG; }
  if (n == 0) return MCR_OK;
  std::vector<float> hull((size_t)n * 16); std::vector<int> cnt(n);
  float* VA = hull.data(); float* VB = VA + (size_t)n * 4; float* NA = VB + (size_t)n * 4; float* NB = NA + (size_t)n * 4;
  for (int i = 0; i < n; ++i) {
    float fx[4], fy[4], box[4];
    for (int k = 0; k < 4; ++k) { fx[k] = quads[(size_t)i * 8 + 2 * k]; fy[k] = quads[(size_t)i * 8 + 2 * k + 1]; }
    mcr_tile_hull(fx, fy, box, VA + (size_t)i * 4, VB + (size_t)i * 4, NA + (size_t)i * 4, NB + (size_t)i * 4, &cnt[i]);
  }
  uint8_t* d = nullptr;
  const size_t b_hull = sizeof(float) * 16 * (size_t)n, b_cnt = sizeof(int) * (size_t)n, b_pose = sizeof(float) * 3 * (size_t)n;
  HIPCHK(hipMalloc(&d, b_hull + b_cnt + b_pose + (size_t)n));
  float* d_hull = (float*)d; int* d_cnt = (int*)(d + b_hull); float* d_pose = (float*)(d + b_hull + b_cnt); uint8_t* d_out = d + b_hull + b_cnt + b_pose;
  hipError_t e = hipMemcpy(d_hull, hull.data(), b_hull, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_cnt, cnt.data(), b_cnt, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_pose, poses, b_pose, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_debug_overlap, dim3((n + 255) / 256), dim3(256), 0, 0, h->P.shapes, fixture, n, (const float4*)d_hull, (const float4*)(d_hull + (size_t)n * 4),
                       (const float4*)(d_hull + (size_t)n * 8), (const float4*)(d_hull + (size_t)n * 12), d_cnt, d_pose, d_out);
    e = hipMemcpy(out, d_out, (size_t)n, hipMemcpyDeviceToHost);
  }
  (void)hipFree(d);
  if (e != hipSuccess) { g_err = std::string("mcr_debug_overlap: ") + hipGetErrorString(e); return MCR_ERR_HIP; }
  return MCR_OK;
}
extern "C" int mcr_debug_set(mcr_env* h, int value) { if (!h) return MCR_ERR_ARG; h->P.debug = value; return MCR_OK; }
extern "C" int mcr_timing_enable(mcr_env* h, int enable) { if (!h) return MCR_ERR_ARG; h->timing = enable; return MCR_OK; }
extern "C" int mcr_timing_read(mcr_env* h, double* ms_out, int64_t* launches_out) {
  if (!h) return MCR_ERR_ARG;
  HIPCHK(hipDeviceSynchronize());
  for (auto& t : h->pending) {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) { h->t_ms[t.id] += ms; h->t_n[t.id] += 1; }
    h->free_events.push_back(t.a); h->free_events.push_back(t.b);
  }
  h->pending.clear();
  for (int i = 0; i < MCR_TIMING_SLOTS; ++i) { if (ms_out) ms_out[i] = h->t_ms[i]; if (launches_out) launches_out[i] = h->t_n[i]; h->t_ms[i] = 0; h->t_n[i] = 0; }
  return MCR_OK;
}
