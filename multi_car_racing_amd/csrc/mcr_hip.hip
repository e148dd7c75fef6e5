// mcr_hip.hip — C-ABI implementation (include/mcr.h) over the gfx950 kernels.
// One handle owns one env slice on one device: SoA car state, per-env state, two episode slots per env.
#include "../../include/mcr.h"
#include "mcr_kernels.h"
#include "k_dynamics.h"
#include "k_collide.h"
#include "k_view.h"
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include <cstring>
#include <cstdio>

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
  double t_ms[5]; int64_t t_n[5];
  bool any_reset;
  int ngroups;                // env sub-batches pipelined on internal streams (1 = caller's stream only)
  hipStream_t gstream[8];
  hipEvent_t ev_fork, ev_done[8];
  float* view_scratch;        // per-view spill area of the rasteriser (zoomed-out frames only)
};

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

extern "C" int mcr_create(const mcr_config* cfg, mcr_env** out) {
  if (!cfg || !out) { g_err = "null argument"; return MCR_ERR_ARG; }
  if (cfg->num_envs < 1 || cfg->num_agents < 1 || cfg->num_agents > MCR_MAX_AGENTS) { g_err = "num_envs/num_agents out of range"; return MCR_ERR_ARG; }
  HIPCHK(hipSetDevice(cfg->device));
  mcr_env* h = new mcr_env();
  h->cfg = *cfg; h->timing = 0; h->any_reset = false;
  for (int i = 0; i < 5; ++i) { h->t_ms[i] = 0; h->t_n[i] = 0; }
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
  const size_t o_shapes = carve(sizeof(McrShapes));
  const size_t o_viewp = carve(sizeof(float) * MCR_VIEWP_FLOATS * BN);
  const size_t o_carpoly = carve(sizeof(float) * MCR_CARPOLY_FLOATS * BN);
  const size_t o_vscratch = carve(sizeof(float) * (size_t)VIEW_SCRATCH_FLOATS * BN);
  const size_t o_slots = carve((size_t)B * 2 * MCR_SLOT_BYTES);
  h->slab_bytes = off;
  if (hipMalloc(&h->slab, off) != hipSuccess) { g_err = "hipMalloc failed"; delete h; return MCR_ERR_HIP; }
  (void)hipMemset(h->slab, 0, o_vscratch);
  uint8_t* base = (uint8_t*)h->slab;
  McrParams& P = h->P;
  memset(&P, 0, sizeof(P));
  P.B = B; P.N = N; P.G = G; P.BN = (int)BN;
  P.carf = (float*)(base + o_carf); P.card = (double*)(base + o_card); P.caru = (uint32_t*)(base + o_caru);
  P.env = (McrEnvState*)(base + o_env); P.tile_touch = (uint32_t*)(base + o_touch); P.tile_flags = (uint16_t*)(base + o_tflags);
  P.cc_store = (uint32_t*)(base + o_cc); P.shapes = (const McrShapes*)(base + o_shapes); P.slots = base + o_slots;
  h->view_scratch = (float*)(base + o_vscratch);
  P.viewp = (float*)(base + o_viewp);
  P.carpoly = (float*)(base + o_carpoly);
  P.auto_reset = cfg->auto_reset; P.max_steps = cfg->max_episode_steps; P.car_contacts = cfg->car_contacts;
  P.backwards_flag = cfg->backwards_flag; P.use_ego_color = cfg->use_ego_color; P.h_ratio = cfg->h_ratio;
  McrShapes S; mcr_build_shapes(&S);
  (void)hipMemcpy((void*)P.shapes, &S, sizeof(S), hipMemcpyHostToDevice);
  if (hipHostMalloc((void**)&h->consumed_host, sizeof(int32_t) * B, hipHostMallocMapped) != hipSuccess) { g_err = "hipHostMalloc failed"; (void)hipFree(h->slab); delete h; return MCR_ERR_HIP; }
  memset(h->consumed_host, 0, sizeof(int32_t) * B);
  void* dptr = nullptr;
  (void)hipHostGetDevicePointer(&dptr, h->consumed_host, 0);
  P.consumed_host = (int32_t*)dptr;
  h->consumed_seen = new int32_t[B]();
  // Stream-level pipelining: the dynamics kernel is a long serial dependency chain on few wavefronts while the
  // raster kernel is throughput bound, so env sub-batches on separate streams overlap one group's dynamics
  // with another group's raster/collide.  cfg.num_streams selects the group count (0 = default).
  h->ngroups = cfg->num_streams > 0 ? cfg->num_streams : 1;   // measured: without graph replay the extra launches cost more than the overlap wins
  if (h->ngroups > 8) h->ngroups = 8;
  while (h->ngroups > 1 && B / h->ngroups < 256) h->ngroups >>= 1;
  P.env0 = 0; P.nenv = B;
  if (h->ngroups > 1) {
    for (int g = 0; g < h->ngroups; ++g) { (void)hipStreamCreateWithFlags(&h->gstream[g], hipStreamNonBlocking); (void)hipEventCreateWithFlags(&h->ev_done[g], hipEventDisableTiming); }
    (void)hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming);
  }
  (void)hipDeviceSynchronize();
  *out = h;
  return MCR_OK;
}

extern "C" int mcr_destroy(mcr_env* h) {
  if (!h) return MCR_ERR_ARG;
  (void)hipSetDevice(h->cfg.device);
  (void)hipDeviceSynchronize();
  for (auto& t : h->pending) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
  for (auto e : h->free_events) (void)hipEventDestroy(e);
  if (h->ngroups > 1) { for (int g = 0; g < h->ngroups; ++g) { (void)hipStreamDestroy(h->gstream[g]); (void)hipEventDestroy(h->ev_done[g]); } (void)hipEventDestroy(h->ev_fork); }
  (void)hipFree(h->slab);
  (void)hipHostFree(h->consumed_host);
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
  static const int32_t one = 1;
  for (int i = 0; i < n; ++i) {
    const int e = env_ids ? env_ids[i] : i;
    if (e < 0 || e >= B) { g_err = "env id out of range"; return MCR_ERR_ARG; }
    const int32_t installs = ((volatile int32_t*)h->consumed_host)[e];
    uint8_t* dst = h->P.slots + ((size_t)e * 2 + ((installs & 1) ^ 1)) * MCR_SLOT_BYTES;
    HIPCHK(hipMemcpyAsync(dst, (const uint8_t*)blobs + (size_t)i * MCR_SLOT_BYTES, MCR_SLOT_BYTES, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(&h->P.env[e].staged_ready, &one, sizeof(int32_t), hipMemcpyHostToDevice, st));
  }
  return MCR_OK;
}

static hipEvent_t get_event(mcr_env* h) {
  if (!h->free_events.empty()) { hipEvent_t e = h->free_events.back(); h->free_events.pop_back(); return e; }
  hipEvent_t e; (void)hipEventCreate(&e); return e;
}
#define LAUNCH(kid_, kernel, grid, block, st, ...) LAUNCH_LDS(kid_, kernel, grid, block, 0, st, __VA_ARGS__)
#define LAUNCH_LDS(kid_, kernel, grid, block, lds_, st, ...)                                         \
  do {                                                                                   \
    TimedLaunch tl_; bool tm_ = (h->timing >> (kid_)) & 1;                                              \
    if (tm_) { tl_.id = (kid_); tl_.a = get_event(h); tl_.b = get_event(h); (void)hipEventRecord(tl_.a, st); } \
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), lds_, st, __VA_ARGS__);          \
    if (tm_) { (void)hipEventRecord(tl_.b, st); h->pending.push_back(tl_); }                   \
  } while (0)

// One env sub-range [e0, e0+ne) through the whole step on stream st.
static void launch_group(mcr_env* h, McrParams P, int e0, int ne, hipStream_t st, bool install, bool step_pass, bool reset_pass,
                         int view_flags, int view_only_just_reset) {
  P.env0 = e0; P.nenv = ne;
  const int N = P.N;
  const int dyn_blocks = (ne * P.G + 63) / 64;
  if (install) hipLaunchKernelGGL(k_install, dim3(dyn_blocks), dim3(64), 0, st, P);
  if (step_pass) {
    LAUNCH(0, k_collide, ne, 64, st, P, 0);
    LAUNCH(1, k_dynamics, dyn_blocks, 64, st, P, 0);
  }
  if (reset_pass) {   // the action-less first step of a freshly installed episode (:408)
    LAUNCH(3, k_collide, ne, 64, st, P, 1);
    LAUNCH(4, k_dynamics, dyn_blocks, 64, st, P, 1);
  }
  if (P.obs || view_flags) LAUNCH_LDS(2, k_view, ne * N, VIEW_THREADS, (size_t)N * 12 * 6 * 16, st, P, h->view_scratch, view_flags, view_only_just_reset);
}

static void launch_all(mcr_env* h, const McrParams& P, hipStream_t st, bool install, bool step_pass, bool reset_pass, int view_flags, int only_jr) {
  const int B = P.B;
  if (h->ngroups <= 1) { launch_group(h, P, 0, B, st, install, step_pass, reset_pass, view_flags, only_jr); return; }
  (void)hipEventRecord(h->ev_fork, st);
  const int per = (B + h->ngroups - 1) / h->ngroups;
  for (int g = 0; g < h->ngroups; ++g) {
    const int e0 = g * per; const int ne = (e0 + per <= B) ? per : B - e0;
    if (ne <= 0) break;
    (void)hipStreamWaitEvent(h->gstream[g], h->ev_fork, 0);
    launch_group(h, P, e0, ne, h->gstream[g], install, step_pass, reset_pass, view_flags, only_jr);
    (void)hipEventRecord(h->ev_done[g], h->gstream[g]);
    (void)hipStreamWaitEvent(st, h->ev_done[g], 0);
  }
}

extern "C" int mcr_reset(mcr_env* h, const uint8_t* d_env_mask, uint8_t* d_obs, void* stream) {
  if (!h) { g_err = "null handle"; return MCR_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  McrParams P = h->P;
  P.reset_mask = d_env_mask; P.obs = h->cfg.obs_enabled ? d_obs : nullptr; P.actions = nullptr;
  launch_all(h, P, st, true, false, true, 0, 1);
  HIPCHK(hipGetLastError());
  h->any_reset = true;
  return MCR_OK;
}

extern "C" int mcr_step(mcr_env* h, const float* d_actions, uint8_t* d_obs, double* d_reward, uint8_t* d_done, uint8_t* d_trunc, void* stream) {
  if (!h || !d_reward || !d_done) { g_err = "null argument"; return MCR_ERR_ARG; }
  if (!h->any_reset) { g_err = "step() before reset()"; return MCR_ERR_STATE; }
  hipStream_t st = (hipStream_t)stream;
  McrParams P = h->P;
  P.actions = d_actions; P.obs = h->cfg.obs_enabled ? d_obs : nullptr;
  P.reward_out = d_reward; P.done_out = d_done; P.trunc_out = d_trunc;
  // with auto_reset, finished envs are re-spawned on the device and take the action-less first step of their
  // new episode inside this call; the view kernel always runs (it also owns the backward/on-grass flags)
  launch_all(h, P, st, false, true, P.auto_reset != 0, d_actions ? 1 : 0, 0);
  HIPCHK(hipGetLastError());
  return MCR_OK;
}

extern "C" int mcr_poll_consumed(mcr_env* h, int32_t* env_ids_out, int cap, void* stream) {
  if (!h) return MCR_ERR_ARG;
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
  if (!h || !out) return MCR_ERR_ARG;
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out, h->view_scratch + (size_t)(view + 1) * VIEW_SCRATCH_FLOATS - 64, nbytes, hipMemcpyDeviceToHost));
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
  for (int i = 0; i < 5; ++i) { if (ms_out) ms_out[i] = h->t_ms[i]; if (launches_out) launches_out[i] = h->t_n[i]; h->t_ms[i] = 0; h->t_n[i] = 0; }
  return MCR_OK;
}
