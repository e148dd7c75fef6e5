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
  bool overlap;               // dynamics (latency bound, 128 waves) and raster (throughput bound) on disjoint CU sets
  hipStream_t s_phys, s_view; // CU-masked internal streams
  hipEvent_t ev_collide, ev_phys, ev_view;
  uint32_t serial;
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
  const size_t o_ready = carve(sizeof(uint32_t) * B);
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
  P.ready = (uint32_t*)(base + o_ready);
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
  // Kernel-level overlap.  k_dynamics is a long serial dependency chain on B*N/64 wavefronts (1 per SIMD on 32
  // CUs) whose duration is set by its slowest lane, k_view is throughput bound: give them disjoint CU sets
  // (CU-masked streams) and let every view start as soon as its own env has been published (ready[env]).
  // Disjoint CU sets make the wait deadlock-free: the raster can never occupy the CUs the physics needs.
  P.env0 = 0; P.nenv = B; P.wait_ready = 0; P.serial = 0;
  h->serial = 0; h->overlap = false;
  if (cfg->num_streams == 2 && cfg->obs_enabled) {
    int ncu = 0; (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, cfg->device);
    const int phys_cus = 32;
    if (ncu >= 128) {
      uint32_t mphys[16] = {0}, mview[16] = {0};
      for (int c = 0; c < ncu && c < 512; ++c) { if (c < phys_cus) mphys[c >> 5] |= 1u << (c & 31); else mview[c >> 5] |= 1u << (c & 31); }
      const uint32_t words = (uint32_t)((ncu + 31) / 32);
      if (hipExtStreamCreateWithCUMask(&h->s_phys, words, mphys) == hipSuccess) {
        if (hipExtStreamCreateWithCUMask(&h->s_view, words, mview) == hipSuccess) {
          (void)hipEventCreateWithFlags(&h->ev_collide, hipEventDisableTiming); (void)hipEventCreateWithFlags(&h->ev_phys, hipEventDisableTiming);
          (void)hipEventCreateWithFlags(&h->ev_view, hipEventDisableTiming);
          h->overlap = true;
        } else (void)hipStreamDestroy(h->s_phys);
      }
      (void)hipGetLastError();
    }
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
  if (h->overlap) { (void)hipStreamDestroy(h->s_phys); (void)hipStreamDestroy(h->s_view); (void)hipEventDestroy(h->ev_collide); (void)hipEventDestroy(h->ev_phys); (void)hipEventDestroy(h->ev_view); }
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

// reset(): install -> collide(1) -> dynamics(1) -> view, in order on one stream
static void launch_reset(mcr_env* h, McrParams P, hipStream_t st) {
  const int B = P.B, N = P.N;
  // (in overlap mode every step already joined s_phys/s_view into st, so st is ordered after them; the next
  // step's physics waits for ev_collide recorded on st after this reset)
  const int dyn_blocks = (B * P.G + 63) / 64;
  P.wait_ready = 0;
  hipLaunchKernelGGL(k_install, dim3(dyn_blocks), dim3(64), 0, st, P);
  LAUNCH(3, k_collide, B, 64, st, P, 1);
  LAUNCH(4, k_dynamics, dyn_blocks, 64, st, P, 1);
  if (P.obs) LAUNCH_LDS(2, k_view, B * N, VIEW_THREADS, (size_t)N * 12 * 6 * 16, st, P, h->view_scratch, 0, 1);
}

// step(): collide -> dynamics [-> auto-reset pass] -> view, all on the caller's stream by default.
// Optional overlap (cfg.num_streams == 2): the dynamics chain (latency bound, B*N/64 wavefronts) and the raster
// (throughput bound) run concurrently on CU-masked internal streams with disjoint CU sets; each view is gated IN
// THE KERNEL on its env's ready word, so the spin can never starve the physics of CUs.  MEASURED on MI355X /
// ROCm 7.0 runtime: every cross-stream event hop costs ~60-70 us here, which eats the overlap (0.72 ms/step vs
// 0.65 ms/step serial at B=4096,N=2) — so it stays off until launches are replayed from a hipGraph.
static void launch_step(mcr_env* h, McrParams P, hipStream_t st, int view_flags) {
  const int B = P.B, N = P.N;
  const int dyn_blocks = (B * P.G + 63) / 64;
  const bool ov = h->overlap && P.obs != nullptr;
  P.serial = ++h->serial; P.wait_ready = ov ? 1 : 0;
  LAUNCH(0, k_collide, B, 64, st, P, 0);
  hipStream_t sp = st, sv = st;
  if (ov) {
    sp = h->s_phys; sv = h->s_view;
    (void)hipEventRecord(h->ev_collide, st);
    (void)hipStreamWaitEvent(sp, h->ev_collide, 0); (void)hipStreamWaitEvent(sv, h->ev_collide, 0);
  }
  LAUNCH(1, k_dynamics, dyn_blocks, 64, sp, P, 0);
  if (P.auto_reset) {   // envs re-spawned by pass 0 take the action-less first step of their new episode (:408)
    LAUNCH(3, k_collide, B, 64, sp, P, 1);
    LAUNCH(4, k_dynamics, dyn_blocks, 64, sp, P, 1);
  }
  if (P.obs || view_flags) LAUNCH_LDS(2, k_view, B * N, VIEW_THREADS, (size_t)N * 12 * 6 * 16, sv, P, h->view_scratch, view_flags, 0);
  if (ov) {
    (void)hipEventRecord(h->ev_phys, sp); (void)hipEventRecord(h->ev_view, sv);
    (void)hipStreamWaitEvent(st, h->ev_phys, 0); (void)hipStreamWaitEvent(st, h->ev_view, 0);
  }
}

extern "C" int mcr_reset(mcr_env* h, const uint8_t* d_env_mask, uint8_t* d_obs, void* stream) {
  if (!h) { g_err = "null handle"; return MCR_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  McrParams P = h->P;
  P.reset_mask = d_env_mask; P.obs = h->cfg.obs_enabled ? d_obs : nullptr; P.actions = nullptr;
  launch_reset(h, P, st);
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
  launch_step(h, P, st, d_actions ? 1 : 0);
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
