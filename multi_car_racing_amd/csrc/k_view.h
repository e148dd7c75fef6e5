// k_view.h — one 256-thread workgroup per agent view (env, agent): the 96x96 ego-frame software
// rasteriser that replaces pyglet/OpenGL (multi_car_racing.py:511-604, 613-674; gym Car.draw), plus the
// per-agent backward/on-grass bookkeeping of :446-495 (its result only reaches pixels one step later).
//
// HBM traffic per view: read the env's road_poly once (2 x float4 + u32 per quad, coalesced), ~50 scalars of
// car state, write 27,648 B of packed RGB with 16-byte-per-lane stores.  Everything else lives in ~30 KB of
// LDS so that 5 workgroups (20 waves) share a CU:
//   1. camera (:540-556) -> 2x3 world->pixel matrix (f32, like the GL pipeline);
//   2. cull + setup: threads stride over quads, transform, reject by pixel bbox (incl. "contains no pixel
//      centre"), store oriented edge equations of the survivors (LDS; rare overflow spills to a per-view HBM
//      scratch) and the car polygons (12 per car);
//   3. bin: the thread that set a polygon up appends it to the lists of the 8x8-pixel bins it can touch
//      (box-vs-convex test, LDS atomics; order is irrelevant because the highest draw index wins);
//   4. shade: one wave per bin, lane = pixel; list walking is wave-uniform (LDS broadcast reads).  Background
//      (playfield + checker) is analytic in world space; road/kerb: highest road_poly index wins (== painter's
//      order); then cars; then the HUD in window space.  Result: one palette index per pixel (u8 framebuffer);
//   5. write-out: palette -> packed RGB, 16 B per lane, three byte-phase patterns.
// Sampling rule: pixel centres; a pixel belongs to a convex polygon iff all oriented edge functions are >= 0.
#pragma once
#include "mcr_kernels.h"

namespace view {

#define VIEW_THREADS 256
#define VIS_LDS 112                  // road survivors kept in LDS; more spill to HBM scratch (zoomed-out frames)
#define BIN_CAP 14                   // entries per 8x8 bin list; overflow -> the bin walks every survivor
#define CAR_KEY 1024                 // bin-list ids >= CAR_KEY are car polygons (drawn after every road quad)
#define NBINS 144
#define CARPOLY_CAP (MCR_MAX_AGENTS * 12)
#define VIEW_SCRATCH_FLOATS (MCR_QUAD_CAP * 16)   // per-view spill area (3 x float4 edge eq + info)

// palette
enum { PAL_BLACK = 0, PAL_GRASS0, PAL_GRASS1, PAL_ROAD0, PAL_ROAD1, PAL_ROAD2, PAL_WHITE, PAL_RED255, PAL_WHEELWHITE,
       PAL_CAR0, PAL_BLUE255 = PAL_CAR0 + 8, PAL_PURPLE, PAL_GREEN255, PAL_COUNT };

__device__ __forceinline__ uint32_t rgb(uint32_t r, uint32_t g, uint32_t b) { return r | (g << 8) | (b << 16); }
// GL float colour -> unorm8: round-to-nearest of c*255 evaluated on the f32 value
__device__ __forceinline__ uint32_t c8(double c) { return (uint32_t)floor((double)(float)c * 255.0 + 0.5); }

__device__ __forceinline__ uint32_t palette_rgb(int i) {
  switch (i) {
    case PAL_BLACK: return 0;
    case PAL_GRASS0: return rgb(c8(0.4), c8(0.8), c8(0.4));
    case PAL_GRASS1: return rgb(c8(0.4), c8(0.9), c8(0.4));
    case PAL_ROAD0: { uint32_t g = c8(0.4); return rgb(g, g, g); }
    case PAL_ROAD1: { uint32_t g = c8(0.4 + 0.01); return rgb(g, g, g); }
    case PAL_ROAD2: { uint32_t g = c8(0.4 + 0.01 * 2); return rgb(g, g, g); }
    case PAL_WHITE: return rgb(255, 255, 255);
    case PAL_RED255: return rgb(255, 0, 0);
    case PAL_WHEELWHITE: { uint32_t g = c8(0.3); return rgb(g, g, g); }
    case PAL_BLUE255: return rgb(0, 0, 255);
    case PAL_PURPLE: return rgb(c8(0.2), 0, 255);
    case PAL_GREEN255: return rgb(0, 255, 0);
    default: break;
  }
  if (i >= PAL_CAR0 && i < PAL_CAR0 + 8) {   // CAR_COLORS (:67-70)
    const int k = i - PAL_CAR0; const uint32_t v = c8(0.8);
    const uint32_t r = (k == 0 || k == 4 || k == 6 || k == 7) ? v : 0, g = (k == 2 || k == 3 || k == 4 || k == 7) ? v : 0,
                   b = (k == 1 || k == 3 || k == 4 || k == 6) ? v : 0;
    return rgb(r, g, b);
  }
  return 0;
}

struct Cam { float m00, m01, m10, m11, tx, ty; };   // pixel = M * world + t (already scaled by 96/1000, 96/800)

// oriented edge equations of a convex polygon given pixel-space vertices; returns false if degenerate
__device__ __forceinline__ bool edge_setup(const float* px, const float* py, int n, float* e /*[n*3]*/) {
  float area = 0.0f;
  for (int i = 0; i < n; ++i) { int j = (i + 1 == n) ? 0 : i + 1; area += px[i] * py[j] - px[j] * py[i]; }
  if (area == 0.0f) return false;
  const float sg = area > 0.0f ? 1.0f : -1.0f;
  for (int i = 0; i < n; ++i) {
    int j = (i + 1 == n) ? 0 : i + 1;
    float ex = px[j] - px[i], ey = py[j] - py[i];
    float A = -sg * ey, B = sg * ex;
    e[i * 3 + 0] = A; e[i * 3 + 1] = B; e[i * 3 + 2] = -(A * px[i] + B * py[i]);
  }
  return true;
}

// does the pixel-centre lattice have a point in [lo, hi] (clipped to rows/cols [c0, c1])?
__device__ __forceinline__ bool centre_range(float lo, float hi, int c0, int c1, int& i0, int& i1) {
  i0 = (int)ceilf(lo - 0.5f); i1 = (int)floorf(hi - 0.5f);
  if (i0 < c0) i0 = c0;
  if (i1 > c1) i1 = c1;
  return i0 <= i1;
}

}  // namespace view

// One 4-edge record vs one pixel centre: all oriented edge functions >= 0 (no short-circuit: one LDS burst).
__device__ __forceinline__ bool inside4(const float4 a, const float4 b, const float4 c, float cx, float cy) {
  const float e0 = a.x * cx + a.y * cy + a.z, e1 = a.w * cx + b.x * cy + b.y, e2 = b.z * cx + b.w * cy + c.x, e3 = c.y * cx + c.z * cy + c.w;
  return fminf(fminf(e0, e1), fminf(e2, e3)) >= 0.0f;
}
// conservative convex-vs-box test: false if some edge function is negative on the whole pixel-centre box
__device__ __forceinline__ bool box_may_touch(const float* e, int n, float X0, float X1, float Y0, float Y1) {
  bool out = false;
  for (int k = 0; k < n; ++k) {
    const float A = e[k * 3], B = e[k * 3 + 1], C = e[k * 3 + 2];
    const float m = A * (A >= 0.0f ? X1 : X0) + B * (B >= 0.0f ? Y1 : Y0) + C;
    out = out || (m < 0.0f);
  }
  return !out;
}

// flags_mode: 1 = evaluate the backward/on-grass block (:446-495) for this agent.
// dynamic LDS: car polygon records, N*12 x 6 float4 (8 edges each, padded with always-true edges)
__global__ __launch_bounds__(VIEW_THREADS) void k_view(McrParams p, float* __restrict__ scratch, int flags_mode, int only_just_reset) {
  using namespace view;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int N = p.N, BN = p.BN;
  // XCD-aware mapping: workgroup b runs on XCD b % 8, so hand the N views of one env to workgroups b, b+8,
  // b+16, ... — they share that XCD's L2 for the env's road_poly instead of fetching it once per XCD.
  int vw;
  {
    const int b = blockIdx.x, full = (p.nenv / 8) * 8 * N;
    if (b < full) { const int grp = b / (8 * N), r = b - grp * 8 * N; vw = (p.env0 + grp * 8 + (r & 7)) * N + (r >> 3); }
    else vw = p.env0 * N + b;
  }
  const int env = vw / N, agent = vw % N;
  const McrEnvState es = p.env[env];
  if (!es.active) return;
  if (only_just_reset && !es.just_reset) return;
  const uint8_t* __restrict__ slot = p.slots + ((size_t)env * 2 + es.slot) * MCR_SLOT_BYTES;
  const McrSlotHeader* H = (const McrSlotHeader*)slot;
  const int T = H->T, P = H->P;
  const int ci = env * N + agent;
  const McrShapes& S = *p.shapes;
  const int dbg = p.debug;

  extern __shared__ __attribute__((aligned(16))) float4 car8[];             // [N*12][6]
  __shared__ __attribute__((aligned(16))) uint8_t fb[96 * 96];
  __shared__ __attribute__((aligned(16))) float4 qe[VIS_LDS][3];          // 4 oriented edges (A,B,C) of each surviving quad
  __shared__ uint32_t qinfo[VIS_LDS];                                       // bins(16) | quad index(10) << 3 | colour(3)
  __shared__ uint16_t bins[NBINS][BIN_CAP];
  __shared__ int bcnt[NBINS];
  __shared__ uint32_t cinfo[CARPOLY_CAP];                                   // 0x100 | palette ; 0 = not drawn
  __shared__ uint32_t pal[32];
  __shared__ double red_d[4]; __shared__ int red_i[4];
  __shared__ int nvis, any_inside, next_bin;

  if (tid == 0) { nvis = 0; any_inside = 0; next_bin = 4; }
  if (tid < 32) pal[tid] = palette_rgb(tid);
  if (tid < NBINS) bcnt[tid] = 0;
  const uint32_t old_flags = p.caru[CU_FLAGS * BN + ci];
  const float hvx = p.carf[(CF_VX + 0) * BN + ci], hvy = p.carf[(CF_VY + 0) * BN + ci], ha = p.carf[(CF_A + 0) * BN + ci];
  const bool draw = p.obs != nullptr;
  const float* __restrict__ vp = p.viewp + (size_t)ci * MCR_VIEWP_FLOATS;
  float m00 = 0, m01 = 0, m10 = 0, m11 = 0, ctx = 0, cty = 0;
  if (draw) { m00 = vp[VP_CAM + 0]; m01 = vp[VP_CAM + 1]; m10 = vp[VP_CAM + 2]; m11 = vp[VP_CAM + 3]; ctx = vp[VP_CAM + 4]; cty = vp[VP_CAM + 5]; }
  float* __restrict__ spill = scratch + (size_t)vw * VIEW_SCRATCH_FLOATS;
  __syncthreads();

  // ---- car polygons (Car.draw): per car 4x(wheel box, white stripe) then 4 hull polys, cars in id order.
  // Threads 0..12N-1 set one polygon up each, then join the road loop.
  if (draw && tid < N * 12) {
    const int k = tid, c = k / 12, j = k % 12;
    const int cj = env * N + c;
    uint32_t info = 0;
    float lx[8], ly[8]; int n = 0; uint32_t colr = 0; Xf xf;
    if (j < 8) {
      const int wk = j >> 1;
      const V2 cc = v2(p.carf[(CF_CX + 1 + wk) * BN + cj], p.carf[(CF_CY + 1 + wk) * BN + cj]);
      xf = xf_of(cc, p.carf[(CF_A + 1 + wk) * BN + cj], v2(0.0f, 0.0f));
      if ((j & 1) == 0) { n = S.wheel.n; for (int i = 0; i < n; ++i) { lx[i] = S.wheel.vx[i]; ly[i] = S.wheel.vy[i]; } colr = PAL_BLACK; }
      else {
        const double ph = p.card[(CD_PHASE + wk) * BN + cj];
        const double a1 = ph, a2 = ph + 1.2;
        const double s1 = sin(a1), s2 = sin(a2); double c1 = cos(a1), c2 = cos(a2);
        if (!(s1 > 0 && s2 > 0)) {
          if (s1 > 0) c1 = dyn::np_sign(c1);
          if (s2 > 0) c2 = dyn::np_sign(c2);
          n = 4;
          lx[0] = (float)(-MCR_WHEEL_W * MCR_SIZE); ly[0] = (float)(+MCR_WHEEL_R * c1 * MCR_SIZE);
          lx[1] = (float)(+MCR_WHEEL_W * MCR_SIZE); ly[1] = (float)(+MCR_WHEEL_R * c1 * MCR_SIZE);
          lx[2] = (float)(+MCR_WHEEL_W * MCR_SIZE); ly[2] = (float)(+MCR_WHEEL_R * c2 * MCR_SIZE);
          lx[3] = (float)(-MCR_WHEEL_W * MCR_SIZE); ly[3] = (float)(+MCR_WHEEL_R * c2 * MCR_SIZE);
          colr = PAL_WHEELWHITE;
        }
      }
    } else {
      const int hk = j - 8;
      const V2 cc = v2(p.carf[(CF_CX + 0) * BN + cj], p.carf[(CF_CY + 0) * BN + cj]);
      xf = xf_of(cc, p.carf[(CF_A + 0) * BN + cj], v2(S.hull_lcx, S.hull_lcy));
      n = S.hull[hk].n; for (int i = 0; i < n; ++i) { lx[i] = S.hull[hk].vx[i]; ly[i] = S.hull[hk].vy[i]; }
      colr = PAL_CAR0 + (c & 7);                                              // CAR_COLORS[c % 8] (:402)
      if (p.use_ego_color) colr = (c == agent) ? PAL_CAR0 + 0 : PAL_CAR0 + 1; // (:560-563)
    }
    if (n > 0 && !(dbg & 4)) {
      float px[8], py[8];
      float x0 = MCR_MAXFLT, x1 = -MCR_MAXFLT, y0 = MCR_MAXFLT, y1 = -MCR_MAXFLT;
      for (int i = 0; i < n; ++i) {
        const V2 w = xmul(xf, v2(lx[i], ly[i]));                 // trans*v in f32, as pybox2d returns it
        px[i] = m00 * w.x + m01 * w.y + ctx; py[i] = m10 * w.x + m11 * w.y + cty;
        x0 = fminf(x0, px[i]); x1 = fmaxf(x1, px[i]); y0 = fminf(y0, py[i]); y1 = fmaxf(y1, py[i]);
      }
      float e[24];
      int ix0, ix1, iy0, iy1;
      if (centre_range(x0, x1, 0, 95, ix0, ix1) && centre_range(y0, y1, 12, 95, iy0, iy1) && edge_setup(px, py, n, e)) {
        for (int i = n * 3; i < 24; i += 3) { e[i] = 0.0f; e[i + 1] = 0.0f; e[i + 2] = 1.0f; }     // always-true padding edges
#pragma unroll
        for (int i = 0; i < 6; ++i) car8[k * 6 + i] = make_float4(e[i * 4], e[i * 4 + 1], e[i * 4 + 2], e[i * 4 + 3]);
        info = 0x100u | colr;
        for (int by = iy0 >> 3; by <= (iy1 >> 3); ++by)
          for (int bx = ix0 >> 3; bx <= (ix1 >> 3); ++bx) {
            const float X0 = (float)(bx * 8) + 0.5f, Y0 = (float)(by * 8) + 0.5f;
            if (!box_may_touch(e, n, X0, X0 + 7.0f, Y0, Y0 + 7.0f)) continue;
            const int b = by * 12 + bx;
            const int at = atomicAdd(&bcnt[b], 1);
            if (at < BIN_CAP) bins[b][at] = (uint16_t)(CAR_KEY + k);
          }
      }
    }
    cinfo[k] = info;
  }

  // ---- road quads: cull + edge setup + bin
  if (draw && !(dbg & 16)) {
    const float4* QA = (const float4*)(slot + MCR_OFF_QA); const float4* QB = (const float4*)(slot + MCR_OFF_QB);
    const uint32_t* QM = (const uint32_t*)(slot + MCR_OFF_QMETA);
    const uint16_t* tflags = p.tile_flags + (size_t)env * MCR_TILE_CAP;
    for (int q = tid; q < P; q += VIEW_THREADS) {
      const float4 a = QA[q], b = QB[q];
      const float wx[4] = {a.x, a.z, b.x, b.z}, wy[4] = {a.y, a.w, b.y, b.w};
      float px[4], py[4];
      float x0 = MCR_MAXFLT, x1 = -MCR_MAXFLT, y0 = MCR_MAXFLT, y1 = -MCR_MAXFLT;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        px[i] = m00 * wx[i] + m01 * wy[i] + ctx; py[i] = m10 * wx[i] + m11 * wy[i] + cty;
        x0 = fminf(x0, px[i]); x1 = fmaxf(x1, px[i]); y0 = fminf(y0, py[i]); y1 = fmaxf(y1, py[i]);
      }
      int ix0, ix1, iy0, iy1;
      if (!centre_range(x0, x1, 0, 95, ix0, ix1) || !centre_range(y0, y1, 12, 95, iy0, iy1)) continue;   // rows < 12: HUD bar
      float e[12];
      if (!edge_setup(px, py, 4, e)) continue;
      const uint32_t meta = QM[q];
      uint32_t col = meta & 0xffu; const uint32_t tile1 = meta >> 8;
      if (tile1 && (tflags[tile1 - 1] & 0x100u)) col = MCR_COL_ROAD0;           // touched tile -> ROAD_COLOR (:102-104)
      const uint32_t info = ((uint32_t)q << 3) | col;
      const int s = atomicAdd(&nvis, 1);
      if (s < VIS_LDS) {
        qe[s][0] = make_float4(e[0], e[1], e[2], e[3]); qe[s][1] = make_float4(e[4], e[5], e[6], e[7]); qe[s][2] = make_float4(e[8], e[9], e[10], e[11]);
        qinfo[s] = info;
      } else {
        float4* d = (float4*)(spill + (size_t)(s - VIS_LDS) * 16);
        d[0] = make_float4(e[0], e[1], e[2], e[3]); d[1] = make_float4(e[4], e[5], e[6], e[7]); d[2] = make_float4(e[8], e[9], e[10], e[11]);
        d[3] = make_float4(__uint_as_float(info), 0.0f, 0.0f, 0.0f);
      }
      for (int by = iy0 >> 3; by <= (iy1 >> 3); ++by)
        for (int bx = ix0 >> 3; bx <= (ix1 >> 3); ++bx) {
          const float X0 = (float)(bx * 8) + 0.5f, Y0 = (float)(by * 8) + 0.5f;
          if (!box_may_touch(e, 4, X0, X0 + 7.0f, Y0, Y0 + 7.0f)) continue;
          const int bb = by * 12 + bx;
          const int at = atomicAdd(&bcnt[bb], 1);
          if (at < BIN_CAP) bins[bb][at] = (uint16_t)s;
        }
    }
  }

  // ---- backward / on-grass flags (:446-495) from the post-solve pose; they reach pixels one step later
  const bool do_flags = flags_mode && !(dbg & 1);
  if (do_flags) {
    const Xf hxf = xf_of(v2(p.carf[(CF_CX + 0) * BN + ci], p.carf[(CF_CY + 0) * BN + ci]), ha, v2(S.hull_lcx, S.hull_lcy));
    const double px = (double)hxf.p.x, py = (double)hxf.p.y;
    const double* TX = (const double*)(slot + MCR_OFF_TRACK_X); const double* TY = (const double*)(slot + MCR_OFF_TRACK_Y); const double* TB = (const double*)(slot + MCR_OFF_TRACK_B);
    const double* TC = (const double*)(slot + MCR_OFF_TRACK_C); const double* TS = (const double*)(slot + MCR_OFF_TRACK_S);
    const uint32_t* TCNT = (const uint32_t*)(slot + MCR_OFF_TCNT);
    const float4* TAABB = (const float4*)(slot + MCR_OFF_TAABB);
    const double TW = 40 / MCR_SCALE, TBW = 8 / MCR_SCALE;
    const float fpx = hxf.p.x, fpy = hxf.p.y, margin = (float)(8 / MCR_SCALE) + 0.01f;
    double bd = 1e300; int bi = 0x7fffffff;
    bool inside = false;
    for (int t = tid; t < T; t += VIEW_THREADS) {
      const double x1 = TX[t], y1 = TY[t];
      const double dx = px - x1, dy = py - y1;
      const double d = sqrt(dx * dx + dy * dy);            // np.linalg.norm(..., axis=1) then argmin (:465-467)
      if (d < bd) { bd = d; bi = t; }
      // strict-interior point-in-quad over road_poly (shapely `within`, :470-472) on the f64 polygons the
      // reference builds (:313-317 tile, :329-333 kerb); a f32 AABB (+kerb width) prefilter skips far tiles
      const float4 bb = TAABB[t];
      if (fpx < bb.x - margin || fpx > bb.z + margin || fpy < bb.y - margin || fpy > bb.w + margin) continue;
      const int u = t == 0 ? T - 1 : t - 1;
      const double c1 = TC[t], s1 = TS[t], x2 = TX[u], y2 = TY[u], c2 = TC[u], s2 = TS[u];
      const int nqd = (TCNT[t] & 0x100u) ? 2 : 1;
      for (int k = 0; k < nqd; ++k) {
        double X[4], Y[4];
        if (k == 0) {
          X[0] = x1 - TW * c1; Y[0] = y1 - TW * s1; X[1] = x1 + TW * c1; Y[1] = y1 + TW * s1;
          X[2] = x2 + TW * c2; Y[2] = y2 + TW * s2; X[3] = x2 - TW * c2; Y[3] = y2 - TW * s2;
        } else {
          const double side = dyn::np_sign(TB[u] - TB[t]);
          const double w0 = side * TW, w1 = side * (TW + TBW);
          X[0] = x1 + w0 * c1; Y[0] = y1 + w0 * s1; X[1] = x1 + w1 * c1; Y[1] = y1 + w1 * s1;
          X[2] = x2 + w1 * c2; Y[2] = y2 + w1 * s2; X[3] = x2 + w0 * c2; Y[3] = y2 + w0 * s2;
        }
        bool pos = true, neg = true;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int j = (i + 1) & 3;
          const double cr = (X[j] - X[i]) * (py - Y[i]) - (Y[j] - Y[i]) * (px - X[i]);
          if (!(cr > 0)) pos = false;
          if (!(cr < 0)) neg = false;
        }
        inside = inside || pos || neg;
      }
    }
    if (__any(inside) && lane == 0) atomicOr(&any_inside, 1);
    // wave argmin (first index on ties), partials combined after the barrier
    for (int o = 32; o > 0; o >>= 1) {
      const double d2 = __shfl_xor(bd, o); const int i2 = __shfl_xor(bi, o);
      if (d2 < bd || (d2 == bd && i2 < bi)) { bd = d2; bi = i2; }
    }
    if (lane == 0) { red_d[wave] = bd; red_i[wave] = bi; }
  }
  __syncthreads();
  const int nq = nvis;
  const int nq_lds = nq < VIS_LDS ? nq : VIS_LDS;

  // ---- finish the flags on one lane of the last wave while the other waves start shading
  if (do_flags && tid == VIEW_THREADS - 1) {
    const double* TB = (const double*)(slot + MCR_OFF_TRACK_B);
    double bd = red_d[0]; int bi = red_i[0];
    for (int w = 1; w < 4; ++w) if (red_d[w] < bd || (red_d[w] == bd && red_i[w] < bi)) { bd = red_d[w]; bi = red_i[w]; }
    const double TWO_PI = 2 * 3.141592653589793, PI = 3.141592653589793;
    double car_angle;
    const double vx = (double)hvx, vy = (double)hvy;
    if (sqrt(vx * vx + vy * vy) > 0.5) car_angle = -atan2(vx, vy); else car_angle = (double)ha;
    car_angle = fmod(car_angle + TWO_PI, TWO_PI); if (car_angle < 0) car_angle += TWO_PI;
    double desired = TB[bi];
    if (H->cw) desired += PI;
    desired = fmod(desired + TWO_PI, TWO_PI); if (desired < 0) desired += TWO_PI;
    double diff = fabs(desired - car_angle);
    if (diff > PI) diff = fabs(diff - TWO_PI);
    uint32_t f = 0;
    if (diff > PI / 2) f |= 1u;
    if (!any_inside) f |= 2u;
    p.caru[CU_FLAGS * BN + ci] = f;
  }
  if (!draw) return;

  // ---- shade: one wave per 8x8 bin (lane = pixel), bins handed out dynamically
  {
    const bool show_flag = (old_flags & 1u) && p.backwards_flag;
    const float kx = 96.0f / 1000.0f, ky = 96.0f / 800.0f;
    float fe[9];
    { const float fx[3] = {900.0f * kx, 925.0f * kx, 950.0f * kx}, fy[3] = {30.0f * ky, 70.0f * ky, 30.0f * ky}; edge_setup(fx, fy, 3, fe); }
    const float hud_top = vp[VP_HUDTOP];
    const int lx = lane & 7, ly = lane >> 3;
    const float inv_kgrid = 1.0f / (float)(MCR_PLAYFIELD / 20.0), PF = (float)MCR_PLAYFIELD;
    const float ax = vp[VP_INV + 0], bx_ = vp[VP_INV + 1], cx0 = vp[VP_INV + 2], ay = vp[VP_INV + 3], by_ = vp[VP_INV + 4], cy0 = vp[VP_INV + 5];
    const int ncar = N * 12;
    int b = wave;
    while (b < NBINS) {
      const int bxi = b % 12, byi = b / 12;
      const int ix = bxi * 8 + lx, iy = byi * 8 + ly;                 // GL pixel coords (origin bottom-left)
      const float cx = (float)ix + 0.5f, cy = (float)iy + 0.5f;
      uint32_t col = PAL_BLACK;
      if (byi >= 1) {                                                   // bin row 0 (y < 8) is entirely under the HUD bar
        const float wx = ax * cx + bx_ * cy + cx0, wy = ay * cx + by_ * cy + cy0;
        if (fabsf(wx) <= PF && fabsf(wy) <= PF) {
          const int gx = (int)floorf(wx * inv_kgrid), gy = (int)floorf(wy * inv_kgrid);
          col = (((gx | gy) & 1) == 0) ? PAL_GRASS1 : PAL_GRASS0;
        }
        // highest draw index covering the pixel wins (painter's order): road_poly index, then car polygons
        const int cnt = bcnt[b];
        int best = -1;
        if (cnt <= BIN_CAP) {
          for (int k = 0; k < cnt; ++k) {
            const int s = bins[b][k];
            if (s >= CAR_KEY) {
              const float4* r = &car8[(s - CAR_KEY) * 6];
              const bool in = inside4(r[0], r[1], r[2], cx, cy) && inside4(r[3], r[4], r[5], cx, cy);
              const int key = (s << 5) | (int)(cinfo[s - CAR_KEY] & 31u);
              if (in && key > best) best = key;
            } else if (s < VIS_LDS) {
              const int key = (int)qinfo[s] << 2;                        // (q << 5) | colour << 2
              if (inside4(qe[s][0], qe[s][1], qe[s][2], cx, cy) && key > best && !(dbg & 2)) best = key;
            } else {
              const float4* d = (const float4*)(spill + (size_t)(s - VIS_LDS) * 16);
              const int key = (int)__float_as_uint(d[3].x) << 2;
              if (inside4(d[0], d[1], d[2], cx, cy) && key > best) best = key;
            }
          }
        } else {
          for (int s = 0; s < nq_lds; ++s) {
            const int key = (int)qinfo[s] << 2;
            if (inside4(qe[s][0], qe[s][1], qe[s][2], cx, cy) && key > best) best = key;
          }
          for (int s = VIS_LDS; s < nq; ++s) {
            const float4* d = (const float4*)(spill + (size_t)(s - VIS_LDS) * 16);
            const int key = (int)__float_as_uint(d[3].x) << 2;
            if (inside4(d[0], d[1], d[2], cx, cy) && key > best) best = key;
          }
          for (int k = 0; k < ncar; ++k) {
            const uint32_t ci2 = cinfo[k];
            if (!ci2) continue;
            const float4* r = &car8[k * 6];
            const int key = ((CAR_KEY + k) << 5) | (int)(ci2 & 31u);
            if (inside4(r[0], r[1], r[2], cx, cy) && inside4(r[3], r[4], r[5], cx, cy) && key > best) best = key;
          }
        }
        if (best >= 0) {
          if (best >= (CAR_KEY << 5)) col = (uint32_t)best & 31u;
          else {
            const uint32_t bc = ((uint32_t)best >> 2) & 7u;
            col = bc == MCR_COL_ROAD0 ? PAL_ROAD0 : bc == MCR_COL_ROAD1 ? PAL_ROAD1 : bc == MCR_COL_ROAD2 ? PAL_ROAD2 : bc == MCR_COL_KERB_WHITE ? PAL_WHITE : PAL_RED255;
          }
        }
        if (iy < 12) col = PAL_BLACK;                                   // the HUD bar (window y < 100) covers the scene
      }
      if ((float)(byi * 8) < hud_top && cy < hud_top) {
        // HUD (window space, drawn last): gauges in draw order (a tall gauge may poke above the bar), then the flag
        const uint32_t ind_col[7] = {PAL_WHITE, PAL_BLUE255, PAL_BLUE255, PAL_PURPLE, PAL_PURPLE, PAL_GREEN255, PAL_RED255};
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          const float x0 = vp[VP_IND + i * 4], x1 = vp[VP_IND + i * 4 + 1], y0 = vp[VP_IND + i * 4 + 2], y1 = vp[VP_IND + i * 4 + 3];
          if (x1 > x0 && y1 > y0 && cx >= x0 && cx <= x1 && cy >= y0 && cy <= y1) col = ind_col[i];
        }
        if (show_flag && (fe[0] * cx + fe[1] * cy + fe[2] >= 0.0f) && (fe[3] * cx + fe[4] * cy + fe[5] >= 0.0f) && (fe[6] * cx + fe[7] * cy + fe[8] >= 0.0f)) col = PAL_BLUE255;
      }
      fb[(95 - iy) * 96 + ix] = (uint8_t)col;                          // arr[::-1] (:602)
      int nb = 0;
      if (lane == 0) nb = atomicAdd(&next_bin, 1);
      b = __shfl(nb, 0);
    }
  }
  __syncthreads();
  if (dbg & 8) return;

  // ---- packed RGB write-out: 16 B per lane = 6 pixels' worth of bytes in one of three phases
  uint4* __restrict__ out = (uint4*)(p.obs + (size_t)vw * (96 * 96 * 3));
  for (int ch = tid; ch < 96 * 96 * 3 / 16; ch += VIEW_THREADS) {
    const int o = ch * 16; const int p0 = o / 3; const int ph = o - p0 * 3;
    uint32_t c[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) c[j] = pal[fb[p0 + j]];
    uint32_t w0, w1, w2, w3;
    if (ph == 0) { w0 = c[0] | (c[1] << 24); w1 = (c[1] >> 8) | (c[2] << 16); w2 = (c[2] >> 16) | (c[3] << 8); w3 = c[4] | (c[5] << 24); }
    else if (ph == 1) { w0 = (c[0] >> 8) | (c[1] << 16); w1 = (c[1] >> 16) | (c[2] << 8); w2 = c[3] | (c[4] << 24); w3 = (c[4] >> 8) | (c[5] << 16); }
    else { w0 = (c[0] >> 16) | (c[1] << 8); w1 = c[2] | (c[3] << 24); w2 = (c[3] >> 8) | (c[4] << 16); w3 = (c[4] >> 16) | (c[5] << 8); }
    out[ch] = make_uint4(w0, w1, w2, w3);
  }
}
