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
//   3. bin: one thread per 8x8-pixel bin builds that bin's survivor list (box-vs-convex test);
//   4. shade: one wave per bin, lane = pixel; list walking is wave-uniform (LDS broadcast reads).  Background
//      (playfield + checker) is analytic in world space; road/kerb: highest road_poly index wins (== painter's
//      order); then cars; then the HUD in window space.  Result: one palette index per pixel (u8 framebuffer);
//   5. write-out: palette -> packed RGB, 16 B per lane, three byte-phase patterns.
// Sampling rule: pixel centres; a pixel belongs to a convex polygon iff all oriented edge functions are >= 0.
#pragma once
#include "mcr_kernels.h"

namespace view {

#define VIEW_THREADS 256
#define VIS_LDS 160                  // survivors kept in LDS; more spill to HBM scratch (zoomed-out frames)
#define BIN_CAP 24                   // entries per 8x8 bin list; overflow -> the bin walks every survivor
#define NBINS 144
#define CARPOLY_CAP (MCR_MAX_AGENTS * 12)
#define VIEW_SCRATCH_FLOATS (MCR_QUAD_CAP * 13)   // per-view spill area (edge eq + info)

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

// flags_mode: 1 = evaluate the backward/on-grass block (:446-495) for this agent.
__global__ __launch_bounds__(VIEW_THREADS) void k_view(McrParams p, float* scratch, int flags_mode, int only_just_reset) {
  using namespace view;
  const int vw = blockIdx.x, tid = threadIdx.x;
  const int N = p.N, BN = p.BN;
  const int env = vw / N, agent = vw % N;
  const McrEnvState es = p.env[env];
  if (!es.active) return;
  if (only_just_reset && !es.just_reset) return;
  const uint8_t* slot = p.slots + ((size_t)env * 2 + es.slot) * MCR_SLOT_BYTES;
  const McrSlotHeader* H = (const McrSlotHeader*)slot;
  const int T = H->T, P = H->P;
  const int ci = env * N + agent;
  const McrShapes& S = *p.shapes;

  __shared__ __attribute__((aligned(16))) uint8_t fb[96 * 96];
  __shared__ __attribute__((aligned(16))) float qe[VIS_LDS][12];          // also reused as reduction scratch by the flags block
  __shared__ uint32_t qinfo[VIS_LDS];                                       // bins(16) | quad index(10) << 3 | colour(3)
  __shared__ uint16_t bins[NBINS][BIN_CAP];
  __shared__ int bcnt[NBINS];
  __shared__ float ce[CARPOLY_CAP][24];                                     // up to 8 edges per car polygon
  __shared__ float cbb[CARPOLY_CAP][4];
  __shared__ uint32_t cinfo[CARPOLY_CAP];                                   // 0x10000 | (nedges << 8) | palette ; 0 = skip
  __shared__ float carbox[MCR_MAX_AGENTS][4];
  __shared__ uint32_t pal[32];
  __shared__ int nvis;
  __shared__ int any_inside;

  if (tid == 0) { nvis = 0; any_inside = 0; }
  if (tid < 32) pal[tid] = palette_rgb(tid);
  if (tid < MCR_MAX_AGENTS) { carbox[tid][0] = MCR_MAXFLT; carbox[tid][1] = 0.0f; carbox[tid][2] = MCR_MAXFLT; carbox[tid][3] = 0.0f; }
  const uint32_t old_flags = p.caru[CU_FLAGS * BN + ci];

  // ---- camera (:540-556).  f64 exactly as CPython, then the f32 values GL receives.
  const float hcx = p.carf[(CF_CX + 0) * BN + ci], hcy = p.carf[(CF_CY + 0) * BN + ci], ha = p.carf[(CF_A + 0) * BN + ci];
  const float hvx = p.carf[(CF_VX + 0) * BN + ci], hvy = p.carf[(CF_VY + 0) * BN + ci], hw = p.carf[(CF_W + 0) * BN + ci];
  const Xf hxf = xf_of(v2(hcx, hcy), ha, v2(S.hull_lcx, S.hull_lcy));
  Cam cam; float fz, fcs, fsn, ftx, fty;
  {
    const double t = es.t;
    const double zoom = 0.1 * MCR_SCALE * fmax(1 - t, 0.0) + MCR_ZOOM * MCR_SCALE * fmin(t, 1.0);
    const double sx = (double)hxf.p.x, sy = (double)hxf.p.y;
    double angle = -(double)ha;
    const double vx = (double)hvx, vy = (double)hvy;
    if (sqrt(vx * vx + vy * vy) > 0.5) angle = atan2(vx, vy);
    const double ttx = MCR_WINDOW_W / 2 - (sx * zoom * cos(angle) - sy * zoom * sin(angle));
    const double tty = MCR_WINDOW_H * p.h_ratio - (sx * zoom * sin(angle) + sy * zoom * cos(angle));
    ftx = (float)ttx; fty = (float)tty; fz = (float)zoom;
    const float fdeg = (float)(57.29577951308232 * angle);
    const double rad = (double)fdeg * (3.14159265358979323846 / 180.0);
    fcs = (float)cos(rad); fsn = (float)sin(rad);
    const float kx = 96.0f / 1000.0f, ky = 96.0f / 800.0f;
    cam.m00 = fcs * fz * kx; cam.m01 = -fsn * fz * kx; cam.tx = ftx * kx;
    cam.m10 = fsn * fz * ky; cam.m11 = fcs * fz * ky; cam.ty = fty * ky;
  }
  __syncthreads();

  // ---- backward / on-grass flags (:446-495) from the post-solve pose; they reach pixels one step later
  uint32_t new_flags = old_flags; bool write_flags = false;
  if (flags_mode) {
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
      const int nq = (TCNT[t] & 0x100u) ? 2 : 1;
      for (int k = 0; k < nq; ++k) {
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
    if (inside) atomicOr(&any_inside, 1);
    double* red_d = (double*)&qe[0][0]; int* red_i = (int*)(red_d + VIEW_THREADS);
    red_d[tid] = bd; red_i[tid] = bi;
    __syncthreads();
    for (int s = VIEW_THREADS / 2; s > 0; s >>= 1) {
      if (tid < s) {
        const double d2 = red_d[tid + s]; const int i2 = red_i[tid + s];
        if (d2 < red_d[tid] || (d2 == red_d[tid] && i2 < red_i[tid])) { red_d[tid] = d2; red_i[tid] = i2; }
      }
      __syncthreads();
    }
    if (tid == 0) {
      const double TWO_PI = 2 * 3.141592653589793, PI = 3.141592653589793;
      double car_angle;
      const double vx = (double)hvx, vy = (double)hvy;
      if (sqrt(vx * vx + vy * vy) > 0.5) car_angle = -atan2(vx, vy); else car_angle = (double)ha;
      car_angle = fmod(car_angle + TWO_PI, TWO_PI); if (car_angle < 0) car_angle += TWO_PI;
      double desired = TB[red_i[0]];
      if (H->cw) desired += PI;
      desired = fmod(desired + TWO_PI, TWO_PI); if (desired < 0) desired += TWO_PI;
      double diff = fabs(desired - car_angle);
      if (diff > PI) diff = fabs(diff - TWO_PI);
      new_flags = 0;
      if (diff > PI / 2) new_flags |= 1u;
      if (!any_inside) new_flags |= 2u;
      write_flags = true;
    }
    __syncthreads();
  }
  if (write_flags) p.caru[CU_FLAGS * BN + ci] = new_flags;
  if (p.obs == nullptr) return;

  // ---- road quads: cull + edge setup
  float* spill = scratch + (size_t)vw * VIEW_SCRATCH_FLOATS;
  {
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
        px[i] = cam.m00 * wx[i] + cam.m01 * wy[i] + cam.tx; py[i] = cam.m10 * wx[i] + cam.m11 * wy[i] + cam.ty;
        x0 = fminf(x0, px[i]); x1 = fmaxf(x1, px[i]); y0 = fminf(y0, py[i]); y1 = fmaxf(y1, py[i]);
      }
      int ix0, ix1, iy0, iy1;
      if (!centre_range(x0, x1, 0, 95, ix0, ix1) || !centre_range(y0, y1, 12, 95, iy0, iy1)) continue;   // rows < 12: HUD bar
      float e[12];
      if (!edge_setup(px, py, 4, e)) continue;
      const uint32_t meta = QM[q];
      uint32_t col = meta & 0xffu; const uint32_t tile1 = meta >> 8;
      if (tile1 && (tflags[tile1 - 1] & 0x100u)) col = MCR_COL_ROAD0;           // touched tile -> ROAD_COLOR (:102-104)
      const uint32_t info = ((uint32_t)(ix0 >> 3) << 28) | ((uint32_t)(ix1 >> 3) << 24) | ((uint32_t)(iy0 >> 3) << 20) | ((uint32_t)(iy1 >> 3) << 16) |
                            ((uint32_t)q << 3) | col;
      const int s = atomicAdd(&nvis, 1);
      if (s < VIS_LDS) {
#pragma unroll
        for (int i = 0; i < 12; ++i) qe[s][i] = e[i];
        qinfo[s] = info;
      } else {
        float* d = spill + (size_t)(s - VIS_LDS) * 13;
#pragma unroll
        for (int i = 0; i < 12; ++i) d[i] = e[i];
        d[12] = __uint_as_float(info);
      }
    }
  }
  // ---- car polygons (Car.draw): per car 4x(wheel box, white stripe) then 4 hull polys, cars in id order
  for (int k = tid; k < N * 12; k += VIEW_THREADS) {
    const int c = k / 12, j = k % 12;
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
    if (n > 0) {
      float px[8], py[8];
      float x0 = MCR_MAXFLT, x1 = -MCR_MAXFLT, y0 = MCR_MAXFLT, y1 = -MCR_MAXFLT;
      for (int i = 0; i < n; ++i) {
        const V2 w = xmul(xf, v2(lx[i], ly[i]));                 // trans*v in f32, as pybox2d returns it
        px[i] = cam.m00 * w.x + cam.m01 * w.y + cam.tx; py[i] = cam.m10 * w.x + cam.m11 * w.y + cam.ty;
        x0 = fminf(x0, px[i]); x1 = fmaxf(x1, px[i]); y0 = fminf(y0, py[i]); y1 = fmaxf(y1, py[i]);
      }
      float e[24];
      int a0, a1, b0, b1;
      if (centre_range(x0, x1, 0, 95, a0, a1) && centre_range(y0, y1, 12, 95, b0, b1) && edge_setup(px, py, n, e)) {
        for (int i = 0; i < n * 3; ++i) ce[k][i] = e[i];
        cbb[k][0] = x0; cbb[k][1] = x1; cbb[k][2] = y0; cbb[k][3] = y1;
        info = 0x10000u | ((uint32_t)n << 8) | colr;
        // per-car pixel bbox via integer atomics on positive floats (on-screen boxes, offset by +1000 px)
        const float ox0 = fmaxf(x0, -900.0f) + 1000.0f, ox1 = fminf(x1, 900.0f) + 1000.0f, oy0 = fmaxf(y0, -900.0f) + 1000.0f, oy1 = fminf(y1, 900.0f) + 1000.0f;
        atomicMin((int*)&carbox[c][0], __float_as_int(ox0)); atomicMax((int*)&carbox[c][1], __float_as_int(ox1));
        atomicMin((int*)&carbox[c][2], __float_as_int(oy0)); atomicMax((int*)&carbox[c][3], __float_as_int(oy1));
      }
    }
    cinfo[k] = info;
  }
  __syncthreads();
  const int nq = nvis;

  // ---- bin: thread b owns the 8x8 bin b and collects the survivors whose quad can touch it
  if (tid < NBINS) {
    const int bx = tid % 12, by = tid / 12;
    const float X0 = (float)(bx * 8) + 0.5f, X1 = X0 + 7.0f, Y0 = (float)(by * 8) + 0.5f, Y1 = Y0 + 7.0f;   // pixel-centre box
    int cnt = 0;
    for (int s = 0; s < nq; ++s) {
      const uint32_t inf = s < VIS_LDS ? qinfo[s] : __float_as_uint(spill[(size_t)(s - VIS_LDS) * 13 + 12]);
      const int bx0 = inf >> 28, bx1 = (inf >> 24) & 15, by0 = (inf >> 20) & 15, by1 = (inf >> 16) & 15;
      if (bx < bx0 || bx > bx1 || by < by0 || by > by1) continue;
      const float* e = s < VIS_LDS ? qe[s] : spill + (size_t)(s - VIS_LDS) * 13;
      bool out = false;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float A = e[k * 3], B = e[k * 3 + 1], C = e[k * 3 + 2];
        const float m = A * (A >= 0.0f ? X1 : X0) + B * (B >= 0.0f ? Y1 : Y0) + C;     // max of the edge function over the box
        out = out || (m < 0.0f);
      }
      if (out) continue;
      if (cnt < BIN_CAP) bins[tid][cnt] = (uint16_t)s;
      ++cnt;
    }
    bcnt[tid] = cnt;
  }
  __syncthreads();

  // ---- HUD values (:634-674) in pixel units (window x*0.096, y*0.12)
  const float kx = 96.0f / 1000.0f, ky = 96.0f / 800.0f;
  const double sW = MCR_WINDOW_W / 40.0, hH = MCR_WINDOW_H / 40.0;
  float ind_x0[7], ind_x1[7], ind_y0[7], ind_y1[7];
  {
    const double speed = sqrt((double)hvx * (double)hvx + (double)hvy * (double)hvy);
    const double vals[5] = {0.02 * speed, 0.01 * p.card[(CD_OMEGA + 0) * BN + ci], 0.01 * p.card[(CD_OMEGA + 1) * BN + ci],
                            0.01 * p.card[(CD_OMEGA + 2) * BN + ci], 0.01 * p.card[(CD_OMEGA + 3) * BN + ci]};
    const double places[5] = {5, 7, 8, 9, 10};
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      ind_x0[i] = (float)((places[i] + 0) * sW) * kx; ind_x1[i] = (float)((places[i] + 1) * sW) * kx;
      const float ya = (float)(hH + hH * vals[i]) * ky, yb = (float)hH * ky;
      ind_y0[i] = fminf(ya, yb); ind_y1[i] = fmaxf(ya, yb);
    }
    const double jang = (double)(p.carf[(CF_A + 1) * BN + ci] - ha);
    const double hv[2] = {-10.0 * jang, -0.8 * (double)hw};
    const double hp[2] = {20, 30};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float xa = (float)((hp[i] + 0) * sW) * kx, xb = (float)((hp[i] + hv[i]) * sW) * kx;
      ind_x0[5 + i] = fminf(xa, xb); ind_x1[5 + i] = fmaxf(xa, xb);
      ind_y0[5 + i] = (float)(2 * hH) * ky; ind_y1[5 + i] = (float)(4 * hH) * ky;
    }
  }
  const uint32_t ind_col[7] = {PAL_WHITE, PAL_BLUE255, PAL_BLUE255, PAL_PURPLE, PAL_PURPLE, PAL_GREEN255, PAL_RED255};
  const bool show_flag = (old_flags & 1u) && p.backwards_flag;
  float fe[9];
  { const float fx[3] = {900.0f * kx, 925.0f * kx, 950.0f * kx}, fy[3] = {30.0f * ky, 70.0f * ky, 30.0f * ky}; edge_setup(fx, fy, 3, fe); }
  float hud_top = 12.0f;
#pragma unroll
  for (int i = 0; i < 7; ++i) hud_top = fmaxf(hud_top, ind_y1[i] + 1.0f);

  // ---- shade: one wave per bin, lane = pixel of the 8x8 bin
  {
    const int wave = tid >> 6, lane = tid & 63;
    const int lx = lane & 7, ly = lane >> 3;
    const float inv_z = 1.0f / fz;
    const float kgrid = (float)(MCR_PLAYFIELD / 20.0), PF = (float)MCR_PLAYFIELD;
    // world = R^T (W - t) / zoom, W = pixel centre * (1000/96, 800/96)
    const float ax = fcs * (1000.0f / 96.0f) * inv_z, bx_ = fsn * (800.0f / 96.0f) * inv_z, cx0 = -(fcs * ftx + fsn * fty) * inv_z;
    const float ay = -fsn * (1000.0f / 96.0f) * inv_z, by_ = fcs * (800.0f / 96.0f) * inv_z, cy0 = (fsn * ftx - fcs * fty) * inv_z;
    for (int b = wave; b < NBINS; b += 4) {
      const int bxi = b % 12, byi = b / 12;
      const int ix = bxi * 8 + lx, iy = byi * 8 + ly;                 // GL pixel coords (origin bottom-left)
      const float cx = (float)ix + 0.5f, cy = (float)iy + 0.5f;
      uint32_t col = PAL_BLACK;
      if (byi >= 1) {                                                   // bin row 0 (y < 8) is entirely under the HUD bar
        if (iy >= 12) {
          const float wx = ax * cx + bx_ * cy + cx0, wy = ay * cx + by_ * cy + cy0;
          if (fabsf(wx) <= PF && fabsf(wy) <= PF) {
            const int gx = (int)floorf(wx / kgrid), gy = (int)floorf(wy / kgrid);
            col = (((gx | gy) & 1) == 0) ? PAL_GRASS1 : PAL_GRASS0;
          }
        }
        // road / kerbs: highest road_poly index covering the pixel wins (painter's order)
        const int cnt = bcnt[b];
        int best = -1;
        if (cnt <= BIN_CAP) {
          for (int k = 0; k < cnt; ++k) {
            const int s = bins[b][k];
            const float* e = s < VIS_LDS ? qe[s] : spill + (size_t)(s - VIS_LDS) * 13;
            const uint32_t inf = s < VIS_LDS ? qinfo[s] : __float_as_uint(e[12]);
            const bool in = (e[0] * cx + e[1] * cy + e[2] >= 0.0f) && (e[3] * cx + e[4] * cy + e[5] >= 0.0f) &&
                            (e[6] * cx + e[7] * cy + e[8] >= 0.0f) && (e[9] * cx + e[10] * cy + e[11] >= 0.0f);
            const int key = (int)(inf & 0xffffu);
            if (in && key > best) best = key;
          }
        } else {
          for (int s = 0; s < nq; ++s) {
            const float* e = s < VIS_LDS ? qe[s] : spill + (size_t)(s - VIS_LDS) * 13;
            const uint32_t inf = s < VIS_LDS ? qinfo[s] : __float_as_uint(e[12]);
            const bool in = (e[0] * cx + e[1] * cy + e[2] >= 0.0f) && (e[3] * cx + e[4] * cy + e[5] >= 0.0f) &&
                            (e[6] * cx + e[7] * cy + e[8] >= 0.0f) && (e[9] * cx + e[10] * cy + e[11] >= 0.0f);
            const int key = (int)(inf & 0xffffu);
            if (in && key > best) best = key;
          }
        }
        if (best >= 0 && iy >= 12) {
          const uint32_t bc = (uint32_t)best & 7u;
          col = bc == MCR_COL_ROAD0 ? PAL_ROAD0 : bc == MCR_COL_ROAD1 ? PAL_ROAD1 : bc == MCR_COL_ROAD2 ? PAL_ROAD2 : bc == MCR_COL_KERB_WHITE ? PAL_WHITE : PAL_RED255;
        }
        // cars
        const float BX0 = (float)(bxi * 8) + 1000.0f, BX1 = BX0 + 8.0f, BY0 = (float)(byi * 8) + 1000.0f, BY1 = BY0 + 8.0f;
        for (int c = 0; c < N; ++c) {
          if (carbox[c][0] > BX1 || carbox[c][1] < BX0 || carbox[c][2] > BY1 || carbox[c][3] < BY0) continue;
          for (int k = c * 12; k < c * 12 + 12; ++k) {
            const uint32_t inf = cinfo[k];
            if (!inf) continue;
            if (cx < cbb[k][0] || cx > cbb[k][1] || cy < cbb[k][2] || cy > cbb[k][3]) continue;
            const int n = (int)((inf >> 8) & 0xffu);
            bool in = true;
            for (int i = 0; i < n; ++i) in = in && (ce[k][i * 3] * cx + ce[k][i * 3 + 1] * cy + ce[k][i * 3 + 2] >= 0.0f);
            if (in && iy >= 12) col = inf & 0xffu;
          }
        }
      }
      if (cy < hud_top) {
        // HUD (window space, drawn last): bar rows are already black; gauges in draw order (a tall gauge may
        // poke above the bar), then the backwards flag
#pragma unroll
        for (int i = 0; i < 7; ++i)
          if (ind_x1[i] > ind_x0[i] && ind_y1[i] > ind_y0[i] && cx >= ind_x0[i] && cx <= ind_x1[i] && cy >= ind_y0[i] && cy <= ind_y1[i]) col = ind_col[i];
        if (show_flag && (fe[0] * cx + fe[1] * cy + fe[2] >= 0.0f) && (fe[3] * cx + fe[4] * cy + fe[5] >= 0.0f) && (fe[6] * cx + fe[7] * cy + fe[8] >= 0.0f)) col = PAL_BLUE255;
      }
      fb[(95 - iy) * 96 + ix] = (uint8_t)col;                          // arr[::-1] (:602)
    }
  }
  __syncthreads();

  // ---- packed RGB write-out: 16 B per lane = 6 pixels' worth of bytes in one of three phases
  uint4* out = (uint4*)(p.obs + (size_t)vw * (96 * 96 * 3));
  for (int ch = tid; ch < 96 * 96 * 3 / 16; ch += VIEW_THREADS) {
    const int o = ch * 16; const int p0 = o / 3; const int ph = o - p0 * 3;
    uint32_t c[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) { const int pi = p0 + j; c[j] = pal[fb[pi < 96 * 96 ? pi : 96 * 96 - 1]]; }
    uint32_t w0, w1, w2, w3;
    if (ph == 0) { w0 = c[0] | (c[1] << 24); w1 = (c[1] >> 8) | (c[2] << 16); w2 = (c[2] >> 16) | (c[3] << 8); w3 = c[4] | (c[5] << 24); }
    else if (ph == 1) { w0 = (c[0] >> 8) | (c[1] << 16); w1 = (c[1] >> 16) | (c[2] << 8); w2 = c[3] | (c[4] << 24); w3 = (c[4] >> 8) | (c[5] << 16); }
    else { w0 = (c[0] >> 16) | (c[1] << 8); w1 = c[2] | (c[3] << 24); w2 = (c[3] >> 8) | (c[4] << 16); w3 = (c[4] >> 16) | (c[5] << 8); }
    out[ch] = make_uint4(w0, w1, w2, w3);
  }
}
