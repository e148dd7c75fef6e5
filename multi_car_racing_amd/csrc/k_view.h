// k_view.h — one 256-thread workgroup per agent view (env, agent): the 96x96 ego-frame software
// rasteriser that replaces pyglet/OpenGL (multi_car_racing.py:511-604, 613-674; gym Car.draw), plus the
// per-agent backward/on-grass bookkeeping of :446-495 (its result only reaches pixels one step later).
//
// Pipeline inside the workgroup (all in LDS, one HBM pass in, one out):
//   1. camera: zoom/rotation/translation of :540-556 -> a 2x3 world->pixel matrix (f32, like GL);
//   2. cull+setup: threads stride over the env's road_poly quads (2 x float4 + meta per quad, coalesced),
//      transform the 4 vertices, reject quads whose pixel bbox misses the visible rows, and append
//      oriented edge equations of the survivors to an LDS list;  car polygons (12 per car) likewise;
//   3. shade: every thread owns pixels (tid + 256k); background (playfield + checker squares) is evaluated
//      analytically in world space, road quads by max-index-wins over the LDS list (== painter's order),
//      then cars, then the HUD bar/indicators in window space;
//   4. write-out: packed RGB rows are emitted as 16-byte-per-lane coalesced stores from the LDS framebuffer.
// Sampling rule: pixel centres, a pixel belongs to a polygon iff all oriented edge functions are >= 0.
#pragma once
#include "mcr_kernels.h"

namespace view {

#define VIEW_THREADS 256
#define VIS_CAP MCR_QUAD_CAP
#define CARPOLY_CAP (MCR_MAX_AGENTS * 12)

__device__ __forceinline__ uint32_t rgb(uint32_t r, uint32_t g, uint32_t b) { return r | (g << 8) | (b << 16); }
// GL float colour -> unorm8: round-to-nearest of c*255 evaluated on the f32 value
__device__ __forceinline__ uint32_t c8(double c) { return (uint32_t)floor((double)(float)c * 255.0 + 0.5); }

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

}  // namespace view

// flags_mode: 1 = evaluate the backward/on-grass block (:446-495) for this agent after drawing.
__global__ __launch_bounds__(VIEW_THREADS) void k_view(McrParams p, int flags_mode, int only_just_reset) {
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

  __shared__ uint32_t fb[96 * 96];
  __shared__ float qe[VIS_CAP][12];
  __shared__ uint32_t qinfo[VIS_CAP];        // (quad index << 8) | colour id   (index order == painter's order)
  __shared__ float ce[CARPOLY_CAP][24];      // up to 8 edges
  __shared__ float cbb[CARPOLY_CAP][4];
  __shared__ uint32_t cinfo[CARPOLY_CAP];    // (nedges << 24) | rgb ; 0 = skip
  __shared__ int nvis;
  __shared__ double red_d[VIEW_THREADS]; __shared__ int red_i[VIEW_THREADS];
  __shared__ int any_inside;

  if (tid == 0) { nvis = 0; any_inside = 0; }
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

  const bool draw = p.obs != nullptr;
  if (draw) {
    // ---- road quads: cull + edge setup
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
      if (x1 < 0.0f || x0 > 96.0f || y1 < 12.0f || y0 > 96.0f) continue;     // rows < 12 are under the HUD bar
      float e[12];
      if (!edge_setup(px, py, 4, e)) continue;
      const uint32_t meta = QM[q];
      uint32_t col = meta & 0xffu; const uint32_t tile1 = meta >> 8;
      if (tile1 && (tflags[tile1 - 1] & 0x100u)) col = MCR_COL_ROAD0;           // touched tile -> ROAD_COLOR (:102-104)
      const int s = atomicAdd(&nvis, 1);
#pragma unroll
      for (int i = 0; i < 12; ++i) qe[s][i] = e[i];
      qinfo[s] = ((uint32_t)q << 8) | col;
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
        if ((j & 1) == 0) { n = S.wheel.n; for (int i = 0; i < n; ++i) { lx[i] = S.wheel.vx[i]; ly[i] = S.wheel.vy[i]; } colr = rgb(0, 0, 0); }
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
            colr = rgb(c8(0.3), c8(0.3), c8(0.3));
          }
        }
      } else {
        const int hk = j - 8;
        const V2 cc = v2(p.carf[(CF_CX + 0) * BN + cj], p.carf[(CF_CY + 0) * BN + cj]);
        xf = xf_of(cc, p.carf[(CF_A + 0) * BN + cj], v2(S.hull_lcx, S.hull_lcy));
        n = S.hull[hk].n; for (int i = 0; i < n; ++i) { lx[i] = S.hull[hk].vx[i]; ly[i] = S.hull[hk].vy[i]; }
        // CAR_COLORS[c % 8] (:67-70, :402); use_ego_color (:560-563)
        const int cc8 = c & 7;
        uint32_t r = (cc8 == 0 || cc8 == 4 || cc8 == 6 || cc8 == 7) ? c8(0.8) : 0;
        uint32_t g = (cc8 == 2 || cc8 == 3 || cc8 == 4 || cc8 == 7) ? c8(0.8) : 0;
        uint32_t b = (cc8 == 1 || cc8 == 3 || cc8 == 4 || cc8 == 6) ? c8(0.8) : 0;
        if (p.use_ego_color) { r = (c == agent) ? c8(0.8) : 0; g = 0; b = (c == agent) ? 0 : c8(0.8); }
        colr = rgb(r, g, b);
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
        if (!(x1 < 0.0f || x0 > 96.0f || y1 < 12.0f || y0 > 96.0f) && edge_setup(px, py, n, e)) {
          for (int i = 0; i < n * 3; ++i) ce[k][i] = e[i];
          cbb[k][0] = x0; cbb[k][1] = x1; cbb[k][2] = y0; cbb[k][3] = y1;
          info = ((uint32_t)n << 24) | colr;
        }
      }
      cinfo[k] = info;
    }
    __syncthreads();

    // ---- HUD values (:634-674) in pixel units (window x*0.096, y*0.12)
    const float kx = 96.0f / 1000.0f, ky = 96.0f / 800.0f;
    const double sW = MCR_WINDOW_W / 40.0, hH = MCR_WINDOW_H / 40.0;
    float ind_x0[7], ind_x1[7], ind_y0[7], ind_y1[7]; uint32_t ind_col[7];
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
      ind_col[0] = rgb(255, 255, 255); ind_col[1] = ind_col[2] = rgb(0, 0, 255); ind_col[3] = ind_col[4] = rgb(c8(0.2), 0, 255);
      const double jang = (double)(p.carf[(CF_A + 1) * BN + ci] - ha);
      const double hv[2] = {-10.0 * jang, -0.8 * (double)hw};
      const double hp[2] = {20, 30};
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float xa = (float)((hp[i] + 0) * sW) * kx, xb = (float)((hp[i] + hv[i]) * sW) * kx;
        ind_x0[5 + i] = fminf(xa, xb); ind_x1[5 + i] = fmaxf(xa, xb);
        ind_y0[5 + i] = (float)(2 * hH) * ky; ind_y1[5 + i] = (float)(4 * hH) * ky;
      }
      ind_col[5] = rgb(0, 255, 0); ind_col[6] = rgb(255, 0, 0);
    }
    const bool show_flag = (old_flags & 1u) && p.backwards_flag;
    float fe[9];
    { const float fx[3] = {900.0f * kx, 925.0f * kx, 950.0f * kx}, fy[3] = {30.0f * ky, 70.0f * ky, 30.0f * ky}; edge_setup(fx, fy, 3, fe); }

    float hud_top = 12.0f;
#pragma unroll
    for (int i = 0; i < 7; ++i) hud_top = fmaxf(hud_top, ind_y1[i] + 1.0f);
    const int nq = nvis;
    const uint32_t GRASS0 = rgb(c8(0.4), c8(0.8), c8(0.4)), GRASS1 = rgb(c8(0.4), c8(0.9), c8(0.4));
    const float inv_z = 1.0f / fz;
    const float kgrid = (float)(MCR_PLAYFIELD / 20.0), PF = (float)MCR_PLAYFIELD;
    for (int pix = tid; pix < 96 * 96; pix += VIEW_THREADS) {
      const int iy = pix / 96, ix = pix - iy * 96;       // GL pixel coords (origin bottom-left)
      const float cx = (float)ix + 0.5f, cy = (float)iy + 0.5f;
      uint32_t col = 0;
      if (iy >= 12) {
        // background in world space: playfield quad + 20x20 lighter squares (:615-627)
        const float Wx = cx * (1000.0f / 96.0f) - ftx, Wy = cy * (800.0f / 96.0f) - fty;
        const float wx = (fcs * Wx + fsn * Wy) * inv_z, wy = (-fsn * Wx + fcs * Wy) * inv_z;
        if (fabsf(wx) <= PF && fabsf(wy) <= PF) {
          const int gx = (int)floorf(wx / kgrid), gy = (int)floorf(wy / kgrid);
          col = (((gx | gy) & 1) == 0) ? GRASS1 : GRASS0;
        }
        // road / kerbs: highest road_poly index covering the pixel wins (painter's order)
        int best = -1; uint32_t bcol = 0;
        for (int s = 0; s < nq; ++s) {
          const float* e = qe[s];
          const bool in = (e[0] * cx + e[1] * cy + e[2] >= 0.0f) && (e[3] * cx + e[4] * cy + e[5] >= 0.0f) &&
                          (e[6] * cx + e[7] * cy + e[8] >= 0.0f) && (e[9] * cx + e[10] * cy + e[11] >= 0.0f);
          const uint32_t inf = qinfo[s];
          if (in && (int)(inf >> 8) > best) { best = (int)(inf >> 8); bcol = inf & 0xffu; }
        }
        if (best >= 0) {
          const uint32_t g0 = c8(0.4), g1 = c8(0.4 + 0.01), g2 = c8(0.4 + 0.01 * 2);
          col = bcol == MCR_COL_ROAD0 ? rgb(g0, g0, g0) : bcol == MCR_COL_ROAD1 ? rgb(g1, g1, g1) : bcol == MCR_COL_ROAD2 ? rgb(g2, g2, g2)
                : bcol == MCR_COL_KERB_WHITE ? rgb(255, 255, 255) : rgb(255, 0, 0);
        }
        // cars
        for (int k = 0; k < N * 12; ++k) {
          const uint32_t inf = cinfo[k];
          if (!inf) continue;
          if (cx < cbb[k][0] || cx > cbb[k][1] || cy < cbb[k][2] || cy > cbb[k][3]) continue;
          const int n = (int)(inf >> 24);
          bool in = true;
          for (int i = 0; i < n; ++i) in = in && (ce[k][i * 3] * cx + ce[k][i * 3 + 1] * cy + ce[k][i * 3 + 2] >= 0.0f);
          if (in) col = inf & 0xffffffu;
        }
      }
      if (cy < hud_top) {
        // HUD (window space, drawn last): black bar rows are already 0; indicators in draw order (a tall gauge
        // may poke above the bar), then the backwards flag
#pragma unroll
        for (int i = 0; i < 7; ++i)
          if (ind_x1[i] > ind_x0[i] && ind_y1[i] > ind_y0[i] && cx >= ind_x0[i] && cx <= ind_x1[i] && cy >= ind_y0[i] && cy <= ind_y1[i]) col = ind_col[i];
        if (show_flag && (fe[0] * cx + fe[1] * cy + fe[2] >= 0.0f) && (fe[3] * cx + fe[4] * cy + fe[5] >= 0.0f) && (fe[6] * cx + fe[7] * cy + fe[8] >= 0.0f)) col = rgb(0, 0, 255);
      }
      fb[(95 - iy) * 96 + ix] = col;                      // arr[::-1] (:602)
    }
    __syncthreads();
    // ---- packed RGB write-out, 16 B per lane, fully coalesced
    uint4* out = (uint4*)(p.obs + (size_t)vw * (96 * 96 * 3));
    for (int ch = tid; ch < 96 * 96 * 3 / 16; ch += VIEW_THREADS) {
      const int o = ch * 16;
      uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int bi = o + j; const int px = bi / 3; const int c = bi - px * 3;
        w[j >> 2] |= ((fb[px] >> (8 * c)) & 0xffu) << (8 * (j & 3));
      }
      out[ch] = make_uint4(w[0], w[1], w[2], w[3]);
    }
  }

  // ---- backward / on-grass flags (:446-495), visible one step later through the HUD flag
  if (flags_mode) {
    const double px = (double)hxf.p.x, py = (double)hxf.p.y;
    const double* TX = (const double*)(slot + MCR_OFF_TRACK_X); const double* TY = (const double*)(slot + MCR_OFF_TRACK_Y); const double* TB = (const double*)(slot + MCR_OFF_TRACK_B);
    double bd = 1e300; int bi = 0x7fffffff;
    for (int t = tid; t < T; t += VIEW_THREADS) {
      const double dx = px - TX[t], dy = py - TY[t];
      const double d = sqrt(dx * dx + dy * dy);
      if (d < bd) { bd = d; bi = t; }
    }
    red_d[tid] = bd; red_i[tid] = bi;
    // strict-interior point-in-quad over all road_poly (shapely `within`), on the f64 polygons the reference
    // hands to shapely: tile quad (:313-317) and kerb quad (:329-333) rebuilt from the host's f64 cos/sin(beta)
    const double* TC = (const double*)(slot + MCR_OFF_TRACK_C); const double* TS = (const double*)(slot + MCR_OFF_TRACK_S);
    const uint32_t* TCNT = (const uint32_t*)(slot + MCR_OFF_TCNT);
    const double TW = 40 / MCR_SCALE, TBW = 8 / MCR_SCALE;
    bool inside = false;
    for (int t = tid; t < T; t += VIEW_THREADS) {
      const int u = t == 0 ? T - 1 : t - 1;
      const double x1 = TX[t], y1 = TY[t], c1 = TC[t], s1 = TS[t], x2 = TX[u], y2 = TY[u], c2 = TC[u], s2 = TS[u];
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
    __syncthreads();
    for (int s = VIEW_THREADS / 2; s > 0; s >>= 1) {
      if (tid < s) {
        const double d2 = red_d[tid + s]; const int i2 = red_i[tid + s];
        if (d2 < red_d[tid] || (d2 == red_d[tid] && i2 < red_i[tid])) { red_d[tid] = d2; red_i[tid] = i2; }
      }
      __syncthreads();
    }
    if (tid == 0) {
      const double TWO_PI = 2 * 3.141592653589793;
      const double PI = 3.141592653589793;
      double car_angle;
      const double vx = (double)hvx, vy = (double)hvy;
      if (sqrt(vx * vx + vy * vy) > 0.5) car_angle = -atan2(vx, vy); else car_angle = (double)ha;
      car_angle = fmod(car_angle + TWO_PI, TWO_PI); if (car_angle < 0) car_angle += TWO_PI;
      double desired = TB[red_i[0]];
      if (H->cw) desired += PI;
      desired = fmod(desired + TWO_PI, TWO_PI); if (desired < 0) desired += TWO_PI;
      double diff = fabs(desired - car_angle);
      if (diff > PI) diff = fabs(diff - TWO_PI);
      uint32_t f = 0;
      if (diff > PI / 2) f |= 1u;
      if (!any_inside) f |= 2u;
      p.caru[CU_FLAGS * BN + ci] = f;
    }
  }
}
