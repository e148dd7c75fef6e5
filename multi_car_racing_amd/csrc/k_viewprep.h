// k_viewprep.h — what the rasteriser needs from a car's pose — the per-car view record (camera 2x3 + inverse, HUD rectangles,
// grass range; multi_car_racing.py:540-556, 634-674, 615-627) and the world-space vertices of the 12 Car.draw polygons — as a
// kernel of its own in the three-chain step instead of the last 12 us of the step's longest kernel (k_dynamics' epilogue: one
// lane per car, eight f64 sin/cos): it runs on the third stream in front of the main envs' bookkeeping kernel and raster, whose
// chain has slack (the step's critical path after the dynamics is the resume chain, on the caller's stream).  One lane per car like the epilogue (a wavefront per car was tried: 8192 wavefronts each issuing the whole f64 camera
// code for one car cost k_flags 20 us).  Same expressions, operand order and types as the epilogue of dynamics_block (the
// list chains, the single-stream step and the reset pass still use that one): identical records.  VP_SCORE and VP_OLDFLAGS
// stay with k_dynamics (it holds the values).
#pragma once
#include "mcr_kernels.h"

__device__ __forceinline__ double vprep_sign(double x) { return x > 0.0 ? 1.0 : (x < 0.0 ? -1.0 : 0.0); }

// (carf / card: per-car SoA fields at stride BN — the live state, or the state the cars of a terminal entry ended their episode with)
// `parts`: bit 0 camera + HUD rectangles + grass range + the hull's four polygons, bits 1..4 wheel 0..3 (box + stripe) — the list chains hand the
// parts of a car to five lanes (viewprep_list_block), everybody else computes all of them on the car's lane
__device__ __forceinline__ void viewprep_car(const McrShapes& S, const float* __restrict__ carf, const double* __restrict__ card, const int BN, const int ci,
                                             float* __restrict__ viewp, float* __restrict__ carpoly, const double h_ratio, const double t_now, const uint32_t parts = 31u) {
  struct { const float* carf; const double* card; double h_ratio; } p = {carf, card, h_ratio};
  float* vp = viewp + (size_t)ci * MCR_VIEWP_FLOATS;
  float4* cp4 = (float4*)(carpoly + (size_t)ci * MCR_CARPOLY_FLOATS);
  int* cnt = (int*)(carpoly + (size_t)ci * MCR_CARPOLY_FLOATS + MCR_CARPOLY_NOFF);
  if (parts & 1u) {
    const float hcx = p.carf[(CF_CX + 0) * BN + ci], hcy = p.carf[(CF_CY + 0) * BN + ci], ha = p.carf[(CF_A + 0) * BN + ci];
    const float hvx = p.carf[(CF_VX + 0) * BN + ci], hvy = p.carf[(CF_VY + 0) * BN + ci];
    const Xf hxf = xf_of(v2(hcx, hcy), ha, v2(S.hull_lcx, S.hull_lcy));
    const double zoom = 0.1 * MCR_SCALE * fmax(1 - t_now, 0.0) + MCR_ZOOM * MCR_SCALE * fmin(t_now, 1.0);
    const double sx = (double)hxf.p.x, sy = (double)hxf.p.y;
    double angle = -(double)ha;
    const double vx = (double)hvx, vy = (double)hvy;
    const double speed = sqrt(vx * vx + vy * vy);
    if (speed > 0.5) angle = atan2(vx, vy);
    double sin_a, cos_a; mcr_sincos_core(angle, &sin_a, &cos_a);
    const double ttx = MCR_WINDOW_W / 2 - (sx * zoom * cos_a - sy * zoom * sin_a);
    const double tty = MCR_WINDOW_H * p.h_ratio - (sx * zoom * sin_a + sy * zoom * cos_a);
    const float ftx = (float)ttx, fty = (float)tty, fz = (float)zoom;
    const float fdeg = (float)(57.29577951308232 * angle);
    const double rad = (double)fdeg * (3.14159265358979323846 / 180.0);
    double sin_r, cos_r; mcr_sincos_core(rad, &sin_r, &cos_r);
    const float fcs = (float)cos_r, fsn = (float)sin_r;
    const float kx = 96.0f / 1000.0f, ky = 96.0f / 800.0f;
    vp[VP_CAM + 0] = fcs * fz * kx; vp[VP_CAM + 1] = -fsn * fz * kx; vp[VP_CAM + 2] = fsn * fz * ky; vp[VP_CAM + 3] = fcs * fz * ky;
    vp[VP_CAM + 4] = ftx * kx; vp[VP_CAM + 5] = fty * ky;
    const float inv_z = 1.0f / fz;
    const float i0 = fcs * (1000.0f / 96.0f) * inv_z, i1 = fsn * (800.0f / 96.0f) * inv_z, i2 = -(fcs * ftx + fsn * fty) * inv_z;
    const float i3 = -fsn * (1000.0f / 96.0f) * inv_z, i4 = fcs * (800.0f / 96.0f) * inv_z, i5 = (fsn * ftx - fcs * fty) * inv_z;
    vp[VP_INV + 0] = i0; vp[VP_INV + 1] = i1; vp[VP_INV + 2] = i2; vp[VP_INV + 3] = i3; vp[VP_INV + 4] = i4; vp[VP_INV + 5] = i5;
    {
      const float hk = 0.5f / (float)(MCR_PLAYFIELD / 20.0);
      const float aU = i0 * hk, bU = i1 * hk, cU = i2 * hk;
      const float aV = i3 * hk, bV = i4 * hk, cV = i5 * hk;
      float umin = MCR_MAXFLT, umax = -MCR_MAXFLT, vmin = MCR_MAXFLT, vmax = -MCR_MAXFLT;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float X = (k & 1) ? 96.0f : 0.0f, Y = (k & 2) ? 96.0f : 12.0f;
        const float u = aU * X + bU * Y + cU, v = aV * X + bV * Y + cV;
        umin = fminf(umin, u); umax = fmaxf(umax, u); vmin = fminf(vmin, v); vmax = fmaxf(vmax, v);
      }
      const float mu = fabsf(aU) + fabsf(bU) + 1e-3f, mv = fabsf(aV) + fabsf(bV) + 1e-3f;
      const bool inside_field = umin - mu >= -10.0f && umax + mu <= 10.0f && vmin - mv >= -10.0f && vmax + mv <= 10.0f;
      int a0 = (int)ceilf(umin - mu - 0.5f), a1 = (int)floorf(umax + mu), b0 = (int)ceilf(vmin - mv - 0.5f), b1 = (int)floorf(vmax + mv);
      a0 = max(a0, -10); a1 = min(a1, 9); b0 = max(b0, -10); b1 = min(b1, 9);
      vp[VP_GRASS + 0] = __int_as_float(a0); vp[VP_GRASS + 1] = __int_as_float(max(a1 - a0 + 1, 0));
      vp[VP_GRASS + 2] = __int_as_float(b0); vp[VP_GRASS + 3] = __int_as_float(max(b1 - b0 + 1, 0));
      vp[VP_GRASS + 4] = __int_as_float(inside_field ? 1 : 0);
    }
  }
  if (parts & 1u) {
    const float hvx = p.carf[(CF_VX + 0) * BN + ci], hvy = p.carf[(CF_VY + 0) * BN + ci];
    const float ha = p.carf[(CF_A + 0) * BN + ci], w1a = p.carf[(CF_A + 1) * BN + ci], hw = p.carf[(CF_W + 0) * BN + ci];
    const double vx = (double)hvx, vy = (double)hvy;
    const double speed = sqrt(vx * vx + vy * vy);
    const float kx = 96.0f / 1000.0f, ky = 96.0f / 800.0f;
    const double sW = MCR_WINDOW_W / 40.0, hH = MCR_WINDOW_H / 40.0;
    const double vals[5] = {0.02 * speed, 0.01 * p.card[(CD_OMEGA + 0) * BN + ci], 0.01 * p.card[(CD_OMEGA + 1) * BN + ci],
                            0.01 * p.card[(CD_OMEGA + 2) * BN + ci], 0.01 * p.card[(CD_OMEGA + 3) * BN + ci]};
    const double places[5] = {5, 7, 8, 9, 10};
    float hud_top = 12.0f;
#pragma unroll
    for (int i = 0; i < 5; ++i) {                                   // vertical_ind (:643-648)
      const float ya = (float)(hH + hH * vals[i]) * ky, yb = (float)hH * ky;
      vp[VP_IND + i * 4 + 0] = (float)((places[i] + 0) * sW) * kx; vp[VP_IND + i * 4 + 1] = (float)((places[i] + 1) * sW) * kx;
      vp[VP_IND + i * 4 + 2] = fminf(ya, yb); vp[VP_IND + i * 4 + 3] = fmaxf(ya, yb);
      hud_top = fmaxf(hud_top, fmaxf(ya, yb) + 1.0f);
    }
    const double jang = (double)(w1a - ha);
    const double hv[2] = {-10.0 * jang, -0.8 * (double)hw};
    const double hp[2] = {20, 30};
#pragma unroll
    for (int i = 0; i < 2; ++i) {                                   // horiz_ind (:649-654)
      const float xa = (float)((hp[i] + 0) * sW) * kx, xb = (float)((hp[i] + hv[i]) * sW) * kx;
      vp[VP_IND + (5 + i) * 4 + 0] = fminf(xa, xb); vp[VP_IND + (5 + i) * 4 + 1] = fmaxf(xa, xb);
      vp[VP_IND + (5 + i) * 4 + 2] = (float)(2 * hH) * ky; vp[VP_IND + (5 + i) * 4 + 3] = (float)(4 * hH) * ky;
    }
    vp[VP_HUDTOP] = hud_top;
  }
  for (int k = 0; k < 4; ++k) {
    if (!((parts >> (1 + k)) & 1u)) continue;
    const Xf wxf = xf_of(v2(p.carf[(CF_CX + 1 + k) * BN + ci], p.carf[(CF_CY + 1 + k) * BN + ci]), p.carf[(CF_A + 1 + k) * BN + ci], v2(0.0f, 0.0f));
    V2 w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = xmul(wxf, v2(S.wheel.vx[i], S.wheel.vy[i]));
    float4* box = cp4 + (2 * k) * 4;
    box[0] = make_float4(w[0].x, w[0].y, w[1].x, w[1].y); box[1] = make_float4(w[2].x, w[2].y, w[3].x, w[3].y);
    box[2] = make_float4(w[3].x, w[3].y, w[3].x, w[3].y); box[3] = box[2];
    cnt[2 * k] = S.wheel.n;
    const double ph = p.card[(CD_PHASE + k) * BN + ci];
    const double a1 = ph, a2 = ph + 1.2;
    double s1, s2, c1, c2; mcr_sincos_core(a1, &s1, &c1); mcr_sincos_core(a2, &s2, &c2);
    int ns = 0;
    if (!(s1 > 0 && s2 > 0)) {
      if (s1 > 0) c1 = vprep_sign(c1);
      if (s2 > 0) c2 = vprep_sign(c2);
      ns = 4;
      const float lx[4] = {(float)(-MCR_WHEEL_W * MCR_SIZE), (float)(+MCR_WHEEL_W * MCR_SIZE), (float)(+MCR_WHEEL_W * MCR_SIZE), (float)(-MCR_WHEEL_W * MCR_SIZE)};
      const float ly[4] = {(float)(+MCR_WHEEL_R * c1 * MCR_SIZE), (float)(+MCR_WHEEL_R * c1 * MCR_SIZE), (float)(+MCR_WHEEL_R * c2 * MCR_SIZE), (float)(+MCR_WHEEL_R * c2 * MCR_SIZE)};
      V2 u[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) u[i] = xmul(wxf, v2(lx[i], ly[i]));
      float4* stripe = cp4 + (2 * k + 1) * 4;
      stripe[0] = make_float4(u[0].x, u[0].y, u[1].x, u[1].y); stripe[1] = make_float4(u[2].x, u[2].y, u[3].x, u[3].y);
      stripe[2] = make_float4(u[3].x, u[3].y, u[3].x, u[3].y); stripe[3] = stripe[2];
    }
    cnt[2 * k + 1] = ns;
  }
  if (parts & 1u)
  for (int k = 0; k < 4; ++k) {
    const Xf hxf = xf_of(v2(p.carf[(CF_CX + 0) * BN + ci], p.carf[(CF_CY + 0) * BN + ci]), p.carf[(CF_A + 0) * BN + ci], v2(S.hull_lcx, S.hull_lcy));
    const McrPoly& P = S.hull[k];
    const int n = P.n;
    V2 w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { if (i < n) w[i] = xmul(hxf, v2(P.vx[i], P.vy[i])); else w[i] = w[i - 1 < 0 ? 0 : i - 1]; }
    float4* hp = cp4 + (8 + k) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) hp[i] = make_float4(w[2 * i].x, w[2 * i].y, w[2 * i + 1].x, w[2 * i + 1].y);
    cnt[8 + k] = n;
  }
}

// one lane per car of the main launch's envs (role 1 semantics: not the contact chain's, not the deferred, not the re-spawned ones)
__device__ __forceinline__ void viewprep_block(const McrParams& p, const int blk) {
  const int g = blk * 64 + threadIdx.x;
  const int env = mcr_env_of_slot(p, g / p.G), agent = g % p.G;
  if (env >= p.env0 + p.nenv || agent >= p.N) return;
  const McrEnvState es = p.env[env];
  if (!es.active || es.just_reset) return;
  viewprep_car(*p.shapes, p.carf, p.card, p.BN, env * p.N + agent, p.viewp, p.carpoly, p.h_ratio, es.t);
}
// The view records and car polygons of a list chain's envs (roles 2 / 3, k_list_chain.h), right behind their dynamics: the chain is the
// step's critical path and its wavefront holds a handful of cars — a car's record is ten f64 sincos on ONE lane (13 us) when the car's lane
// computes it; here FIVE lanes share it (camera + HUD + hull | wheel 0..3), reading the state the car lanes have just written back.  The same
// code as everywhere (viewprep_car): identical records.  VP_SCORE / VP_OLDFLAGS are written by the dynamics (it holds the values).
__device__ __forceinline__ void viewprep_list_block(const McrParams& p, const int blk) {
  if (p.obs == nullptr) return;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");       // (same wavefront, same L1: the write-back above is complete and visible to the other lanes)
  const int lane = threadIdx.x & 63, ncars = p.list_envs_per_block * p.N;
  for (int base = 0; base < ncars * 8; base += 64) {
    const int w = base + lane, c = w >> 3, part = w & 7;
    if (c >= ncars || part >= 5) continue;
    const int env = mcr_env_of_slot(p, blk * p.list_envs_per_block + c / p.N);
    if (env >= p.env0 + p.nenv) continue;
    const McrEnvState es = p.env[env];
    if (!es.active || es.resetting) continue;                   // (an env that ended its episode here: the reset pass draws up its first record)
    viewprep_car(*p.shapes, p.carf, p.card, p.BN, env * p.N + c % p.N, p.viewp, p.carpoly, p.h_ratio, es.t, 1u << part);
  }
}
// Terminal entry of env `env` (if this step's dynamics made one): view records and car polygons from the state its cars ended the episode
// with (lanes 0 .. N-1), and the tiles' recolour flags before the reset pass clears them.  Called by the env's reset pass, one wavefront.
__device__ __forceinline__ void term_prepare(const McrParams& p, const int env) {
  if (p.term_idx == nullptr || env >= p.env0 + p.nenv) return;
  const McrEnvState es = p.env[env];
  if (!es.active || !es.resetting) return;
  const int tidx = p.term_idx[env];
  if (tidx < 0) return;
  const int lane = threadIdx.x & 63;
  const uint32_t* __restrict__ src = (const uint32_t*)(p.tile_flags + (size_t)env * MCR_TILE_CAP);
  uint32_t* __restrict__ dst = (uint32_t*)(p.term_tflags + (size_t)tidx * MCR_TILE_CAP);
  for (int i = lane; i < MCR_TILE_CAP / 2; i += 64) dst[i] = src[i];
  if (lane < p.N) viewprep_car(*p.shapes, p.term_carf, p.term_card, p.term_cap * p.N, tidx * p.N + lane, p.term_viewp, p.term_carpoly, p.h_ratio, p.term_env[tidx].t);
}
__global__ __launch_bounds__(64) void k_viewprep(McrParams p) { viewprep_block(p, (int)blockIdx.x); }
