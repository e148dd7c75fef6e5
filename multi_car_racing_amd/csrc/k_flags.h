// k_flags.h — one wavefront per car: the backward / on-grass bookkeeping of multi_car_racing.py:446-495 on the poses
// the step just produced.  Its results (`driving_backward`, `driving_on_grass`) have no reward effect (K_BACKWARD = 0,
// :78); they reach pixels one step later (the HUD flag is drawn from LAST step's value, which k_dynamics hands to the
// raster in the view record) and the host through mcr_get_env_state.  Round 1 computed them inside the raster kernel;
// they are a per-car O(T + P) scan with f64 tails (atan2, fmod) that cost the raster registers, barriers and a serial
// lane — as a kernel of its own the scan is 8,192 independent wavefronts, launched per chain as soon as that chain's
// dynamics is done (the main envs' launch runs beside the reset pass, before the raster needs the machine).
//   nearest track point (:465-467, np.linalg.norm + argmin = first minimum): f32 distances of all tiles, wave minimum,
//     exact f64 distance for the tiles within the rounding band of that minimum, lowest index among exact ties;
//   on grass (:470-472, shapely Point.within = strict interior): f32 bbox prefilter, exact f64 test on the polygon the
//     reference builds for the tile / kerb;
//   heading vs track direction (:449-495) by one lane.
#pragma once
#include "k_raster_common.h"
#include "k_touch.h"

// (env_of_list >= 0: the env was looked up by the caller — the list rasters' bookkeeping workgroups, k_view.h — and blk is the agent)
// the touch verdict of a listed env on a wavefront of its own (the list rasters' bookkeeping workgroups: beside the env's flag scans, not behind one)
__device__ __forceinline__ void verdict_block(const McrParams& p, const int env) {
  if (p.part_next == nullptr || env >= p.env0 + p.nenv) return;
  const bool active = p.env[env].active != 0;
  const bool v = mcr_touch_verdict(p, env);
  if ((threadIdx.x & 63) == 0) mcr_set_verdict(p, env, active && v);
}
// (env_of_list >= 0 ...; verdict_elsewhere: another wavefront settles the env's touch verdict)
__device__ __forceinline__ void flags_block(const McrParams& p, const int blk, const int env_of_list = -1, const bool verdict_elsewhere = false) {
  const int lane = threadIdx.x & 63;
  const int N = p.N, BN = p.BN;
  // roles as in the other step kernels: 0 every env, 1 the main launch's envs, 2 / 3 the contact / deferred lists
  const int env = env_of_list >= 0 ? env_of_list : mcr_env_of_slot(p, blk / N);
  if (env >= p.env0 + p.nenv) return;
  const int ci = env * N + (env_of_list >= 0 ? blk : blk % N);
  const McrEnvState es = p.env[env];
  // The wavefront of an env's first car also settles the env's touch verdict for the NEXT step (k_touch.h): the poses this
  // step ended with are the ones the next contact pass sees.  Main launch: the main dynamics has written 0 for every env
  // whose hulls are far apart (nearly all) and 2 where the exact test is needed; list launches: always the exact test.
  // The mark is requested here and looked at when the scan below is done (its latency is off the wavefront's chain).
  const bool vwave = p.part_next != nullptr && !verdict_elsewhere && (env_of_list >= 0 ? blk : blk % N) == 0;
  uint32_t vmark = 0;
  if (vwave) vmark = p.role == 1 ? (uint32_t)p.part_next[env] : 2u;
  auto settle_verdict = [&]() {
    if (vwave && vmark == 2u) {
      const bool v = mcr_touch_verdict(p, env);
      if (lane == 0) mcr_set_verdict(p, env, es.active && v);
    }
  };
  if (!es.active || es.just_reset) { settle_verdict(); return; }   // reset() -> step(None) skips the block (:435); a re-spawned car keeps its zeroed flags
  const uint8_t* __restrict__ slot = p.slots + ((size_t)env * 2 + es.slot) * MCR_SLOT_BYTES;
  const McrSlotHeader* H = (const McrSlotHeader*)slot;
  const int T = H->T, P = H->P;
  const McrShapes& S = *p.shapes;
  const float hvx = p.carf[(CF_VX + 0) * BN + ci], hvy = p.carf[(CF_VY + 0) * BN + ci], ha = p.carf[(CF_A + 0) * BN + ci];
  const Xf hxf = xf_of(v2(p.carf[(CF_CX + 0) * BN + ci], p.carf[(CF_CY + 0) * BN + ci]), ha, v2(S.hull_lcx, S.hull_lcy));
  const float fpx = hxf.p.x, fpy = hxf.p.y;                  // hull.position (body origin)
  const double dpx = (double)fpx, dpy = (double)fpy;
  const float4* __restrict__ QA = (const float4*)(slot + MCR_OFF_QA); const float4* __restrict__ QB = (const float4*)(slot + MCR_OFF_QB);
  const uint32_t* __restrict__ QM = (const uint32_t*)(slot + MCR_OFF_QMETA);
  const double* __restrict__ TX = (const double*)(slot + MCR_OFF_TRACK_X); const double* __restrict__ TY = (const double*)(slot + MCR_OFF_TRACK_Y);

  // Only the road_poly blocks (runs of MCR_QBLK consecutive entries; boxes from the track generator, as in the raster)
  // that can hold the nearest track point — or the car itself — are scanned.  Per block: a lower bound of the distance to
  // anything in it (distance to the box) and, for full blocks (they hold at least 8 tiles, whose track point lies on the
  // tile's leading edge, inside the box), an upper bound of the distance to its nearest track point (distance to the
  // box's farthest corner).  A block whose lower bound exceeds the smallest upper bound cannot matter.
  unsigned long long visit;
  {
    float4 bb = make_float4(1.0f, 1.0f, -1.0f, -1.0f);
    if (lane < MCR_QUAD_CAP / MCR_QBLK) bb = ((const float4*)(slot + MCR_OFF_QBLK))[lane];
    const bool nonempty = bb.x <= bb.z;
    const float lx = fmaxf(fmaxf(bb.x - fpx, fpx - bb.z), 0.0f), ly = fmaxf(fmaxf(bb.y - fpy, fpy - bb.w), 0.0f);
    const float hx = fmaxf(fabsf(fpx - bb.x), fabsf(fpx - bb.z)), hy = fmaxf(fabsf(fpy - bb.y), fabsf(fpy - bb.w));
    const float lb2 = lx * lx + ly * ly;
    float ub2 = (nonempty && (lane + 1) * MCR_QBLK <= P) ? hx * hx + hy * hy : MCR_MAXFLT;
    for (int o = 32; o > 0; o >>= 1) ub2 = fminf(ub2, __shfl_xor(ub2, o));
    const float reach = ub2 < 1e30f ? sqrtf(ub2) * (1.0f + 1e-5f) + 0.01f : 1e18f;
    visit = __ballot(nonempty && lb2 <= reach * reach);
  }
  // the visited blocks' entries, four blocks at a time: lane -> entry q (or -1)
  auto next_entries = [&](unsigned long long& m) -> int {
    int b[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) { b[g] = m ? (int)__builtin_ctzll(m) : -1; m &= m - 1ull; }
    const int g = lane >> 4;
    const int blk_ = g == 0 ? b[0] : g == 1 ? b[1] : g == 2 ? b[2] : b[3];
    const int q = blk_ * MCR_QBLK + (lane & (MCR_QBLK - 1));
    return (blk_ >= 0 && q < P) ? q : -1;
  };
  // pass 1: on-grass + f32 distance to every tile's track point (~ midpoint of the tile's leading edge)
  bool inside = false;
  float dmin = MCR_MAXFLT;
  for (unsigned long long m = visit; m;) {
    const int q = next_entries(m);
    if (q < 0) continue;
    const float4 a = QA[q], b = QB[q];
    const uint32_t meta = QM[q];
    const uint32_t tile1 = (meta >> 8) & 0x3ffu, owner1 = meta >> 18;
    const float bx0 = fminf(fminf(a.x, a.z), fminf(b.x, b.z)) - 0.02f, bx1 = fmaxf(fmaxf(a.x, a.z), fmaxf(b.x, b.z)) + 0.02f;
    const float by0 = fminf(fminf(a.y, a.w), fminf(b.y, b.w)) - 0.02f, by1 = fmaxf(fmaxf(a.y, a.w), fmaxf(b.y, b.w)) + 0.02f;
    if (!inside && fpx >= bx0 && fpx <= bx1 && fpy >= by0 && fpy <= by1)
      inside = point_in_road_poly_f64(slot, (int)(tile1 ? tile1 : owner1) - 1, T, tile1 == 0, dpx, dpy);
    if (tile1) {
      const float mx = 0.5f * (a.x + a.z), my = 0.5f * (a.y + a.w);
      const float ddx = fpx - mx, ddy = fpy - my;
      dmin = fminf(dmin, ddx * ddx + ddy * ddy);
    }
  }
  for (int o = 32; o > 0; o >>= 1) dmin = fminf(dmin, __shfl_xor(dmin, o));
  const bool any_inside = __any(inside) != 0;
  // pass 2: exact distance (f64, as np.linalg.norm evaluates it) of the tiles within the f32 error band of the minimum
  const float band = sqrtf(dmin) * (1.0f + 1e-5f) + 2e-3f;
  const float thr = band * band;
  double bd = 1e300; int bi = 0x7fffffff;
  for (unsigned long long m = visit; m;) {
    const int q = next_entries(m);
    if (q < 0) continue;
    const uint32_t tile1 = (QM[q] >> 8) & 0x3ffu;
    if (!tile1) continue;
    const float4 a = QA[q];
    const float mx = 0.5f * (a.x + a.z), my = 0.5f * (a.y + a.w);
    const float ddx = fpx - mx, ddy = fpy - my;
    if (ddx * ddx + ddy * ddy <= thr) {
      const int t = (int)tile1 - 1;
      const double dx = dpx - TX[t], dy = dpy - TY[t];
      const double d = sqrt(dx * dx + dy * dy);
      if (d < bd || (d == bd && t < bi)) { bd = d; bi = t; }
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const double od = __shfl_xor(bd, o); const int oi = __shfl_xor(bi, o);
    if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
  }
  settle_verdict();
  if (lane != 0) return;
  if (bi == 0x7fffffff) {                                      // cannot happen with a valid track; exact full scan keeps the result defined
    bd = 1e300; bi = 0;
    for (int t = 0; t < T; ++t) { const double dx = dpx - TX[t], dy = dpy - TY[t]; const double d = sqrt(dx * dx + dy * dy); if (d < bd) { bd = d; bi = t; } }
  }
  const double* TB = (const double*)(slot + MCR_OFF_TRACK_B);
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

// one wavefront per car (the list launches of roles >= 2 call flags_block from k_list_chain.h)
// 8192 wavefronts = one round on 1024 SIMDs at 8 wavefronts each: the kernel must stay within 64 VGPRs
#ifndef MCR_DEVICE_FUNCTIONS_ONLY
__global__ __launch_bounds__(64, 8) void k_flags(McrParams p) { flags_block(p, (int)blockIdx.x); }
#endif
