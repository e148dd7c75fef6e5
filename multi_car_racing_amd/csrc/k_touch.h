// k_touch.h — "does any car<->car fixture pair of this env touch?" at the poses in HBM, i.e. whether the contact pass
// (k_collide.h, b2Contact::Update for the dynamic pairs) is going to find a manifold there: the same fixture
// transforms, the same boxes (+-0.05), the same pair filter (wheel<->wheel never collide) and the same narrowphase
// (cc::collide_polygons), so the two always agree (k_collide counts disagreements in counters[4]).
// Why: in the three-chain step the main dynamics launch runs BESIDE the contact pass; which envs it must leave to the
// contact chain has to be known before either starts.  The verdict only needs the poses a step ends with, so the
// bookkeeping kernels of step t (k_flags.h; the list chains) evaluate it for step t+1.
#pragma once
#include "k_carcontacts.h"
#include "k_world.h"

// whole wavefront (64 lanes, lane = car * 8 + fixture as in k_collide); returns the wave-uniform verdict
__device__ __forceinline__ bool mcr_touch_verdict(const McrParams& p, const int env) {
  const int lane = threadIdx.x & 63, N = p.N, BN = p.BN;
  if (!p.car_contacts || N < 2) return false;
  const McrShapes& S = *p.shapes;
  const int npairs = N * (N - 1) / 2;
  // lane L < npairs <-> car pair (a, b), a < b, in the order (0,1) (0,2) .. (0,N-1) (1,2) ..
  int pa = 0, pb = 1;
  { int r = lane; for (pa = 0; pa < N - 1; ++pa) { const int cnt = N - 1 - pa; if (r < cnt) break; r -= cnt; } pb = pa + 1 + r; }
  const bool is_pair = lane < npairs;
  {
    // coarse exit before any fixture is transformed: a car lies inside the discs of radius S.pad[0] around its hull's
    // centre of mass and S.pad[1] around each wheel's; two cars whose discs stay 0.2 apart (polygon radii and box slack
    // are 0.02 and 0.05) cannot have overlapping car boxes, let alone touch.  Lane = car * 8 + body (bodies 0..4) holds a
    // centre; lane = car pair tests its nine disc pairs (hull-hull, hull_a-wheel_b x4, wheel_a-hull_b x4).
    const int cc_ = lane >> 3, body = lane & 7;
    float bx = 0.0f, by = 0.0f;
    const bool have = cc_ < N && body < 5;
    if (have) { bx = p.carf[(CF_CX + body) * BN + env * N + cc_]; by = p.carf[(CF_CY + body) * BN + env * N + cc_]; }
    bool close = false;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int ba = k < 5 ? 0 : k - 4, bb = k < 5 ? k : 0;
      const int la = is_pair ? pa * 8 + ba : 0, lb = is_pair ? pb * 8 + bb : 0;
      const float ax = __shfl(bx, la), ay = __shfl(by, la), ox = __shfl(bx, lb), oy = __shfl(by, lb);
      const float r = (ba == 0 ? S.pad[0] : S.pad[1]) + (bb == 0 ? S.pad[0] : S.pad[1]) + 0.2f;
      close = close || (is_pair && (ax - ox) * (ax - ox) + (ay - oy) * (ay - oy) <= r * r);
    }
    if (!__any(close)) return false;
  }
  const int c = lane >> 3, fi = lane & 7;
  float4 fxf = make_float4(0.0f, 0.0f, 0.0f, 1.0f), fbox = make_float4(MCR_MAXFLT, MCR_MAXFLT, -MCR_MAXFLT, -MCR_MAXFLT);
  float lox = MCR_MAXFLT, loy = MCR_MAXFLT, hix = -MCR_MAXFLT, hiy = -MCR_MAXFLT;
  if (c < N) {
    const int ci = env * N + c;
    const int body = fi < 4 ? 0 : fi - 3;
    const McrPoly& P = fi < 4 ? S.hull[fi] : S.wheel;
    const V2 cc_ = v2(p.carf[(CF_CX + body) * BN + ci], p.carf[(CF_CY + body) * BN + ci]);
    const float a = p.carf[(CF_A + body) * BN + ci];
    const V2 lc = body == 0 ? v2(S.hull_lcx, S.hull_lcy) : v2(0.0f, 0.0f);
    const Xf xf = xf_of(cc_, a, lc);
    fxf = make_float4(xf.p.x, xf.p.y, xf.q.s, xf.q.c);
    for (int i = 0; i < P.n; ++i) {
      const V2 w = xmul(xf, v2(P.vx[i], P.vy[i]));
      lox = mcr_min(lox, w.x); loy = mcr_min(loy, w.y); hix = mcr_max(hix, w.x); hiy = mcr_max(hiy, w.y);
    }
    fbox = make_float4(lox - 0.05f, loy - 0.05f, hix + 0.05f, hiy + 0.05f);
  }
  for (int o = 1; o < 8; o <<= 1) {
    lox = mcr_min(lox, __shfl_xor(lox, o)); loy = mcr_min(loy, __shfl_xor(loy, o));
    hix = mcr_max(hix, __shfl_xor(hix, o)); hiy = mcr_max(hiy, __shfl_xor(hiy, o));
  }
  const float cb0 = lox - 0.05f, cb1 = loy - 0.05f, cb2 = hix + 0.05f, cb3 = hiy + 0.05f;     // the car's box, on all 8 lanes of the car
  // cheap exit, as in k_collide: no pair of car boxes overlaps -> no fixture pair can touch (lane = car pair)
  unsigned long long pmask;
  {
    const int la = is_pair ? pa * 8 : 0, lb = is_pair ? pb * 8 : 0;
    const float a0 = __shfl(cb0, la), a1 = __shfl(cb1, la), a2 = __shfl(cb2, la), a3 = __shfl(cb3, la);
    const float b0 = __shfl(cb0, lb), b1 = __shfl(cb1, lb), b2 = __shfl(cb2, lb), b3 = __shfl(cb3, lb);
    pmask = __ballot(is_pair && !(a0 > b2 || a2 < b0 || a1 > b3 || a3 < b1));
  }
  // fixtureA of a pair is the fixture with the lower proxy id (k_collide.h: `flipped`), and b2CollidePolygons is not symmetric in its
  // arguments (the reference-face tie-break 0.98 * sepA + 0.0005): the verdict has to ask in the order the contact pass will
  const McrPidTables pidt = mcr_pid_tables(p, env, p.slots + ((size_t)env * 2 + p.env[env].slot) * MCR_SLOT_BYTES);
  // the fixture pairs of the car pairs whose boxes overlap, 8 x 8 at a time (lane = fixture of a * 8 + fixture of b).  "Does any
  // pair touch" does not depend on the order the pairs are looked at: k_collide, which also has to ORDER its manifolds, walks the
  // pairs differently and must arrive at the same answer (it counts the envs where it does not).
  while (pmask) {
    const int q = (int)__builtin_ctzll(pmask); pmask &= pmask - 1ull;
    int a = 0, r = q;
    for (a = 0; a < N - 1; ++a) { const int cnt = N - 1 - a; if (r < cnt) break; r -= cnt; }
    const int b = a + 1 + r;
    const int fa = lane >> 3, fb = lane & 7;
    const bool valid = !(fa >= 4 && fb >= 4);                              // wheel vs wheel: filtered
    const int ia = a * 8 + fa, ib = b * 8 + fb;
    const float4 A = make_float4(__shfl(fbox.x, ia), __shfl(fbox.y, ia), __shfl(fbox.z, ia), __shfl(fbox.w, ia));
    const float4 Bb = make_float4(__shfl(fbox.x, ib), __shfl(fbox.y, ib), __shfl(fbox.z, ib), __shfl(fbox.w, ib));
    const float4 ta = make_float4(__shfl(fxf.x, ia), __shfl(fxf.y, ia), __shfl(fxf.z, ia), __shfl(fxf.w, ia));
    const float4 tb = make_float4(__shfl(fxf.x, ib), __shfl(fxf.y, ib), __shfl(fxf.z, ib), __shfl(fxf.w, ib));
    bool hit = false;
    if (valid && !(A.x > Bb.z || A.z < Bb.x || A.y > Bb.w || A.w < Bb.y)) {
      const McrPoly& pa_ = fa < 4 ? S.hull[fa] : S.wheel; const McrPoly& pb_ = fb < 4 ? S.hull[fb] : S.wheel;
      Xf xa, xb;
      xa.p = v2(ta.x, ta.y); xa.q.s = ta.z; xa.q.c = ta.w; xb.p = v2(tb.x, tb.y); xb.q.s = tb.z; xb.q.c = tb.w;
      cc::Manifold M; M.n = 0; M.type = 0; M.pl[0] = M.pl[1] = v2(0.0f, 0.0f); M.id[0] = M.id[1] = 0; M.localNormal = M.localPoint = v2(0.0f, 0.0f);
      if (pidt.has && pidt.fix[ib] < pidt.fix[ia]) cc::collide_polygons(M, pb_, xb, pa_, xa); else cc::collide_polygons(M, pa_, xa, pb_, xb);
      hit = M.n > 0;
    }
    if (__any(hit)) return true;
  }
  return false;
}

// all envs at once: what mcr_step launches first when the verdicts may be stale (after reset(), reset_envs(), a state
// restore, a step without actions)
#ifndef MCR_DEVICE_FUNCTIONS_ONLY
__global__ __launch_bounds__(64) void k_touch(McrParams p) {
  const int env = p.env0 + (int)blockIdx.x;
  if (env >= p.env0 + p.nenv) return;
  const bool v = p.env[env].active ? mcr_touch_verdict(p, env) : false;
  if (threadIdx.x == 0) { p.part[env] = v ? 1 : 0; if (v && p.fuse_collide) p.clist[1 + atomicAdd(&p.clist[0], 1)] = env; }     // (... and THIS step's contact list: the caller zeroed its count)
}
#endif
