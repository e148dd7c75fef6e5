// k_dynamics.h — one wavefront lane per car: Car.steer/gas/brake + Car.step (binary64, as CPython),
// then the Box2D island solve for the car's 5 bodies and 4 revolute joints (binary32, as Box2D):
// integrate velocities, warm start, 180 Gauss-Seidel velocity iterations, integrate positions,
// <=60 position iterations with the early exit, sleep bookkeeping; then the env bookkeeping of
// multi_car_racing.py:433-443,497-507 across the env's lane group, TimeLimit, and the auto-reset install.
//
// Reference: multi_car_racing.py:418-429; gym car_dynamics.py Car.{steer,gas,brake,step};
// Box2D 2.3 b2Island::Solve, b2RevoluteJoint::{Init,Solve}VelocityConstraints/SolvePositionConstraints.
//
// State lives in registers for the whole step (256 VGPRs, one wavefront per SIMD): the 180x4 joint iterations are
// a serial dependency chain, so the kernel is bound by the VALU issue of a single wave, not by HBM; loads/stores are
// one coalesced 4- or 8-byte access per SoA field per lane.  The launch lasts as long as its slowest wavefront,
// which is why envs with car<->car contacts (role 2) and envs with a crawling position loop (deferral, role 3)
// run in launches of their own on other streams (mcr_hip.hip: launch_step).
// dynamics_block<CC>: CC = false is the build without any car<->car contact code (main launch of the three-chain step, resume chain);
// CC = true adds the contact solver (sequential Gauss-Seidel over the env's manifolds by its leader lane, bodies exchanged through LDS,
// constraints in the order of b2World::Solve's island search: the records arrive in it, a car's joints are permuted into it).
#pragma once
#include "mcr_kernels.h"
#include "k_carcontacts.h"
#include <type_traits>

namespace dyn {

typedef float f2 __attribute__((ext_vector_type(2)));

struct Joint {
  float ix, iy, iz, im;          // accumulated impulse (x,y,z) + motor impulse (warm start state)
  float motorSpeed;
  int limit;                     // 0 inactive, 1 at lower, 2 at upper
  // per-step temporaries
  float rAx, rAy, rBx, rBy;
  float nrAy;                    // -rAy: cross(w, rA) = w * (nrAy, rAx) and cross(rA, P) = rAx * Py + nrAy * Px as one packed multiply each
  float exx, eyx, ezx, eyy, ezy, ezz;   // symmetric K (b2Mat33 m_mass): ex=(exx,eyx,ezx) ey=(eyx,eyy,ezy) ez=(ezx,ezy,ezz)
  float motorMass;
  // pieces of b2Mat33::Solve33 / Solve22 that depend on K only: Box2D recomputes them in every call, here they are
  // evaluated once per step with the same expressions (identical values, ~20 instructions less per joint sweep)
  float cyz0, cyz1, cyz2, idet33, idet22;
};

__device__ __forceinline__ float d3(float ax, float ay, float az, float bx, float by, float bz) { return ax * bx + ay * by + az * bz; }

// K-only parts of b2Mat33::Solve33 / Solve22: cross(ey, ez), 1/det (0 stays 0), same expression order as Box2D
__device__ __forceinline__ void solve_prepare(Joint& J) {
  const float ex0 = J.exx, ex1 = J.eyx, ex2 = J.ezx, ey0 = J.eyx, ey1 = J.eyy, ey2 = J.ezy, ez0 = J.ezx, ez1 = J.ezy, ez2 = J.ezz;
  J.cyz0 = ey1 * ez2 - ey2 * ez1; J.cyz1 = ey2 * ez0 - ey0 * ez2; J.cyz2 = ey0 * ez1 - ey1 * ez0;
  float det = d3(ex0, ex1, ex2, J.cyz0, J.cyz1, J.cyz2);
  if (det != 0.0f) det = 1.0f / det;
  J.idet33 = det;
  float det2 = J.exx * J.eyy - J.eyx * J.eyx;
  if (det2 != 0.0f) det2 = 1.0f / det2;
  J.idet22 = det2;
}
// b2Mat33::Solve33
__device__ __forceinline__ void solve33(const Joint& J, float bx, float by, float bz, float& x, float& y, float& z) {
  // ex=(exx,eyx,ezx)  ey=(eyx,eyy,ezy)  ez=(ezx,ezy,ezz)
  const float ex0 = J.exx, ex1 = J.eyx, ex2 = J.ezx;
  const float ey0 = J.eyx, ey1 = J.eyy, ey2 = J.ezy;
  const float ez0 = J.ezx, ez1 = J.ezy, ez2 = J.ezz;
  const float det = J.idet33;
  x = det * d3(bx, by, bz, J.cyz0, J.cyz1, J.cyz2);
  // cross(b, ez)
  float p0 = by * ez2 - bz * ez1, p1 = bz * ez0 - bx * ez2, p2 = bx * ez1 - by * ez0;
  y = det * d3(ex0, ex1, ex2, p0, p1, p2);
  // cross(ey, b)
  float r0 = ey1 * bz - ey2 * by, r1 = ey2 * bx - ey0 * bz, r2 = ey0 * by - ey1 * bx;
  z = det * d3(ex0, ex1, ex2, r0, r1, r2);
}
// b2Mat33::Solve22
__device__ __forceinline__ void solve22(const Joint& J, float bx, float by, float& x, float& y) {
  const float a11 = J.exx, a12 = J.eyx, a21 = J.eyx, a22 = J.eyy;
  const float det = J.idet22;
  x = det * (a22 * bx - a12 * by);
  y = det * (a11 * by - a21 * bx);
}

struct Body { float cx, cy, a, vx, vy, w; };

// The wheel's joint anchor is its own centre of mass (localAnchorB = localCenterB = 0), so Box2D's
// rB = R(aB)*(0-0) is (+-0, +-0): every term it feeds is an exact zero and contributes nothing but the SIGN of a
// zero.  The kernels below therefore drop rB (one sincos less per joint evaluation); results are identical up to
// -0.0 vs +0.0, which compare equal and never reach a division or a sign test.

// b2RevoluteJoint::InitVelocityConstraints (enableMotor, enableLimit, limits +-0.4, dtRatio 1)
__device__ __forceinline__ void joint_init(Joint& J, Body& A, Body& B, float anchx, float anchy, float lcx, float lcy,
                                           float mA, float iA, float mB, float iB) {
  Rot qA = rot_of(A.a);
  V2 rA = rmul(qA, v2(anchx, anchy) - v2(lcx, lcy));
  J.rAx = rA.x; J.rAy = rA.y; J.rBx = 0.0f; J.rBy = 0.0f; J.nrAy = -rA.y;
  J.exx = mA + mB + J.rAy * J.rAy * iA;
  J.eyx = -J.rAy * J.rAx * iA;
  J.ezx = -J.rAy * iA;
  J.eyy = mA + mB + J.rAx * J.rAx * iA;
  J.ezy = J.rAx * iA;
  J.ezz = iA + iB;
  J.motorMass = iA + iB;
  if (J.motorMass > 0.0f) J.motorMass = 1.0f / J.motorMass;
  solve_prepare(J);
  float jointAngle = B.a - A.a;
  if (jointAngle <= -0.4f) { if (J.limit != 1) J.iz = 0.0f; J.limit = 1; }
  else if (jointAngle >= 0.4f) { if (J.limit != 2) J.iz = 0.0f; J.limit = 2; }
  else { J.limit = 0; J.iz = 0.0f; }
  // warm start
  float Px = J.ix, Py = J.iy;
  A.vx = A.vx - mA * Px; A.vy = A.vy - mA * Py;
  A.w -= iA * ((J.rAx * Py - J.rAy * Px) + J.im + J.iz);
  B.vx = B.vx + mB * Px; B.vy = B.vy + mB * Py;
  B.w += iB * (J.im + J.iz);
}

// b2RevoluteJoint::SolveVelocityConstraints, Box2D's scalar form — what the contact chains run: their sweep loops are short of registers and the
// pairs of joint_velocity (below) push state into scratch there (contact envs' velocity phase 186 -> 197 us measured).  LIMITS = false: for joints whose limit is known to be inactive in every lane of the wavefront
// (the caller checked: J.limit == 0 — the state is a constant of the step) — the same operations as the !lim path below, without the
// selects and masks that carry the other path's results along.
template <bool LIMITS = true>
__device__ __forceinline__ void joint_velocity_scalar(Joint& J, Body& A, Body& B, float mA, float iA, float mB, float iB, float maxImpulse) {
  float vAx = A.vx, vAy = A.vy, wA = A.w, vBx = B.vx, vBy = B.vy, wB = B.w;
  {
    float Cdot = wB - wA - J.motorSpeed;
    float impulse = -J.motorMass * Cdot;
    float old = J.im;
    J.im = mcr_clamp(J.im + impulse, -maxImpulse, maxImpulse);
    impulse = J.im - old;
    wA -= iA * impulse; wB += iB * impulse;
  }
  // Cdot1 = vB + cross(wB, rB) - vA - cross(wA, rA)   with rB == 0
  const float c1x = (vBx - vAx) - (-wA * J.rAy);
  const float c1y = (vBy - vAy) - (wA * J.rAx);
  // Box2D's two branches (limit active: 3x3 solve, and a 2x2 re-solve when the limit impulse would change sign; limit
  // inactive: 2x2 solve) share the 2x2 solve and the application of the impulse here — same expressions and operand
  // order per lane, but a wavefront whose lanes disagree about the limit state (front wheels of a batch of cars with
  // random steering: nearly always) no longer runs both copies of them.
  const bool lim = LIMITS && J.limit != 0;
  float impx = 0.0f, impy = 0.0f, impz = 0.0f;
  float rhsx = -c1x, rhsy = -c1y;
  bool two = !lim;
  if (lim) {
    float c2 = wB - wA;
    float sx, sy, sz; solve33(J, c1x, c1y, c2, sx, sy, sz);
    impx = -sx; impy = -sy; impz = -sz;
    float newImpulse = J.iz + impz;
    bool reduce = (J.limit == 1) ? (newImpulse < 0.0f) : (newImpulse > 0.0f);
    if (reduce) {
      rhsx = -c1x + J.iz * J.ezx; rhsy = -c1y + J.iz * J.ezy;
      impz = -J.iz;
      J.iz = 0.0f;
      two = true;
    } else J.iz += impz;
  }
  if (two) solve22(J, rhsx, rhsy, impx, impy);
  J.ix += impx; J.iy += impy;
  vAx = vAx - mA * impx; vAy = vAy - mA * impy;
  {
    const float t = J.rAx * impy - J.rAy * impx;
    wA -= iA * (lim ? t + impz : t);
  }
  vBx = vBx + mB * impx; vBy = vBy + mB * impy;
  if (lim) wB += iB * impz;
  A.vx = vAx; A.vy = vAy; A.w = wA; B.vx = vBx; B.vy = vBy; B.w = wB;
}

// b2RevoluteJoint::SolveVelocityConstraints.  LIMITS = false: for joints whose limit is known to be inactive in every lane of the wavefront
// (the caller checked: J.limit == 0 — the state is a constant of the step).
// Written on PAIRS of floats (v_pk_mul_f32 / v_pk_add_f32: one issue slot for two lanes of a vector — and issue slots of one wavefront are what
// the 180 sweeps cost): the same IEEE operations on the same operands as Box2D's scalar code, grouped so that the two components of a
// vector expression travel together.  What makes the grouping exact:
//   a - b == a + (-b) == -(b - a),  (-a) * b == a * (-b) == -(a * b),  a * b == b * a,  a + b == b + a   (bit for bit, signs of zeros included);
//   x + (-0.0f) == x for every x (so a lane without an active limit adds an impulse of -0.0f instead of skipping the addition).
// Solve33's two cross products and the dot products that consume them run as (cross(b, ez), cross(ey, b)) pairs: with the second halves of
// the constant pairs NEGATED (E_k = (ez_k, -ey_k)), (p_i, r_i) = s * E_j - t * E_k for scalars s, t out of (bx, by, bz) — plain packed
// operations with a broadcast scalar, no mixed signs.
template <bool LIMITS = true>
__device__ __forceinline__ void joint_velocity(Joint& J, Body& A, Body& B, float mA, float iA, float mB, float iB, float maxImpulse) {
  f2 vA = {A.vx, A.vy}, vB = {B.vx, B.vy};
  float wA = A.w, wB = B.w;
  {
    float Cdot = wB - wA - J.motorSpeed;
    float impulse = -J.motorMass * Cdot;
    float old = J.im;
    J.im = __builtin_amdgcn_fmed3f(J.im + impulse, -maxImpulse, maxImpulse);   // b2Clamp as ONE instruction: the median of (x, -M, M) is max(-M, min(x, M)) for every x that is a number
    impulse = J.im - old;
    wA -= iA * impulse; wB += iB * impulse;
  }
  const f2 nr = {J.nrAy, J.rAx};                     // cross(w, rA) = w * nr
  // Cdot1 = vB + cross(wB, rB) - vA - cross(wA, rA)   with rB == 0
  const f2 c1 = (vB - vA) - wA * nr;
  const f2 kd = {J.eyy, J.exx};
  f2 imp;
  float impz = -0.0f;
  if constexpr (LIMITS) {
    // Box2D's two branches (limit active: 3x3 solve, and a 2x2 re-solve when the limit impulse would change sign; limit inactive: 2x2 solve)
    // computed side by side for every lane and selected: lanes are free on a wavefront that is alone on its SIMD, branches are not
    const bool lim = J.limit != 0;
    const float bz = wB - wA;
    const f2 E0 = {J.ezx, -J.eyx}, E1 = {J.ezy, -J.eyy}, E2 = {J.ezz, -J.ezy};
    const f2 cyz01 = {J.cyz0, J.cyz1};
    // b2Mat33::Solve33(c1x, c1y, bz)
    const f2 m = c1 * cyz01;
    const float sx = J.idet33 * ((m.x + m.y) + bz * J.cyz2);
    const f2 pr0 = c1.y * E2 - bz * E1;              // (by ez2 - bz ez1, ey1 bz - ey2 by)
    const f2 pr1 = bz * E0 - c1.x * E2;              // (bz ez0 - bx ez2, ey2 bx - ey0 bz)
    const f2 pr2 = c1.x * E1 - c1.y * E0;            // (bx ez1 - by ez0, ey0 by - ey1 bx)
    const f2 syz = J.idet33 * ((J.exx * pr0 + J.eyx * pr1) + J.ezx * pr2);
    const float newImpulse = J.iz + (-syz.y);
    // at the lower limit the impulse may not become negative, at the upper one not positive (x < 0 <=> -x > 0: one sign flip, one compare)
    const bool reduce = lim && __uint_as_float(__float_as_uint(newImpulse) ^ (J.limit == 1 ? 0x80000000u : 0u)) > 0.0f;
    // the 2x2 solve for BOTH right-hand sides it can get (no limit: -c1; the limit impulse taken back: -c1 + iz (ezx, ezy)) beside the 3x3
    // solve instead of behind its verdict: eight more packed operations, five fewer on the sweep's dependency chain — which is what a
    // wavefront alone on its SIMD pays for
    const f2 ezxy = {J.ezx, J.ezy};
    const f2 rhsN = -c1, rhsR = -c1 + J.iz * ezxy;
    const f2 s22N = J.idet22 * (kd * rhsN - J.eyx * rhsN.yx), s22R = J.idet22 * (kd * rhsR - J.eyx * rhsR.yx);
    const f2 base = {lim ? -sx : s22N.x, lim ? -syz.x : s22N.y};
    imp.x = reduce ? s22R.x : base.x; imp.y = reduce ? s22R.y : base.y;
    impz = lim ? (reduce ? -J.iz : -syz.y) : -0.0f;
    J.iz = reduce ? 0.0f : J.iz + impz;              // (no limit: 0 + -0 = 0)
  } else {
    const f2 rhs = -c1;
    imp = J.idet22 * (kd * rhs - J.eyx * rhs.yx);
  }
  f2 ixy = {J.ix, J.iy};
  ixy = ixy + imp;
  J.ix = ixy.x; J.iy = ixy.y;
  vA = vA - mA * imp;
  {
    const f2 tt = nr * imp;                          // cross(rA, imp) = rAx impy - rAy impx
    const float t = tt.y + tt.x;
    wA -= iA * (LIMITS ? t + impz : t);
  }
  vB = vB + mB * imp;
  if constexpr (LIMITS) wB += iB * impz;
  A.vx = vA.x; A.vy = vA.y; A.w = wA; B.vx = vB.x; B.vy = vB.y; B.w = wB;
}

// the form a build runs: pairs in the builds without contact code (main launch, resume chain: registers to spare), Box2D's scalar form in the
// builds that carry the contact chains (dynamics_block<CC = true>: their code generation stays what round 5 tuned)
template <bool PK, bool LIMITS = true>
__device__ __forceinline__ void joint_velocity_as(Joint& J, Body& A, Body& B, float mA, float iA, float mB, float iB, float maxImpulse) {
  if constexpr (PK) joint_velocity<LIMITS>(J, A, B, mA, iA, mB, iB, maxImpulse); else joint_velocity_scalar<LIMITS>(J, A, B, mA, iA, mB, iB, maxImpulse);
}

// sin/cos of the hull angle, remembered across joint_position calls: the hull is ~100x heavier than a wheel, so its
// angle is bit-identical from one joint to the next most of the time (always, in the long marginal position loops)
// and the f64 sincos — the bulk of a position sweep — is only re-evaluated when the argument actually changed.
struct RotCache { float a; Rot q; };
// b2RevoluteJoint::SolvePositionConstraints
__device__ __forceinline__ bool joint_position(const Joint& J, Body& A, Body& B, float anchx, float anchy, float lcx, float lcy,
                                               float mA, float iA, float mB, float iB, RotCache& rc) {
  float cAx = A.cx, cAy = A.cy, aA = A.a, cBx = B.cx, cBy = B.cy, aB = B.a;
  float angularError = 0.0f, positionError = 0.0f;
  if (J.limit != 0) {
    float angle = aB - aA;
    float limitImpulse = 0.0f;
    if (J.limit == 1) {
      float C = angle - (-0.4f); angularError = -C;
      C = mcr_clamp(C + B2_ANGULAR_SLOP, -B2_MAX_ANGULAR_CORRECTION, 0.0f);
      limitImpulse = -J.motorMass * C;
    } else {
      float C = angle - 0.4f; angularError = C;
      C = mcr_clamp(C - B2_ANGULAR_SLOP, 0.0f, B2_MAX_ANGULAR_CORRECTION);
      limitImpulse = -J.motorMass * C;
    }
    aA -= iA * limitImpulse; aB += iB * limitImpulse;
  }
  {
    if (__float_as_int(aA) != __float_as_int(rc.a)) { rc.q = rot_of(aA); rc.a = aA; }   // bit compare; rc.a starts as a NaN
    const Rot qA = rc.q;
    V2 rA = rmul(qA, v2(anchx, anchy) - v2(lcx, lcy));
    float Cx = (cBx - cAx) - rA.x, Cy = (cBy - cAy) - rA.y;
    // positionError = sqrtf(Cx Cx + Cy Cy) is only ever compared with b2_linearSlop: sqrtf is monotonic and correctly rounded, so
    // "sqrtf(x) <= 0.005f" is "x <= T" with T the largest float whose root rounds to 0.005f or below (0x37d1b718: the next float's root
    // is 0.0050000004f; tests/test_oracle_pinning.py checks it against sqrtf) — the square root (12 instructions of every correction) goes
    positionError = Cx * Cx + Cy * Cy;
    float k11 = mA + mB + iA * rA.y * rA.y;
    float k12 = -iA * rA.x * rA.y;
    float k22 = mA + mB + iA * rA.x * rA.x;
    float det = k11 * k22 - k12 * k12;
    if (det != 0.0f) det = 1.0f / det;
    float ix = -(det * (k22 * Cx - k12 * Cy)), iy = -(det * (k11 * Cy - k12 * Cx));
    cAx = cAx - mA * ix; cAy = cAy - mA * iy; aA -= iA * (rA.x * iy - rA.y * ix);
    cBx = cBx + mB * ix; cBy = cBy + mB * iy;
  }
  A.cx = cAx; A.cy = cAy; A.a = aA; B.cx = cBx; B.cy = cBy; B.a = aB;
  return positionError <= __int_as_float(0x37d1b718) && angularError <= B2_ANGULAR_SLOP;
}

__device__ __forceinline__ double np_sign(double x) { return x > 0.0 ? 1.0 : (x < 0.0 ? -1.0 : 0.0); }

// ---------------------------------------------------------------------------------------------------------
// car<->car contact constraints, executed by the env's leader lane on LDS-resident body state.
// xs[comp*5 + body][lane]: comp 0..2 = (vx, vy, w) or (cx, cy, a) of `body` of the car owned by `lane`.
#define DYN_VC_POOL (MCR_SIDE_ENVS_PER_WAVE * MCR_CC_MAX)
// mass data of the two body kinds, held in registers by the kernel (reading the shape table from memory inside the
// contact sweeps would put a load on the critical path of every contact of every sweep)
struct CcMass { float mH, iH, mW, iW, lcx, lcy; };
__device__ __forceinline__ void cc_masses(const CcMass& S, int body, float& m, float& i, V2& lc) {
  if (body == 0) { m = S.mH; i = S.iH; lc = v2(S.lcx, S.lcy); }
  else { m = S.mW; i = S.iW; lc = v2(0.0f, 0.0f); }
}
// ---- island order of a car's joints (k_collide.h: b2World::Solve's depth-first search; store[2..3], 2 bits per position).  The joint
// code below is written for the order 3,2,1,0 with the state of joint q and of its wheel in registers J[q], b[q + 1]; a car whose
// search entered through another wheel gets its (joint, wheel) register sets PERMUTED for the solve instead — slot 3 holds what is
// solved first — and put back after it: a 5-comparator sorting network of per-lane selects, run only in waves that hold such a car.
// What goes by wheel identity in between: the joint anchors (anchor_of) and the body rows of the contact exchange (xmap).
__device__ __forceinline__ void cswapf(bool sw, float& a, float& b) { const float t = sw ? b : a; b = sw ? a : b; a = t; }
__device__ __forceinline__ void cswapi(bool sw, int& a, int& b) { const int t = sw ? b : a; b = sw ? a : b; a = t; }
__device__ __forceinline__ void slot_exchange(Joint* J, Body* b, int* key, int* tag, int i, int j) {
  const bool sw = key[i] > key[j];
  cswapi(sw, key[i], key[j]); cswapi(sw, tag[i], tag[j]);
  cswapf(sw, J[i].ix, J[j].ix); cswapf(sw, J[i].iy, J[j].iy); cswapf(sw, J[i].iz, J[j].iz); cswapf(sw, J[i].im, J[j].im);
  cswapf(sw, J[i].motorSpeed, J[j].motorSpeed); cswapi(sw, J[i].limit, J[j].limit);
  Body& x = b[i + 1]; Body& y = b[j + 1];
  cswapf(sw, x.cx, y.cx); cswapf(sw, x.cy, y.cy); cswapf(sw, x.a, y.a); cswapf(sw, x.vx, y.vx); cswapf(sw, x.vy, y.vy); cswapf(sw, x.w, y.w);
}
// sorts the four slots ascending by key; tag rides along
__device__ __forceinline__ void slot_sort(Joint* J, Body* b, int* key, int* tag) {
  slot_exchange(J, b, key, tag, 0, 1); slot_exchange(J, b, key, tag, 2, 3); slot_exchange(J, b, key, tag, 0, 2);
  slot_exchange(J, b, key, tag, 1, 3); slot_exchange(J, b, key, tag, 1, 2);
}
__device__ __forceinline__ float pick4(int i, float a0, float a1, float a2, float a3) { return i == 0 ? a0 : i == 1 ? a1 : i == 2 ? a2 : a3; }
// row of body `body` (0 hull, 1 + wheel) of the car in lane L of the exchange arrays: xmap[L] = slot of wheel w at bits 2w
__device__ __forceinline__ int cc_row(const int* xmap, int L, int body) { return body == 0 ? 0 : 1 + ((xmap[L] >> (2 * (body - 1))) & 3); }
// exchange rows of a contact's two bodies, for bits 16.. / 20.. of the key of its LDS copy
__device__ __forceinline__ uint32_t cc_rows_of(uint32_t key, int leader_lane, const int* xmap) {
  const int LA = leader_lane + (int)(key & 15u), LB = leader_lane + (int)((key >> 8) & 15u);
  return ((uint32_t)cc_row(xmap, LA, cc::fixture_body((int)((key >> 4) & 15u))) << 16) | ((uint32_t)cc_row(xmap, LB, cc::fixture_body((int)((key >> 12) & 15u))) << 20);
}
// b2ContactSolver ctor + InitializeVelocityConstraints + WarmStart for one stored manifold
__device__ inline void cc_init(const CcMass& S, const uint32_t* rec, int rec_index, int leader_lane, float (*xp)[64], float (*xv)[64], const int* xmap, float* vc) {
  const uint32_t key = rec[0];
  const int carA = key & 15, fixA = (key >> 4) & 15, carB = (key >> 8) & 15, fixB = (key >> 12) & 15;
  const int LA = leader_lane + carA, LB = leader_lane + carB;
  // (rows of the exchange arrays; row 0 is the hull, so the masses below go by the row as well)
  const int bA = cc_row(xmap, LA, cc::fixture_body(fixA)), bB = cc_row(xmap, LB, cc::fixture_body(fixB));
  const int type = rec[1] & 255; int n = (int)(rec[1] >> 8);
  float mA, iA, mB, iB; V2 lcA, lcB; cc_masses(S, bA, mA, iA, lcA); cc_masses(S, bB, mB, iB, lcB);
  const V2 cA = v2(xp[0 * 5 + bA][LA], xp[1 * 5 + bA][LA]); const float aA = xp[2 * 5 + bA][LA];
  const V2 cB = v2(xp[0 * 5 + bB][LB], xp[1 * 5 + bB][LB]); const float aB = xp[2 * 5 + bB][LB];
  Xf xfA, xfB; xfA.q = rot_of(aA); xfB.q = rot_of(aB);
  xfA.p = cA - rmul(xfA.q, lcA); xfB.p = cB - rmul(xfB.q, lcB);
  const V2 localNormal = v2(__uint_as_float(rec[2]), __uint_as_float(rec[3])), localPoint = v2(__uint_as_float(rec[4]), __uint_as_float(rec[5]));
  // b2WorldManifold::Initialize
  V2 normal, pts[2]; pts[0] = pts[1] = v2(0.0f, 0.0f);
  const float rA_ = B2_POLYGON_RADIUS, rB_ = B2_POLYGON_RADIUS;
  if (type == 1) {
    normal = rmul(xfA.q, localNormal);
    const V2 planePoint = xmul(xfA, localPoint);
#pragma unroll
    for (int j = 0; j < 2; ++j) {                 // fixed trip count + guard: pts[] stays in registers (a `j < n` loop put it in scratch)
      if (j < n) {
        const V2 clip = xmul(xfB, v2(__uint_as_float(rec[6 + j * 5]), __uint_as_float(rec[6 + j * 5 + 1])));
        const V2 pa = clip + (rA_ - dot(clip - planePoint, normal)) * normal;
        const V2 pb = clip - rB_ * normal;
        pts[j] = 0.5f * (pa + pb);
      }
    }
  } else {
    normal = rmul(xfB.q, localNormal);
    const V2 planePoint = xmul(xfB, localPoint);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (j < n) {
        const V2 clip = xmul(xfA, v2(__uint_as_float(rec[6 + j * 5]), __uint_as_float(rec[6 + j * 5 + 1])));
        const V2 pb = clip + (rB_ - dot(clip - planePoint, normal)) * normal;
        const V2 pa = clip - rA_ * normal;
        pts[j] = 0.5f * (pa + pb);
      }
    }
    normal = -normal;
  }
  const V2 tangent = cross(normal, 1.0f);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float* q = vc + (j == 0 ? cc::VC_P0 : cc::VC_P1);
    if (j < n) {
      const V2 rA = pts[j] - cA, rB = pts[j] - cB;
      const float rnA = cross(rA, normal), rnB = cross(rB, normal);
      const float kN = mA + mB + iA * rnA * rnA + iB * rnB * rnB;
      const float rtA = cross(rA, tangent), rtB = cross(rB, tangent);
      const float kT = mA + mB + iA * rtA * rtA + iB * rtB * rtB;
      q[0] = rA.x; q[1] = rA.y; q[2] = rB.x; q[3] = rB.y;
      q[4] = 1.0f * __uint_as_float(rec[6 + j * 5 + 2]); q[5] = 1.0f * __uint_as_float(rec[6 + j * 5 + 3]);
      q[6] = kN > 0.0f ? 1.0f / kN : 0.0f; q[7] = kT > 0.0f ? 1.0f / kT : 0.0f;
    } else { for (int t = 0; t < 8; ++t) q[t] = 0.0f; }
  }
  if (n == 2) {
    const float* p0 = vc + cc::VC_P0; const float* p1 = vc + cc::VC_P1;
    const float rn1A = cross(v2(p0[0], p0[1]), normal), rn1B = cross(v2(p0[2], p0[3]), normal);
    const float rn2A = cross(v2(p1[0], p1[1]), normal), rn2B = cross(v2(p1[2], p1[3]), normal);
    const float k11 = mA + mB + iA * rn1A * rn1A + iB * rn1B * rn1B;
    const float k22 = mA + mB + iA * rn2A * rn2A + iB * rn2B * rn2B;
    const float k12 = mA + mB + iA * rn1A * rn2A + iB * rn1B * rn2B;
    if (k11 * k11 < 1000.0f * (k11 * k22 - k12 * k12)) {
      vc[cc::VC_K11] = k11; vc[cc::VC_K12] = k12; vc[cc::VC_K22] = k22;
      float det = k11 * k22 - k12 * k12; if (det != 0.0f) det = 1.0f / det;
      vc[cc::VC_NM11] = det * k22; vc[cc::VC_NM12] = -det * k12; vc[cc::VC_NM22] = det * k11;
    } else n = 1;
  }
  vc[cc::VC_NX] = normal.x; vc[cc::VC_NY] = normal.y;
  ((int*)vc)[cc::VC_N] = n; ((int*)vc)[cc::VC_LA] = LA * 8 + bA; ((int*)vc)[cc::VC_LB] = LB * 8 + bB; ((int*)vc)[cc::VC_REC] = rec_index;
  // WarmStart
  V2 vA = v2(xv[0 * 5 + bA][LA], xv[1 * 5 + bA][LA]); float wA = xv[2 * 5 + bA][LA];
  V2 vB = v2(xv[0 * 5 + bB][LB], xv[1 * 5 + bB][LB]); float wB = xv[2 * 5 + bB][LB];
  for (int j = 0; j < n; ++j) {
    const float* q = vc + (j == 0 ? cc::VC_P0 : cc::VC_P1);
    const V2 P = q[4] * normal + q[5] * tangent;
    wA -= iA * cross(v2(q[0], q[1]), P); vA = vA - mA * P;
    wB += iB * cross(v2(q[2], q[3]), P); vB = vB + mB * P;
  }
  xv[0 * 5 + bA][LA] = vA.x; xv[1 * 5 + bA][LA] = vA.y; xv[2 * 5 + bA][LA] = wA;
  xv[0 * 5 + bB][LB] = vB.x; xv[1 * 5 + bB][LB] = vB.y; xv[2 * 5 + bB][LB] = wB;
}

// b2ContactSolver::SolveVelocityConstraints for one contact.  The record and both bodies are pulled into
// registers with a few wide LDS reads, and only the accumulated impulses + body velocities are written back
// (one LDS round trip per contact per iteration instead of one per scalar).
__device__ inline void cc_velocity(const CcMass& S, float* __restrict__ vcf, float (* __restrict__ xv)[64]) {
  const float4* __restrict__ v4 = (const float4*)vcf;
  const float4 r0 = v4[0], r1 = v4[1], r2 = v4[2], r3 = v4[3], r4 = v4[4], r5 = v4[5], r6 = v4[6];
  // layout: [0]=nx [1]=ny [2]=n [3]=LA | [4]=LB [5..12]=P0 | [13..20]=P1 | [21]=k11 [22]=k12 [23]=k22 [24]=nm11 [25]=nm12 [26]=nm22
  const int n = __float_as_int(r0.z);
  const int sa = __float_as_int(r0.w), sb = __float_as_int(r1.x);
  const int LA = sa >> 3, bA = sa & 7, LB = sb >> 3, bB = sb & 7;
  float mA, iA, mB, iB; V2 lcA, lcB; cc_masses(S, bA, mA, iA, lcA); cc_masses(S, bB, mB, iB, lcB);
  V2 vA = v2(xv[0 * 5 + bA][LA], xv[1 * 5 + bA][LA]); float wA = xv[2 * 5 + bA][LA];
  V2 vB = v2(xv[0 * 5 + bB][LB], xv[1 * 5 + bB][LB]); float wB = xv[2 * 5 + bB][LB];
  const V2 normal = v2(r0.x, r0.y); const V2 tangent = cross(normal, 1.0f);
  const float friction = sqrtf(0.2f * 0.2f);
  // point 0: rA rB nImp tImp normalMass tangentMass = words 5..12 ; point 1 = words 13..20
  const V2 r1A = v2(r1.y, r1.z), r1B = v2(r1.w, r2.x); float n1 = r2.y, t1 = r2.z; const float nm1 = r2.w, tm1 = r3.x;
  const V2 r2A = v2(r3.y, r3.z), r2B = v2(r3.w, r4.x); float n2 = r4.y, t2 = r4.z; const float nm2 = r4.w, tm2 = r5.x;
  {
    const V2 dv = vB + cross(wB, r1B) - vA - cross(wA, r1A);
    const float vt = dot(dv, tangent) - 0.0f;
    float lambda = tm1 * (-vt);
    const float maxF = friction * n1;
    const float newImp = mcr_clamp(t1 + lambda, -maxF, maxF);
    lambda = newImp - t1; t1 = newImp;
    const V2 P = lambda * tangent;
    vA = vA - mA * P; wA -= iA * cross(r1A, P);
    vB = vB + mB * P; wB += iB * cross(r1B, P);
  }
  if (n == 2) {
    const V2 dv = vB + cross(wB, r2B) - vA - cross(wA, r2A);
    const float vt = dot(dv, tangent) - 0.0f;
    float lambda = tm2 * (-vt);
    const float maxF = friction * n2;
    const float newImp = mcr_clamp(t2 + lambda, -maxF, maxF);
    lambda = newImp - t2; t2 = newImp;
    const V2 P = lambda * tangent;
    vA = vA - mA * P; wA -= iA * cross(r2A, P);
    vB = vB + mB * P; wB += iB * cross(r2B, P);
  }
  if (n == 1) {
    const V2 dv = vB + cross(wB, r1B) - vA - cross(wA, r1A);
    const float vn = dot(dv, normal);
    float lambda = -nm1 * (vn - 0.0f);
    const float newImp = mcr_max(n1 + lambda, 0.0f);
    lambda = newImp - n1; n1 = newImp;
    const V2 P = lambda * normal;
    vA = vA - mA * P; wA -= iA * cross(r1A, P);
    vB = vB + mB * P; wB += iB * cross(r1B, P);
  } else {
    const V2 a = v2(n1, n2);
    const V2 dv1 = vB + cross(wB, r1B) - vA - cross(wA, r1A);
    const V2 dv2 = vB + cross(wB, r2B) - vA - cross(wA, r2A);
    float vn1 = dot(dv1, normal), vn2 = dot(dv2, normal);
    const float k11 = r5.y, k12 = r5.z, k22 = r5.w, nm11 = r6.x, nm12 = r6.y, nm22 = r6.z;
    V2 b = v2(vn1 - 0.0f, vn2 - 0.0f);
    b = b - v2(k11 * a.x + k12 * a.y, k12 * a.x + k22 * a.y);
    V2 x; bool ok = false;
    for (;;) {
      x = -v2(nm11 * b.x + nm12 * b.y, nm12 * b.x + nm22 * b.y);
      if (x.x >= 0.0f && x.y >= 0.0f) { ok = true; break; }
      x.x = -nm1 * b.x; x.y = 0.0f; vn1 = 0.0f; vn2 = k12 * x.x + b.y;
      if (x.x >= 0.0f && vn2 >= 0.0f) { ok = true; break; }
      x.x = 0.0f; x.y = -nm2 * b.y; vn1 = k12 * x.y + b.x; vn2 = 0.0f;
      if (x.y >= 0.0f && vn1 >= 0.0f) { ok = true; break; }
      x.x = 0.0f; x.y = 0.0f; vn1 = b.x; vn2 = b.y;
      if (vn1 >= 0.0f && vn2 >= 0.0f) { ok = true; break; }
      break;
    }
    if (ok) {
      const V2 d = x - a;
      const V2 P1 = d.x * normal, P2 = d.y * normal;
      vA = vA - mA * (P1 + P2); wA -= iA * (cross(r1A, P1) + cross(r2A, P2));
      vB = vB + mB * (P1 + P2); wB += iB * (cross(r1B, P1) + cross(r2B, P2));
      n1 = x.x; n2 = x.y;
    }
  }
  vcf[cc::VC_P0 + 4] = n1; vcf[cc::VC_P0 + 5] = t1; vcf[cc::VC_P1 + 4] = n2; vcf[cc::VC_P1 + 5] = t2;
  xv[0 * 5 + bA][LA] = vA.x; xv[1 * 5 + bA][LA] = vA.y; xv[2 * 5 + bA][LA] = wA;
  xv[0 * 5 + bB][LB] = vB.x; xv[1 * 5 + bB][LB] = vB.y; xv[2 * 5 + bB][LB] = wB;
}

// b2ContactSolver::SolvePositionConstraints for one contact; returns its min separation
// (rec: the step's LDS copy of the record — its key carries the two bodies' exchange rows at bits 16 and 20, see cc_rows_of)
__device__ inline float cc_position(const CcMass& S, const uint32_t* rec, int leader_lane, float (*xp)[64]) {
  const uint32_t key = rec[0];
  const int carA = key & 15, carB = (key >> 8) & 15;
  const int LA = leader_lane + carA, LB = leader_lane + carB;
  const int bA = (key >> 16) & 7, bB = (key >> 20) & 7;
  const int type = rec[1] & 255; const int n = (int)(rec[1] >> 8);
  float mA, iA, mB, iB; V2 lcA, lcB; cc_masses(S, bA, mA, iA, lcA); cc_masses(S, bB, mB, iB, lcB);
  V2 cA = v2(xp[0 * 5 + bA][LA], xp[1 * 5 + bA][LA]); float aA = xp[2 * 5 + bA][LA];
  V2 cB = v2(xp[0 * 5 + bB][LB], xp[1 * 5 + bB][LB]); float aB = xp[2 * 5 + bB][LB];
  const V2 localNormal = v2(__uint_as_float(rec[2]), __uint_as_float(rec[3])), localPoint = v2(__uint_as_float(rec[4]), __uint_as_float(rec[5]));
  float minSep = 0.0f;
  for (int j = 0; j < n; ++j) {
    Xf xfA, xfB; xfA.q = rot_of(aA); xfB.q = rot_of(aB);
    xfA.p = cA - rmul(xfA.q, lcA); xfB.p = cB - rmul(xfB.q, lcB);
    const V2 lp = v2(__uint_as_float(rec[6 + j * 5]), __uint_as_float(rec[6 + j * 5 + 1]));
    V2 normal, point; float separation;
    if (type == 1) {
      normal = rmul(xfA.q, localNormal);
      const V2 planePoint = xmul(xfA, localPoint);
      const V2 clip = xmul(xfB, lp);
      separation = dot(clip - planePoint, normal) - B2_POLYGON_RADIUS - B2_POLYGON_RADIUS;
      point = clip;
    } else {
      normal = rmul(xfB.q, localNormal);
      const V2 planePoint = xmul(xfB, localPoint);
      const V2 clip = xmul(xfA, lp);
      separation = dot(clip - planePoint, normal) - B2_POLYGON_RADIUS - B2_POLYGON_RADIUS;
      point = clip;
      normal = -normal;
    }
    const V2 rA = point - cA, rB = point - cB;
    minSep = mcr_min(minSep, separation);
    const float C = mcr_clamp(B2_BAUMGARTE * (separation + B2_LINEAR_SLOP), -B2_MAX_LINEAR_CORRECTION, 0.0f);
    const float rnA = cross(rA, normal), rnB = cross(rB, normal);
    const float K = mA + mB + iA * rnA * rnA + iB * rnB * rnB;
    const float impulse = K > 0.0f ? -C / K : 0.0f;
    const V2 P = impulse * normal;
    cA = cA - mA * P; aA -= iA * cross(rA, P);
    cB = cB + mB * P; aB += iB * cross(rB, P);
  }
  xp[0 * 5 + bA][LA] = cA.x; xp[1 * 5 + bA][LA] = cA.y; xp[2 * 5 + bA][LA] = aA;
  xp[0 * 5 + bB][LB] = cB.x; xp[1 * 5 + bB][LB] = cB.y; xp[2 * 5 + bB][LB] = aB;
  return minSep;
}


// ---------------------------------------------------------------------------------------------------------
// The contact chain's form of the two solvers above (UNI): ONE env per wavefront (k_list_chain, role 2), so everything about a manifold —
// which bodies touch, the constraint record — is wave-uniform, and the wavefront's idle lanes are put to use: for the contact phase of a
// sweep the env's bodies are laid out ONE BODY PER LANE (lane car * 5 + row holds that body's three values; the car lanes transpose their
// registers through LDS once per sweep, in and out).  A manifold's two bodies are then read with v_readlane at a SCALAR lane index — the
// register file is the crossbar: no leader lane, no LDS round trip or barrier on the dependent chain of the contacts, no branches —, every
// lane runs the manifold's arithmetic on the same values, and the two owner lanes keep their body's result with a select.
// cmeta (lane i holds manifold i's): vc n [0,2) | body lane A [2,8) | body lane B [8,14) | A is a hull [14] | B is a hull [15] | island [16,20) |
// manifold n [20,22) | type [22,24)
__device__ __forceinline__ float rl(float v, int L) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), L)); }
// b2ContactSolver::SolveVelocityConstraints for manifold i — the arithmetic of cc_velocity, statement for statement.  (cx, cy, cw): this
// lane's body (vx, vy, w).  R: the step's constants of the constraint (the LDS record is never written during the sweeps: the caller loads
// it a manifold ahead); imp: lane i holds the accumulated impulses of manifold i.  Every lane holds the same values, so the solver's
// case analysis branches on the scalar unit (ANYL: "some lane", which is "every lane").
struct VcRec { float4 r0, r1, r2, r3, r4, r5, r6; };
struct CcImp { float n1, t1, n2, t2; };
__device__ __forceinline__ VcRec vc_load(const float* __restrict__ vcf) {
  const float4* __restrict__ v4 = (const float4*)vcf;
  VcRec R; R.r0 = v4[0]; R.r1 = v4[1]; R.r2 = v4[2]; R.r3 = v4[3]; R.r4 = v4[4]; R.r5 = v4[5]; R.r6 = v4[6];
  return R;
}
#define ANYL(c) (__ballot(c) != 0ull)
// Written on PAIRS of floats (see joint_velocity: only sign-symmetric and commutative regroupings of the operations of cc_velocity above — cross(w, r) =
// w (-r.y, r.x), cross(r, P) = ((-r.y, r.x) P).y + ((-r.y, r.x) P).x, the block solver's 2x2 products as column pairs times a broadcast scalar):
// --actions drive 5.41 -> 5.55 M env-steps/s, N=8 +0.5 %; bit-identical (the contact / pile-up / island-order tests).
__device__ __forceinline__ void cc_velocity_bl(const CcMass& S, const VcRec& R, const int meta, CcImp& imp, const int i, const int lane, float& cx, float& cy, float& cw) {
  const float4 r0 = R.r0, r1 = R.r1, r2 = R.r2, r3 = R.r3, r4 = R.r4, r5 = R.r5, r6 = R.r6;
  const int n = meta & 3;
  const int ia = (meta >> 2) & 63, ib = (meta >> 8) & 63;
  const bool hA = (meta >> 14) & 1, hB = (meta >> 15) & 1;
  const float mA = hA ? S.mH : S.mW, iA = hA ? S.iH : S.iW, mB = hB ? S.mH : S.mW, iB = hB ? S.iH : S.iW;
  f2 vA = {rl(cx, ia), rl(cy, ia)}; float wA = rl(cw, ia);
  f2 vB = {rl(cx, ib), rl(cy, ib)}; float wB = rl(cw, ib);
  float n1 = rl(imp.n1, i), t1 = rl(imp.t1, i), n2 = rl(imp.n2, i), t2 = rl(imp.t2, i);
  const f2 normal = {r0.x, r0.y}, tangent = {r0.y, -r0.x};                    // cross(normal, 1.0f)
  const float friction = sqrtf(0.2f * 0.2f);
  // (-r.y, r.x) of the four arms
  const f2 p1A = {-r1.z, r1.y}, p1B = {-r2.x, r1.w}; const float nm1 = r2.w, tm1 = r3.x;
  const f2 p2A = {-r3.z, r3.y}, p2B = {-r4.x, r3.w}; const float nm2 = r4.w, tm2 = r5.x;
  auto crs = [](const f2 pr, const f2 P) -> float { const f2 tt = pr * P; return tt.y + tt.x; };     // cross(r, P)
  auto dot2 = [](const f2 a_, const f2 b_) -> float { const f2 m = a_ * b_; return m.x + m.y; };
  {
    const f2 dv = ((vB + wB * p1B) - vA) - wA * p1A;
    const float vt = dot2(dv, tangent);
    float lambda = tm1 * (-vt);
    const float maxF = friction * n1;
    const float newImp = mcr_clamp(t1 + lambda, -maxF, maxF);
    lambda = newImp - t1; t1 = newImp;
    const f2 P = lambda * tangent;
    vA = vA - mA * P; wA -= iA * crs(p1A, P);
    vB = vB + mB * P; wB += iB * crs(p1B, P);
  }
  if (n == 2) {
    const f2 dv = ((vB + wB * p2B) - vA) - wA * p2A;
    const float vt = dot2(dv, tangent);
    float lambda = tm2 * (-vt);
    const float maxF = friction * n2;
    const float newImp = mcr_clamp(t2 + lambda, -maxF, maxF);
    lambda = newImp - t2; t2 = newImp;
    const f2 P = lambda * tangent;
    vA = vA - mA * P; wA -= iA * crs(p2A, P);
    vB = vB + mB * P; wB += iB * crs(p2B, P);
  }
  if (n == 1) {
    const f2 dv = ((vB + wB * p1B) - vA) - wA * p1A;
    const float vn = dot2(dv, normal);
    float lambda = -nm1 * vn;
    const float newImp = mcr_max(n1 + lambda, 0.0f);
    lambda = newImp - n1; n1 = newImp;
    const f2 P = lambda * normal;
    vA = vA - mA * P; wA -= iA * crs(p1A, P);
    vB = vB + mB * P; wB += iB * crs(p1B, P);
  } else {
    const f2 a = {n1, n2};
    const f2 dv1 = ((vB + wB * p1B) - vA) - wA * p1A;
    const f2 dv2 = ((vB + wB * p2B) - vA) - wA * p2A;
    float vn1 = dot2(dv1, normal), vn2 = dot2(dv2, normal);
    const float k12 = r5.z;
    const f2 Kc1 = {r5.y, r5.z}, Kc2 = {r5.z, r5.w}, NMc1 = {r6.x, r6.y}, NMc2 = {r6.y, r6.z};
    f2 bb = {vn1, vn2};
    bb = bb - (a.x * Kc1 + a.y * Kc2);
    f2 x; bool ok = false;
    for (;;) {
      x = -(bb.x * NMc1 + bb.y * NMc2);
      if (ANYL(x.x >= 0.0f && x.y >= 0.0f)) { ok = true; break; }
      x.x = -nm1 * bb.x; x.y = 0.0f; vn1 = 0.0f; vn2 = k12 * x.x + bb.y;
      if (ANYL(x.x >= 0.0f && vn2 >= 0.0f)) { ok = true; break; }
      x.x = 0.0f; x.y = -nm2 * bb.y; vn1 = k12 * x.y + bb.x; vn2 = 0.0f;
      if (ANYL(x.y >= 0.0f && vn1 >= 0.0f)) { ok = true; break; }
      x.x = 0.0f; x.y = 0.0f; vn1 = bb.x; vn2 = bb.y;
      if (ANYL(vn1 >= 0.0f && vn2 >= 0.0f)) { ok = true; break; }
      break;
    }
    if (ok) {
      const f2 d = x - a;
      const f2 P1 = d.x * normal, P2 = d.y * normal;
      const f2 P12 = P1 + P2;
      vA = vA - mA * P12; wA -= iA * (crs(p1A, P1) + crs(p2A, P2));
      vB = vB + mB * P12; wB += iB * (crs(p1B, P1) + crs(p2B, P2));
      n1 = x.x; n2 = x.y;
    }
  }
  const bool own = lane == i;
  imp.n1 = own ? n1 : imp.n1; imp.t1 = own ? t1 : imp.t1; imp.n2 = own ? n2 : imp.n2; imp.t2 = own ? t2 : imp.t2;
  const bool isA = lane == ia, isB = lane == ib;
  cx = isA ? vA.x : (isB ? vB.x : cx); cy = isA ? vA.y : (isB ? vB.y : cy); cw = isA ? wA : (isB ? wB : cw);
}

// b2ContactSolver::SolvePositionConstraints for one manifold (rec: the step's LDS copy of the stored record); returns its min separation.
// (px, py, pa): this lane's body (c.x, c.y, angle).  Every body lane evaluates the rotation of ITS OWN angle: one f64 sincos on the chain per
// point instead of two, read out of the two owner lanes.
__device__ __forceinline__ float cc_position_bl(const CcMass& S, const uint32_t* __restrict__ rec, const int meta, const int lane, float& px, float& py, float& pa) {
  const uint4* __restrict__ q4 = (const uint4*)rec;
  const uint4 w0 = q4[0], w1 = q4[1], w2 = q4[2], w3 = q4[3];
  const int ia = (meta >> 2) & 63, ib = (meta >> 8) & 63;
  const bool hA = (meta >> 14) & 1, hB = (meta >> 15) & 1;
  const int n = (meta >> 20) & 3, type = (meta >> 22) & 3;
  const float mA = hA ? S.mH : S.mW, iA = hA ? S.iH : S.iW, mB = hB ? S.mH : S.mW, iB = hB ? S.iH : S.iW;
  const V2 lcA = hA ? v2(S.lcx, S.lcy) : v2(0.0f, 0.0f), lcB = hB ? v2(S.lcx, S.lcy) : v2(0.0f, 0.0f);
  V2 cA = v2(rl(px, ia), rl(py, ia)); float aA = rl(pa, ia);
  V2 cB = v2(rl(px, ib), rl(py, ib)); float aB = rl(pa, ib);
  (void)w0.x; (void)w0.y;
  const V2 localNormal = v2(__uint_as_float(w0.z), __uint_as_float(w0.w)), localPoint = v2(__uint_as_float(w1.x), __uint_as_float(w1.y));
  const V2 lp0 = v2(__uint_as_float(w1.z), __uint_as_float(w1.w)), lp1 = v2(__uint_as_float(w2.w), __uint_as_float(w3.x));   // words 6,7 and 11,12
  const bool isA = lane == ia, isB = lane == ib;
  float minSep = 0.0f;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    if (j < n) {
      const Rot qm = rot_of(isA ? aA : (isB ? aB : pa));
      Xf xfA, xfB;
      xfA.q.s = rl(qm.s, ia); xfA.q.c = rl(qm.c, ia); xfB.q.s = rl(qm.s, ib); xfB.q.c = rl(qm.c, ib);
      xfA.p = cA - rmul(xfA.q, lcA); xfB.p = cB - rmul(xfB.q, lcB);
      const V2 lp = j == 0 ? lp0 : lp1;
      V2 normal, point; float separation;
      if (type == 1) {
        normal = rmul(xfA.q, localNormal);
        const V2 planePoint = xmul(xfA, localPoint);
        const V2 clip = xmul(xfB, lp);
        separation = dot(clip - planePoint, normal) - B2_POLYGON_RADIUS - B2_POLYGON_RADIUS;
        point = clip;
      } else {
        normal = rmul(xfB.q, localNormal);
        const V2 planePoint = xmul(xfB, localPoint);
        const V2 clip = xmul(xfA, lp);
        separation = dot(clip - planePoint, normal) - B2_POLYGON_RADIUS - B2_POLYGON_RADIUS;
        point = clip;
        normal = -normal;
      }
      const V2 rA = point - cA, rB = point - cB;
      minSep = mcr_min(minSep, separation);
      const float C = mcr_clamp(B2_BAUMGARTE * (separation + B2_LINEAR_SLOP), -B2_MAX_LINEAR_CORRECTION, 0.0f);
      const float rnA = cross(rA, normal), rnB = cross(rB, normal);
      const float K = mA + mB + iA * rnA * rnA + iB * rnB * rnB;
      const float impulse = K > 0.0f ? -C / K : 0.0f;
      const V2 P = impulse * normal;
      cA = cA - mA * P; aA -= iA * cross(rA, P);
      cB = cB + mB * P; aB += iB * cross(rB, P);
    }
  }
  px = isA ? cA.x : (isB ? cB.x : px); py = isA ? cA.y : (isB ? cB.y : py); pa = isA ? aA : (isB ? aB : pa);
  return minSep;
}

}  // namespace dyn

// mode 0: regular step (bookkeeping, TimeLimit, auto-reset install)
// mode 1: the action-less step of reset() (:408) for envs whose `resetting` flag is set
// debug bit 8 (256): lane 0 of every wavefront stamps the clock per phase (0 start, 1 state loaded + Car.step +
// velocity integration, 2 velocity sweeps done, 3 position loop done, 4 end) into p.dbg_stamps[block][8] (main launch, then the launches of roles 2, 3, 4)
#ifndef MCR_NO_PARK_WAIT
#define MCR_NO_PARK_WAIT 0            // (1: the missing wait of rounds 3-4, to see tests/test_gpu_parity.py::test_a_late_contact_pass... fail)
#endif
#define UNI_I(x) __builtin_amdgcn_readfirstlane(x)
#define DYN_STAMP(i) do { if ((p.debug & 256) && mode == 0 && threadIdx.x == 0) p.dbg_stamps[((p.role >= 2 ? (p.B * p.G + 63) / 64 + (p.role - 2) * ((p.B + 1) / 2) : 0) + blk) * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
// CC = false: a launch that cannot hold an env with touching car<->car contacts (the main launch of the three-chain step; contacts off;
// N = 1) — none of the contact code is compiled in, and the contact-free loops keep the registers and the schedule they get alone
// UNI (with CC): the launch holds ONE env per wavefront (the contact chain, role 2): the contact sweeps run in the uniform form above
template <bool CC, bool UNI = false, bool COOP = false>
__device__ __forceinline__ void dynamics_block(const McrParams& p, const int mode, const int blk) {
  static_assert(CC || !UNI, "the uniform contact sweeps are part of the contact build");
  using namespace dyn;
  DYN_STAMP(0);
  // LDS used only by waves that contain a touching car<->car pair
  __shared__ __attribute__((aligned(16))) float xv[15][64], xp[15][64];            // body exchange: (vx,vy,w) / (cx,cy,a) x 5 bodies per lane
  __shared__ __attribute__((aligned(16))) float vcpool[DYN_VC_POOL][cc::VC_SIZE];
  __shared__ uint32_t pcrec[DYN_VC_POOL][16];      // manifold records (key, type|n, local normal/point, 2 points) for the position sweeps
  __shared__ int xisl[64], xjok[64];
  __shared__ float xms[64];
  __shared__ int xmap[64];                            // per lane: slot of wheel w's registers at bits 2w (island order of the joints)
  const int g = blk * 64 + threadIdx.x;
  const int env = mcr_env_of_slot(p, mcr_dyn_slot(p, blk), true), agent = g % p.G;
  const int env_end = p.env0 + p.nenv;
  bool lane_ok = env < env_end && agent < p.N;
  const int ci = lane_ok ? env * p.N + agent : 0;
  const int BN = p.BN;
  // Deferral (contact side stream configuration): the main launch (role 1) gives the position loop p.defer_after
  // sweeps; an env with a car that is still iterating then (a slow marginal crawl, ~0.06 % of the env-steps, which
  // would otherwise hold the whole launch for up to 60 sweeps) parks its state and goes on the deferred list; the
  // resume launch (role 3, own stream) reloads it, runs the remaining sweeps and the whole tail of the step.
  const bool resume = p.role == 3 && mode == 0;
  const int defer_cap = (p.role == 1 && mode == 0) ? p.defer_after : 0;
  McrEnvState es;
  if (env < env_end) es = p.env[env]; else { es.active = 0; es.resetting = 0; }
  bool run = lane_ok && es.active;
  if (mode == 1) run = run && es.resetting;
  // Thaw: an env that finished while the host had not staged its next episode yet was frozen (inactive, zero
  // outputs).  As soon as the staged slot is filled, the next step's main launch re-spawns it exactly like the
  // auto-reset of a `done` step does (install -> reset pass -> first observation); reward/done of that step are 0.
  const bool frozen_now = lane_ok && mode == 0 && p.role <= 1 && !es.active && es.frozen && p.auto_reset;
  const bool thaw = frozen_now && es.staged_ready;
  if (frozen_now && agent == 0) atomicAdd(&p.counters[3], 1ull);          // env-steps that produced nothing

  const McrShapes& S = *p.shapes;
  const float mH = S.hull_invMass, iH = S.hull_invI, mW = S.wheel_invMass, iW = S.wheel_invI;
  const float lcx = S.hull_lcx, lcy = S.hull_lcy;
  const CcMass CM{mH, iH, mW, iW, lcx, lcy};
  const float h = (float)(1.0 / MCR_FPS);
  const double dt = 1.0 / MCR_FPS;

  Body b[5]; Joint J[4];
  double gas[2] = {0, 0}, steer = 0, brake = 0, omega[4] = {0, 0, 0, 0}, phase[4] = {0, 0, 0, 0};
  float sleepT[5] = {0, 0, 0, 0, 0};
  uint32_t onroad = 0;
#pragma unroll
  for (int k = 0; k < 5; ++k) { b[k].cx = b[k].cy = b[k].a = b[k].vx = b[k].vy = b[k].w = 0.0f; }
#pragma unroll
  for (int k = 0; k < 4; ++k) { J[k].ix = J[k].iy = J[k].iz = J[k].im = 0.0f; J[k].limit = 0; J[k].motorSpeed = 0.0f; }

  if (run) {
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      b[k].cx = p.carf[(CF_CX + k) * BN + ci]; b[k].cy = p.carf[(CF_CY + k) * BN + ci]; b[k].a = p.carf[(CF_A + k) * BN + ci];
      b[k].vx = p.carf[(CF_VX + k) * BN + ci]; b[k].vy = p.carf[(CF_VY + k) * BN + ci]; b[k].w = p.carf[(CF_W + k) * BN + ci];
      sleepT[k] = p.carf[(CF_SLEEP + k) * BN + ci];
    }
    uint32_t lim = p.caru[CU_LIMIT * BN + ci];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      J[k].ix = p.carf[(CF_JIX + k) * BN + ci]; J[k].iy = p.carf[(CF_JIY + k) * BN + ci];
      J[k].iz = p.carf[(CF_JIZ + k) * BN + ci]; J[k].im = p.carf[(CF_JM + k) * BN + ci];
      J[k].limit = (lim >> (2 * k)) & 3;
      omega[k] = p.card[(CD_OMEGA + k) * BN + ci]; phase[k] = p.card[(CD_PHASE + k) * BN + ci];
    }
    gas[0] = p.card[(CD_GAS + 0) * BN + ci]; gas[1] = p.card[(CD_GAS + 1) * BN + ci];
    steer = p.card[CD_STEER * BN + ci]; brake = p.card[CD_BRAKE * BN + ci];
    onroad = p.caru[CU_ONROAD * BN + ci];

  }
  if (run) {
    if (!resume) {
    // ---- controls (:418-424) — the reference negates the steering input
    if (mode == 0 && p.actions) {
      double a0 = (double)p.actions[ci * 3 + 0], a1 = (double)p.actions[ci * 3 + 1], a2 = (double)p.actions[ci * 3 + 2];
      steer = -a0;
      double gg = fmin(fmax(a1, 0.0), 1.0);
#pragma unroll
      for (int k = 0; k < 2; ++k) { double diff = gg - gas[k]; if (diff > 0.1) diff = 0.1; gas[k] += diff; }
      brake = a2;
    }

    // ---- Car.step(dt): tyre model, f64.  Friction uses the PREVIOUS Collide's contact set (CU_ONROAD).
    const double ENGINE_POWER = 100000000 * MCR_SIZE * MCR_SIZE;
    const double WHEEL_MOI = 4000 * MCR_SIZE * MCR_SIZE;
    const double FRICTION_LIMIT = 1000000 * MCR_SIZE * MCR_SIZE;
    const double WHEEL_RAD = MCR_WHEEL_R * MCR_SIZE;
    const double KF = 205000 * MCR_SIZE * MCR_SIZE;
    float fx[4], fy[4];
    uint32_t skid = 0;                                           // bit k: wheel k skids ("Skid trace" of Car.step, particles only)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      double wsteer = (k < 2) ? steer : 0.0;
      double wgas = (k >= 2) ? gas[k - 2] : 0.0;
      double jangle = (double)(b[k + 1].a - b[0].a);
      double dir = np_sign(wsteer - jangle);
      double val = fabs(wsteer - jangle);
      J[k].motorSpeed = (float)(dir * fmin(50.0 * val, 3.0));
      double friction_limit = FRICTION_LIMIT * 0.6;
      if ((onroad >> k) & 1u) friction_limit = fmax(friction_limit, FRICTION_LIMIT * 1.0);
      Rot q = rot_of(b[k + 1].a);
      double forw0 = (double)(-q.s), forw1 = (double)q.c, side0 = (double)q.c, side1 = (double)q.s;
      double vx = (double)b[k + 1].vx, vy = (double)b[k + 1].vy;
      double vf = forw0 * vx + forw1 * vy;
      double vs = side0 * vx + side1 * vy;
      double om = omega[k];
      om += dt * ENGINE_POWER * wgas / WHEEL_MOI / (fabs(om) + 5.0);
      if (brake >= 0.9) om = 0;
      else if (brake > 0) {
        double d = -np_sign(om);
        double v = 15 * brake;
        if (fabs(v) > fabs(om)) v = fabs(om);
        om += d * v;
      }
      phase[k] += om * dt;
      double vr = om * WHEEL_RAD;
      double f_force = -vf + vr;
      double p_force = -vs;
      f_force *= KF; p_force *= KF;
      double force = sqrt(f_force * f_force + p_force * p_force);
      if (fabs(force) > 2.0 * friction_limit) skid |= 1u << k;
      if (fabs(force) > friction_limit) {
        f_force /= force; p_force /= force;
        force = friction_limit;
        f_force *= force; p_force *= force;
      }
      om -= dt * f_force * WHEEL_RAD / WHEEL_MOI;
      omega[k] = om;
      fx[k] = (float)(p_force * side0 + f_force * forw0);
      fy[k] = (float)(p_force * side1 + f_force * forw1);
    }
    if (p.particles) {                                           // w.position: the wheel bodies have their origin at the centre
      uint32_t* pc = p.particles + (size_t)ci * MCR_PART_WORDS;
      for (int k = 0; k < 4; ++k) mcr_particle_step(pc, k, ((skid >> k) & 1u) != 0u, !((onroad >> k) & 1u), b[k + 1].cx, b[k + 1].cy);
    }

    // ---- b2Island::Solve: integrate velocities
    b[0].vx = b[0].vx + h * (mH * 0.0f); b[0].vy = b[0].vy + h * (mH * 0.0f); b[0].w += h * iH * 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      b[k + 1].vx = b[k + 1].vx + h * (mW * (0.0f + fx[k])); b[k + 1].vy = b[k + 1].vy + h * (mW * (0.0f + fy[k]));
      b[k + 1].w += h * iW * 0.0f;
    }
    }   // !resume
  }

  // ---- car<->car contacts of this env (manifolds prepared by the collide kernel).  The common case — no
  // touching pair in the whole wave — skips every contact block below with one wave-uniform branch.
  const int lane = threadIdx.x;
  const int leader_lane = lane - agent;
  uint32_t* store = p.cc_store + (size_t)(env < env_end ? env : 0) * (MCR_CC_MAX * MCR_CC_WORDS + 4);
  int ccn = 0;
  // (cc_mode: the main launch's envs are those whose verdict — mcr_touch_verdict, evaluated by last step's bookkeeping on the
  // same poses with the same arithmetic — says that no car<->car fixture pair touches: k_collide, running beside this
  // launch, finds none either; its store[0] is not read)
  if (CC && run && p.car_contacts && p.N > 1 && !(p.cc_mode && mode == 0 && p.role == 1)) ccn = (int)store[0];
  const bool wave_cc = CC && __any(ccn > 0) != 0;
  DYN_STAMP(1);
  int pool_base = 0;
  int isl = agent;                        // island id of this car = lowest car id linked to it by touching contacts
  int cmeta = 0; CcImp cimp; cimp.n1 = cimp.t1 = cimp.n2 = cimp.t2 = 0.0f;   // UNI: lane i holds the description and the accumulated impulses of manifold i
  if (wave_cc) {
    // LDS pool of velocity-constraint records: exclusive prefix sum of the leaders' needs across the wave
    int need = (agent == 0) ? ccn : 0;
    int incl = need;
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
    pool_base = incl - need;
    if (agent == 0 && pool_base + ccn > DYN_VC_POOL) { ccn = DYN_VC_POOL - pool_base; if (ccn < 0) ccn = 0; store[1] = 2u; mcr_raise(p, ST_CC_OVERFLOW); }
    ccn = __shfl(ccn, leader_lane); pool_base = __shfl(pool_base, leader_lane);
    // island order of this car's joints (3,2,1,0 unless a contact says otherwise); wq[t] = the wheel slot t holds.  (Everything that
    // knows about the permutation lives in the contact branches: the contact-free loops keep their registers.)
    int jord8 = 0x1b, wq[4] = {0, 1, 2, 3};
    if (ccn > 0) jord8 = (int)((store[2 + (agent >> 2)] >> ((agent & 3) * 8)) & 255u);
    const bool wave_perm = __any(jord8 != 0x1b) != 0;
    if (wave_perm) {
      int key[4];                                       // joint q goes to slot 3 - (its position in the island order)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int pos = 0;
#pragma unroll
        for (int t = 1; t < 4; ++t) if (((jord8 >> (2 * t)) & 3) == q) pos = t;
        key[q] = 3 - pos; wq[q] = q;
      }
      slot_sort(J, b, key, wq);
    }
    {
      int m = 0;
#pragma unroll
      for (int t = 0; t < 4; ++t) m |= t << (2 * wq[t]);
      xmap[lane] = m;
    }
    if (ccn > 0) {
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        xv[0 * 5 + k][lane] = b[k].vx; xv[1 * 5 + k][lane] = b[k].vy; xv[2 * 5 + k][lane] = b[k].w;
        xp[0 * 5 + k][lane] = b[k].cx; xp[1 * 5 + k][lane] = b[k].cy; xp[2 * 5 + k][lane] = b[k].a;
      }
    }
    __syncthreads();
    if (ccn > 0 && agent == 0) {
      int root[MCR_MAX_AGENTS];
#pragma unroll
      for (int c = 0; c < MCR_MAX_AGENTS; ++c) root[c] = c;
      for (int i = 0; i < ccn; ++i) {
        const uint32_t* rec = store + 4 + i * MCR_CC_WORDS;
        cc_init(CM, rec, i, leader_lane, xp, xv, xmap, vcpool[pool_base + i]);
#pragma unroll
        for (int w = 0; w < 16; ++w) pcrec[pool_base + i][w] = rec[w];      // once per step instead of one HBM round trip per sweep
        pcrec[pool_base + i][0] |= cc_rows_of(rec[0], leader_lane, xmap);   // (looked up once, not in every position sweep)
        // union-find over cars (b2World::Solve island DFS through touching contacts)
        const int ca = rec[0] & 15, cb = (rec[0] >> 8) & 15;
        int ra = ca, rb = cb;
        for (int t = 0; t < MCR_MAX_AGENTS; ++t) {
          int na = ra, nb = rb;
#pragma unroll
          for (int c = 0; c < MCR_MAX_AGENTS; ++c) { if (c == ra) na = root[c]; if (c == rb) nb = root[c]; }
          ra = na; rb = nb;
        }
        if (ra != rb) {
          const int hi = ra > rb ? ra : rb, lo = ra > rb ? rb : ra;
#pragma unroll
          for (int c = 0; c < MCR_MAX_AGENTS; ++c) if (c == hi) root[c] = lo;
        }
      }
#pragma unroll
      for (int c = 0; c < MCR_MAX_AGENTS; ++c) {
        int r = c;
        for (int t = 0; t < MCR_MAX_AGENTS; ++t) {
          int nr = r;
#pragma unroll
          for (int d = 0; d < MCR_MAX_AGENTS; ++d) if (d == r) nr = root[d];
          r = nr;
        }
        if (c < p.N) xisl[leader_lane + c] = r;
      }
    }
    __syncthreads();
    if (ccn > 0) {
      isl = xisl[lane];
#pragma unroll
      for (int k = 0; k < 5; ++k) { b[k].vx = xv[0 * 5 + k][lane]; b[k].vy = xv[1 * 5 + k][lane]; b[k].w = xv[2 * 5 + k][lane]; }
    }
    if constexpr (UNI) {
      // lane i keeps the description of manifold i (cmeta, see cc_velocity_bl); body lane of (car c, exchange row r) = 5 c + r
      // (the env's cars sit in lanes 0 .. N - 1, its leader is lane 0 and its pool starts at 0: the lanes beyond the env hold manifolds too)
      if (lane < UNI_I(ccn)) {
        const float* vc = vcpool[lane];
        const int sa = ((const int*)vc)[cc::VC_LA], sb = ((const int*)vc)[cc::VC_LB];
        const uint32_t* rec = pcrec[lane];
        const int ra = sa & 7, rb = sb & 7;
        cmeta = ((const int*)vc)[cc::VC_N] | (((sa >> 3) * 5 + ra) << 2) | (((sb >> 3) * 5 + rb) << 8) | ((ra == 0 ? 1 : 0) << 14) | ((rb == 0 ? 1 : 0) << 15) |
                (xisl[(int)(rec[0] & 15u)] << 16) | ((int)((rec[1] >> 8) & 3u) << 20) | ((int)(rec[1] & 3u) << 22);
        cimp.n1 = vc[cc::VC_P0 + 4]; cimp.t1 = vc[cc::VC_P0 + 5]; cimp.n2 = vc[cc::VC_P1 + 4]; cimp.t2 = vc[cc::VC_P1 + 5];
      }
    }
    if (run && !resume) {
      // joints, slot order 3,2,1,0 = the car's island order
#pragma unroll
      for (int q = 3; q >= 0; --q) {
        float ax = S.anchor_x[q], ay = S.anchor_y[q];
        if (wave_perm) { ax = pick4(wq[q], S.anchor_x[0], S.anchor_x[1], S.anchor_x[2], S.anchor_x[3]); ay = pick4(wq[q], S.anchor_y[0], S.anchor_y[1], S.anchor_y[2], S.anchor_y[3]); }
        joint_init(J[q], b[0], b[q + 1], ax, ay, lcx, lcy, mH, iH, mW, iW);
      }
    }
  }

  bool positionSolved = false;
  if (run && !resume && !wave_cc) {
    // joints, island order 3,2,1,0
#pragma unroll
    for (int q = 3; q >= 0; --q) joint_init(J[q], b[0], b[q + 1], S.anchor_x[q], S.anchor_y[q], lcx, lcy, mH, iH, mW, iW);
  }
  if (resume && run && agent == 0) atomicAdd(&p.counters[1], 1ull);
  if (resume) {   // the position solver needs nothing from InitVelocityConstraints but the motor mass (limit state is stored)
#pragma unroll
    for (int q = 0; q < 4; ++q) { float mm = iH + iW; if (mm > 0.0f) mm = 1.0f / mm; J[q].motorMass = mm; }
  }
  const float maxImpulse = h * (float)(180 * 900 * MCR_SIZE * MCR_SIZE);
  if (resume) {
    // velocity phase already done by the main launch
  } else if (!wave_cc) {
    // hot loop of the common case: nothing but the four revolute joints, state in registers
    if (mode == 1) {
      // reset pass (step(None) on freshly spawned cars): the iteration map is a pure function of (velocities,
      // accumulated impulses); once one sweep leaves all of them unchanged every later sweep does too, so the
      // remaining sweeps are skipped — same final state as the full 180, a fraction of the serial chain.
      for (int it = 0; it < 180; ++it) {
        bool changed = false;
        if (run) {
          Body ob[5]; float oi[16];
#pragma unroll
          for (int k = 0; k < 5; ++k) ob[k] = b[k];
#pragma unroll
          for (int q = 0; q < 4; ++q) { oi[q * 4] = J[q].ix; oi[q * 4 + 1] = J[q].iy; oi[q * 4 + 2] = J[q].iz; oi[q * 4 + 3] = J[q].im; }
#pragma unroll
          for (int q = 3; q >= 0; --q) joint_velocity_as<!CC>(J[q], b[0], b[q + 1], mH, iH, mW, iW, maxImpulse);
#pragma unroll
          for (int k = 0; k < 5; ++k) changed = changed || ob[k].vx != b[k].vx || ob[k].vy != b[k].vy || ob[k].w != b[k].w;
#pragma unroll
          for (int q = 0; q < 4; ++q) changed = changed || oi[q * 4] != J[q].ix || oi[q * 4 + 1] != J[q].iy || oi[q * 4 + 2] != J[q].iz || oi[q * 4 + 3] != J[q].im;
        }
        if (!__any(changed)) break;
      }
    } else if (run) {
      // The rear wheels (joints 2, 3: not steered, a motor holds them straight) reach their limits in crashes only: a wavefront none of
      // whose cars has one there runs its 180 sweeps with the limit-free form of those two joints (36 instructions less per sweep).
      // ... and a wavefront none of whose cars has ANY joint at a limit (a policy that does not hold the steering at its stop: the steered joints
      // sit at +-0.4 rad only while |steer| = 1 has been held for ~0.1 s) runs the limit-free form of all four: 111 instructions per sweep
      // instead of 199.  (Random actions over [-1, 1]: nearly every wavefront of 64 cars holds a steered joint at its stop.)
      if (!CC && !__any(J[0].limit != 0 || J[1].limit != 0 || J[2].limit != 0 || J[3].limit != 0)) {
        for (int it = 0; it < 180; ++it) {
#pragma unroll
          for (int q = 3; q >= 0; --q) joint_velocity_as<!CC, false>(J[q], b[0], b[q + 1], mH, iH, mW, iW, maxImpulse);
        }
      } else if (!__any(J[2].limit != 0 || J[3].limit != 0)) {
        for (int it = 0; it < 180; ++it) {
          joint_velocity_as<!CC, false>(J[3], b[0], b[4], mH, iH, mW, iW, maxImpulse);
          joint_velocity_as<!CC, false>(J[2], b[0], b[3], mH, iH, mW, iW, maxImpulse);
          joint_velocity_as<!CC, true>(J[1], b[0], b[2], mH, iH, mW, iW, maxImpulse);
          joint_velocity_as<!CC, true>(J[0], b[0], b[1], mH, iH, mW, iW, maxImpulse);
        }
      } else {
        for (int it = 0; it < 180; ++it) {
#pragma unroll
          for (int q = 3; q >= 0; --q) joint_velocity_as<!CC>(J[q], b[0], b[q + 1], mH, iH, mW, iW, maxImpulse);
        }
      }
    }
  } else {
    const int vel_iters = (p.debug & 128) ? 2 : 180;          // debug bit 7: timing experiments only
#ifdef MCR_POSLOOP_PROFILE
    unsigned long long vt[4] = {0, 0, 0, 0}, vt0 = 0;
#define VP_MARK(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); vt[i] += now_ - vt0; vt0 = now_; } while (0)
#define VP_BEGIN() do { vt0 = __builtin_readcyclecounter(); } while (0)
#else
#define VP_MARK(i) do {} while (0)
#define VP_BEGIN() do {} while (0)
#endif
    if constexpr (UNI) {
      const int ccnu = UNI_I(ccn);
      float4* const xt = (float4*)&xv[0][0];              // the transposition buffer: xt[body lane] = (vx, vy, w, -)
      // (a car's joints reach their limits at full steering lock or in a crash: the env none of whose joints is at one — the state is a constant
      // of the step — runs its sweeps with the limit-free form of all four, chosen ONCE: no branch inside the loop)
      // (... and so is the manifold count: an env with up to FOUR manifolds — all but a handful per rollout — keeps their constants in registers for
      // the whole phase: 530-570 ticks per manifold and sweep with one or two, 680-810 with three or four (registers run out: the joints' state
      // starts to travel through AGPRs); beyond four they rotate through a record loaded a manifold ahead (960-1190 ticks: ~26 register moves,
      // 12 LDS reads and ~40 AGPR moves per manifold).  Measured and not kept: two records used alternately instead of the rotation (1080
      // ticks), the constants in lane registers read out with v_readlane (slower than the LDS prefetch); round 6: for three or more manifolds the
      // constants RE-READ from LDS in every sweep, three manifolds at a time, nothing of them live across the joints — the loop's AGPR moves fall
      // from 124 to 14 per sweep (ISA) and --actions drive from 5.56 to 5.21 M env-steps/s (same box, alternating, twice): the reads' latency
      // sits on the chain.  tools/posloop_profile.py VEL=1)
      auto sweeps = [&](auto lim_tag, auto cnt_tag) {
        constexpr bool LIM = decltype(lim_tag)::value;
        constexpr int CNT = decltype(cnt_tag)::value;     // 1 .. 4: exactly that many manifolds; 0: any number
        VcRec cur = vc_load(vcpool[0]);
        VcRec second = cur, third = cur, fourth = cur;
        if constexpr (CNT >= 2) second = vc_load(vcpool[1]);
        if constexpr (CNT >= 3) third = vc_load(vcpool[2]);
        if constexpr (CNT >= 4) fourth = vc_load(vcpool[3]);
        const int meta0 = __builtin_amdgcn_readlane(cmeta, 0), meta1 = __builtin_amdgcn_readlane(cmeta, 1), meta2 = __builtin_amdgcn_readlane(cmeta, 2), meta3 = __builtin_amdgcn_readlane(cmeta, 3);
        for (int it = 0; it < vel_iters; ++it) {
          VP_BEGIN();
          if (run) {
#pragma unroll
            for (int q = 3; q >= 0; --q) joint_velocity_as<!LIM, LIM>(J[q], b[0], b[q + 1], mH, iH, mW, iW, maxImpulse);   // pairs for the limit-free form only (with limits: N=8 +0.3 %, N=4 -1 %, drive -0.3 %: nothing)
          }
          VP_MARK(0);
          if (run) {
#pragma unroll
            for (int k = 0; k < 5; ++k) xt[agent * 5 + k] = make_float4(b[k].vx, b[k].vy, b[k].w, 0.0f);
          }
          const float4 c4 = xt[lane];
          float cx = c4.x, cy = c4.y, cw = c4.z;
          VP_MARK(1);
          if constexpr (CNT == 1) cc_velocity_bl(CM, cur, meta0, cimp, 0, lane, cx, cy, cw);
          else if constexpr (CNT == 2) { cc_velocity_bl(CM, cur, meta0, cimp, 0, lane, cx, cy, cw); cc_velocity_bl(CM, second, meta1, cimp, 1, lane, cx, cy, cw); }
          else if constexpr (CNT == 3) { cc_velocity_bl(CM, cur, meta0, cimp, 0, lane, cx, cy, cw); cc_velocity_bl(CM, second, meta1, cimp, 1, lane, cx, cy, cw); cc_velocity_bl(CM, third, meta2, cimp, 2, lane, cx, cy, cw); }
          else if constexpr (CNT == 4) { cc_velocity_bl(CM, cur, meta0, cimp, 0, lane, cx, cy, cw); cc_velocity_bl(CM, second, meta1, cimp, 1, lane, cx, cy, cw); cc_velocity_bl(CM, third, meta2, cimp, 2, lane, cx, cy, cw); cc_velocity_bl(CM, fourth, meta3, cimp, 3, lane, cx, cy, cw); }
          else
          for (int i = 0; i < ccnu; ++i) {
            const VcRec nxt = vc_load(vcpool[i + 1 == ccnu ? 0 : i + 1]);     // the next manifold's constants, a manifold ahead of their use
            cc_velocity_bl(CM, cur, __builtin_amdgcn_readlane(cmeta, i), cimp, i, lane, cx, cy, cw);
            cur = nxt;
          }
          VP_MARK(2);
          xt[lane] = make_float4(cx, cy, cw, 0.0f);
          if (run) {
#pragma unroll
            for (int k = 0; k < 5; ++k) { const float4 v = xt[agent * 5 + k]; b[k].vx = v.x; b[k].vy = v.y; b[k].w = v.z; }
          }
          VP_MARK(3);
        }
      };
      bool any_limit = false;
#pragma unroll
      for (int q = 0; q < 4; ++q) any_limit = any_limit || J[q].limit != 0;
      const bool lim_any = __any(any_limit) != 0;
      using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>; using I4 = std::integral_constant<int, 4>;
      if (ccnu == 1) { if (lim_any) sweeps(std::true_type{}, I1{}); else sweeps(std::false_type{}, I1{}); }
      else if (ccnu == 2) { if (lim_any) sweeps(std::true_type{}, I2{}); else sweeps(std::false_type{}, I2{}); }
      else if (ccnu == 3) { if (lim_any) sweeps(std::true_type{}, I3{}); else sweeps(std::false_type{}, I3{}); }
      else if (ccnu == 4) { if (lim_any) sweeps(std::true_type{}, I4{}); else sweeps(std::false_type{}, I4{}); }
      else { if (lim_any) sweeps(std::true_type{}, I0{}); else sweeps(std::false_type{}, I0{}); }
      if (lane < ccnu) { float* vc = vcpool[lane]; vc[cc::VC_P0 + 4] = cimp.n1; vc[cc::VC_P0 + 5] = cimp.t1; vc[cc::VC_P1 + 4] = cimp.n2; vc[cc::VC_P1 + 5] = cimp.t2; }   // for StoreImpulses
    } else
    for (int it = 0; it < vel_iters; ++it) {
      VP_BEGIN();
      if (run) {
#pragma unroll
        for (int q = 3; q >= 0; --q) joint_velocity_scalar(J[q], b[0], b[q + 1], mH, iH, mW, iW, maxImpulse);
      }
      VP_MARK(0);
      if (ccn > 0 && !(p.debug & 1024)) {
#pragma unroll
        for (int k = 0; k < 5; ++k) { xv[0 * 5 + k][lane] = b[k].vx; xv[1 * 5 + k][lane] = b[k].vy; xv[2 * 5 + k][lane] = b[k].w; }
      }
      __syncthreads();
      VP_MARK(1);
      if (ccn > 0 && agent == 0 && !(p.debug & 512)) for (int i = 0; i < ccn; ++i) cc_velocity(CM, vcpool[pool_base + i], xv);
      __syncthreads();
      VP_MARK(2);
      if (ccn > 0 && !(p.debug & 1024)) {
#pragma unroll
        for (int k = 0; k < 5; ++k) { b[k].vx = xv[0 * 5 + k][lane]; b[k].vy = xv[1 * 5 + k][lane]; b[k].w = xv[2 * 5 + k][lane]; }
      }
      VP_MARK(3);
    }
#ifdef MCR_POSLOOP_PROFILE
    if ((p.debug & 256) && (p.debug & 65536) && mode == 0 && threadIdx.x == 0 && p.role == 2) {
      unsigned long long* o = p.dbg_stamps + ((size_t)((p.B * p.G + 63) / 64) + blk) * 8;
      o[5] = vt[0] | (vt[1] << 32); o[6] = vt[2] | (vt[3] << 32); o[7] = (unsigned long long)ccn;
    }
#endif
  }
  if (wave_cc && ccn > 0 && agent == 0) {            // StoreImpulses
    for (int i = 0; i < ccn; ++i) {
      const float* vc = vcpool[pool_base + i];
      uint32_t* rec = store + 4 + i * MCR_CC_WORDS;
      const int n = ((const int*)vc)[cc::VC_N];
      for (int j = 0; j < n; ++j) { const float* q = vc + (j == 0 ? cc::VC_P0 : cc::VC_P1); rec[6 + j * 5 + 2] = __float_as_uint(q[4]); rec[6 + j * 5 + 3] = __float_as_uint(q[5]); }
    }
  }
  DYN_STAMP(2);
  if (run && !resume) {
    // integrate positions
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      float tx = h * b[k].vx, ty = h * b[k].vy;
      if (tx * tx + ty * ty > B2_MAX_TRANSLATION * B2_MAX_TRANSLATION) {
        float ratio = B2_MAX_TRANSLATION / sqrtf(tx * tx + ty * ty);
        b[k].vx = ratio * b[k].vx; b[k].vy = ratio * b[k].vy;
      }
      float rot = h * b[k].w;
      if (rot * rot > B2_MAX_ROTATION * B2_MAX_ROTATION) { float ratio = B2_MAX_ROTATION / fabsf(rot); b[k].w *= ratio; }
      b[k].cx = b[k].cx + h * b[k].vx; b[k].cy = b[k].cy + h * b[k].vy; b[k].a += h * b[k].w;
    }
  }
  // position iterations with the per-island early exit
  RotCache hull_rot; hull_rot.a = __int_as_float(0x7fc00000); hull_rot.q.s = 0.0f; hull_rot.q.c = 1.0f;
  bool unfinished = false;                // deferral: this lane used up the main launch's sweeps without an outcome
  if (!wave_cc) {
    if (run) {
      int it0 = 0, pos_iters = (p.debug & 64) ? 2 : 60;       // debug bit 6: cap the position iterations (timing experiments only)
      if (defer_cap > 0 && defer_cap < pos_iters) pos_iters = defer_cap;
      bool stuck = false;
      if (resume) {
        const uint32_t stt = p.defer_state[ci];               // 0: keep iterating, 1: solved, 2: failed at a fixed point
        it0 = (stt == 0u) ? p.defer_after : 60;
        positionSolved = stt == 1u;
      }
      int it = it0;
      for (; it < pos_iters; ++it) {
        // (the fixed-point test below costs 50 of a sweep's 720 instructions: it runs on every fourth sweep — the main launch's last one
        // among them — since a fixed point, once reached, is still there three sweeps later, with the same poses)
        const bool probe = (it & 3) == 2;
        float ox[5], oy[5], oa[5];                              // (written and read on probing sweeps only)
        if (probe) {
#pragma unroll
          for (int k = 0; k < 5; ++k) { ox[k] = b[k].cx; oy[k] = b[k].cy; oa[k] = b[k].a; }
        }
        bool ok = true;
#pragma unroll
        for (int q = 3; q >= 0; --q) {
          bool jo = joint_position(J[q], b[0], b[q + 1], S.anchor_x[q], S.anchor_y[q], lcx, lcy, mH, iH, mW, iW, hull_rot);
          ok = ok && jo;
        }
        if (ok) { positionSolved = true; break; }
        // A sweep is a pure function of the 15 position values.  One that fails AND moves nothing (the marginal
        // case: an error one ulp above its slop whose correction rounds away) would repeat identically up to
        // iteration 60 — stop here with the same outcome (positionSolved stays false, positions as they are).
        if (probe) {
          bool moved = false;
#pragma unroll
          for (int k = 0; k < 5; ++k) moved = moved || ox[k] != b[k].cx || oy[k] != b[k].cy || oa[k] != b[k].a;
          if (!moved) { stuck = true; break; }
        }
      }
      unfinished = defer_cap > 0 && defer_cap < 60 && !positionSolved && !stuck && it == pos_iters;
      if (defer_cap > 0) p.defer_state[ci] = positionSolved ? 1u : (stuck ? 2u : 0u);
    }
  } else {
    bool active = run;
    int wq[4];                                           // the wheel each slot holds (see the permutation above)
    {
      const int m = xmap[lane];
#pragma unroll
      for (int t = 0; t < 4; ++t) { wq[t] = 0;
#pragma unroll
        for (int w = 1; w < 4; ++w) if (((m >> (2 * w)) & 3) == t) wq[t] = w; }
    }
    const bool wave_perm = __any(xmap[lane] != 0xe4) != 0;
    // Island bookkeeping of the sweeps in registers: imask = the lanes of this car's island (constant over the loop); who is still
    // iterating, whose joints are within tolerance and who moved travel as ballots, the leader's per-island contact verdicts as one
    // shuffle — no LDS round trips (they were a quarter of a sweep: 8 dependent reads per lane at N = 8).
    unsigned long long imask = 1ull << lane;
    if (ccn > 0) {
      imask = 0ull;
      for (int c = 0; c < p.N; ++c) if (xisl[leader_lane + c] == isl) imask |= 1ull << (leader_lane + c);
      if (agent == 0) for (int i = 0; i < ccn; ++i) pcrec[pool_base + i][0] |= (uint32_t)xisl[leader_lane + (int)(pcrec[pool_base + i][0] & 15u)] << 24;   // the contact's island
    }
    __syncthreads();
    // the joints' anchors of this car's slot order, ONCE per step and opaque to the optimiser: left to itself it re-reads them from the
    // shapes table inside the position sweeps (registers are short here) — 20 global loads and 350 SGPR reloads per sweep, each load a
    // round trip to L2 on the sweep's dependent chain (found in round 5's ISA; the sweep took 5.9 us)
    float jax[4], jay[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      jax[q] = S.anchor_x[q]; jay[q] = S.anchor_y[q];
      if (wave_perm) { jax[q] = pick4(wq[q], S.anchor_x[0], S.anchor_x[1], S.anchor_x[2], S.anchor_x[3]); jay[q] = pick4(wq[q], S.anchor_y[0], S.anchor_y[1], S.anchor_y[2], S.anchor_y[3]); }
      asm volatile("" : "+v"(jax[q]), "+v"(jay[q]));
    }
    const int pos_iters_cc = (p.debug & 64) ? 2 : 60;
#ifdef MCR_POSLOOP_PROFILE        // build-time diagnostic (MCR_EXTRA_CFLAGS=-DMCR_POSLOOP_PROFILE, tools/posloop_profile.py): where a contact position sweep goes
    unsigned long long pt[5] = {0, 0, 0, 0, 0}, pt0 = 0; int pn = 0;
#define PP_MARK(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); pt[i] += now_ - pt0; pt0 = now_; } while (0)
#define PP_BEGIN() do { pt0 = __builtin_readcyclecounter(); ++pn; } while (0)
#else
#define PP_MARK(i) do {} while (0)
#define PP_BEGIN() do {} while (0)
#endif
    if constexpr (UNI) {
      // the uniform form: contacts (every lane, on the owners' registers), then the joints, island verdicts by ballots — no LDS traffic but
      // the constants of the manifolds
      const int ccnu = UNI_I(ccn);
      float4* const xt = (float4*)&xp[0][0];              // the transposition buffer: xt[body lane] = (c.x, c.y, angle, -)
      for (int it = 0; it < pos_iters_cc; ++it) {
        const unsigned long long actm = __ballot(active);
        if (!actm) break;
        PP_BEGIN();
        // (the fixed-point probe — did the sweep move anything? — on every fourth sweep, the last one among them: a fixed point is still
        // there three sweeps later, with the same poses; 15 values kept and compared, a twelfth of a sweep)
        const bool probe = (it & 3) == 3;
        float ox[5], oy[5], oa[5];
        if (probe) {
#pragma unroll
          for (int k = 0; k < 5; ++k) { ox[k] = b[k].cx; oy[k] = b[k].cy; oa[k] = b[k].a; }
        }
        PP_MARK(0);
        float myMin = 0.0f;                               // min separation over the contacts of this car's island
        {
          if (run) {
#pragma unroll
            for (int k = 0; k < 5; ++k) xt[agent * 5 + k] = make_float4(b[k].cx, b[k].cy, b[k].a, 0.0f);
          }
          const float4 p4 = xt[lane];
          float px = p4.x, py = p4.y, pa = p4.z;
          for (int i = 0; i < ccnu; ++i) {
            const int meta = __builtin_amdgcn_readlane(cmeta, i);
            const int r = (meta >> 16) & 15;
            if (!((actm >> r) & 1ull)) continue;
            const float ms = cc_position_bl(CM, pcrec[i], meta, lane, px, py, pa);
            myMin = (isl == r) ? mcr_min(myMin, ms) : myMin;
          }
          xt[lane] = make_float4(px, py, pa, 0.0f);
          if (active) {
#pragma unroll
            for (int k = 0; k < 5; ++k) { const float4 v = xt[agent * 5 + k]; b[k].cx = v.x; b[k].cy = v.y; b[k].a = v.z; }
          }
        }
        PP_MARK(1);
        bool jointsOk = true;
        if (active) {
#pragma unroll
          for (int q = 3; q >= 0; --q) {
            bool jo = joint_position(J[q], b[0], b[q + 1], jax[q], jay[q], lcx, lcy, mH, iH, mW, iW, hull_rot);
            jointsOk = jointsOk && jo;
          }
        }
        PP_MARK(2);
        bool moved = true;
        if (probe) {
          moved = false;
#pragma unroll
          for (int k = 0; k < 5; ++k) moved = moved || ox[k] != b[k].cx || oy[k] != b[k].cy || oa[k] != b[k].a;
        }
        const unsigned long long okm = __ballot(jointsOk), mvm = probe ? __ballot(moved) : ~0ull;
        if (active) {
          bool ok = jointsOk, mv = moved;
          if (ccn > 0) { ok = myMin >= -3.0f * B2_LINEAR_SLOP && (okm & imask) == imask; mv = (mvm & imask) != 0ull; }
          if (ok) { positionSolved = true; active = false; }
          else if (!mv) active = false;
        }
        PP_MARK(3);
      }
    } else {
    for (int it = 0; it < pos_iters_cc; ++it) {
      const unsigned long long actm = __ballot(active);
      if (!actm) break;
      PP_BEGIN();
      const bool in_cc = active && ccn > 0;
      float ox[5], oy[5], oa[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) { ox[k] = b[k].cx; oy[k] = b[k].cy; oa[k] = b[k].a; }
      if (in_cc) {
#pragma unroll
        for (int k = 0; k < 5; ++k) { xp[0 * 5 + k][lane] = b[k].cx; xp[1 * 5 + k][lane] = b[k].cy; xp[2 * 5 + k][lane] = b[k].a; }
      }
      __syncthreads();
      PP_MARK(0);
      int cok = 0xff;                                    // leader: bit r = the contacts of island r are within tolerance
      if (ccn > 0 && agent == 0) {
        // contacts of the still-iterating islands, in contact order; min separation per island
        float minSep[MCR_MAX_AGENTS];
#pragma unroll
        for (int c = 0; c < MCR_MAX_AGENTS; ++c) minSep[c] = 0.0f;
        for (int i = 0; i < ccn; ++i) {
          const uint32_t* rec = pcrec[pool_base + i];
          const int r = (int)((rec[0] >> 24) & 15u);
          if (!((actm >> (leader_lane + r)) & 1ull)) continue;
          const float ms = cc_position(CM, rec, leader_lane, xp);
#pragma unroll
          for (int c = 0; c < MCR_MAX_AGENTS; ++c) if (c == r) minSep[c] = mcr_min(minSep[c], ms);
        }
        cok = 0;
#pragma unroll
        for (int c = 0; c < MCR_MAX_AGENTS; ++c) cok |= (minSep[c] >= -3.0f * B2_LINEAR_SLOP) ? (1 << c) : 0;
      }
      __syncthreads();
      PP_MARK(1);
      bool jointsOk = true;
      if (active) {
        if (in_cc) {
#pragma unroll
          for (int k = 0; k < 5; ++k) { b[k].cx = xp[0 * 5 + k][lane]; b[k].cy = xp[1 * 5 + k][lane]; b[k].a = xp[2 * 5 + k][lane]; }
        }
#pragma unroll
        for (int q = 3; q >= 0; --q) {
          bool jo = joint_position(J[q], b[0], b[q + 1], jax[q], jay[q], lcx, lcy, mH, iH, mW, iW, hull_rot);
          jointsOk = jointsOk && jo;
        }
      }
      // fixed point: an island whose sweep failed without moving any of its bodies would repeat that sweep
      // identically up to iteration 60 — it stops here with the same outcome (see the contact-free loop above)
      PP_MARK(2);
      bool moved = false;
#pragma unroll
      for (int k = 0; k < 5; ++k) moved = moved || ox[k] != b[k].cx || oy[k] != b[k].cy || oa[k] != b[k].a;
      {
        const unsigned long long okm = __ballot(jointsOk), mvm = __ballot(moved);
        const int cokl = __shfl(cok, leader_lane);
        if (active) {
          bool ok = jointsOk, mv = moved;
          if (ccn > 0) { ok = ((cokl >> isl) & 1) != 0 && (okm & imask) == imask; mv = (mvm & imask) != 0ull; }
          if (ok) { positionSolved = true; active = false; }
          else if (!mv) active = false;
        }
      }
      __syncthreads();
      PP_MARK(3);
    }
    }   // !UNI
#ifdef MCR_POSLOOP_PROFILE
    if ((p.debug & 256) && !(p.debug & 65536) && mode == 0 && threadIdx.x == 0 && p.role == 2) {
      unsigned long long* o = p.dbg_stamps + ((size_t)((p.B * p.G + 63) / 64) + blk) * 8;
      o[5] = pt[0] | ((unsigned long long)pn << 48); o[6] = pt[1] | (pt[2] << 32); o[7] = pt[3] | ((unsigned long long)ccn << 48);
    }
#endif
    if (wave_perm) {                                     // the (joint, wheel) register sets back where the rest of the step expects them
      int tag[4] = {0, 0, 0, 0};
      slot_sort(J, b, wq, tag);
    }
  }
  if (defer_cap > 0) {
    int dfr = unfinished ? 1 : 0;
    for (int o = 1; o < p.G; o <<= 1) dfr |= __shfl_xor(dfr, o);          // env-wide: its cars finish the step together
    if (!dfr && lane_ok && agent == 0) p.dpart[env] = 0;                  // (the mark of an earlier step's deferral)
    if (dfr && run && p.cc_mode && mode == 0 && p.role == 1 && !MCR_NO_PARK_WAIT) {
      // Parking overwrites the env's ENTRY poses, which the contact pass — running beside this launch — reads: not before it is through with
      // the env.  (It nearly always is by now, 90 us into this kernel; a contact pass held up for longer — a machine full of other work — is
      // not: found in round 5 as rollouts that diverged when the contact chain filled every SIMD at the step's begin.)
      const int bound = (p.debug & 4096) ? (1 << 14) : (1 << 24);
      const int epoch = mcr_epoch(p);
      int spin = 0;
      for (; spin < bound && __hip_atomic_load(&p.collide_epoch[env], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch; ++spin) __builtin_amdgcn_s_sleep(8);
      if (spin == bound) { atomicAdd(&p.counters[5], 1ull); mcr_raise(p, ST_SPIN_GIVEUP); }
    }
    if (dfr && run) {
      // park the post-velocity-phase state exactly as the resume launch's prologue reloads it
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        p.carf[(CF_CX + k) * BN + ci] = b[k].cx; p.carf[(CF_CY + k) * BN + ci] = b[k].cy; p.carf[(CF_A + k) * BN + ci] = b[k].a;
        p.carf[(CF_VX + k) * BN + ci] = b[k].vx; p.carf[(CF_VY + k) * BN + ci] = b[k].vy; p.carf[(CF_W + k) * BN + ci] = b[k].w;
      }
      uint32_t lim = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        p.carf[(CF_JIX + k) * BN + ci] = J[k].ix; p.carf[(CF_JIY + k) * BN + ci] = J[k].iy;
        p.carf[(CF_JIZ + k) * BN + ci] = J[k].iz; p.carf[(CF_JM + k) * BN + ci] = J[k].im;
        lim |= (uint32_t)J[k].limit << (2 * k);
        p.card[(CD_OMEGA + k) * BN + ci] = omega[k]; p.card[(CD_PHASE + k) * BN + ci] = phase[k];
      }
      p.caru[CU_LIMIT * BN + ci] = lim;
      p.card[(CD_GAS + 0) * BN + ci] = gas[0]; p.card[(CD_GAS + 1) * BN + ci] = gas[1];
      p.card[CD_STEER * BN + ci] = steer; p.card[CD_BRAKE * BN + ci] = brake;
      if (agent == 0) { p.dpart[env] = 1; p.dlist[1 + atomicAdd(&p.dlist[0], 1)] = env; atomicAdd(&p.counters[0], 1ull); }   // the main launches after this one skip it
    }
    if (dfr) { run = false; lane_ok = false; }                            // nothing below is this launch's business
  }
  DYN_STAMP(3);
  float minSleep = MCR_MAXFLT;
  if (run) {
    // sleep (b2Island::Solve tail).  Car.step re-wakes every body next step, so "asleep" reduces to:
    // zero the velocities and restart the timers.  The decision is per island.
    const float linTol2 = B2_LINEAR_SLEEP_TOL * B2_LINEAR_SLEEP_TOL, angTol2 = B2_ANGULAR_SLEEP_TOL * B2_ANGULAR_SLEEP_TOL;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      if (b[k].w * b[k].w > angTol2 || (b[k].vx * b[k].vx + b[k].vy * b[k].vy) > linTol2) { sleepT[k] = 0.0f; minSleep = 0.0f; }
      else { sleepT[k] += h; minSleep = mcr_min(minSleep, sleepT[k]); }
    }
  }
  bool island_solved = positionSolved;
  if (wave_cc) {                          // members of a merged island decide together
    if (ccn > 0) { xms[lane] = minSleep; xjok[lane] = positionSolved ? 1 : 0; }
    __syncthreads();
    if (run && ccn > 0)
      for (int c = 0; c < p.N; ++c) if (xisl[leader_lane + c] == isl) { minSleep = mcr_min(minSleep, xms[leader_lane + c]); island_solved = island_solved && (xjok[leader_lane + c] != 0); }
  }
  if (run && minSleep >= B2_TIME_TO_SLEEP && island_solved) {
#pragma unroll
    for (int k = 0; k < 5; ++k) { sleepT[k] = 0.0f; b[k].vx = 0.0f; b[k].vy = 0.0f; b[k].w = 0.0f; }
  }

  // ---- env bookkeeping (:433-443, :497-507) + TimeLimit, across the env's lane group
  bool done = false, trunc = false, respawn = false;
  double step_reward = 0.0, reward = 0.0, prev_reward = 0.0, epret = 0.0;
  uint32_t tvc = 0, flags = 0;
  if (p.cc_mode && mode == 0 && p.role == 1 && run)               // k_collide pass 0 runs beside this launch: wait until it is through with this env
  {
    // p.epoch is the handle's step counter: no earlier pass can have left the same value behind.  The wait is bounded (~3 s; the
    // contact pass was enqueued before this launch, takes ~25 us and — mcr_hip.hip gates cc_mode on it — always finds room beside
    // this launch's one wavefront per SIMD); a give-up is REPORTED (status word -> mcr_step fails, the handle falls back to the
    // contact pass in front), never silent.  debug bit 12 shortens the bound (tests).
    const int bound = (p.debug & 4096) ? (1 << 14) : (1 << 24);
    const int epoch = mcr_epoch(p);
    int spin = 0;
    if (p.debug & 2048) { for (; spin < bound && __hip_atomic_load(&p.collide_epoch[env], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != epoch; ++spin) __builtin_amdgcn_s_sleep(8); }
    else { for (; spin < bound && __hip_atomic_load(&p.collide_epoch[env], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch; ++spin) __builtin_amdgcn_s_sleep(8); }
    if (spin == bound) { atomicAdd(&p.counters[5], 1ull); mcr_raise(p, ST_SPIN_GIVEUP); }
  }
  const bool cc_wait = p.cc_mode && mode == 0 && p.role == 1;
  uint32_t onroad_new = 0;
  if (run && cc_wait) {                                           // k_collide's three words of this car: device-scope loads (see k_collide)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    reward = __longlong_as_double((long long)__hip_atomic_load((unsigned long long*)&p.card[CD_REWARD * BN + ci], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    tvc = __hip_atomic_load(&p.caru[CU_TVC * BN + ci], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    onroad_new = __hip_atomic_load(&p.caru[CU_ONROAD_NEW * BN + ci], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    prev_reward = p.card[CD_PREV_REWARD * BN + ci]; flags = p.caru[CU_FLAGS * BN + ci];
  } else if (run) {
    reward = p.card[CD_REWARD * BN + ci]; prev_reward = p.card[CD_PREV_REWARD * BN + ci]; tvc = p.caru[CU_TVC * BN + ci]; flags = p.caru[CU_FLAGS * BN + ci];
    onroad_new = p.caru[CU_ONROAD_NEW * BN + ci];
  }
  const double reward_shown = reward;      // the score label is drawn (:431) before this step's -0.1 (:437)
  int T = 0;
  if (env < env_end && es.active) T = ((const McrSlotHeader*)(p.slots + ((size_t)env * 2 + es.slot) * MCR_SLOT_BYTES))->T;
  if (mode == 0) {
    const bool has_action = p.actions != nullptr;
    int d = 0;
    if (run && has_action) {
      reward -= 0.1;
      step_reward = reward - prev_reward;
      prev_reward = reward;
      if ((int)tvc == T) d = 1;
      // hull.position = xf.p
      Xf xh = xf_of(v2(b[0].cx, b[0].cy), b[0].a, v2(lcx, lcy));
      double px = (double)xh.p.x, py = (double)xh.p.y;
      if (fabs(px) > MCR_PLAYFIELD || fabs(py) > MCR_PLAYFIELD) { d = 1; step_reward = -100; }
    }
    for (int o = 1; o < p.G; o <<= 1) d |= __shfl_xor(d, o);
    done = d != 0;
    if (run && has_action) {
      int steps = es.steps + 1;
      if (p.max_steps > 0 && steps >= p.max_steps) { trunc = !done; done = true; }
    }
    if (run && has_action) {   // episode statistics: what a RecordEpisodeStatistics wrapper would report at `done`
      epret = p.card[CD_EPRET * BN + ci] + step_reward;
      if (done) {
        if (p.ep_return_out) p.ep_return_out[ci] = epret;
        if (p.ep_len_out && agent == 0) p.ep_len_out[env] = es.steps + 1;
        if (agent == 0) atomicAdd(&p.stats[0], 1.0);
        atomicAdd(&p.stats[1], epret);
      }
    }
    if (lane_ok && es.active) {
      p.reward_out[ci] = step_reward;
      if (agent == 0) { p.done_out[env] = done ? 1 : 0; if (p.trunc_out) p.trunc_out[env] = trunc ? 1 : 0; }
    } else if (lane_ok) {
      p.reward_out[ci] = 0.0;
      if (agent == 0) { p.done_out[env] = 0; if (p.trunc_out) p.trunc_out[env] = 0; }
    }
    respawn = (run && done && p.auto_reset && es.staged_ready) || thaw;
  }
  // ---- terminal observation (mcr_set_terminal_obs): an episode that ends with a re-spawn leaves an ENTRY — what its frames need of the state
  // the cars end it with — before the spawn poses overwrite that state; the env's reset pass and the list raster of its chain do the rest
  int tidx = -1;
  if (mode == 0 && p.term_cnt != nullptr && p.obs != nullptr) {
    const bool fin = run && done && respawn;                        // (a thaw re-spawns without an ending)
    if (fin && agent == 0) {
      tidx = atomicAdd(&p.term_cnt[0], 1);
      if (tidx < p.term_cap) {
        const int chain = p.role == 2 ? 1 : 0;
        p.term_list[chain * p.term_cap + atomicAdd(&p.term_cnt[1 + chain], 1)] = tidx;
        p.term_ids[tidx] = env;
        McrTermEnv te; te.t = es.t + 1.0 / MCR_FPS; te.slot = es.slot; te.env = env; te.consumed = es.consumed + 1; te.pad = 0;
        p.term_env[tidx] = te;
      } else tidx = -1;
    }
    tidx = __shfl(tidx, leader_lane);
    if (respawn && agent == 0) p.term_idx[env] = fin ? tidx : -1;
    if (fin && tidx >= 0) {
      const int tn = p.term_cap * p.N, tc = tidx * p.N + agent;
#pragma unroll
      for (int k = 0; k < 5; ++k) { p.term_carf[(CF_CX + k) * tn + tc] = b[k].cx; p.term_carf[(CF_CY + k) * tn + tc] = b[k].cy; p.term_carf[(CF_A + k) * tn + tc] = b[k].a; }
      p.term_carf[(CF_VX + 0) * tn + tc] = b[0].vx; p.term_carf[(CF_VY + 0) * tn + tc] = b[0].vy; p.term_carf[(CF_W + 0) * tn + tc] = b[0].w;
#pragma unroll
      for (int k = 0; k < 4; ++k) { p.term_card[(CD_OMEGA + k) * tn + tc] = omega[k]; p.term_card[(CD_PHASE + k) * tn + tc] = phase[k]; }
      float* tvp = p.term_viewp + (size_t)tc * MCR_VIEWP_FLOATS;
      tvp[VP_SCORE] = __int_as_float(mcr_label_value(reward_shown));
      tvp[VP_OLDFLAGS] = __uint_as_float(flags);
    }
  }

  // ---- env state update by the group leader
  if (lane_ok && agent == 0 && (es.active || thaw) && (mode == 0 || es.resetting)) {
    McrEnvState* E = &p.env[env];
    if (mode == 0) {
      if (respawn) {
        E->slot = es.slot ^ 1; E->staged_ready = 0; E->consumed = es.consumed + 1; E->resetting = 1; E->just_reset = 1;
        E->t = 0.0; E->steps = 0; E->active = 1; E->frozen = 0;
        // (with a terminal entry, the host learns that the staged episode was consumed when the entry's frames are drawn — the end of the
        // step —: they read the episode slot the env leaves, which the host refills as soon as it sees the counter)
        if (tidx < 0) p.consumed_host[env] = es.consumed + 1;
      } else {
        E->t = es.t + 1.0 / MCR_FPS;
        if (p.actions) E->steps = es.steps + 1;
        E->just_reset = 0;
        if (done && p.auto_reset) { E->active = 0; E->frozen = 1; mcr_raise(p, ST_FROZEN); if (p.part_next) { p.part[env] = 0; p.part_next[env] = 0; } }   // no staged episode yet (host refill late): frozen until it arrives, see `thaw` (a frozen env is the main launch's)
      }
      // raster launch order: zoomed-out frames (first second of an episode, :540-542) cost several times a normal
      // one, so their workgroups go FIRST (front of vorder) and cannot end up as the launch's tail
      // (only the main launch fills the list: envs of the contact / resume launches are drawn by launches of their own, and
      // an append from those streams would race with the main raster launch that is reading the counts)
      // A re-spawned env of the main launch: with respawn_list its reset pass and first observation are list launches
      // (role 4) of their own, beside the main raster
      if (p.role < 2 && respawn && p.respawn_list) p.rlist[1 + atomicAdd(&p.rlist[0], 1)] = env;
      else if (p.role < 2 && (respawn || !(done && p.auto_reset))) {
        const bool heavy = respawn || es.t + 1.0 / MCR_FPS < 1.0;
        const int vslot = respawn ? (es.slot ^ 1) : es.slot;
        const int vP = ((const McrSlotHeader*)(p.slots + ((size_t)env * 2 + vslot) * MCR_SLOT_BYTES))->P;
        const int entry = env | (vslot << MCR_VORDER_SLOT_SHIFT) | (vP << MCR_VORDER_P_SHIFT);
        if (heavy) p.vorder[atomicAdd(&p.vcount[0], 1)] = entry;
        else p.vorder[p.B - 1 - atomicAdd(&p.vcount[1], 1)] = entry;
      }
    } else {
      E->t = es.t + 1.0 / MCR_FPS;
      E->resetting = 0;
    }
  }

  if (run || thaw) {
  if (respawn) {
    // Car(world, angle, x, y): hull at the pose, wheels at UNROTATED offsets with the same angle
    const McrSlotHeader* H = (const McrSlotHeader*)(p.slots + ((size_t)env * 2 + (es.slot ^ 1)) * MCR_SLOT_BYTES);
    double sa = H->spawn[agent][0], sx = H->spawn[agent][1], sy = H->spawn[agent][2];
    float fa = (float)sa;
    Rot q = rot_of(fa);
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      float px = (k == 0) ? (float)sx : (float)(sx + (k == 1 || k == 3 ? -55 : 55) * MCR_SIZE);
      float py = (k == 0) ? (float)sy : (float)(sy + (k <= 2 ? 80 : -82) * MCR_SIZE);
      Xf xf; xf.p = v2(px, py); xf.q = q;
      V2 lc = (k == 0) ? v2(lcx, lcy) : v2(0.0f, 0.0f);
      V2 c = xmul(xf, lc);
      b[k].cx = c.x; b[k].cy = c.y; b[k].a = fa; b[k].vx = b[k].vy = b[k].w = 0.0f; sleepT[k] = 0.0f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { J[k].ix = J[k].iy = J[k].iz = J[k].im = 0.0f; J[k].limit = 0; omega[k] = 0; phase[k] = 0; }
    gas[0] = gas[1] = 0; steer = 0; brake = 0; onroad = 0;
    reward = 0; prev_reward = 0; tvc = 0; flags = 0; epret = 0;
    if (p.particles) mcr_particles_clear(p.particles + (size_t)ci * MCR_PART_WORDS);
    p.caru[CU_TVC * BN + ci] = 0;
  }
  // ---- write back
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    p.carf[(CF_CX + k) * BN + ci] = b[k].cx; p.carf[(CF_CY + k) * BN + ci] = b[k].cy; p.carf[(CF_A + k) * BN + ci] = b[k].a;
    p.carf[(CF_VX + k) * BN + ci] = b[k].vx; p.carf[(CF_VY + k) * BN + ci] = b[k].vy; p.carf[(CF_W + k) * BN + ci] = b[k].w;
    p.carf[(CF_SLEEP + k) * BN + ci] = sleepT[k];
  }
  uint32_t lim = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    p.carf[(CF_JIX + k) * BN + ci] = J[k].ix; p.carf[(CF_JIY + k) * BN + ci] = J[k].iy;
    p.carf[(CF_JIZ + k) * BN + ci] = J[k].iz; p.carf[(CF_JM + k) * BN + ci] = J[k].im;
    lim |= (uint32_t)J[k].limit << (2 * k);
    p.card[(CD_OMEGA + k) * BN + ci] = omega[k]; p.card[(CD_PHASE + k) * BN + ci] = phase[k];
  }
  p.caru[CU_LIMIT * BN + ci] = lim;
  p.caru[CU_ONROAD * BN + ci] = respawn ? 0u : onroad_new;                          // this step's Collide is what the next Car.step sees
  p.card[(CD_GAS + 0) * BN + ci] = gas[0]; p.card[(CD_GAS + 1) * BN + ci] = gas[1];
  p.card[CD_STEER * BN + ci] = steer; p.card[CD_BRAKE * BN + ci] = brake;
  if (mode == 0) {
    p.card[CD_REWARD * BN + ci] = reward; p.card[CD_PREV_REWARD * BN + ci] = prev_reward;
    if (p.actions != nullptr || respawn) p.card[CD_EPRET * BN + ci] = epret;
    if (respawn) p.caru[CU_FLAGS * BN + ci] = 0;
  }
  // ---- per-car view parameters for the rasteriser: one lane per agent view here instead of one redundant
  // evaluation per raster thread there.  Camera (:540-556): f64 exactly as CPython evaluates it, then the f32
  // values gym's Transform hands to glTranslatef/glRotatef/glScalef; HUD rectangles (:634-674).
  float bxl = MCR_MAXFLT, byl = MCR_MAXFLT, bxh = -MCR_MAXFLT, byh = -MCR_MAXFLT;   // world box of the car's draw polygons (= its fixtures)
  bool have_box = false;
  // (three-chain step, main launch: the record and the polygons are produced by k_viewprep, beside the bookkeeping kernel,
  // and only the two words this kernel holds are written here)
  // (... list chains, COOP: by viewprep_list_block right behind this function, on five lanes per car)
  const bool prep_later = (p.viewprep_in_flags && p.role == 1 && mode == 0) || (COOP && mode == 0);
  if (p.obs != nullptr && !respawn && prep_later) {
    float* vp = p.viewp + (size_t)ci * MCR_VIEWP_FLOATS;
    vp[VP_SCORE] = __int_as_float(mcr_label_value(reward_shown));
    vp[VP_OLDFLAGS] = __uint_as_float(flags);
  }
  if (p.obs != nullptr && !respawn && !prep_later) {
    float* vp = p.viewp + (size_t)ci * MCR_VIEWP_FLOATS;
    const Xf hxf = xf_of(v2(b[0].cx, b[0].cy), b[0].a, v2(lcx, lcy));
    const double t = es.t + 1.0 / MCR_FPS;
    const double zoom = 0.1 * MCR_SCALE * fmax(1 - t, 0.0) + MCR_ZOOM * MCR_SCALE * fmin(t, 1.0);
    const double sx = (double)hxf.p.x, sy = (double)hxf.p.y;
    double angle = -(double)b[0].a;
    const double vx = (double)b[0].vx, vy = (double)b[0].vy;
    const double speed = sqrt(vx * vx + vy * vy);
    if (speed > 0.5) angle = atan2(vx, vy);
    double sin_a, cos_a; mcr_sincos_core(angle, &sin_a, &cos_a);     // |angle| is a few turns at most; only pixels depend on it
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
    // pixel centre -> world:  world = R^T (W - t) / zoom,  W = centre * (1000/96, 800/96)
    const float inv_z = 1.0f / fz;
    vp[VP_INV + 0] = fcs * (1000.0f / 96.0f) * inv_z; vp[VP_INV + 1] = fsn * (800.0f / 96.0f) * inv_z; vp[VP_INV + 2] = -(fcs * ftx + fsn * fty) * inv_z;
    vp[VP_INV + 3] = -fsn * (1000.0f / 96.0f) * inv_z; vp[VP_INV + 4] = fcs * (800.0f / 96.0f) * inv_z; vp[VP_INV + 5] = (fsn * ftx - fcs * fty) * inv_z;
    const double sW = MCR_WINDOW_W / 40.0, hH = MCR_WINDOW_H / 40.0;
    const double vals[5] = {0.02 * speed, 0.01 * omega[0], 0.01 * omega[1], 0.01 * omega[2], 0.01 * omega[3]};
    const double places[5] = {5, 7, 8, 9, 10};
    float hud_top = 12.0f;
#pragma unroll
    for (int i = 0; i < 5; ++i) {                                   // vertical_ind (:643-648)
      const float ya = (float)(hH + hH * vals[i]) * ky, yb = (float)hH * ky;
      vp[VP_IND + i * 4 + 0] = (float)((places[i] + 0) * sW) * kx; vp[VP_IND + i * 4 + 1] = (float)((places[i] + 1) * sW) * kx;
      vp[VP_IND + i * 4 + 2] = fminf(ya, yb); vp[VP_IND + i * 4 + 3] = fmaxf(ya, yb);
      hud_top = fmaxf(hud_top, fmaxf(ya, yb) + 1.0f);
    }
    const double jang = (double)(b[1].a - b[0].a);
    const double hv[2] = {-10.0 * jang, -0.8 * (double)b[0].w};
    const double hp[2] = {20, 30};
#pragma unroll
    for (int i = 0; i < 2; ++i) {                                   // horiz_ind (:649-654)
      const float xa = (float)((hp[i] + 0) * sW) * kx, xb = (float)((hp[i] + hv[i]) * sW) * kx;
      vp[VP_IND + (5 + i) * 4 + 0] = fminf(xa, xb); vp[VP_IND + (5 + i) * 4 + 1] = fmaxf(xa, xb);
      vp[VP_IND + (5 + i) * 4 + 2] = (float)(2 * hH) * ky; vp[VP_IND + (5 + i) * 4 + 3] = (float)(4 * hH) * ky;
    }
    vp[VP_HUDTOP] = hud_top;
    vp[VP_SCORE] = __int_as_float(mcr_label_value(reward_shown));
    vp[VP_OLDFLAGS] = __uint_as_float(flags);                       // :669-674 draws the flag from the value the PREVIOUS step computed
    {
      // Light grass squares the viewport can see + "is the whole viewport inside the playfield", from the inverse camera at
      // the four corners of the scene rectangle, in checker units U = world.x / (2k), V = world.y / (2k), k = PLAYFIELD / 20:
      // the playfield is |U|,|V| <= 10 and light square m covers [m, m + 0.5] (:615-627).  One pixel of slack.
      const float hk = 0.5f / (float)(MCR_PLAYFIELD / 20.0);
      const float aU = vp[VP_INV + 0] * hk, bU = vp[VP_INV + 1] * hk, cU = vp[VP_INV + 2] * hk;
      const float aV = vp[VP_INV + 3] * hk, bV = vp[VP_INV + 4] * hk, cV = vp[VP_INV + 5] * hk;
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
    // world-space vertices of the 12 Car.draw polygons (trans*v in f32, as pybox2d hands them to the viewer).
    // The record is AoS on purpose (the raster reads a car's 832 bytes as one run); every lane writes it with
    // 16-byte stores — a polygon is 4 of them — and each distinct vertex is transformed once (padding repeats the last).
    float4* cp4 = (float4*)(p.carpoly + (size_t)ci * MCR_CARPOLY_FLOATS);
    int counts[12];
    have_box = true;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const Xf wxf = xf_of(v2(b[k + 1].cx, b[k + 1].cy), b[k + 1].a, v2(0.0f, 0.0f));
      V2 w[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { w[i] = xmul(wxf, v2(S.wheel.vx[i], S.wheel.vy[i])); bxl = mcr_min(bxl, w[i].x); bxh = mcr_max(bxh, w[i].x); byl = mcr_min(byl, w[i].y); byh = mcr_max(byh, w[i].y); }
      float4* box = cp4 + (2 * k) * 4;
      box[0] = make_float4(w[0].x, w[0].y, w[1].x, w[1].y); box[1] = make_float4(w[2].x, w[2].y, w[3].x, w[3].y);
      box[2] = make_float4(w[3].x, w[3].y, w[3].x, w[3].y); box[3] = box[2];
      counts[2 * k] = S.wheel.n;
      const double a1 = phase[k], a2 = phase[k] + 1.2;
      double s1, s2, c1, c2; mcr_sincos_core(a1, &s1, &c1); mcr_sincos_core(a2, &s2, &c2);   // phases stay far below the core's 1.6e6 rad range
      int ns = 0;
      if (!(s1 > 0 && s2 > 0)) {
        if (s1 > 0) c1 = np_sign(c1);
        if (s2 > 0) c2 = np_sign(c2);
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
      counts[2 * k + 1] = ns;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int n = __builtin_amdgcn_readfirstlane(S.hull[k].n);     // the shape table is the same for every lane: scalar loads
      V2 w[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) { if (i < n) w[i] = xmul(hxf, v2(S.hull[k].vx[i], S.hull[k].vy[i])); else w[i] = w[i - 1 < 0 ? 0 : i - 1]; }
#pragma unroll
      for (int i = 0; i < 8; ++i) { bxl = mcr_min(bxl, w[i].x); bxh = mcr_max(bxh, w[i].x); byl = mcr_min(byl, w[i].y); byh = mcr_max(byh, w[i].y); }
      float4* hp = cp4 + (8 + k) * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) hp[i] = make_float4(w[2 * i].x, w[2 * i].y, w[2 * i + 1].x, w[2 * i + 1].y);
      counts[8 + k] = n;
    }
    float4* cn = cp4 + MCR_CARPOLY_NOFF / 4;
#pragma unroll
    for (int i = 0; i < 3; ++i) cn[i] = make_float4(__int_as_float(counts[4 * i]), __int_as_float(counts[4 * i + 1]), __int_as_float(counts[4 * i + 2]), __int_as_float(counts[4 * i + 3]));
  }
  // ---- the next step's touch verdict (k_touch.h), cheap half: can ANY fixture pair of two cars of this env touch?  (2: maybe —
  // the bookkeeping kernel runs the exact test; 0: no — that is the verdict.)  A conservative filter: the boxes of the
  // draw polygons (the fixtures' own vertices) with 0.2 of slack, or — no observations, or a fresh spawn — the hull discs of
  // mcr_touch_verdict's first exit.
  if (mode == 0 && p.role == 1 && p.part_next && p.car_contacts && p.N > 1) {
    const int lead = (int)threadIdx.x - agent;
    const float r0 = 2.0f * (fmaxf(S.pad[0], 2.5f + S.pad[1]) + 0.1f);
    bool near = false;
    for (int a = 0; a < p.N - 1; ++a)
      for (int c = a + 1; c < p.N; ++c) {
        const float ax = __shfl(b[0].cx, lead + a), ay = __shfl(b[0].cy, lead + a), ox = __shfl(b[0].cx, lead + c), oy = __shfl(b[0].cy, lead + c);
        bool nr = (ax - ox) * (ax - ox) + (ay - oy) * (ay - oy) <= r0 * r0;
        const float a0 = __shfl(bxl, lead + a), a1 = __shfl(byl, lead + a), a2 = __shfl(bxh, lead + a), a3 = __shfl(byh, lead + a);
        const float c0 = __shfl(bxl, lead + c), c1 = __shfl(byl, lead + c), c2 = __shfl(bxh, lead + c), c3 = __shfl(byh, lead + c);
        const bool boxes = __shfl((int)have_box, lead + a) && __shfl((int)have_box, lead + c);
        if (boxes) nr = nr && !(a0 > c2 + 0.2f || a2 + 0.2f < c0 || a1 > c3 + 0.2f || a3 + 0.2f < c1);
        near = near || nr;
      }
    // (an env re-spawned in this step whose reset pass is a list launch of its own, role 4: that launch settles the verdict on
    // the poses it ends with — the bookkeeping kernel of the main envs, which may run beside it, must leave the env alone)
    if (agent == 0) p.part_next[env] = (near && !(respawn && p.respawn_list)) ? 2 : 0;
  }
  }   // run
  DYN_STAMP(4);
  if (mode == 0 && lane_ok && agent == 0 && (run || thaw)) MCR_TRACE(p, env, p.role == 2 ? 0 : 1, p.role);   // (slot 0: stepped by the contact chain, 1: by anybody else)

}

// main launches (roles 0 / 1): 64 / G envs per wavefront (the list launches of roles >= 2 call dynamics_block from k_list_chain.h)
template <bool CC>
__global__ __launch_bounds__(64) void k_dynamics(McrParams p, int mode) {
  // one wavefront per SIMD whose serial chain IS the step's critical path: it goes before the wavefronts of the kernels that run beside it
  // on the same SIMDs (k_collide's 4096, the raster's tail) whenever both can issue
  __builtin_amdgcn_s_setprio(3);
  if (mode == 0 && blockIdx.x == 0 && threadIdx.x == 0) {         // the next step's lists: every reader of these buffers finished last step
    if (p.soft_sync && p.role == 1) mcr_post(p, W_BEGIN);         // the caller's stream is here: the side stream may start this step
    // (fuse_collide: the next step's contact list is filled DURING this step, by verdict writers that may run while this kernel — whose
    // stores sit in its XCD's L2 until it ends — is still going: that list is emptied a step earlier, by the side stream's last kernel)
    if (p.role == 1 && !p.fuse_collide) p.clist_next[0] = 0;
    for (int i = 0; i < 4; ++i) if (p.next_counts[i]) *p.next_counts[i] = 0;
    if (p.term_cnt_next) { p.term_cnt_next[0] = 0; p.term_cnt_next[1] = 0; p.term_cnt_next[2] = 0; }
  }
  dynamics_block<CC>(p, mode, (int)blockIdx.x);
  if (mode == 0 && p.post_dyn) {
    // W_DYN from inside the kernel: every workgroup (= wavefront) releases what it stored (agent scope: its XCD's L2 is written back), then
    // counts itself; the last one posts.  The HIP memory model's release pattern — the consumer's kernels behind k_await start with the usual acquire.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (threadIdx.x == 0) {
      const int done = __hip_atomic_fetch_add(&p.sync_words[W_DYN_COUNT * 16], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
      if (done == (int)gridDim.x) { __hip_atomic_store(&p.sync_words[W_DYN_COUNT * 16], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); mcr_post(p, W_DYN); }
    }
  }
}

__global__ void k_mark_staged(McrParams p, const int32_t* __restrict__ ids, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p.env[ids ? ids[i] : i].staged_ready = 1;
}

// Explicit reset(): install the staged episode for masked envs and spawn the cars; the caller then
// runs collide(pass 1) -> dynamics(mode 1) -> view to complete `return self.step(None)[0]` (:408).
__global__ __launch_bounds__(64) void k_install(McrParams p) {
  const int g = blockIdx.x * 64 + threadIdx.x;
  const int env = mcr_env_of_slot(p, g / p.G), agent = g % p.G;
  if (env >= p.env0 + p.nenv || agent >= p.N) return;
  if (p.reset_mask && !p.reset_mask[env]) return;
  McrEnvState es = p.env[env];
  if (!es.staged_ready) return;
  const int ci = env * p.N + agent, BN = p.BN;
  const McrShapes& S = *p.shapes;
  const McrSlotHeader* H = (const McrSlotHeader*)(p.slots + ((size_t)env * 2 + (es.slot ^ 1)) * MCR_SLOT_BYTES);
  double sa = H->spawn[agent][0], sx = H->spawn[agent][1], sy = H->spawn[agent][2];
  float fa = (float)sa;
  Rot q = rot_of(fa);
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    float px = (k == 0) ? (float)sx : (float)(sx + (k == 1 || k == 3 ? -55 : 55) * MCR_SIZE);
    float py = (k == 0) ? (float)sy : (float)(sy + (k <= 2 ? 80 : -82) * MCR_SIZE);
    Xf xf; xf.p = v2(px, py); xf.q = q;
    V2 lc = (k == 0) ? v2(S.hull_lcx, S.hull_lcy) : v2(0.0f, 0.0f);
    V2 c = xmul(xf, lc);
    p.carf[(CF_CX + k) * BN + ci] = c.x; p.carf[(CF_CY + k) * BN + ci] = c.y; p.carf[(CF_A + k) * BN + ci] = fa;
    p.carf[(CF_VX + k) * BN + ci] = 0.0f; p.carf[(CF_VY + k) * BN + ci] = 0.0f; p.carf[(CF_W + k) * BN + ci] = 0.0f;
    p.carf[(CF_SLEEP + k) * BN + ci] = 0.0f;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    p.carf[(CF_JIX + k) * BN + ci] = 0.0f; p.carf[(CF_JIY + k) * BN + ci] = 0.0f; p.carf[(CF_JIZ + k) * BN + ci] = 0.0f; p.carf[(CF_JM + k) * BN + ci] = 0.0f;
    p.card[(CD_OMEGA + k) * BN + ci] = 0.0; p.card[(CD_PHASE + k) * BN + ci] = 0.0;
  }
  p.card[(CD_GAS + 0) * BN + ci] = 0.0; p.card[(CD_GAS + 1) * BN + ci] = 0.0; p.card[CD_STEER * BN + ci] = 0.0; p.card[CD_BRAKE * BN + ci] = 0.0;
  p.card[CD_REWARD * BN + ci] = 0.0; p.card[CD_PREV_REWARD * BN + ci] = 0.0; p.card[CD_EPRET * BN + ci] = 0.0;
  p.caru[CU_LIMIT * BN + ci] = 0; p.caru[CU_ONROAD * BN + ci] = 0; p.caru[CU_ONROAD_NEW * BN + ci] = 0; p.caru[CU_TVC * BN + ci] = 0; p.caru[CU_FLAGS * BN + ci] = 0;
  if (p.particles) mcr_particles_clear(p.particles + (size_t)ci * MCR_PART_WORDS);
  if (agent == 0) {
    McrEnvState* E = &p.env[env];
    E->slot = es.slot ^ 1; E->staged_ready = 0; E->consumed = es.consumed + 1; E->resetting = 1; E->just_reset = 1; E->active = 1; E->frozen = 0;
    E->t = 0.0; E->steps = 0;
    p.consumed_host[env] = es.consumed + 1;
  }
}
