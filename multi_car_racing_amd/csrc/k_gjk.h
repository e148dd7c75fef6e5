// k_gjk.h — Box2D's sensor predicate on the device: b2TestOverlap = b2Distance (GJK) with an empty simplex cache, radii
// applied after the iteration, touching iff distance < 10 * b2_epsilon.  Behind multi_car_racing.py:428 ->
// b2ContactManager::Collide -> b2Contact::Update (sensor branch) -> b2TestOverlap(shapeA, indexA, shapeB, indexB, xfA, xfB).
// Restated from the published Box2D 2.3.x sources (b2Distance.cpp: b2DistanceProxy::GetSupport, b2Simplex::{ReadCache with
// count 0, Solve2, Solve3, GetSearchDirection, GetWitnessPoints}, k_maxIters = 20, duplicate-support termination; the
// "ensure progress" exit is commented out there), expression by expression in f32 — the oracle holds the same restatement
// on the host, and tests/test_gpu_parity.py runs >= 1e6 threshold-band poses through both with 0 differences.
//
// Proxy A is the TILE (fixtureA of a tile<->car contact is the fixture with the lower broadphase proxy id, and tiles are
// created before the cars: DESIGN.md 4); its body transform is the identity (CreateStaticBody() at the origin, vertices in
// world coordinates), so b2MulT(xfA.q, -d) = -d and b2Mul(xfA, v) = v up to the sign of a zero.  Proxy B is the car
// fixture: supports in body-local coordinates, simplex points in world space.
#pragma once
#include "mcr_kernels.h"

namespace gjk {

struct SV { float wAx, wAy, wBx, wBy, wx, wy, a; int iA, iB; };

__device__ __forceinline__ void tile_vertex(const float4 va, const float4 vb, int i, float& x, float& y) {
  x = i == 0 ? va.x : i == 1 ? va.z : i == 2 ? vb.x : vb.z;
  y = i == 0 ? va.y : i == 1 ? va.w : i == 2 ? vb.y : vb.w;
}
// b2DistanceProxy::GetSupport: first index of the largest dot product
__device__ __forceinline__ int support_tile(const float4 va, const float4 vb, int tn, float dx, float dy) {
  int best = 0; float bestValue = va.x * dx + va.y * dy;
#pragma unroll
  for (int i = 1; i < 4; ++i) {
    float x, y; tile_vertex(va, vb, i, x, y);
    const float value = x * dx + y * dy;
    if (i < tn && value > bestValue) { best = i; bestValue = value; }
  }
  return best;
}
__device__ __forceinline__ int support_poly(const McrPoly* __restrict__ P, float dx, float dy) {
  int best = 0; float bestValue = P->vx[0] * dx + P->vy[0] * dy;
  const int n = P->n;
  for (int i = 1; i < n; ++i) { const float value = P->vx[i] * dx + P->vy[i] * dy; if (value > bestValue) { best = i; bestValue = value; } }
  return best;
}
// SWAP: the car fixture is proxy A and the tile proxy B (a world that lives across reset() hands out proxy ids off its tree's free list:
// a car fixture's can be the lower one, include/mcr.h: mcr_world) — iA then indexes the fixture, iB the tile
template <bool SWAP>
__device__ __forceinline__ SV make_vertex(const float4 va, const float4 vb, int iA, const McrPoly* __restrict__ PB, const float4 xfB, int iB) {
  SV v; v.iA = iA; v.iB = iB;
  float tx, ty; tile_vertex(va, vb, SWAP ? iB : iA, tx, ty);
  const float lx = PB->vx[SWAP ? iA : iB], ly = PB->vy[SWAP ? iA : iB];
  const float fx = (xfB.w * lx - xfB.z * ly) + xfB.x, fy = (xfB.z * lx + xfB.w * ly) + xfB.y;     // b2Mul(xf, v) of the car fixture's vertex
  if (SWAP) { v.wAx = fx; v.wAy = fy; v.wBx = tx; v.wBy = ty; } else { v.wAx = tx; v.wAy = ty; v.wBx = fx; v.wBy = fy; }
  v.wx = v.wBx - v.wAx; v.wy = v.wBy - v.wAy; v.a = 1.0f;
  return v;
}

// tile: hull vertices v0 v1 | v2 v3 (CCW, b2PolygonShape::Set order), tn = 3 or 4;  PB: the car fixture in body-local
// coordinates;  xfB = (p.x, p.y, sin, cos) of its body.  Both shapes have radius b2_polygonRadius.
template <bool SWAP>
__device__ __forceinline__ bool touching_as(const float4 va, const float4 vb, const int tn, const McrPoly* __restrict__ PB, const float4 xfB) {
  SV v0 = make_vertex<SWAP>(va, vb, 0, PB, xfB, 0), v1 = v0, v2 = v0;
  int count = 1;
  int saveA0 = 0, saveA1 = 0, saveA2 = 0, saveB0 = 0, saveB1 = 0, saveB2 = 0, saveCount = 0;
  int iter = 0;
  while (iter < 20) {
    saveCount = count;
    saveA0 = v0.iA; saveB0 = v0.iB; saveA1 = v1.iA; saveB1 = v1.iB; saveA2 = v2.iA; saveB2 = v2.iB;
    if (count == 2) {                                                      // b2Simplex::Solve2
      const float e12x = v1.wx - v0.wx, e12y = v1.wy - v0.wy;
      const float d12_2 = -(v0.wx * e12x + v0.wy * e12y);
      if (d12_2 <= 0.0f) { v0.a = 1.0f; count = 1; }
      else {
        const float d12_1 = v1.wx * e12x + v1.wy * e12y;
        if (d12_1 <= 0.0f) { v1.a = 1.0f; count = 1; v0 = v1; }
        else { const float inv = 1.0f / (d12_1 + d12_2); v0.a = d12_1 * inv; v1.a = d12_2 * inv; count = 2; }
      }
    } else if (count == 3) {                                               // b2Simplex::Solve3
      const float w1x = v0.wx, w1y = v0.wy, w2x = v1.wx, w2y = v1.wy, w3x = v2.wx, w3y = v2.wy;
      const float e12x = w2x - w1x, e12y = w2y - w1y;
      const float w1e12 = w1x * e12x + w1y * e12y, w2e12 = w2x * e12x + w2y * e12y;
      const float d12_1 = w2e12, d12_2 = -w1e12;
      const float e13x = w3x - w1x, e13y = w3y - w1y;
      const float w1e13 = w1x * e13x + w1y * e13y, w3e13 = w3x * e13x + w3y * e13y;
      const float d13_1 = w3e13, d13_2 = -w1e13;
      const float e23x = w3x - w2x, e23y = w3y - w2y;
      const float w2e23 = w2x * e23x + w2y * e23y, w3e23 = w3x * e23x + w3y * e23y;
      const float d23_1 = w3e23, d23_2 = -w2e23;
      const float n123 = e12x * e13y - e12y * e13x;
      const float d123_1 = n123 * (w2x * w3y - w2y * w3x), d123_2 = n123 * (w3x * w1y - w3y * w1x), d123_3 = n123 * (w1x * w2y - w1y * w2x);
      if (d12_2 <= 0.0f && d13_2 <= 0.0f) { v0.a = 1.0f; count = 1; }
      else if (d12_1 > 0.0f && d12_2 > 0.0f && d123_3 <= 0.0f) { const float inv = 1.0f / (d12_1 + d12_2); v0.a = d12_1 * inv; v1.a = d12_2 * inv; count = 2; }
      else if (d13_1 > 0.0f && d13_2 > 0.0f && d123_2 <= 0.0f) { const float inv = 1.0f / (d13_1 + d13_2); v0.a = d13_1 * inv; v2.a = d13_2 * inv; count = 2; v1 = v2; }
      else if (d12_1 <= 0.0f && d23_2 <= 0.0f) { v1.a = 1.0f; count = 1; v0 = v1; }
      else if (d13_1 <= 0.0f && d23_1 <= 0.0f) { v2.a = 1.0f; count = 1; v0 = v2; }
      else if (d23_1 > 0.0f && d23_2 > 0.0f && d123_1 <= 0.0f) { const float inv = 1.0f / (d23_1 + d23_2); v1.a = d23_1 * inv; v2.a = d23_2 * inv; count = 2; v0 = v2; }
      else { const float inv = 1.0f / (d123_1 + d123_2 + d123_3); v0.a = d123_1 * inv; v1.a = d123_2 * inv; v2.a = d123_3 * inv; count = 3; }
    }
    if (count == 3) break;                                                 // the origin lies in the triangle: overlap
    float dx, dy;                                                          // b2Simplex::GetSearchDirection
    if (count == 1) { dx = -v0.wx; dy = -v0.wy; }
    else {
      const float e12x = v1.wx - v0.wx, e12y = v1.wy - v0.wy;
      const float sgn = e12x * (-v0.wy) - e12y * (-v0.wx);
      if (sgn > 0.0f) { dx = -1.0f * e12y; dy = 1.0f * e12x; }            // b2Cross(1.0f, e12)
      else { dx = 1.0f * e12y; dy = -1.0f * e12x; }                        // b2Cross(e12, 1.0f)
    }
    if (dx * dx + dy * dy < B2_EPSILON * B2_EPSILON) break;
    int iA, iB;
    if (SWAP) {
      const float ndx = -dx, ndy = -dy;
      iA = support_poly(PB, xfB.w * ndx + xfB.z * ndy, -xfB.z * ndx + xfB.w * ndy);        // b2MulT(xfA.q, -d), A the car fixture
      iB = support_tile(va, vb, tn, dx, dy);                                                // b2MulT(identity, d)
    } else {
      iA = support_tile(va, vb, tn, -dx, -dy);                                              // b2MulT(identity, -d)
      iB = support_poly(PB, xfB.w * dx + xfB.z * dy, -xfB.z * dx + xfB.w * dy);             // b2MulT(xfB.q, d)
    }
    const SV nv = make_vertex<SWAP>(va, vb, iA, PB, xfB, iB);
    ++iter;
    bool duplicate = (saveCount > 0 && iA == saveA0 && iB == saveB0) || (saveCount > 1 && iA == saveA1 && iB == saveB1) ||
                     (saveCount > 2 && iA == saveA2 && iB == saveB2);
    if (duplicate) break;
    if (count == 1) v1 = nv; else v2 = nv;
    ++count;
  }
  float pAx, pAy, pBx, pBy;                                                // b2Simplex::GetWitnessPoints
  if (count == 1) { pAx = v0.wAx; pAy = v0.wAy; pBx = v0.wBx; pBy = v0.wBy; }
  else if (count == 2) {
    pAx = v0.a * v0.wAx + v1.a * v1.wAx; pAy = v0.a * v0.wAy + v1.a * v1.wAy;
    pBx = v0.a * v0.wBx + v1.a * v1.wBx; pBy = v0.a * v0.wBy + v1.a * v1.wBy;
  } else {
    pAx = (v0.a * v0.wAx + v1.a * v1.wAx) + v2.a * v2.wAx; pAy = (v0.a * v0.wAy + v1.a * v1.wAy) + v2.a * v2.wAy;
    pBx = pAx; pBy = pAy;
  }
  const float ddx = pBx - pAx, ddy = pBy - pAy;
  float distance = sqrtf(ddx * ddx + ddy * ddy);
  const float rr = B2_POLYGON_RADIUS + B2_POLYGON_RADIUS;
  if (distance > rr && distance > B2_EPSILON) distance -= rr; else distance = 0.0f;      // input.useRadii
  return distance < 10.0f * B2_EPSILON;
}
// tile = proxy A (a fresh world: tiles are created before the cars) / car fixture = proxy A
__device__ __forceinline__ bool touching(const float4 va, const float4 vb, const int tn, const McrPoly* __restrict__ PB, const float4 xfB) { return touching_as<false>(va, vb, tn, PB, xfB); }
__device__ __forceinline__ bool touching_fixture_first(const float4 va, const float4 vb, const int tn, const McrPoly* __restrict__ PB, const float4 xfB) { return touching_as<true>(va, vb, tn, PB, xfB); }

}  // namespace gjk
