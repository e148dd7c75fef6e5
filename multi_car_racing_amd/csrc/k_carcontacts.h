// k_carcontacts.h — rigid car<->car contacts on the device: manifold generation (b2CollidePolygons +
// b2ClipSegmentToLine) used by the collide kernel, and the contact-constraint maths (b2ContactSolver:
// b2WorldManifold, velocity constraints with the 2-point block solver, position constraints) used by the
// dynamics kernel.  Box2D 2.3.x semantics; hull/wheel polygons have radius b2_polygonRadius, friction
// sqrt(0.2*0.2), restitution 0.
//
// Orderings Box2D derives from broadphase proxy ids / intrusive lists are DEFINED:
// fixture pair (A,B) with A the lower (car, fixture), contacts sorted by (carA, fixA, carB, fixB).
//
// Stored manifold record (MCR_CC_WORDS u32 words per touching pair, per env, in HBM):
//   0 key = carA | fixA<<4 | carB<<8 | fixB<<12      1 type(1 faceA,2 faceB) | n<<8
//   2,3 localNormal   4,5 localPoint
//   6..10  point0: local.x local.y nImp tImp id      11..15 point1
#pragma once
#include "mcr_kernels.h"

namespace cc {

typedef McrPoly LPoly;     // polygons are read in place from the shape table in HBM (dynamic indexing is a plain load)
__device__ __forceinline__ int fixture_body(int fix) { return fix < 4 ? 0 : fix - 3; }
__device__ __forceinline__ Xf xf_mulT(const Xf& A, const Xf& B) {   // b2MulT(A,B)
  Xf C; C.q.s = A.q.c * B.q.s - A.q.s * B.q.c; C.q.c = A.q.c * B.q.c + A.q.s * B.q.s;
  C.p = rmulT(A.q, B.p - A.p); return C;
}
__device__ __forceinline__ uint32_t mkid(int ia, int ib, int ta, int tb) { return (uint32_t)(ia & 255) | ((uint32_t)(ib & 255) << 8) | ((uint32_t)ta << 16) | ((uint32_t)tb << 24); }
__device__ __forceinline__ uint32_t flipid(uint32_t id) { return ((id >> 8) & 255u) | ((id & 255u) << 8) | (((id >> 24) & 255u) << 16) | (((id >> 16) & 255u) << 24); }

__device__ inline float find_max_separation(int* edge, const LPoly& p1, const Xf& xf1, const LPoly& p2, const Xf& xf2) {
  const Xf xf = xf_mulT(xf2, xf1);
  int best = 0; float maxSep = -MCR_MAXFLT;
  for (int i = 0; i < p1.n; ++i) {
    const V2 n = rmul(xf.q, v2(p1.nx[i], p1.ny[i]));
    const V2 v1 = xmul(xf, v2(p1.vx[i], p1.vy[i]));
    float si = MCR_MAXFLT;
    for (int j = 0; j < p2.n; ++j) { const float sij = dot(n, v2(p2.vx[j], p2.vy[j]) - v1); if (sij < si) si = sij; }
    if (si > maxSep) { maxSep = si; best = i; }
  }
  *edge = best; return maxSep;
}
struct ClipV { V2 v; uint32_t id; };
// b2ClipSegmentToLine without run-time indexing of out[] (which would put the clip vertices in scratch memory): the kept
// input vertices first (in order), then the intersection point — exactly the sequence the original appends
__device__ __forceinline__ int clip_segment(ClipV out[2], const ClipV in[2], V2 normal, float offset, int vertexIndexA) {
  const float d0 = dot(normal, in[0].v) - offset, d1 = dot(normal, in[1].v) - offset;
  const bool k0 = d0 <= 0.0f, k1 = d1 <= 0.0f, cut = d0 * d1 < 0.0f;
  ClipV I; I.v = in[0].v; I.id = in[0].id;
  if (cut) {
    const float interp = d0 / (d0 - d1);
    I.v = in[0].v + interp * (in[1].v - in[0].v);
    I.id = mkid(vertexIndexA, (in[0].id >> 8) & 255, 0, 1);
  }
  out[0] = k0 ? in[0] : (k1 ? in[1] : I);
  out[1] = (k0 && k1) ? in[1] : I;
  return (k0 ? 1 : 0) + (k1 ? 1 : 0) + (cut ? 1 : 0);
}
struct Manifold { int type, n; V2 localNormal, localPoint; V2 pl[2]; uint32_t id[2]; };

// b2CollidePolygons (2.3.1+ form: brute-force max separation, k_tol = 0.1*linearSlop)
__device__ inline void collide_polygons(Manifold& M, const LPoly& pA, const Xf& xfA, const LPoly& pB, const Xf& xfB) {
  M.n = 0; M.type = 0;
  const float totalRadius = B2_POLYGON_RADIUS + B2_POLYGON_RADIUS;
  int edgeA = 0; const float sepA = find_max_separation(&edgeA, pA, xfA, pB, xfB);
  if (sepA > totalRadius) return;
  int edgeB = 0; const float sepB = find_max_separation(&edgeB, pB, xfB, pA, xfA);
  if (sepB > totalRadius) return;
  const LPoly *p1, *p2; Xf xf1, xf2; int edge1; int flip;
  const float k_tol = 0.1f * B2_LINEAR_SLOP;
  if (sepB > sepA + k_tol) { p1 = &pB; p2 = &pA; xf1 = xfB; xf2 = xfA; edge1 = edgeB; M.type = 2; flip = 1; }
  else { p1 = &pA; p2 = &pB; xf1 = xfA; xf2 = xfB; edge1 = edgeA; M.type = 1; flip = 0; }
  ClipV inc[2];
  {
    const V2 normal1 = rmulT(xf2.q, rmul(xf1.q, v2(p1->nx[edge1], p1->ny[edge1])));
    int index = 0; float minDot = MCR_MAXFLT;
    for (int i = 0; i < p2->n; ++i) { const float d = dot(normal1, v2(p2->nx[i], p2->ny[i])); if (d < minDot) { minDot = d; index = i; } }
    const int i1 = index, i2 = i1 + 1 < p2->n ? i1 + 1 : 0;
    inc[0].v = xmul(xf2, v2(p2->vx[i1], p2->vy[i1])); inc[0].id = mkid(edge1, i1, 1, 0);
    inc[1].v = xmul(xf2, v2(p2->vx[i2], p2->vy[i2])); inc[1].id = mkid(edge1, i2, 1, 0);
  }
  const int iv1 = edge1, iv2 = edge1 + 1 < p1->n ? edge1 + 1 : 0;
  V2 v11 = v2(p1->vx[iv1], p1->vy[iv1]), v12 = v2(p1->vx[iv2], p1->vy[iv2]);
  V2 localTangent = v12 - v11;
  { const float len = length(localTangent); if (len >= B2_EPSILON) { const float inv = 1.0f / len; localTangent.x *= inv; localTangent.y *= inv; } }
  const V2 localNormal = cross(localTangent, 1.0f);
  const V2 planePoint = 0.5f * (v11 + v12);
  const V2 tangent = rmul(xf1.q, localTangent);
  const V2 normal = cross(tangent, 1.0f);
  v11 = xmul(xf1, v11); v12 = xmul(xf1, v12);
  const float frontOffset = dot(normal, v11);
  const float sideOffset1 = -dot(tangent, v11) + totalRadius;
  const float sideOffset2 = dot(tangent, v12) + totalRadius;
  ClipV c1[2], c2[2];
  if (clip_segment(c1, inc, -tangent, sideOffset1, iv1) < 2) return;
  if (clip_segment(c2, c1, tangent, sideOffset2, iv2) < 2) return;
  M.localNormal = localNormal; M.localPoint = planePoint;
  int pc = 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float separation = dot(normal, c2[i].v) - frontOffset;
    if (separation <= totalRadius) {
      const V2 lp = xmulT(xf2, c2[i].v);
      const uint32_t id = flip ? flipid(c2[i].id) : c2[i].id;
      if (pc == 0) { M.pl[0] = lp; M.id[0] = id; } else { M.pl[1] = lp; M.id[1] = id; }
      ++pc;
    }
  }
  M.n = pc;
}

// ------------------------------------------------------------------ solver side (dynamics kernel)
// velocity-constraint record in LDS (32 floats)
enum { VC_NX = 0, VC_NY, VC_N /*int*/, VC_LA /*int: exchange slot of body A*/, VC_LB,
       VC_P0 = 5 /* rAx rAy rBx rBy nImp tImp normalMass tangentMass */, VC_P1 = 13,
       VC_K11 = 21, VC_K12, VC_K22, VC_NM11, VC_NM12, VC_NM22, VC_REC /*int: record index in the env store*/, VC_SIZE = 32 };

}  // namespace cc
