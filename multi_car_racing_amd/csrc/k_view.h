// k_view.h — one 256-thread workgroup per agent view (env, agent): the 96x96 ego-frame software
// rasteriser that replaces pyglet/OpenGL (multi_car_racing.py:511-604, 613-674; gym Car.draw), plus the
// per-agent backward/on-grass bookkeeping of :446-495 (its result only reaches pixels one step later).
//
// HBM traffic per view: read the env's road_poly once (2 x float4 + u32 per quad, coalesced), the camera / HUD record
// and the car polygons k_dynamics prepared, write 27,648 B of packed RGB as dwordx3 stores.  Everything else lives
// in ~25 KB of LDS so that 5 workgroups (20 waves) share a CU:
//   1. workgroup -> view through the raster order list of k_dynamics (zoomed-out "heavy" envs first), XCD-aware so
//      that the N views of an env share one XCD's L2;
//   2. cull in two passes: every thread transforms its quads and rejects by pixel bbox (incl. "contains no pixel
//      centre"), survivors are ballot-compacted; then consecutive threads set the survivors up (oriented edge
//      equations -> LDS, rare overflow spills to a per-view HBM scratch).  Car polygons arrive as padded 8-gons in
//      world space (12 per car).  The backward/on-grass bookkeeping rides on the same quad pass: f32 prefilter on
//      the quads already in registers, exact f64 only for the 1-3 candidates;
//   3. bin: the thread that set a polygon up appends it to the lists of the 8x16-pixel bins it can touch
//      (box-vs-convex test, LDS atomics; order is irrelevant because the highest draw key wins);
//   4. shade: one wave per bin, each lane two pixels (rows y, y+8 share every edge product); bin ids, list
//      lengths and entries are wave-uniform (one vector LDS read per list + v_readlane).  Background (playfield +
//      checker) is analytic and classified per bin; road/kerb: highest road_poly index wins (== painter's order);
//      then cars; then the HUD in window space (scalar bin-column ranges).  Result: one palette index per pixel;
//   5. write-out: 4 pixels (one aligned palette word) -> 12 packed RGB bytes per lane, contiguous across lanes.
// Sampling rule: pixel centres; a pixel belongs to a convex polygon iff all oriented edge functions are >= 0.
#pragma once
#include "mcr_kernels.h"

namespace view {

#define VIEW_THREADS 256
// VIS_LDS: road survivors kept in LDS, more spill to HBM scratch (zoomed-out frames); BIN_CAP: entries per bin list,
// overflow -> the bin walks every survivor.  Both are template parameters of the raster body: few agents leave LDS
// for larger tables at 6 workgroups/CU, many agents need the LDS for their car polygons.
#define CAR_KEY 1024                 // bin-list ids >= CAR_KEY are car polygons (drawn after every road quad)
#define NBINS 72                     // 12 x 6 bins of 8 x 16 pixels: one wave shades a bin, each lane two pixels (y, y+8)
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
typedef float f2 __attribute__((ext_vector_type(2)));     // pixel pairs for packed-f32 arithmetic (v_pk_fma_f32)
#define UNI(x) __builtin_amdgcn_readfirstlane(x)
// per-phase s_memtime stamps of thread 0 (debug bit 32) into the tail of the view's spill area
#define PHASE_STAMP(i) do { if ((dbg & 32) && tid == 0) ((unsigned long long*)(spill + VIEW_SCRATCH_FLOATS - 64))[i] = __builtin_readcyclecounter(); } while (0)

// strict-interior point-in-quad (shapely `within`, mcr.py:470-472) on the f64 polygon the reference builds for
// tile t (kerb == false, :313-317) or for its kerb (kerb == true, :329-333)
__device__ inline bool point_in_road_poly_f64(const uint8_t* __restrict__ slot, int t, int T, bool kerb, double px, double py) {
  const double* TX = (const double*)(slot + MCR_OFF_TRACK_X); const double* TY = (const double*)(slot + MCR_OFF_TRACK_Y); const double* TB = (const double*)(slot + MCR_OFF_TRACK_B);
  const double* TC = (const double*)(slot + MCR_OFF_TRACK_C); const double* TS = (const double*)(slot + MCR_OFF_TRACK_S);
  const double TW = 40 / MCR_SCALE, TBW = 8 / MCR_SCALE;
  const int u = t == 0 ? T - 1 : t - 1;
  const double x1 = TX[t], y1 = TY[t], c1 = TC[t], s1 = TS[t], x2 = TX[u], y2 = TY[u], c2 = TC[u], s2 = TS[u];
  double X[4], Y[4];
  if (!kerb) {
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
  return pos || neg;
}

// flags_mode: 1 = evaluate the backward/on-grass block (:446-495) for this agent.
// dynamic LDS: car polygon records, N*12 x 6 float4 (8 edges each, padded with always-true edges)
template <int VIS_LDS, int BIN_CAP>
__device__ __forceinline__ void view_body(const McrParams& p, float* __restrict__ scratch, const int flags_mode, const int only_just_reset) {
  using namespace view;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = UNI(tid >> 6);
  const int N = p.N, BN = p.BN;
  // XCD-aware mapping: workgroup b runs on XCD b % 8, so hand the N views of one env to workgroups b, b+8,
  // b+16, ... — they share that XCD's L2 for the env's road_poly instead of fetching it once per XCD.
  int vw;
  if (p.role >= 2) {                 // side streams / late pass: views of the envs in the contact / deferred lists
    const int e = mcr_env_of_slot(p, blockIdx.x / N);
    if (e >= p.env0 + p.nenv) return;
    vw = e * N + (int)(blockIdx.x % N);
  } else if (p.use_vorder) {         // step path: heavy (zoomed-out) envs first, see k_dynamics
    const int b = blockIdx.x, grp = b / (8 * N), r = b - grp * 8 * N;
    const int idx = grp * 8 + (r & 7), nh = p.vcount[0], nn = p.vcount[1];
    if (idx >= nh + nn) return;
    vw = p.vorder[idx < nh ? idx : p.B - 1 - (idx - nh)] * N + (r >> 3);
  } else {
    const int b = blockIdx.x, full = (p.nenv / 8) * 8 * N;
    if (b < full) { const int grp = b / (8 * N), r = b - grp * 8 * N; vw = (p.env0 + grp * 8 + (r & 7)) * N + (r >> 3); }
    else vw = p.env0 * N + b;
    if (vw >= (p.env0 + p.nenv) * N) return;     // grids are rounded up to whole groups of 8 envs
  }
  const int env = vw / N, agent = vw % N;
  if (p.role == 1 && p.part[env]) return;
  const McrEnvState es = p.env[env];
  if (!es.active) return;
  if (only_just_reset && !es.just_reset) return;
  // side-stream raster: a contact env re-spawned by this step's dynamics is drawn after its (late) reset pass
  if (p.role >= 2 && !only_just_reset && es.resetting) return;
  const uint8_t* __restrict__ slot = p.slots + ((size_t)env * 2 + es.slot) * MCR_SLOT_BYTES;
  const McrSlotHeader* H = (const McrSlotHeader*)slot;
  const int T = H->T, P = H->P;
  const int ci = env * N + agent;
  const McrShapes& S = *p.shapes;
  const int dbg = p.debug;

  extern __shared__ __attribute__((aligned(16))) float4 car8[];             // [N*12][6]
  __shared__ __attribute__((aligned(16))) uint8_t fb[96 * 96];
  __shared__ __attribute__((aligned(16))) float4 qe[VIS_LDS][3];          // 4 oriented edges (A,B,C) of each surviving quad
  __shared__ uint32_t qinfo[VIS_LDS];                                       // quad index << 5 | palette index
  __shared__ uint16_t bins[NBINS][BIN_CAP];
  __shared__ int bcnt[NBINS];
  __shared__ uint16_t surv[MCR_QUAD_CAP];
  __shared__ uint32_t cinfo[CARPOLY_CAP];                                   // 0x100 | palette ; 0 = not drawn
  __shared__ uint32_t pal[32];
  __shared__ float fmin_w[4];
  __shared__ double cand_d[16]; __shared__ int cand_i[16];
  __shared__ int nsurv, ncand, any_inside;
  __shared__ float hud[32];                                                 // 7 gauge rectangles (x0 x1 y0 y1) + hud_top

  if (tid == 0) { nsurv = 0; ncand = 0; any_inside = 0; }
  if (tid < 32) pal[tid] = palette_rgb(tid);
  if (tid < NBINS) bcnt[tid] = 0;
  if (p.obs != nullptr && tid >= 64 && tid < 64 + 29) hud[tid - 64] = p.viewp[(size_t)vw * MCR_VIEWP_FLOATS + VP_IND + (tid - 64)];
  const uint32_t old_flags = p.caru[CU_FLAGS * BN + ci];
  const float hvx = p.carf[(CF_VX + 0) * BN + ci], hvy = p.carf[(CF_VY + 0) * BN + ci], ha = p.carf[(CF_A + 0) * BN + ci];
  const bool draw = p.obs != nullptr;
  const bool do_flags = flags_mode && !es.just_reset && !(dbg & 1);      // reset() -> step(None) skips the block (:435)
  const float* __restrict__ vp = p.viewp + (size_t)ci * MCR_VIEWP_FLOATS;
  float m00 = 0, m01 = 0, m10 = 0, m11 = 0, ctx = 0, cty = 0;
  if (draw) { m00 = vp[VP_CAM + 0]; m01 = vp[VP_CAM + 1]; m10 = vp[VP_CAM + 2]; m11 = vp[VP_CAM + 3]; ctx = vp[VP_CAM + 4]; cty = vp[VP_CAM + 5]; }
  float* __restrict__ spill = scratch + (size_t)vw * VIEW_SCRATCH_FLOATS;
  // hull.position (body origin) for the bookkeeping block
  float fpx = 0.0f, fpy = 0.0f;
  if (do_flags) { const Xf hxf = xf_of(v2(p.carf[(CF_CX + 0) * BN + ci], p.carf[(CF_CY + 0) * BN + ci]), ha, v2(S.hull_lcx, S.hull_lcy)); fpx = hxf.p.x; fpy = hxf.p.y; }
  const double dpx = (double)fpx, dpy = (double)fpy;
  PHASE_STAMP(0);
  __syncthreads();
  PHASE_STAMP(1);

  const float4* __restrict__ QA = (const float4*)(slot + MCR_OFF_QA); const float4* __restrict__ QB = (const float4*)(slot + MCR_OFF_QB);
  const uint32_t* __restrict__ QM = (const uint32_t*)(slot + MCR_OFF_QMETA);

  // ---- pass 1 over road_poly: cull (ballot-compacted survivor list) + bookkeeping prefilter
  float cd2[3] = {MCR_MAXFLT, MCR_MAXFLT, MCR_MAXFLT}; int cti[3] = {-1, -1, -1};       // nearest-track-point candidates (f32)
  bool inside = false;
  // all three quads of the thread are requested before the first is used: one exposed HBM/L2 latency instead of three
  float4 qa_[3], qb_[3]; uint32_t qm_[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int q = tid + r * VIEW_THREADS;
    if (q < P) { qa_[r] = QA[q]; qb_[r] = QB[q]; qm_[r] = QM[q]; }
    else { qa_[r] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); qb_[r] = qa_[r]; qm_[r] = 0u; }
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int q = tid + r * VIEW_THREADS;
    bool keep = false;
    if (q < P) {
      const float4 a = qa_[r], b = qb_[r];
      const uint32_t meta = qm_[r];
      if (draw && !(dbg & 16)) {
        float x0 = MCR_MAXFLT, x1 = -MCR_MAXFLT, y0 = MCR_MAXFLT, y1 = -MCR_MAXFLT;
        const float wx[4] = {a.x, a.z, b.x, b.z}, wy[4] = {a.y, a.w, b.y, b.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float px = __builtin_fmaf(m00, wx[i], __builtin_fmaf(m01, wy[i], ctx)), py = __builtin_fmaf(m10, wx[i], __builtin_fmaf(m11, wy[i], cty));
          x0 = fminf(x0, px); x1 = fmaxf(x1, px); y0 = fminf(y0, py); y1 = fmaxf(y1, py);
        }
        int i0, i1, j0, j1;
        keep = centre_range(x0, x1, 0, 95, i0, i1) && centre_range(y0, y1, 12, 95, j0, j1);   // rows < 12: HUD bar
      }
      if (do_flags) {
        const uint32_t tile1 = (meta >> 8) & 0x3ffu, owner1 = meta >> 18;
        // on-grass: f32 bbox of the quad (+margin for the f32 rounding of its vertices) before the exact f64 test
        const float bx0 = fminf(fminf(a.x, a.z), fminf(b.x, b.z)) - 0.02f, bx1 = fmaxf(fmaxf(a.x, a.z), fmaxf(b.x, b.z)) + 0.02f;
        const float by0 = fminf(fminf(a.y, a.w), fminf(b.y, b.w)) - 0.02f, by1 = fmaxf(fmaxf(a.y, a.w), fmaxf(b.y, b.w)) + 0.02f;
        if (fpx >= bx0 && fpx <= bx1 && fpy >= by0 && fpy <= by1)
          inside = inside || point_in_road_poly_f64(slot, (int)(tile1 ? tile1 : owner1) - 1, T, tile1 == 0, dpx, dpy);
        if (tile1) {   // track point ~ midpoint of the tile's leading edge (v0,v1): f32 estimate of the distance
          const float mx = 0.5f * (a.x + a.z), my = 0.5f * (a.y + a.w);
          const float ddx = fpx - mx, ddy = fpy - my;
          cd2[r] = ddx * ddx + ddy * ddy; cti[r] = (int)tile1 - 1;
        }
      }
    }
    const unsigned long long mask = __ballot(keep);
    if (mask) {
      int base = 0;
      if (lane == 0) base = atomicAdd(&nsurv, __popcll(mask));
      base = __shfl(base, 0);
      if (keep) surv[base + __popcll(mask & ((1ull << lane) - 1ull))] = (uint16_t)q;
    }
  }
  if (do_flags) {
    if (__any(inside) && lane == 0) atomicOr(&any_inside, 1);
    float m = fminf(cd2[0], fminf(cd2[1], cd2[2]));
    for (int o = 32; o > 0; o >>= 1) m = fminf(m, __shfl_xor(m, o));
    if (lane == 0) fmin_w[wave] = m;
  }

  PHASE_STAMP(2);
  PHASE_STAMP(3);
  __syncthreads();
  PHASE_STAMP(4);

  // ---- car polygons (Car.draw): world vertices come from k_dynamics; the LAST 12N threads set one polygon up each,
  // in the same phase as pass 2 — survivors fill the threads from 0 upwards (~100-150 of them on a normal frame), so
  // the two set-ups usually run on different wavefronts instead of one after the other
  if (draw && tid >= VIEW_THREADS - N * 12) {
    const int k = tid - (VIEW_THREADS - N * 12), c = k / 12, j = k % 12;
    const float* __restrict__ cp = p.carpoly + (size_t)(env * N + c) * MCR_CARPOLY_FLOATS;
    // one burst: 8 vertices (4 x float4) + the vertex count, all issued before anything is consumed
    const float* cv = cp + j * 16;
    const float4 v01 = ((const float4*)cv)[0], v23 = ((const float4*)cv)[1], v45 = ((const float4*)cv)[2], v67 = ((const float4*)cv)[3];
    const int n = __float_as_int(cp[MCR_CARPOLY_NOFF + j]);
    uint32_t info = 0;
    if (n > 0 && !(dbg & 4)) {
      uint32_t colr;
      if (j < 8) colr = (j & 1) ? PAL_WHEELWHITE : PAL_BLACK;
      else { colr = PAL_CAR0 + (c & 7); if (p.use_ego_color) colr = (c == agent) ? PAL_CAR0 + 0 : PAL_CAR0 + 1; }   // :402, :560-563
      // every car polygon arrives padded to 8 vertices (last vertex repeated): the extra edges are degenerate
      // (A = B = C = 0 -> always "inside"), so all loops below are fixed-size and fully unrolled
      const float wxs[8] = {v01.x, v01.z, v23.x, v23.z, v45.x, v45.z, v67.x, v67.z}, wys[8] = {v01.y, v01.w, v23.y, v23.w, v45.y, v45.w, v67.y, v67.w};
      float px[8], py[8];
      float x0 = MCR_MAXFLT, x1 = -MCR_MAXFLT, y0 = MCR_MAXFLT, y1 = -MCR_MAXFLT;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        px[i] = __builtin_fmaf(m00, wxs[i], __builtin_fmaf(m01, wys[i], ctx)); py[i] = __builtin_fmaf(m10, wxs[i], __builtin_fmaf(m11, wys[i], cty));
        x0 = fminf(x0, px[i]); x1 = fmaxf(x1, px[i]); y0 = fminf(y0, py[i]); y1 = fmaxf(y1, py[i]);
      }
      float e[24];
      int ix0, ix1, iy0, iy1;
      float area = 0.0f;
#pragma unroll
      for (int i = 0; i < 8; ++i) { const int j = (i + 1) & 7; area += px[i] * py[j] - px[j] * py[i]; }
      if (centre_range(x0, x1, 0, 95, ix0, ix1) && centre_range(y0, y1, 12, 95, iy0, iy1) && area != 0.0f) {
        const float sg = area > 0.0f ? 1.0f : -1.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int j = (i + 1) & 7;
          const float ex = px[j] - px[i], ey = py[j] - py[i];
          const float A = -sg * ey, B = sg * ex;
          e[i * 3 + 0] = A; e[i * 3 + 1] = B; e[i * 3 + 2] = -(A * px[i] + B * py[i]);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) car8[k * 6 + i] = make_float4(e[i * 4], e[i * 4 + 1], e[i * 4 + 2], e[i * 4 + 3]);
        info = 0x100u | colr;
        for (int by = iy0 >> 4; by <= (iy1 >> 4); ++by)
          for (int bx = ix0 >> 3; bx <= (ix1 >> 3); ++bx) {
            const float X0 = (float)(bx * 8) + 0.5f, Y0 = (float)(by * 16) + 0.5f, X1 = X0 + 7.0f, Y1 = Y0 + 15.0f;
            bool out = false;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
              const float A = e[kk * 3], B = e[kk * 3 + 1], C = e[kk * 3 + 2];
              out = out || (A * (A >= 0.0f ? X1 : X0) + B * (B >= 0.0f ? Y1 : Y0) + C < 0.0f);
            }
            if (out) continue;
            const int b = by * 12 + bx;
            const int at = atomicAdd(&bcnt[b], 1);
            if (at < BIN_CAP) bins[b][at] = (uint16_t)(CAR_KEY + k);
          }
      }
    }
    cinfo[k] = info;
  }
  // ---- pass 2: dense set-up of the survivors (edge equations, colour, bins)
  const int nq = nsurv;
  const int nq_lds = nq < VIS_LDS ? nq : VIS_LDS;
  if (draw) {
    const uint16_t* __restrict__ tflags = p.tile_flags + (size_t)env * MCR_TILE_CAP;
    for (int s = tid; s < nq; s += VIEW_THREADS) {
      const int q = surv[s];
      const float4 a = QA[q], b = QB[q];
      const float wx[4] = {a.x, a.z, b.x, b.z}, wy[4] = {a.y, a.w, b.y, b.w};
      float px[4], py[4];
      float x0 = MCR_MAXFLT, x1 = -MCR_MAXFLT, y0 = MCR_MAXFLT, y1 = -MCR_MAXFLT;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        px[i] = __builtin_fmaf(m00, wx[i], __builtin_fmaf(m01, wy[i], ctx)); py[i] = __builtin_fmaf(m10, wx[i], __builtin_fmaf(m11, wy[i], cty));
        x0 = fminf(x0, px[i]); x1 = fmaxf(x1, px[i]); y0 = fminf(y0, py[i]); y1 = fmaxf(y1, py[i]);
      }
      int ix0, ix1, iy0, iy1;
      centre_range(x0, x1, 0, 95, ix0, ix1); centre_range(y0, y1, 12, 95, iy0, iy1);
      float e[12];
      const bool ok = edge_setup(px, py, 4, e);
      if (!ok) { for (int i = 0; i < 12; i += 3) { e[i] = 0.0f; e[i + 1] = 0.0f; e[i + 2] = -1.0f; } }   // never covers a pixel
      const uint32_t meta = QM[q];
      uint32_t col = meta & 0xffu; const uint32_t tile1 = (meta >> 8) & 0x3ffu;
      if (tile1) {                                                              // touched tile -> ROAD_COLOR (:102-104)
        const uint32_t w = ((const uint32_t*)tflags)[(tile1 - 1) >> 1];
        if ((w >> (((tile1 - 1) & 1u) * 16u)) & 0x100u) col = MCR_COL_ROAD0;
      }
      // draw-order key: quad index above the 5-bit palette index, so "highest key wins" also carries the colour
      const uint32_t pal = col == MCR_COL_ROAD0 ? PAL_ROAD0 : col == MCR_COL_ROAD1 ? PAL_ROAD1 : col == MCR_COL_ROAD2 ? PAL_ROAD2 : col == MCR_COL_KERB_WHITE ? PAL_WHITE : PAL_RED255;
      const uint32_t info = ((uint32_t)q << 5) | pal;
      if (s < VIS_LDS) {
        qe[s][0] = make_float4(e[0], e[1], e[2], e[3]); qe[s][1] = make_float4(e[4], e[5], e[6], e[7]); qe[s][2] = make_float4(e[8], e[9], e[10], e[11]);
        qinfo[s] = info;
      } else {
        float4* d = (float4*)(spill + (size_t)(s - VIS_LDS) * 16);
        d[0] = make_float4(e[0], e[1], e[2], e[3]); d[1] = make_float4(e[4], e[5], e[6], e[7]); d[2] = make_float4(e[8], e[9], e[10], e[11]);
        d[3] = make_float4(__uint_as_float(info), 0.0f, 0.0f, 0.0f);
      }
      if (ok)
        for (int by = iy0 >> 4; by <= (iy1 >> 4); ++by)
          for (int bx = ix0 >> 3; bx <= (ix1 >> 3); ++bx) {
            const float X0 = (float)(bx * 8) + 0.5f, Y0 = (float)(by * 16) + 0.5f;
            const float X1 = X0 + 7.0f, Y1 = Y0 + 15.0f;
            bool out = false;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
              const float A = e[kk * 3], B = e[kk * 3 + 1], C = e[kk * 3 + 2];
              out = out || (A * (A >= 0.0f ? X1 : X0) + B * (B >= 0.0f ? Y1 : Y0) + C < 0.0f);
            }
            if (out) continue;
            const int bb = by * 12 + bx;
            const int at = atomicAdd(&bcnt[bb], 1);
            if (at < BIN_CAP) bins[bb][at] = (uint16_t)s;
          }
    }
  }
  // ---- bookkeeping stage 2: exact f64 distance for the tiles whose f32 distance is within the error band of the minimum
  if (do_flags) {
    const float m = fminf(fminf(fmin_w[0], fmin_w[1]), fminf(fmin_w[2], fmin_w[3]));
    const float band = sqrtf(m) * (1.0f + 1e-5f) + 2e-3f;
    const float thr = band * band;
    const double* TX = (const double*)(slot + MCR_OFF_TRACK_X); const double* TY = (const double*)(slot + MCR_OFF_TRACK_Y);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      if (cti[r] >= 0 && cd2[r] <= thr) {
        const double dx = dpx - TX[cti[r]], dy = dpy - TY[cti[r]];
        const double d = sqrt(dx * dx + dy * dy);             // np.linalg.norm(..., axis=1) then argmin (:465-467)
        const int at = atomicAdd(&ncand, 1);
        if (at < 16) { cand_d[at] = d; cand_i[at] = cti[r]; }
      }
    }
  }
  PHASE_STAMP(5);
  __syncthreads();
  PHASE_STAMP(6);

  // ---- bookkeeping stage 3 (one lane of the last wave, while the other waves shade)
  if (do_flags && tid == VIEW_THREADS - 1) {
    const double* TB = (const double*)(slot + MCR_OFF_TRACK_B);
    int nc = ncand; if (nc > 16) nc = 16;
    double bd = 1e300; int bi = 0x7fffffff;
    for (int i = 0; i < nc; ++i) if (cand_d[i] < bd || (cand_d[i] == bd && cand_i[i] < bi)) { bd = cand_d[i]; bi = cand_i[i]; }
    if (ncand > 16 || nc == 0) {          // pathological tie cluster: fall back to the exact full scan
      const double* TX = (const double*)(slot + MCR_OFF_TRACK_X); const double* TY = (const double*)(slot + MCR_OFF_TRACK_Y);
      bd = 1e300; bi = 0;
      for (int t = 0; t < T; ++t) { const double dx = dpx - TX[t], dy = dpy - TY[t]; const double d = sqrt(dx * dx + dy * dy); if (d < bd) { bd = d; bi = t; } }
    }
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

  // Coverage tests use fused multiply-adds: pixel colours are compared against the ideal raster with an ambiguity
  // band (tests/util.py), so the raster is free to evaluate edge functions MORE accurately than mul+add would.
#define FMA(a, b, c) __builtin_fmaf((a), (b), (c))
  // ---- shade: one wave per 8x16 bin, each lane owns the pixels (x, y) and (x, y+8) of the bin.  Bin ids, list
  // lengths and list entries are wave-uniform (scalar registers): list walking costs scalar branches and broadcast
  // LDS reads; the second pixel of a lane re-uses every edge value (e(y+8) = e(y) + 8*B).
  {
    const bool show_flag = (old_flags & 1u) && p.backwards_flag;
    const float kx = 96.0f / 1000.0f, ky = 96.0f / 800.0f;
    float fe[9];
    { const float fx[3] = {900.0f * kx, 925.0f * kx, 950.0f * kx}, fy[3] = {30.0f * ky, 70.0f * ky, 30.0f * ky}; edge_setup(fx, fy, 3, fe); }
    const float hud_top = hud[VP_HUDTOP - VP_IND];
    const int hud_rows = UNI((int)ceilf(hud_top * 0.0625f));             // bin rows that can contain HUD pixels
    int hbx0[7], hbx1[7];                                                // bin columns each gauge can touch (scalar); empty: 1..0
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const float x0 = hud[i * 4], x1 = hud[i * 4 + 1], y0 = hud[i * 4 + 2], y1 = hud[i * 4 + 3];
      const bool live = x1 > x0 && y1 > y0;
      hbx0[i] = UNI(live ? max(0, (int)floorf(x0 * 0.125f)) : 1);
      hbx1[i] = UNI(live ? min(11, (int)floorf(x1 * 0.125f)) : 0);
    }
    const int lx = lane & 7, ly = lane >> 3;
    const float flx = (float)lx + 0.5f, fly = (float)ly + 0.5f;
    // background in "checker units": U = world.x / (2k), V = world.y / (2k) with k = PLAYFIELD/20;
    // playfield <=> |U|,|V| <= 10; light square <=> frac(U) < .5 and frac(V) < .5 (:615-627)
    const float hk = 0.5f / (float)(MCR_PLAYFIELD / 20.0);
    const float aU = vp[VP_INV + 0] * hk, bU = vp[VP_INV + 1] * hk, cU = vp[VP_INV + 2] * hk;
    const float aV = vp[VP_INV + 3] * hk, bV = vp[VP_INV + 4] * hk, cV = vp[VP_INV + 5] * hk;
    const float U_lane = aU * flx + bU * fly + cU, V_lane = aV * flx + bV * fly + cV;
    const float dU8 = 8.0f * bU, dV8 = 8.0f * bV;
    const int fb_lane = (95 - ly) * 96 + lx;
    const int ncar = N * 12;
    // list lengths of this wave's 18 bins, one per lane; a bin's list is fetched with ONE vector LDS read (lane k
    // holds entry k) and handed out with v_readlane, so walking a list costs no LDS round trip per entry
    int cnt_v = 0;
    if (lane < NBINS / 4) { const int yy = lane / 3; cnt_v = bcnt[yy * 12 + ((wave - 2 * yy) & 3) + 4 * (lane - yy * 3)]; }
    // background class per bin, one bin per lane: when the (U,V) bounding box of a bin's pixel centres stays clear of
    // every checker boundary (margin 1e-4 units >> f32 rounding of the per-pixel evaluation) the bin is one colour
    int ucol_v = 0xff;
    if (lane < NBINS / 4) {
      const int yy = lane / 3, xx = ((wave - 2 * yy) & 3) + 4 * (lane - yy * 3);
      const float X0 = (float)(xx * 8) + 0.5f, Y0 = (float)(yy * 16) + 0.5f;
      const float u00 = aU * X0 + bU * Y0 + cU, v00 = aV * X0 + bV * Y0 + cV;
      const float ux = 7.0f * aU, uy = 15.0f * bU, vx = 7.0f * aV, vy = 15.0f * bV;
      const float umin = u00 + fminf(ux, 0.0f) + fminf(uy, 0.0f) - 1e-4f, umax = u00 + fmaxf(ux, 0.0f) + fmaxf(uy, 0.0f) + 1e-4f;
      const float vmin = v00 + fminf(vx, 0.0f) + fminf(vy, 0.0f) - 1e-4f, vmax = v00 + fmaxf(vx, 0.0f) + fmaxf(vy, 0.0f) + 1e-4f;
      if (umin > 10.0f || umax < -10.0f || vmin > 10.0f || vmax < -10.0f) ucol_v = PAL_BLACK;
      else if (umin >= -10.0f && umax <= 10.0f && vmin >= -10.0f && vmax <= 10.0f) {
        const float fu = floorf(umin), fv = floorf(vmin);
        const bool u_dark = umin - fu > 0.5f && umax < fu + 1.0f, v_dark = vmin - fv > 0.5f && vmax < fv + 1.0f;
        const bool u_light = umax < fu + 0.5f, v_light = vmax < fv + 0.5f;
        if (u_dark || v_dark) ucol_v = PAL_GRASS0;
        else if (u_light && v_light) ucol_v = PAL_GRASS1;
      }
    }
    for (int it = 0; it < NBINS / 4; ++it) {
      // wave w takes the bins with (bx + 2*by) % 4 == w: any 2x2 block of bins lands on four different waves
      const int byi = it / 3, bxi = ((wave - 2 * byi) & 3) + 4 * (it - byi * 3);   // scalar
      const int b = byi * 12 + bxi;
      const int list_v = (int)bins[b][lane < BIN_CAP ? lane : 0];
      const float fbx = (float)(bxi * 8), fby = (float)(byi * 16);
      const float cx = fbx + flx, cy0 = fby + fly, cy1 = cy0 + 8.0f;    // pixel centres, GL coords (origin bottom-left)
      const f2 cxx = {cx, cx}, cyy = {cy0, cy1};
      uint32_t col0 = PAL_BLACK, col1 = PAL_BLACK;
      const int ucol = __builtin_amdgcn_readlane(ucol_v, it);
      if (ucol != 0xff) col0 = col1 = (uint32_t)ucol;
      else {
        const float U0 = U_lane + aU * fbx + bU * fby, V0 = V_lane + aV * fbx + bV * fby;
        const float U1 = U0 + dU8, V1 = V0 + dV8;
        if (fabsf(U0) <= 10.0f && fabsf(V0) <= 10.0f) col0 = ((U0 - floorf(U0)) < 0.5f && (V0 - floorf(V0)) < 0.5f) ? PAL_GRASS1 : PAL_GRASS0;
        if (fabsf(U1) <= 10.0f && fabsf(V1) <= 10.0f) col1 = ((U1 - floorf(U1)) < 0.5f && (V1 - floorf(V1)) < 0.5f) ? PAL_GRASS1 : PAL_GRASS0;
      }
      // highest draw index covering the pixel wins (painter's order): road_poly index, then car polygons
      const int cnt = __builtin_amdgcn_readlane(cnt_v, it);
      int best0 = -1, best1 = -1;
      // both pixels of the lane share every edge's coefficients: packed f32 FMAs (v_pk_fma_f32) evaluate an edge
      // function at (cx, cy0) and (cx, cy1) in one instruction
#define PKF(A, X, Y) __builtin_elementwise_fma((f2){(A), (A)}, (X), (Y))
#define EDGE4_2PX(r0, r1, r2, in0, in1)                                                                                   \
      {                                                                                                                  \
        const f2 e0 = PKF(r0.x, cxx, PKF(r0.y, cyy, ((f2){r0.z, r0.z}))), e1 = PKF(r0.w, cxx, PKF(r1.x, cyy, ((f2){r1.y, r1.y}))),  \
                 e2 = PKF(r1.z, cxx, PKF(r1.w, cyy, ((f2){r2.x, r2.x}))), e3 = PKF(r2.y, cxx, PKF(r2.z, cyy, ((f2){r2.w, r2.w})));  \
        in0 = fminf(fminf(e0.x, e1.x), fminf(e2.x, e3.x)) >= 0.0f;                                                       \
        in1 = fminf(fminf(e0.y, e1.y), fminf(e2.y, e3.y)) >= 0.0f;                                                       \
      }
      if (cnt <= BIN_CAP) {
        for (int k = 0; k < cnt; ++k) {
          const int s = __builtin_amdgcn_readlane(list_v, k);
          if (s >= CAR_KEY) {
            const float4* r = &car8[(s - CAR_KEY) * 6];
            bool a0, a1, b0, b1;
            EDGE4_2PX(r[0], r[1], r[2], a0, a1); EDGE4_2PX(r[3], r[4], r[5], b0, b1);
            const int key = (s << 5) | (int)(cinfo[s - CAR_KEY] & 31u);
            if (a0 && b0 && key > best0) best0 = key;
            if (a1 && b1 && key > best1) best1 = key;
          } else if (s < VIS_LDS) {
            const int key = (int)qinfo[s];
            bool a0, a1; EDGE4_2PX(qe[s][0], qe[s][1], qe[s][2], a0, a1);
            if (!(dbg & 2)) { if (a0 && key > best0) best0 = key; if (a1 && key > best1) best1 = key; }
          } else {
            const float4* d = (const float4*)(spill + (size_t)(s - VIS_LDS) * 16);
            const int key = (int)__float_as_uint(d[3].x);
            bool a0, a1; EDGE4_2PX(d[0], d[1], d[2], a0, a1);
            if (a0 && key > best0) best0 = key;
            if (a1 && key > best1) best1 = key;
          }
        }
      } else {
        for (int s = 0; s < nq_lds; ++s) {
          const int key = (int)qinfo[s];
          bool a0, a1; EDGE4_2PX(qe[s][0], qe[s][1], qe[s][2], a0, a1);
          if (a0 && key > best0) best0 = key;
          if (a1 && key > best1) best1 = key;
        }
        for (int s = VIS_LDS; s < nq; ++s) {
          const float4* d = (const float4*)(spill + (size_t)(s - VIS_LDS) * 16);
          const int key = (int)__float_as_uint(d[3].x);
          bool a0, a1; EDGE4_2PX(d[0], d[1], d[2], a0, a1);
          if (a0 && key > best0) best0 = key;
          if (a1 && key > best1) best1 = key;
        }
        for (int k = 0; k < ncar; ++k) {
          const uint32_t ci2 = cinfo[k];
          if (!ci2) continue;
          const float4* r = &car8[k * 6];
          const int key = ((CAR_KEY + k) << 5) | (int)(ci2 & 31u);
          bool a0, a1, b0, b1;
          EDGE4_2PX(r[0], r[1], r[2], a0, a1); EDGE4_2PX(r[3], r[4], r[5], b0, b1);
          if (a0 && b0 && key > best0) best0 = key;
          if (a1 && b1 && key > best1) best1 = key;
        }
      }
#undef EDGE4_2PX
#undef PKF
#undef FMA
      if (best0 >= 0) col0 = (uint32_t)best0 & 31u;
      if (best1 >= 0) col1 = (uint32_t)best1 & 31u;
      if (byi < hud_rows) {
        // the HUD bar (window y < 100) covers the scene; gauges in draw order (a tall gauge may poke above the
        // bar), then the backwards flag — all in window space, drawn last
        if (cy0 < 12.0f) col0 = PAL_BLACK;
        if (cy1 < 12.0f) col1 = PAL_BLACK;
        const uint32_t ind_col[7] = {PAL_WHITE, PAL_BLUE255, PAL_BLUE255, PAL_PURPLE, PAL_PURPLE, PAL_GREEN255, PAL_RED255};
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          if (bxi < hbx0[i] || bxi > hbx1[i]) continue;                // scalar: a gauge is 2.4 px wide, 1-2 bin columns
          const float x0 = hud[i * 4], x1 = hud[i * 4 + 1], y0 = hud[i * 4 + 2], y1 = hud[i * 4 + 3];
          const bool okx = cx >= x0 && cx <= x1;
          if (okx && cy0 >= y0 && cy0 <= y1) col0 = ind_col[i];
          if (okx && cy1 >= y0 && cy1 <= y1) col1 = ind_col[i];
        }
        if (show_flag && byi == 0 && bxi >= 10) {                      // the flag triangle spans x 86.4..91.2, y 3.6..8.4
          if ((fe[0] * cx + fe[1] * cy0 + fe[2] >= 0.0f) && (fe[3] * cx + fe[4] * cy0 + fe[5] >= 0.0f) && (fe[6] * cx + fe[7] * cy0 + fe[8] >= 0.0f)) col0 = PAL_BLUE255;
          if ((fe[0] * cx + fe[1] * cy1 + fe[2] >= 0.0f) && (fe[3] * cx + fe[4] * cy1 + fe[5] >= 0.0f) && (fe[6] * cx + fe[7] * cy1 + fe[8] >= 0.0f)) col1 = PAL_BLUE255;
        }
      }
      const int fo = fb_lane - byi * (16 * 96) + bxi * 8;               // arr[::-1] (:602)
      fb[fo] = (uint8_t)col0; fb[fo - 8 * 96] = (uint8_t)col1;
    }
  }
  PHASE_STAMP(7);
  __syncthreads();
  PHASE_STAMP(8);
  if (dbg & 8) return;

  // ---- packed RGB write-out: 4 pixels (one aligned LDS word of palette indices) -> 12 bytes per lane
  {
    uint32_t* __restrict__ out = (uint32_t*)(p.obs + (size_t)vw * (96 * 96 * 3));
    const uint32_t* fb4 = (const uint32_t*)fb;
    for (int i = tid; i < 96 * 96 / 4; i += VIEW_THREADS) {
      const uint32_t ix4 = fb4[i];
      const uint32_t c0 = pal[ix4 & 255u], c1 = pal[(ix4 >> 8) & 255u], c2 = pal[(ix4 >> 16) & 255u], c3 = pal[ix4 >> 24];
      uint3 w; w.x = c0 | (c1 << 24); w.y = (c1 >> 8) | (c2 << 16); w.z = (c2 >> 16) | (c3 << 8);
      *(uint3*)(out + (size_t)i * 3) = w;
    }
  }
  PHASE_STAMP(9);
}

// N <= 2: 2.3 KB of car polygons in dynamic LDS -> 25 KB per workgroup, 6 workgroups per CU
__global__ __launch_bounds__(VIEW_THREADS, 6) void k_view(McrParams p, float* __restrict__ scratch, int flags_mode, int only_just_reset) {
  view_body<144, 24>(p, scratch, flags_mode, only_just_reset);
}
// N > 2: up to 9.2 KB of car polygons -> smaller survivor / bin tables keep 5 workgroups per CU
__global__ __launch_bounds__(VIEW_THREADS, 5) void k_view_many(McrParams p, float* __restrict__ scratch, int flags_mode, int only_just_reset) {
  view_body<112, 20>(p, scratch, flags_mode, only_just_reset);
}
