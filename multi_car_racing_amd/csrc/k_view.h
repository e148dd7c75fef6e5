// k_view.h — one 256-thread workgroup per env, drawing its N agent views one after the other: the 96x96 ego-frame
// software rasteriser that replaces pyglet/OpenGL (multi_car_racing.py:511-604, 613-674; gym Car.draw).  What the views
// of an env share — episode slot header, tile flags, the road quads in L1/L2 — is fetched once per workgroup.
// (The backward/on-grass bookkeeping of :446-495 that round 1 computed in here is k_flags.h now.)
//
// Round-2 design: a SPAN rasteriser.  Round 1 tested every polygon of a bin list against every pixel of its 8x16 bin
// (64 k pixel tests per view for ~4 k covered pixels) and was issue-bound on three pipes at once (SALU list walking,
// VALU edge functions, broadcast LDS reads; profiles/r02_ubench_issue_rates.txt).  Here the work is proportional to what
// is actually drawn:
//   1. candidates, in rounds of 248 (one thread each): the road_poly quads (read once, coalesced) and the view's
//      "special" polygons — car polygons k_dynamics prepared, gauges, flag, the light grass squares the viewport can
//      see.  A thread transforms its polygon, culls it exactly (does its bbox hold a pixel centre?) and, if it survives,
//      builds a 48-byte record in its own slot: the polygon is scanned along its LONGER bounding-box axis (rows or
//      columns), so a scan line crosses it in a short span; per edge the span bound is affine in the line coordinate,
//      bound(v) = s*v + c, stored once as a lower and once as an upper bound (+-1e30 when the edge is of the other
//      kind), so a span end is a max / min of four FMAs — no per-pixel edge tests at all.  No compaction, no atomics:
//      a culled slot simply has zero lines.  The next round's HBM data is requested before this round's spans are drawn;
//   2. tasks: a block-wide prefix sum over the slots' line counts lays out (record, line) tasks, 64 per wavefront,
//      whatever the polygon sizes are;
//   4. fill: each lane computes its line's span [lo, hi] and writes the polygon's draw key over it with LDS ds_max_u32 —
//      "highest draw key wins" is the painter's order of the reference (playfield, grass squares, road_poly in creation
//      order, cars, HUD), and the palette index rides in the key's low bits;
//   5. resolve: key buffer -> palette -> packed RGB, 4 pixels = 12 bytes per lane, contiguous dwordx3 stores with the
//      vertical flip of :602 folded into the address.
// The playfield base colour and the HUD bar are part of the key-buffer clear; the score label (:665-666) is stamped by
// one wavefront with the top key.  52 KB of LDS per workgroup -> 3 workgroups per CU (LDS is granted in 2 KB steps).
// Sampling rule: pixel centres; a centre belongs to a convex polygon iff it lies inside every edge (closed).  Span
// ends are computed by division instead of evaluating the edge function at the centre; the two agree except for
// centres within ~1e-5 px of an edge — inside the 0.02 px band in which the oracle declares a pixel ambiguous.
#pragma once
#include "k_raster_common.h"

namespace view {

constexpr int KS = 97;                 // key-buffer row stride in words: odd, so that a column of pixels walks all LDS banks
constexpr int RC = 248;                // candidate slots (= record slots) per round
constexpr int TASK4_CAP = 592;         // task entries per chunk; an entry covers 4 consecutive scan lines of one record
constexpr float BIG = 1e30f;
constexpr int NBLK = MCR_QUAD_CAP / MCR_QBLK;   // road_poly culling blocks per track
// A pixel of the key buffer is rank << 24 | RGB: "highest rank wins" (ds_max_u32) is the painter's order of the reference
// and the winner's colour needs no lookup.  Ranks: the playfield quad, the light grass squares, then the candidates in
// slot order (road_poly entries in creation order, cars, gauges, flag: RANK_SLOT0 + slot), the score label on top.
// A view that needs several rounds of candidates flattens what is drawn so far to RANK_FLAT before each further round.
enum { RANK_BASE = 0, RANK_PLAYFIELD = 1, RANK_GRASS = 2, RANK_FLAT = 3, RANK_SLOT0 = 4, RANK_LABEL = 255 };
// record key field: palette index [0,5) | rank [5,13) | HUD polygon, not clipped to the scene rows [13]
__device__ __forceinline__ uint32_t mk_key(int rank, int pal) { return ((uint32_t)rank << 5) | (uint32_t)pal; }
#define REC_HUD (1u << 13)
// record meta word: key field [0,16) | row-scan [16] | chained second half in the next slot [17] | first line [18,25) | lines [25,32)
#define REC_ROW (1u << 16)
#define REC_CHAIN (1u << 17)

// One edge (ax,ay)->(bx,by) of a polygon with orientation sign sg: inside <=> A x + B y + C >= 0.  With u the
// coordinate along the span and v the line coordinate the edge bounds u from below (Au > 0) or above (Au < 0) by
// s v + c; an edge parallel to the spans (Au == 0) bounds nothing (the polygon's line range already accounts for it).
__device__ __forceinline__ void span_edge(float ax, float ay, float bx, float by, float sg, bool row, float& s, float& cl, float& ch) {
  const float ex = bx - ax, ey = by - ay;
  const float A = -sg * ey, B = sg * ex, C = -(A * ax + B * ay);
  const float Au = row ? A : B, Av = row ? B : A;
  const float r = __builtin_amdgcn_rcpf(Au);
  const float c = -C * r;
  s = Au != 0.0f ? -Av * r : 0.0f;
  cl = Au > 0.0f ? c : -BIG;
  ch = Au < 0.0f ? c : BIG;
}
// Span record of four consecutive edges v0->v1->v2->v3->v4 of a convex polygon in pixel space that survived the cull with
// pixel-centre ranges [i0,i1] x [j0,j1] (a quad: v4 = v0; an 8-gon takes two records in adjacent slots, the halves 0..4 and
// 4..7,0, written by two threads that share the polygon's orientation, scan axis and line range).  `area`: the polygon's
// signed area x 2.  Returns the scan-axis / line-range bits of the meta word (0: degenerate, nothing to draw).
__device__ __forceinline__ uint32_t setup_half(const float* x5, const float* y5, float area, int i0, int i1, int j0, int j1,
                                               float4 (*rdat)[3], int slot) {
  uint32_t meta = 0u;
  if (area != 0.0f) {
    const float sg = area > 0.0f ? 1.0f : -1.0f;
    const bool row = (j1 - j0) >= (i1 - i0);                       // scan along the longer axis: many short spans
    const int l0 = row ? j0 : i0, nl = (row ? j1 : i1) - l0 + 1;
    float4 S4, L4, H4;
    span_edge(x5[0], y5[0], x5[1], y5[1], sg, row, S4.x, L4.x, H4.x); span_edge(x5[1], y5[1], x5[2], y5[2], sg, row, S4.y, L4.y, H4.y);
    span_edge(x5[2], y5[2], x5[3], y5[3], sg, row, S4.z, L4.z, H4.z); span_edge(x5[3], y5[3], x5[4], y5[4], sg, row, S4.w, L4.w, H4.w);
    rdat[slot][0] = S4; rdat[slot][1] = L4; rdat[slot][2] = H4;
    meta = (row ? REC_ROW : 0u) | ((uint32_t)l0 << 18) | ((uint32_t)nl << 25);
  }
  return meta;
}

}  // namespace view

// per-phase clock accumulators of thread 0 (the PHASES instantiation, launched when debug bit 32 is set): [view][16] u64,
// summed over the rounds of the view.  A separate instantiation: the accumulators cost 22 VGPRs the kernel does not have.
#define PHASE_ACC(i) do { if constexpr (PHASES) { const unsigned long long now_ = __builtin_readcyclecounter(); pacc[i] += now_ - tprev; tprev = now_; } } while (0)

template <bool PHASES, bool PERSIST>
__global__ __launch_bounds__(VIEW_THREADS, 3) void k_view(McrParams p, unsigned long long* __restrict__ stamps, const int only_just_reset) {
  using namespace view;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = UNI(tid >> 6);
  const int N = p.N;
  // ---- this workgroup's envs.  Main launches: one env per workgroup (work slot = blockIdx; the hardware's dispatcher
  // balances them better than a static assignment can: measured).  PERSIST — the list launches of the side streams, whose
  // lists are short and whose length only the device knows: a small grid, workgroup b draws the envs of work slots
  // b, b + gridDim, b + 2 gridDim, .. one after the other (lane k of every wavefront looks up slot b + k gridDim; the
  // host sizes the grid so that 64 lanes cover the list), and what the next env needs from HBM — header, tile flags,
  // view record, visible-block list, first candidates — is requested while the current one is drawn.
  // Whose business is a work slot?
  //   role >= 2 (side streams): the envs of the contact / deferred lists;  use_vorder (step path): the order k_dynamics
  //   recorded, heavy (zoomed-out) envs first;  role 1: not the envs the side streams draw;  only_just_reset: reset().
  if (PERSIST) __builtin_amdgcn_s_setprio(3);                               // a few envs beside the main launch that fills every CU: they go first
  const int vgrid = (int)gridDim.x - (PERSIST ? p.flags_blocks : 0);        // workgroups that draw; the rest: the list's bookkeeping
  if (PERSIST && (int)blockIdx.x >= vgrid) {
    // the bookkeeping of the list's cars (k_flags.h: backward / on-grass flags, the env's touch verdict for the next step), one wavefront
    // per car.  Contact list for the contact chain's raster, deferred list otherwise (the re-spawned envs' cars take none in this step).
    const int32_t* __restrict__ L = p.role == 2 ? p.clist : p.dlist;
    const int ncars = L[0] * N, per_round = p.flags_blocks * (VIEW_THREADS / 64);
    for (int c = ((int)blockIdx.x - vgrid) * (VIEW_THREADS / 64) + (int)(threadIdx.x >> 6); c < ncars; c += per_round) flags_block(p, c % N, L[1 + c / N]);
    return;
  }
  int my_env = -1, my_slot = 0, my_P = -1, my_agent = 0;
  // List launches with `split_views`: a work slot is ONE VIEW (list entry s / N, agent s % N) instead of an env with its N views —
  // the few envs of a list are the tail of a chain on the step's critical path, and their views side by side take half the time
  // of one after the other (what an env's views share is fetched once per view then).
  const bool split_views = PERSIST && p.split_views != 0;
  {
    const int s = (int)blockIdx.x + (PERSIST ? lane * vgrid : 0);
    int e = -1;
    bool ojr = only_just_reset != 0;
    if (p.role == 5) {
      // the tail of the caller's stream: the deferred envs, then the envs the main dynamics re-spawned (their first observation) — one launch
      const int idx = split_views ? s / N : s, nd = p.dlist[0], nr = p.rlist[0];
      if (idx < nd) e = p.dlist[1 + idx]; else if (idx - nd < nr) { e = p.rlist[1 + idx - nd]; ojr = true; }
      my_agent = split_views ? s % N : 0;
    }
    else if (p.role >= 2) { e = mcr_env_of_slot(p, split_views ? s / N : s); if (e >= p.env0 + p.nenv) e = -1; my_agent = split_views ? s % N : 0; }
    else if (p.use_vorder) {
      // one load instead of a chain of four (list counts -> list entry -> env record -> slot header): the entry k_dynamics left
      // carries the env, its episode slot and the slot's entry count; only envs this launch draws are listed (active, not
      // the contact chain's, not deferred), unused entries hold -1 and every used one is reset below for its next use
      const int v = s < p.B ? p.vorder[s] : -1;
      if (v != -1) { e = v & MCR_VORDER_ENV_MASK; my_slot = (v >> MCR_VORDER_SLOT_SHIFT) & 1; my_P = (v >> MCR_VORDER_P_SHIFT) & 1023; }
    }
    else if (s < p.nenv) e = p.env0 + s;
    if (e >= 0 && !p.use_vorder) {
      if (p.role == 1 && (p.part[e] || p.dpart[e])) e = -1;
    }
    if (e >= 0 && (!p.use_vorder || p.role >= 2)) {
      const McrEnvState es = p.env[e];
      // side-stream raster: a contact env re-spawned by this step's dynamics is drawn after its reset pass
      if (!es.active || (ojr && !es.just_reset) || (p.role >= 2 && !ojr && es.resetting)) e = -1;
      my_slot = es.slot;
    }
    my_env = e;
  }
  unsigned long long todo = PERSIST ? __ballot(my_env >= 0) : (my_env >= 0 ? 1ull : 0ull);
  // (soft_sync: the step's join, see McrParams::await_tail)
  auto join_tail = [&]() { if (PERSIST && p.await_tail && blockIdx.x == 0 && threadIdx.x == 0) { (void)mcr_await(p, W_SIDE); (void)mcr_await(p, W_MAIN); } };
  if (!todo) { join_tail(); return; }
  auto slot_of = [&](int k) -> const uint8_t* {
    return p.slots + ((size_t)__builtin_amdgcn_readlane(my_env, k) * 2 + __builtin_amdgcn_readlane(my_slot, k)) * MCR_SLOT_BYTES;
  };
  int env, P, a_lo;
  const uint8_t* __restrict__ slot;
  { const int k = UNI(__builtin_ctzll(todo)); todo &= todo - 1; env = __builtin_amdgcn_readlane(my_env, k); slot = slot_of(k); a_lo = __builtin_amdgcn_readlane(my_agent, k); }
  // what a candidate needs from HBM, requested one round ahead
  struct Raw { float4 a, b, c, d; uint32_t m; };     // quad: a = v0 v1, b = v2 v3, m = meta | car polygon: a..d = 8 vertices, m = vertex count
  Raw nxt; nxt.a = nxt.b = nxt.c = nxt.d = make_float4(0.0f, 0.0f, 0.0f, 0.0f); nxt.m = 0u;
  P = (!PERSIST && my_P >= 0) ? my_P : ((const McrSlotHeader*)slot)->P;
  const int dbg = p.debug;
  unsigned long long pacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = PHASES ? __builtin_readcyclecounter() : 0ull;

  __shared__ __attribute__((aligned(16))) uint32_t keyb[96 * KS];           // draw key per pixel, GL rows (0 = bottom)
  __shared__ __attribute__((aligned(16))) float4 rdat[RC][3];               // per record: slopes, lower consts, upper consts of 4 edges
  __shared__ uint32_t rmeta[RC];
  __shared__ uint16_t tasks4[TASK4_CAP];                                    // record slot << 5 | group of 4 lines
  __shared__ uint32_t tfl[MCR_TILE_CAP / 2];                                // the env's tile flags (bit 8 / 24: recoloured)
  __shared__ uint32_t palc[32];
  __shared__ float vrec[2][MCR_VIEWP_FLOATS];                               // view record of the view being drawn / the next one
  __shared__ float glo[20], ghi[20];
  __shared__ uint8_t glyphs[80];
  __shared__ int wsum[4];
  __shared__ uint8_t vblk[2][NBLK];                                         // visible blocks of the view being drawn / the next one
  __shared__ int nvb[2];
  static_assert(MCR_TILE_CAP / 2 == VIEW_THREADS, "one tile-flag word per thread");
  // Wavefront 3 lists, per view, the road_poly blocks (runs of MCR_QBLK consecutive entries; boxes from the track
  // generator) the scene rectangle (obs x 0..96, y 12..96) can see: only their entries become candidates.  Separating-
  // axis test both ways — the box in pixel space against the rectangle, the rectangle in world space against the box —
  // with a pixel of slack.
  auto list_blocks = [&](const uint8_t* __restrict__ sl, int vw, int buf) {
    float4 bbox = make_float4(1.0f, 1.0f, -1.0f, -1.0f);
    if (lane < NBLK) bbox = ((const float4*)(sl + MCR_OFF_QBLK))[lane];
    const float* __restrict__ vp = p.viewp + (size_t)vw * MCR_VIEWP_FLOATS;
    const float c0 = vp[VP_CAM + 0], c1 = vp[VP_CAM + 1], c2 = vp[VP_CAM + 2], c3 = vp[VP_CAM + 3], c4 = vp[VP_CAM + 4], c5 = vp[VP_CAM + 5];
    const float v0 = vp[VP_INV + 0], v1 = vp[VP_INV + 1], v2 = vp[VP_INV + 2], v3 = vp[VP_INV + 3], v4 = vp[VP_INV + 4], v5 = vp[VP_INV + 5];
    float pxl = BIG, pxh = -BIG, pyl = BIG, pyh = -BIG, wxl = BIG, wxh = -BIG, wyl = BIG, wyh = -BIG;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float X = (k & 1) ? bbox.z : bbox.x, Y = (k & 2) ? bbox.w : bbox.y;
      const float px = __builtin_fmaf(c0, X, __builtin_fmaf(c1, Y, c4)), py = __builtin_fmaf(c2, X, __builtin_fmaf(c3, Y, c5));
      pxl = fminf(pxl, px); pxh = fmaxf(pxh, px); pyl = fminf(pyl, py); pyh = fmaxf(pyh, py);
      const float U = (k & 1) ? 96.0f : 0.0f, V = (k & 2) ? 96.0f : 12.0f;
      const float wx = v0 * U + v1 * V + v2, wy = v3 * U + v4 * V + v5;
      wxl = fminf(wxl, wx); wxh = fmaxf(wxh, wx); wyl = fminf(wyl, wy); wyh = fmaxf(wyh, wy);
    }
    const float mx = fabsf(v0) + fabsf(v1) + 0.05f, my = fabsf(v3) + fabsf(v4) + 0.05f;
    const bool vis = bbox.x <= bbox.z && pxh >= -1.0f && pxl <= 97.0f && pyh >= 11.0f && pyl <= 97.0f &&
                     bbox.z >= wxl - mx && bbox.x <= wxh + mx && bbox.w >= wyl - my && bbox.y <= wyh + my;
    const unsigned long long mask = __ballot(vis);
    if (vis) vblk[buf][__popcll(mask & ((1ull << lane) - 1ull))] = (uint8_t)lane;
    if (lane == 0) nvb[buf] = __popcll(mask);
  };

  // ---- once per workgroup: palette, glyphs, grass lattice; the first env's tile flags, view record, block list
  if (tid < 32) palc[tid] = PALETTE_RGB[tid];
  if (tid >= 128 && tid < 128 + 77) glyphs[tid - 128] = ((const uint8_t*)LABEL_GLYPHS)[tid - 128];
  tfl[tid] = ((const uint32_t*)(p.tile_flags + (size_t)env * MCR_TILE_CAP))[tid];
  if (tid >= 64 && tid < 64 + MCR_VIEWP_FLOATS) vrec[0][tid - 64] = p.viewp[(size_t)(env * N + a_lo) * MCR_VIEWP_FLOATS + (tid - 64)];
  if (wave == 3) list_blocks(slot, env * N + a_lo, 0);
  // grass lattice as the reference builds it (:620-627): f32(k*x) and f32(k*x + k) for x = -20, -18, .., 18
  if (tid >= 224 && tid < 244) { const double k = MCR_PLAYFIELD / 20.0, x = 2.0 * (double)(tid - 224 - 10); glo[tid - 224] = (float)(k * x + 0); ghi[tid - 224] = (float)(k * x + k); }
  const float kx = 96.0f / 1000.0f, ky = 96.0f / 800.0f;
  // candidate c of a view (env e, episode slot sl with pe road_poly entries) whose visible blocks are
  // vblk[buf][0 .. pv / QBLK): road_poly entry or special; `sb`: first candidate index of the view's specials (they sit
  // at the END of the last round, see below)
  auto quad_of = [&](int c, int buf) -> int { return (int)vblk[buf][c / MCR_QBLK] * MCR_QBLK + (c & (MCR_QBLK - 1)); };
  auto fetch_raw = [&](int c, int pv, int sb, int buf, const uint8_t* __restrict__ sl, int pe, int e) -> Raw {
    Raw r; r.a = r.b = r.c = r.d = make_float4(0.0f, 0.0f, 0.0f, 0.0f); r.m = 0u;
    if (c < pv) {
      const int q = quad_of(c, buf);
      if (q < pe) { r.a = ((const float4*)(sl + MCR_OFF_QA))[q]; r.b = ((const float4*)(sl + MCR_OFF_QB))[q]; r.m = ((const uint32_t*)(sl + MCR_OFF_QMETA))[q]; }
    } else if (c >= sb && c - sb < 14 * N) {
      const int cc = (c - sb) / 14, sl14 = (c - sb) - cc * 14;
      const int j = sl14 < 11 ? sl14 : sl14 - 1;                            // slot 11 sets up the second half of polygon 10 (the 8-gon): same data; slot 13 is a pad
      if (sl14 != 13) {
        const float* __restrict__ cp = p.carpoly + (size_t)(e * N + cc) * MCR_CARPOLY_FLOATS;
        const float4* cv = (const float4*)(cp + j * 16);
        r.a = cv[0]; r.b = cv[1]; r.c = cv[2]; r.d = cv[3];
        r.m = (uint32_t)__float_as_int(cp[MCR_CARPOLY_NOFF + j]);
      }
    }
    return r;
  };

  // slot layout of a view's specials (see the candidate phase): spread over the wavefronts when the view fits one round
  auto spread_layout = [&](int pv, int f, int g) -> bool { return N <= 4 && pv <= 128 - 14 * N && g <= 56 && f - 14 * N <= 64; };
  auto special_of = [&](int c, bool spread, int sb, int f, int g) -> int {   // candidate c -> index among the view's specials, or -1
    if (!spread) return (c >= sb && c - sb < f + g) ? c - sb : -1;
    const int nh = f - 14 * N;                                              // gauges, flag [, playfield]
    if (c >= sb && c < 128) return c - sb;
    if (c >= 192 - nh && c < 192) return 14 * N + c - (192 - nh);
    if (c >= 248 - g && c < 248) return f + c - (248 - g);
    return -1;
  };
  int vs = 0;                                                               // views this workgroup has drawn
#pragma nounroll
  for (;;) {
  // the workgroup's next env, if any
  const bool has_next = PERSIST && todo != 0ull;
  int env_n = env, a_lo_n = 0; const uint8_t* __restrict__ slot_n = slot;
  if (has_next) { const int k = UNI(__builtin_ctzll(todo)); todo &= todo - 1; env_n = __builtin_amdgcn_readlane(my_env, k); slot_n = slot_of(k); a_lo_n = __builtin_amdgcn_readlane(my_agent, k); }
  const int a_hi = split_views ? a_lo + 1 : N;
  uint32_t tfl_n = 0u; int P_nv = 0;
#pragma nounroll
  for (int agent = a_lo; agent < a_hi; ++agent, ++vs) {
    const int vw = env * N + agent;
    const int buf = vs & 1;
    const float* __restrict__ vr = vrec[buf];
    // the view after this one: the env's next agent, or agent 0 of the workgroup's next env
    const bool last = agent + 1 == a_hi;
    const bool nv_ok = !last || has_next;
    const int env_v = last ? env_n : env, vw_v = last ? env_n * N + a_lo_n : vw + 1;
    const uint8_t* __restrict__ slot_v = last ? slot_n : slot;
    PHASE_ACC(0);
    __syncthreads();                                                        // this view's record and block list are in LDS; the previous view's resolve is through with the key buffer
    PHASE_ACC(1);
    if (!PERSIST && vs == 0 && tid == 0 && my_P >= 0) p.vorder[blockIdx.x] = -1;   // every wavefront has read the entry: free it for the step after next
    // the next view's record, block list and (new env) tile flags / entry count travel while this one is drawn
    if (nv_ok && tid >= 64 && tid < 64 + MCR_VIEWP_FLOATS) vrec[buf ^ 1][tid - 64] = p.viewp[(size_t)vw_v * MCR_VIEWP_FLOATS + (tid - 64)];
    if (nv_ok && wave == 3) list_blocks(slot_v, vw_v, buf ^ 1);
    if (last && has_next) { tfl_n = ((const uint32_t*)(p.tile_flags + (size_t)env_n * MCR_TILE_CAP))[tid]; P_nv = ((const McrSlotHeader*)slot_n)->P; }
    const float m00 = vr[VP_CAM + 0], m01 = vr[VP_CAM + 1], m10 = vr[VP_CAM + 2], m11 = vr[VP_CAM + 3], ctx = vr[VP_CAM + 4], cty = vr[VP_CAM + 5];
    const uint32_t old_flags = __float_as_uint(vr[VP_OLDFLAGS]);
    // grass squares the viewport can see / "is the whole viewport inside the playfield" (k_dynamics, from the inverse camera)
    const int mu0 = UNI(__float_as_int(vr[VP_GRASS + 0])), nu = UNI(__float_as_int(vr[VP_GRASS + 1])), mv0 = UNI(__float_as_int(vr[VP_GRASS + 2])), nv = UNI(__float_as_int(vr[VP_GRASS + 3]));
    const bool inside_field = UNI(__float_as_int(vr[VP_GRASS + 4])) != 0;
    // key-buffer clear: HUD bar (window y < 100 = obs rows 0..11, :655-656) and the scene's base colour
    {
      const uint32_t kbar = 0u, kbase = inside_field ? ((uint32_t)RANK_PLAYFIELD << 24) | palc[PAL_GRASS0] : 0u;   // black bar / black beyond the playfield
      uint4* k4 = (uint4*)keyb;
      for (int i = tid; i < 96 * KS / 4; i += VIEW_THREADS) {
        const uint32_t k = i < 12 * KS / 4 ? kbar : kbase;                  // 12 * 97 = 1164 words = 291 uint4: the split is aligned
        k4[i] = make_uint4(k, k, k, k);
      }
    }
    // Candidates of a view: [0, P) the road_poly quads and, right-aligned to the end of the last round (so that they
    // land on the wavefront the quads leave idle), the specials: 14 slots per car (its 12 Car.draw polygons, world
    // vertices from k_dynamics; the 8-gon hull polygon takes slots 10+11), 7 gauges, the flag, the playfield quad (only
    // when the view leaves it), the G light grass squares the viewport can see.
    const int Pv = UNI(nvb[buf]) * MCR_QBLK;                                // road_poly candidates: the entries of the visible blocks
    const int G = nu * nv;
    const int F = 14 * N + 8 + (inside_field ? 0 : 1);
    const int nspec = (F + G + 1) & ~1;                                     // even: the 8-gon's slot pair never straddles two rounds
    const int nround = (Pv + nspec + RC - 1) / RC;
    // A view that fits one round with room to spare (the rule once the camera has zoomed in) SPREADS its specials over
    // the wavefronts instead: cars end at slot 128 (wavefront 1), gauges / flag / playfield at 192 (wavefront 2), grass at
    // 248 (wavefront 3) — every wavefront then runs one kind of candidate, not all of them one after the other.
    const bool spread = spread_layout(Pv, F, G);
    const int SB = spread ? 128 - 14 * N : nround * RC - nspec;             // first slot of the cars (= of the specials when they are contiguous)
    if (vs == 0 && tid < RC) nxt = fetch_raw(tid, Pv, SB, buf, slot, P, env);   // later views: requested while the previous one was drawn
#pragma nounroll
    for (int rd = 0; rd < nround; ++rd) {
      const int c = rd * RC + tid;
      const Raw cur = nxt;
      if (rd > 0) {                                                         // what the earlier rounds drew keeps its colour, below everything to come (but above the grass)
        uint4* k4 = (uint4*)keyb;
        for (int i = tid; i < 96 * KS / 4; i += VIEW_THREADS) {
          uint4 k = k4[i];
          k.x = k.x > (((uint32_t)RANK_FLAT << 24) | 0xffffffu) ? (k.x & 0xffffffu) | ((uint32_t)RANK_FLAT << 24) : k.x;
          k.y = k.y > (((uint32_t)RANK_FLAT << 24) | 0xffffffu) ? (k.y & 0xffffffu) | ((uint32_t)RANK_FLAT << 24) : k.y;
          k.z = k.z > (((uint32_t)RANK_FLAT << 24) | 0xffffffu) ? (k.z & 0xffffffu) | ((uint32_t)RANK_FLAT << 24) : k.z;
          k.w = k.w > (((uint32_t)RANK_FLAT << 24) | 0xffffffu) ? (k.w & 0xffffffu) | ((uint32_t)RANK_FLAT << 24) : k.w;
          k4[i] = k;
        }
      }
      const bool mine = tid < RC;
      uint32_t my_meta = 0u;                                                // a culled slot has no lines
      // Every candidate, whatever its kind, is "4 (or 8) vertices in world or pixel space + a key": a short kind-specific head
      // produces them, ONE common tail transforms, culls and sets the span record up — a wavefront that holds several kinds
      // of candidates (road quads and cars; gauges; grass) runs the tail once, not once per kind.
      float wx[8], wy[8];
      uint32_t key = 0u; int cminY = 12, nn = 0;                            // nn: 0 nothing; 4 / 8 vertices in world space; -4: 4 vertices in pixel space
      bool half2 = false;                                                   // this slot sets up the SECOND half (edges 4..7) of an 8-gon
      const int q = (mine && c < Pv) ? quad_of(c, buf) : P;
      if (q < P) {
        // ---- road_poly entry
        const uint32_t meta = cur.m;
        const uint32_t tile1 = (meta >> 8) & 0x3ffu;
        if (!(dbg & 18)) {
          wx[0] = cur.a.x; wy[0] = cur.a.y; wx[1] = cur.a.z; wy[1] = cur.a.w; wx[2] = cur.b.x; wy[2] = cur.b.y; wx[3] = cur.b.z; wy[3] = cur.b.w;
          uint32_t col = meta & 0xffu;
          if (tile1 && ((tfl[(tile1 - 1) >> 1] >> (((tile1 - 1) & 1u) * 16u)) & 0x100u)) col = MCR_COL_ROAD0;     // touched tile -> ROAD_COLOR (:102-104)
          const uint32_t pal = col == MCR_COL_ROAD0 ? PAL_ROAD0 : col == MCR_COL_ROAD1 ? PAL_ROAD1 : col == MCR_COL_ROAD2 ? PAL_ROAD2 : col == MCR_COL_KERB_WHITE ? PAL_WHITE : PAL_RED255;
          key = mk_key(RANK_SLOT0 + tid, pal); nn = 4;
        }
      } else if (const int sidx = mine ? special_of(c, spread, SB, F, G) : -1; sidx >= 0) {
        // ---- specials
        if (sidx < 14 * N) {                                                // Car.draw polygon
          const int cc = sidx / 14, sl = sidx - cc * 14;
          const int j = sl < 11 ? sl : sl - 1;
          const int n = (int)cur.m;
          if (sl != 13 && n > 0 && !(dbg & 4) && (sl != 11 || n > 4)) {
            wx[0] = cur.a.x; wy[0] = cur.a.y; wx[1] = cur.a.z; wy[1] = cur.a.w; wx[2] = cur.b.x; wy[2] = cur.b.y; wx[3] = cur.b.z; wy[3] = cur.b.w;
            wx[4] = cur.c.x; wy[4] = cur.c.y; wx[5] = cur.c.z; wy[5] = cur.c.w; wx[6] = cur.d.x; wy[6] = cur.d.y; wx[7] = cur.d.z; wy[7] = cur.d.w;
            uint32_t colr;
            if (j < 8) colr = (j & 1) ? PAL_WHEELWHITE : PAL_BLACK;
            else { colr = PAL_CAR0 + (cc & 7); if (p.use_ego_color) colr = (cc == agent) ? PAL_CAR0 + 0 : PAL_CAR0 + 1; }   // :402, :560-563
            // only HULL_POLY3 (slots 10 + 11) has more than 4 vertices (mcr_create checks); k_dynamics pads to 8.  Both of its
            // slots carry the key of the FIRST one: the fill reads the second record through the first one's meta word
            half2 = sl == 11;
            key = mk_key(RANK_SLOT0 + tid - (half2 ? 1 : 0), colr);
            nn = (n > 4 && sl >= 10) ? 8 : 4;
          }
        } else if (sidx < F) {
          const int h = sidx - 14 * N;
          if (h < 7) {                                                      // gauges (:643-663), already in pixel units
            const float gx0 = vr[VP_IND + h * 4], gx1 = vr[VP_IND + h * 4 + 1], gy0 = vr[VP_IND + h * 4 + 2], gy1 = vr[VP_IND + h * 4 + 3];
            if (gx1 > gx0 && gy1 > gy0) {
              wx[0] = gx0; wy[0] = gy0; wx[1] = gx1; wy[1] = gy0; wx[2] = gx1; wy[2] = gy1; wx[3] = gx0; wy[3] = gy1;
              const uint32_t col = h == 0 ? PAL_WHITE : h <= 2 ? PAL_BLUE255 : h <= 4 ? PAL_PURPLE : h == 5 ? PAL_GREEN255 : PAL_RED255;
              key = mk_key(RANK_SLOT0 + tid, col) | REC_HUD; nn = -4; cminY = 0;
            }
          } else if (h == 7) {                                              // backwards flag (:669-674): drawn with last step's flag
            if ((old_flags & 1u) && p.backwards_flag) {
              wx[0] = 900.0f * kx; wy[0] = 30.0f * ky; wx[1] = 925.0f * kx; wy[1] = 70.0f * ky; wx[2] = 950.0f * kx; wy[2] = 30.0f * ky; wx[3] = wx[2]; wy[3] = wy[2];
              key = mk_key(RANK_SLOT0 + tid, PAL_BLUE255) | REC_HUD; nn = -4; cminY = 0;
            }
          } else {                                                          // playfield quad (:615-619), only when the view leaves it
            const float PF = (float)MCR_PLAYFIELD;
            wx[0] = -PF; wy[0] = PF; wx[1] = PF; wy[1] = PF; wx[2] = PF; wy[2] = -PF; wx[3] = -PF; wy[3] = -PF;
            key = mk_key(RANK_PLAYFIELD, PAL_GRASS0); nn = 4;
          }
        } else {                                                            // light grass square (:620-627)
          const int g = sidx - F, iv = g / nu, iu = g - iv * nu;
          const int tu = mu0 + iu + 10, tv = mv0 + iv + 10;
          wx[0] = ghi[tu]; wy[0] = glo[tv]; wx[1] = glo[tu]; wy[1] = glo[tv]; wx[2] = glo[tu]; wy[2] = ghi[tv]; wx[3] = ghi[tu]; wy[3] = ghi[tv];
          key = mk_key(RANK_GRASS, PAL_GRASS1); nn = 4;
        }
      }
      // ---- common tail
      if (nn != 0) {
        const bool eight = nn == 8;
        float px[8], py[8];
        if (nn > 0) {                                                       // camera transform (world -> pixel)
#pragma unroll
          for (int i = 0; i < 4; ++i) { px[i] = __builtin_fmaf(m00, wx[i], __builtin_fmaf(m01, wy[i], ctx)); py[i] = __builtin_fmaf(m10, wx[i], __builtin_fmaf(m11, wy[i], cty)); }
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) { px[i] = wx[i]; py[i] = wy[i]; }
        }
        float x0 = fminf(fminf(px[0], px[1]), fminf(px[2], px[3])), x1 = fmaxf(fmaxf(px[0], px[1]), fmaxf(px[2], px[3]));
        float y0 = fminf(fminf(py[0], py[1]), fminf(py[2], py[3])), y1 = fmaxf(fmaxf(py[0], py[1]), fmaxf(py[2], py[3]));
        float area = (px[0] * py[1] - px[1] * py[0]) + (px[1] * py[2] - px[2] * py[1]) + (px[2] * py[3] - px[3] * py[2]);
        float x5[5], y5[5];
#pragma unroll
        for (int i = 0; i < 4; ++i) { x5[i] = px[i]; y5[i] = py[i]; }
        x5[4] = px[0]; y5[4] = py[0];
        if (eight) {
#pragma unroll
          for (int i = 4; i < 8; ++i) { px[i] = __builtin_fmaf(m00, wx[i], __builtin_fmaf(m01, wy[i], ctx)); py[i] = __builtin_fmaf(m10, wx[i], __builtin_fmaf(m11, wy[i], cty)); }
          x0 = fminf(x0, fminf(fminf(px[4], px[5]), fminf(px[6], px[7]))); x1 = fmaxf(x1, fmaxf(fmaxf(px[4], px[5]), fmaxf(px[6], px[7])));
          y0 = fminf(y0, fminf(fminf(py[4], py[5]), fminf(py[6], py[7]))); y1 = fmaxf(y1, fmaxf(fmaxf(py[4], py[5]), fmaxf(py[6], py[7])));
          area = (area + (px[3] * py[4] - px[4] * py[3])) + ((px[4] * py[5] - px[5] * py[4]) + (px[5] * py[6] - px[6] * py[5]) + (px[6] * py[7] - px[7] * py[6]) + (px[7] * py[0] - px[0] * py[7]));
          x5[4] = px[4]; y5[4] = py[4];
          if (half2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { x5[i] = px[4 + i]; y5[i] = py[4 + i]; }
            x5[4] = px[0]; y5[4] = py[0];
          }
        } else area += px[3] * py[0] - px[0] * py[3];
        int i0, i1, j0, j1;
        if (centre_range(x0, x1, 0, 95, i0, i1) && centre_range(y0, y1, cminY, 95, j0, j1)) {          // rows < 12: HUD bar
          const uint32_t geo = setup_half(x5, y5, area, i0, i1, j0, j1, rdat, tid);
          if (geo && !half2) my_meta = key | geo | (eight ? REC_CHAIN : 0u);
        }
      }
      if (tid < RC) rmeta[tid] = my_meta;
      // line counts in slot order -> block-wide exclusive prefix sum of the 4-line task groups
      const int nl = (int)(my_meta >> 25);
      const int g4 = (nl + 3) >> 2;
      int incl = g4;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
      if (lane == 63) wsum[wave] = incl;
      PHASE_ACC(2);
      __syncthreads();                                                      // records and line counts of the round are in LDS
      PHASE_ACC(3);
      // next round's candidates (or round 0 of the next agent's view: the same quads): their HBM / L2 data travels
      // while this round's spans are drawn
      if (tid < RC) {
        if (rd + 1 < nround) nxt = fetch_raw(c + RC, Pv, SB, buf, slot, P, env);
        else if (nv_ok) {                                                   // round 0 of the next view (its record and block list are in LDS)
          const float* __restrict__ vn = vrec[buf ^ 1];
          const int pvn = UNI(nvb[buf ^ 1]) * MCR_QBLK;
          const int gn = UNI(__float_as_int(vn[VP_GRASS + 1])) * UNI(__float_as_int(vn[VP_GRASS + 3]));
          const int fn = 14 * N + 8 + (UNI(__float_as_int(vn[VP_GRASS + 4])) ? 0 : 1);
          const int nsn = (fn + gn + 1) & ~1;
          const int sbn = spread_layout(pvn, fn, gn) ? 128 - 14 * N : (pvn + nsn + RC - 1) / RC * RC - nsn;
          nxt = fetch_raw(tid, pvn, sbn, buf ^ 1, slot_v, last ? UNI(P_nv) : P, env_v);
        }
      }
      // score label (:665-666): white glyph cells stamped with the top key; 16 x 4 pixel centres cover its window box
      if (rd + 1 == nround && wave == 2) {
        const int lx = 1 + (lane & 15), ly = 4 + (lane >> 4);
        const int value = UNI(__float_as_int(vr[VP_SCORE]));
        if (label_on(value, ((float)lx + 0.5f) * (1000.0f / 96.0f), ((float)ly + 0.5f) * (800.0f / 96.0f), glyphs))
          atomicMax(&keyb[ly * KS + lx], ((uint32_t)RANK_LABEL << 24) | 0xffffffu);
      }
      int off = incl - g4;
      for (int w = 0; w < 4; ++w) if (w < wave) off += wsum[w];
      const int total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
      int t0 = 0;
      do {
        for (int k4 = 0; k4 < g4; ++k4) {
          const int idx = off + k4 - t0;
          if ((unsigned)idx < (unsigned)TASK4_CAP) tasks4[idx] = (uint16_t)((tid << 5) | k4);
        }
        PHASE_ACC(4);
        __syncthreads();
        PHASE_ACC(5);
        const int nlines = min(TASK4_CAP, total - t0) * 4;
        // Two line tasks per lane and trip: the three dependent LDS reads of a task (task entry -> record meta -> record) are
        // issued for both before either is used — the phase is bound by that latency chain, not by the arithmetic.  Lanes
        // without a task read entry 0 (harmless) and end up with an empty span.
        for (int i0 = 0; i0 < nlines; i0 += 2 * VIEW_THREADS) {
          const int iA = i0 + tid, iB = iA + VIEW_THREADS;
          const bool inA = iA < nlines, inB = iB < nlines;
          const uint32_t eA = tasks4[inA ? iA >> 2 : 0], eB = tasks4[inB ? iB >> 2 : 0];
          const int slA = min((int)(eA >> 5), RC - 1), slB = min((int)(eB >> 5), RC - 1);
          const int kA = (int)((eA & 31u) << 2) | (iA & 3), kB = (int)((eB & 31u) << 2) | (iB & 3);
          const uint32_t mA = rmeta[slA], mB = rmeta[slB];
          const float4 SA = rdat[slA][0], LA = rdat[slA][1], HA = rdat[slA][2];
          const float4 SB4 = rdat[slB][0], LB = rdat[slB][1], HB = rdat[slB][2];
          const uint32_t cA = palc[mA & 31u], cB = palc[mB & 31u];
          auto span = [&](bool in, uint32_t meta, int sl, int k, const float4 Sl, const float4 Lo, const float4 Hi, uint32_t col,
                          int& len, int& addr, int& stride, uint32_t& key) {
            len = 0; addr = 0; stride = 1; key = 0u;
            if (in && k < (int)(meta >> 25)) {
              const int line = (int)((meta >> 18) & 127u) + k;
              const float v = (float)line + 0.5f;
              float lo = fmaxf(fmaxf(__builtin_fmaf(Sl.x, v, Lo.x), __builtin_fmaf(Sl.y, v, Lo.y)), fmaxf(__builtin_fmaf(Sl.z, v, Lo.z), __builtin_fmaf(Sl.w, v, Lo.w)));
              float hi = fminf(fminf(__builtin_fmaf(Sl.x, v, Hi.x), __builtin_fmaf(Sl.y, v, Hi.y)), fminf(__builtin_fmaf(Sl.z, v, Hi.z), __builtin_fmaf(Sl.w, v, Hi.w)));
              if (meta & REC_CHAIN) {
                const float4 S2 = rdat[sl + 1][0], L2 = rdat[sl + 1][1], H2 = rdat[sl + 1][2];
                lo = fmaxf(lo, fmaxf(fmaxf(__builtin_fmaf(S2.x, v, L2.x), __builtin_fmaf(S2.y, v, L2.y)), fmaxf(__builtin_fmaf(S2.z, v, L2.z), __builtin_fmaf(S2.w, v, L2.w))));
                hi = fminf(hi, fminf(fminf(__builtin_fmaf(S2.x, v, H2.x), __builtin_fmaf(S2.y, v, H2.y)), fminf(__builtin_fmaf(S2.z, v, H2.z), __builtin_fmaf(S2.w, v, H2.w))));
              }
              key = ((meta << 19) & 0xff000000u) | col;
              const bool row = (meta & REC_ROW) != 0u;
              // rows < 12 belong to the HUD: scene polygons stop at y = 12 (row scans are clipped by their line range)
              const float cmin = (meta & (REC_ROW | REC_HUD)) ? 0.0f : 12.0f;
              lo = fmaxf(lo, cmin); hi = fminf(hi, 96.0f);
              const int a = (int)ceilf(lo - 0.5f), b = (int)floorf(hi - 0.5f);    // pixel centres a+.5 .. b+.5 lie in [lo, hi]
              len = b - a + 1;
              addr = (row ? line : a) * KS + (row ? a : line);
              stride = row ? 1 : KS;
            }
          };
          int lenA, addrA, strideA, lenB, addrB, strideB; uint32_t keyA, keyB;
          span(inA, mA, slA, kA, SA, LA, HA, cA, lenA, addrA, strideA, keyA);
          span(inB, mB, slB, kB, SB4, LB, HB, cB, lenB, addrB, strideB, keyB);
          // a lane leaves a loop when its span is drawn (the wavefront iterates to its longest span)
          // (two pixels per trip: the loop's bookkeeping — 3 scalar + 2 vector instructions — is what the phase issues most of;
          // the second ds_max of an odd span's last trip carries key 0, which changes nothing wherever it lands)
          if (lenA > 0) { int j = 0; do { atomicMax(&keyb[addrA], keyA); atomicMax(&keyb[addrA + strideA], j + 1 < lenA ? keyA : 0u); addrA += 2 * strideA; j += 2; } while (j < lenA); }
          if (lenB > 0) { int j = 0; do { atomicMax(&keyb[addrB], keyB); atomicMax(&keyb[addrB + strideB], j + 1 < lenB ? keyB : 0u); addrB += 2 * strideB; j += 2; } while (j < lenB); }
        }
        PHASE_ACC(6);
        __syncthreads();                                                    // the next chunk / round rewrites tasks and records
        PHASE_ACC(7);
        t0 += TASK4_CAP;
      } while (t0 < total);
    }
    // ---- resolve + packed RGB write-out: 4 pixels -> 12 bytes per lane, rows top-down (arr[::-1], :602)
    if (!(dbg & 8)) {
      uint32_t* __restrict__ out = (uint32_t*)(p.obs + (size_t)vw * (96 * 96 * 3));
      // 9 groups of 4 pixels per thread: the winners' RGB bytes are packed with three byte permutes
#pragma unroll 3
      for (int g0 = 0; g0 < 9; ++g0) {
        const int g = g0 * VIEW_THREADS + tid, r = g / 24, x4 = (g - r * 24) * 4;
        const uint32_t* kp = &keyb[(95 - r) * KS + x4];
        const uint32_t c0 = kp[0], c1 = kp[1], c2 = kp[2], c3 = kp[3];
        uint3 w;
        w.x = __builtin_amdgcn_perm(c1, c0, 0x04020100u); w.y = __builtin_amdgcn_perm(c2, c1, 0x05040201u); w.z = __builtin_amdgcn_perm(c3, c2, 0x06050402u);
        // streaming store: the frames of a step (226 MB at B = 4096, N = 2) are written once and read by nobody on the
        // device — they must not evict the state the step's latency-bound chains live on from L2 / MALL
        typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
        u32x3 wv; wv.x = w.x; wv.y = w.y; wv.z = w.z;
        __builtin_nontemporal_store(wv, (u32x3*)(out + (size_t)g * 3));
      }
    }
    PHASE_ACC(8);
    if (PHASES && tid == 0 && stamps) {
      for (int i = 0; i < 9; ++i) { stamps[(size_t)vw * 16 + i] = pacc[i]; pacc[i] = 0; }
      stamps[(size_t)vw * 16 + 9] = (unsigned long long)nround;
      // placement: wall clock (100 MHz) at the end of the view, workgroup | view number, HW_ID | XCC_ID
      stamps[(size_t)vw * 16 + 10] = __builtin_amdgcn_s_memrealtime();
      stamps[(size_t)vw * 16 + 11] = (unsigned long long)blockIdx.x | ((unsigned long long)vs << 32);
      stamps[(size_t)vw * 16 + 12] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32);
    }
  }
  if (!has_next) break;
  // hand-over: every thread is past the last barrier of the view's span fill, nobody reads the tile flags any more
  env = env_n; slot = slot_n; P = UNI(P_nv); tfl[tid] = tfl_n; a_lo = a_lo_n;
  }
  join_tail();
}
