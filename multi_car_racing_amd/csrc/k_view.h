// k_view.h — one 256-thread workgroup per env, drawing its N agent views one after the other: the 96x96 ego-frame
// software rasteriser that replaces pyglet/OpenGL (multi_car_racing.py:511-604, 613-674; gym Car.draw).  What the views
// of an env share — episode slot header, tile flags, the road quads in L1/L2 — is fetched once per workgroup.
// (The backward/on-grass bookkeeping of :446-495 is k_flags.h.)
//
// A SPAN rasteriser (round 2), laid out in round 4 for FOUR workgroups per CU (<= 40,960 B of LDS, <= 128 VGPRs):
//   1. candidates, in rounds of 168 slots (one thread each; one round is the rule): the road_poly quads of the blocks the
//      viewport can see (read once, coalesced) and the view's "special" polygons — the Car.draw polygons of the cars it can
//      see, the light grass squares, the playfield quad when the view leaves it, the part of a gauge that is taller than the
//      HUD bar.  A thread transforms its polygon, culls it exactly (does its bbox hold a pixel centre?) and, if it
//      survives, builds a 32-byte record in its own slot: the polygon is scanned along its LONGER bounding-box axis (rows
//      or columns), so a scan line crosses it in a short span; per edge the span bound is affine in the line coordinate,
//      bound(v) = s*v + c, and a bit of the record's meta word says whether the edge bounds the span from below or from
//      above — a span end is a max / min of four FMAs, no per-pixel edge tests.  No compaction, no atomics: a culled slot
//      has zero lines.  The next round's (or view's) HBM data is requested before this round's spans are drawn;
//   2. tasks: a prefix sum over the slots' line counts lays out (record, group of 4 lines) tasks, 64 lines per wavefront trip
//      whatever the polygon sizes are;
//   3. fill: each lane computes its line's span [lo, hi] and writes rank << 24 | RGB over it with LDS ds_max_u32 — "highest
//      rank wins" is the painter's order of the reference (playfield, grass squares, road_poly in creation order, cars,
//      gauges) and the winner's colour needs no lookup;
//   4. resolve: 4 key words -> 3 byte permutes -> 12 bytes per lane as one streaming dwordx3 store, the vertical flip of
//      :602 folded into the address.
// The key buffer holds the 84 SCENE rows only (32,592 B).  The 12 rows below them are the HUD (window y < 100: the black
// quad of :638-642 overdraws whatever the scene put there): the fourth wavefront, which has no candidate slots, composes
// them analytically — bar, the 7 gauges in draw order, score label (:665-666), backwards flag — and stores them straight
// into the frame (hud_prep while the other three set their candidates up, hud_store while they lay out the span tasks).
// Sampling rule: pixel centres; a centre belongs to a convex polygon iff it lies inside every edge (closed).  Span
// ends are computed by division instead of evaluating the edge function at the centre; the two agree except for
// centres within ~1e-5 px of an edge — inside the 0.02 px band in which the oracle declares a pixel ambiguous.
#pragma once
#include "k_raster_common.h"

namespace view {

constexpr int KS = 97;                 // key-buffer row stride in words: odd, so that a column of pixels walks all LDS banks
constexpr int ROWS = 84;               // scene rows of the key buffer: GL rows 12 .. 95 (row 0 of the buffer = GL row 12)
constexpr int HUD_ROWS = 12;           // GL rows 0 .. 11 = array rows 84 .. 95: hud_prep() / hud_store()
constexpr int RC = 168;                // candidate slots (= record slots) per round: wavefronts 0, 1 and the first 40 lanes of wavefront 2
constexpr int TASK4_CAP = 320;         // task entries per chunk; an entry covers 4 consecutive scan lines of one record
constexpr int KEYW4 = ROWS * KS / 4;   // the key buffer in 16-byte units
static_assert(ROWS * KS % 4 == 0, "the key buffer is cleared in 16-byte units");
constexpr float BIG = 1e30f;
constexpr int NBLK = MCR_QUAD_CAP / MCR_QBLK;   // road_poly culling blocks per track
constexpr float CAR_RADIUS = 4.5f;     // world units around the middle of the hull's 8-gon that hold every Car.draw polygon: the farthest
                                       // hull corner is 3.46 away, the farthest wheel corner 3.07 (+ a unit for a joint stretched by a crash)
// A pixel of the key buffer is rank << 24 | RGB: "highest rank wins" (ds_max_u32) is the painter's order of the reference
// and the winner's colour needs no lookup.  Ranks: the playfield quad, the light grass squares, then the candidates in
// slot order (road_poly entries in creation order, cars, tall gauges: RANK_SLOT0 + slot).
// A view that needs several rounds of candidates flattens what is drawn so far to RANK_FLAT before each further round.
enum { RANK_BASE = 0, RANK_PLAYFIELD = 1, RANK_GRASS = 2, RANK_FLAT = 3, RANK_SLOT0 = 4 };
// record meta word: edge i bounds the span from below [i] (else from above), i < 4 | row-scan [4] | chained second half in the
// next slot [5] | first line [6,13) | lines [13,20)
#define REC_ROW (1u << 4)
#define REC_CHAIN (1u << 5)
#define REC_L0_SHIFT 6
#define REC_NL_SHIFT 13

// One edge (ax,ay)->(bx,by) of a polygon with orientation sign sg: inside <=> A x + B y + C >= 0.  With u the
// coordinate along the span and v the line coordinate the edge bounds u from below (Au > 0) or above (Au < 0) by
// s v + c; an edge parallel to the spans (Au == 0) bounds nothing (the polygon's line range already accounts for it):
// it is stored as the lower bound -BIG.  Returns 1 for a lower bound.
__device__ __forceinline__ uint32_t span_edge(float ax, float ay, float bx, float by, float sg, bool row, float& s, float& c) {
  const float ex = bx - ax, ey = by - ay;
  const float A = -sg * ey, B = sg * ex, C = -(A * ax + B * ay);
  const float Au = row ? A : B, Av = row ? B : A;
  const float r = __builtin_amdgcn_rcpf(Au);
  s = Au != 0.0f ? -Av * r : 0.0f;
  c = Au != 0.0f ? -C * r : -BIG;
  return Au < 0.0f ? 0u : 1u;
}
// Span record of four consecutive edges v0->v1->v2->v3->(cx,cy) of a convex polygon in key-buffer space that survived the cull
// with pixel-centre ranges [i0,i1] x [j0,j1] (a quad: the closing vertex is v0; an 8-gon takes two records in adjacent slots, the
// halves 0..4 and 4..7,0, written by two threads that share the polygon's orientation, scan axis and line range).  `area`: the
// polygon's signed area x 2.  Returns the meta word without the chain bit (0: degenerate, nothing to draw).
__device__ __forceinline__ uint32_t setup_half(const float* px, const float* py, float cx, float cy, float area, int i0, int i1, int j0, int j1,
                                               float4 (*rdat)[2], int slot) {
  uint32_t meta = 0u;
  if (area != 0.0f) {
    const float sg = area > 0.0f ? 1.0f : -1.0f;
    const bool row = (j1 - j0) >= (i1 - i0);                       // scan along the longer axis: many short spans
    const int l0 = row ? j0 : i0, nl = (row ? j1 : i1) - l0 + 1;
    float4 S4, C4;
    const uint32_t k0 = span_edge(px[0], py[0], px[1], py[1], sg, row, S4.x, C4.x), k1 = span_edge(px[1], py[1], px[2], py[2], sg, row, S4.y, C4.y);
    const uint32_t k2 = span_edge(px[2], py[2], px[3], py[3], sg, row, S4.z, C4.z), k3 = span_edge(px[3], py[3], cx, cy, sg, row, S4.w, C4.w);
    rdat[slot][0] = S4; rdat[slot][1] = C4;
    meta = k0 | (k1 << 1) | (k2 << 2) | (k3 << 3) | (row ? REC_ROW : 0u) | ((uint32_t)l0 << REC_L0_SHIFT) | ((uint32_t)nl << REC_NL_SHIFT);
  }
  return meta;
}
// the other lane of an even / odd lane pair (the two halves of an 8-gon sit in such a pair)
__device__ __forceinline__ float pair_swap(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true)); }

// The backwards flag (:669-674): the triangle (900,30) (925,70) (950,30) in window units = (86.4,3.6) (88.8,8.4) (91.2,3.6) px covers the
// pixel centres x 87..90 of GL row 4, 87..89 of row 5, 88..89 of row 6 and 88 of row 7; no centre lies within 0.13 px of an edge.
// As byte masks of the 4-pixel groups 21 (x 84..87) and 22 (x 88..91) for GL row 4 + r4 (tests/test_oracle_pinning.py derives them from the vertices):
__device__ __forceinline__ uint32_t hud_flag_mask(int cg, int r4) {
  const uint32_t g21 = r4 < 2 ? 0xff000000u : 0u, g22 = r4 == 0 ? 0x00ffffffu : r4 <= 2 ? 0x0000ffffu : 0x000000ffu;
  return cg == 21 ? g21 : cg == 22 ? g22 : 0u;
}

// The 12 HUD rows of a view (render_indicators, :634-674), composed by ONE wavefront and stored straight into the frame (array rows
// 84..95): black bar, the 7 gauge rectangles of the view record in draw order (a later one overwrites an earlier one), the score label,
// the backwards flag.  Lanes 0..47: the 4-pixel group lane % 24 of GL rows 2k + lane / 24, k = 0..5.  A pixel is a colour CLASS in a byte
// (0 black, 1 white, 2 blue, 3 purple, 4 green, 5 red) until three byte permutes turn the four classes of a lane into R, G and B planes.
// What does not depend on the lane's pixels is computed once, by a lane or by the scalar unit: a gauge's pixel-centre ranges by lane i
// (then broadcast), the label's characters as a uniform BCD word.
// Two parts: hud_prep (no memory traffic) and hud_store (six streaming stores per lane); the caller consumes its own outstanding loads
// between them.  On gfx9 stores count in vmcnt like loads: a wait for a load issued BEHIND them would wait for the stores' acknowledgements
// from a memory system the raster keeps saturated.
struct HudState { uint32_t xm[7], rm[7], rows_any, lvalid; unsigned long long lmask; int cg, rsel; };
__device__ __forceinline__ HudState hud_prep(const float* __restrict__ vr, const uint8_t* __restrict__ glyphs, const int lane, const bool flag_on) {
  HudState H;
  // ---- gauges (vertical_ind x 5, horiz_ind x 2, :643-663), lane i < 7: columns [a, b1) and rows [ra, rb1) whose pixel centres the rectangle
  // holds (closed), clipped to the HUD rows, as a | b1 << 8 | row mask << 16 (0 when it holds none)
  uint32_t gw;
  {
    const int i = min(lane, 6);
    const float x0 = vr[VP_IND + i * 4], x1 = vr[VP_IND + i * 4 + 1], y0 = vr[VP_IND + i * 4 + 2], y1 = vr[VP_IND + i * 4 + 3];
    const int a = max((int)ceilf(x0 - 0.5f), 0), b1 = min((int)floorf(x1 - 0.5f), 95) + 1;
    const int ra = max((int)ceilf(y0 - 0.5f), 0), rb1 = min((int)floorf(y1 - 0.5f), HUD_ROWS - 1) + 1;
    gw = (b1 > a && rb1 > ra) ? (uint32_t)a | ((uint32_t)b1 << 8) | (((1u << rb1) - (1u << ra)) << 16) : 0u;
  }
  // ---- score label (:665-666), "%04i" % value: the characters as a uniform BCD word (digit k of |value| in bits [4k, 4k + 4)), then the
  // glyph cells that 16 x 4 pixel centres (x 1..16, GL rows 4..7: they cover its window box) see, one bit per lane
  const int value = UNI(__float_as_int(vr[VP_SCORE]));
  const int neg = value < 0 ? 1 : 0;
  unsigned long long bcd = 0ull; int nd = 0;
  for (unsigned t = (unsigned)(neg ? -(long long)value : value); ; ) { const unsigned q = t / 10u; bcd |= (unsigned long long)(t - q * 10u) << (4 * nd); ++nd; t = q; if (t == 0u) break; }
  nd = max(nd, 4 - neg);                                                     // zero padding of %04i (the sign counts)
  bool lit = false;
  {
    const float fx = ((float)(1 + (lane & 15)) + 0.5f) * (1000.0f / 96.0f) - LABEL_X0, fy = (((float)(4 + (lane >> 4)) + 0.5f) * (800.0f / 96.0f) - LABEL_Y0) * (1.0f / LABEL_CELL_H);
    const int j = (int)floorf(fx * (1.0f / LABEL_ADV));                      // character
    const int c = (int)floorf((fx - (float)j * LABEL_ADV) * (1.0f / LABEL_CELL_W));
    if (fx >= 0.0f && fy >= 0.0f && fy < 7.0f && j < nd + neg && c < 5) {
      const int d = nd - 1 - (j - neg);                                      // digit index, least significant = 0
      const int g = (neg && j == 0) ? 10 : (int)((bcd >> (4 * max(d, 0))) & 15ull);
      lit = (((uint32_t)glyphs[g * 7 + 6 - (int)floorf(fy)] >> (4 - c)) & 1u) != 0u;
    }
  }
  H.lmask = __ballot(lit);
  const int cg = lane >= 24 ? (lane >= 48 ? 23 : lane - 24) : lane;
  H.cg = cg; H.rsel = lane >= 24 ? 1 : 0;
  // x part of the gauges: the bytes of my 4 pixels inside [a, b1).  M(k) = the low k bytes = the high word of 0xffffffff << 8k
  H.rows_any = 0u;
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)gw, i);
    const int lo = min(max((int)(w & 255u) - 4 * cg, 0), 4), hi = min(max((int)((w >> 8) & 255u) - 4 * cg, 0), 4);
    const uint32_t mlo = (uint32_t)((0xffffffffull << (8 * lo)) >> 32), mhi = (uint32_t)((0xffffffffull << (8 * hi)) >> 32);
    H.xm[i] = mhi & ~mlo;
    H.rm[i] = w >> 16; H.rows_any |= H.rm[i];
  }
  if (H.lmask || flag_on) H.rows_any |= 0xf0u;
  // label bits of my group: bit (row - 4) * 16 + (x - 1) of lmask, x = 4 cg + j; x = 0 and x > 16 are outside the sampled window
  H.lvalid = cg == 0 ? 0xeu : cg < 4 ? 0xfu : cg == 4 ? 0x1u : 0u;
  return H;
}
__device__ __forceinline__ void hud_store(const HudState& H, uint32_t* __restrict__ frame, const int lane, const bool flag_on) {
  if (lane >= 48) return;
  const int cg = H.cg, rsel = H.rsel;
  typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
#pragma unroll
  for (int k = 0; k < HUD_ROWS / 2; ++k) {
    const int row = 2 * k + rsel;
    u32x3 wv; wv.x = wv.y = wv.z = 0u;                                       // black bar (:638-642)
    {                                                                        // (no "nothing is drawn in these two rows" shortcut: the six trips are
      uint32_t win = 0u;                                                     // independent dependency chains that the scheduler interleaves — branches between them would stop it)
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const uint32_t cls = i == 0 ? 0x01010101u : i <= 2 ? 0x02020202u : i <= 4 ? 0x03030303u : i == 5 ? 0x04040404u : 0x05050505u;
        const uint32_t m = H.xm[i] & (uint32_t)__builtin_amdgcn_sbfe((int)H.rm[i], row, 1);  // all ones iff the gauge holds my row
        win = (win & ~m) | (cls & m);
      }
      if (k == 2 || k == 3) {                                                // GL rows 4..7: label, then flag
        const int r4 = row - 4;
        uint32_t nib = cg == 0 ? ((uint32_t)(H.lmask >> (r4 * 16)) << 1) : (uint32_t)(H.lmask >> (r4 * 16 + 4 * min(cg, 4) - 1));
        nib &= H.lvalid;
        const uint32_t lm = ((nib * 0x00204081u) & 0x01010101u) * 0xffu;
        win = (win & ~lm) | (0x01010101u & lm);
        if (flag_on) {
          const uint32_t fm = hud_flag_mask(cg, r4);
          win = (win & ~fm) | (0x02020202u & fm);
        }
      }
      // classes -> planes (table byte = the class's channel value: R 0 255 0 51 0 255, G 0 255 0 0 255 0, B 0 255 255 255 0 0) -> packed RGB
      const uint32_t R4 = __builtin_amdgcn_perm(0x0000ff00u, 0x3300ff00u, win), G4 = __builtin_amdgcn_perm(0x000000ffu, 0x0000ff00u, win), B4 = __builtin_amdgcn_perm(0u, 0xffffff00u, win);
      wv.x = __builtin_amdgcn_perm(B4, __builtin_amdgcn_perm(G4, R4, 0x01000400u), 0x03040100u);      // R0 G0 B0 R1
      wv.y = __builtin_amdgcn_perm(B4, __builtin_amdgcn_perm(G4, R4, 0x06020005u), 0x03020500u);      // G1 B1 R2 G2
      wv.z = __builtin_amdgcn_perm(B4, __builtin_amdgcn_perm(G4, R4, 0x00070300u), 0x07020106u);      // B2 R3 G3 B3
    }
    __builtin_nontemporal_store(wv, (u32x3*)(frame + (size_t)((95 - row) * 24 + cg) * 3));
  }
}

}  // namespace view

// per-phase clock accumulators of thread 0 (the PHASES instantiation, launched when debug bit 32 is set): [view][16] u64,
// summed over the rounds of the view.  A separate instantiation: the accumulators cost registers the kernel does not have.
#define PHASE_ACC(i) do { if constexpr (PHASES) { const unsigned long long now_ = __builtin_readcyclecounter(); pacc[i] += now_ - tprev; tprev = now_; } } while (0)

// Main launches (one env per workgroup) are built for 4 workgroups per CU (16 wavefronts: 128 VGPRs, 40,960 B of LDS); the list launches
// (LIST: a handful of workgroups that walk a device-side list, with the cars' bookkeeping inlined) have registers of their own.
// (Round 4 also built the main launch as 1024 workgroups that stay and take envs off a queue in device memory — every view finds its data
// requested a view ahead, the three dependent round trips to HBM of a workgroup's first view are paid once per workgroup instead of once
// per env —: every view then costs what the FIRST view of a workgroup costs here (it carries the next view's block list and fetch), and
// the launch ends with a tail of one env's time (20 us of 80): 102 us instead of 81.  NOTES.md.)
template <bool PHASES, bool LIST>
__global__ __launch_bounds__(VIEW_THREADS, LIST ? 3 : 4) void k_view(McrParams p, unsigned long long* __restrict__ stamps, const int only_just_reset) {
  using namespace view;
  constexpr bool PERSIST = LIST;
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
  if (LIST) __builtin_amdgcn_s_setprio(3);                                  // a few envs beside the main launch that fills every CU: they go first
  const int vgrid = (int)gridDim.x - (LIST ? p.flags_blocks : 0);           // workgroups that draw; the rest: the list's bookkeeping
  if (LIST && (int)blockIdx.x >= vgrid) {
    // the bookkeeping of the list's cars (k_flags.h: backward / on-grass flags, the env's touch verdict for the next step), one wavefront
    // per car.  Contact list for the contact chain's raster, deferred list otherwise (the re-spawned envs' cars take none in this step).
    const int32_t* __restrict__ L = p.role == 2 ? p.clist : p.dlist;
    const int ncars = L[0] * N, per_round = p.flags_blocks * (VIEW_THREADS / 64);
    // (work items: the cars' flag scans, then one touch verdict per env — on different wavefronts, side by side)
    const int nverd = (p.part_next != nullptr && N > 1) ? L[0] : 0;
    for (int c = ((int)blockIdx.x - vgrid) * (VIEW_THREADS / 64) + (int)(threadIdx.x >> 6); c < ncars + nverd; c += per_round) {
      if (c < ncars) flags_block(p, c % N, L[1 + c / N], nverd != 0);
      else verdict_block(p, L[1 + c - ncars]);
    }
    return;
  }
  int my_env = -1, my_slot = 0, my_P = -1, my_agent = 0;
  // (LIST) a TERMINAL item (mcr_kernels.h: McrTermEnv): the frames of an env's finished episode.  my_env addresses the episode slot (the
  // env's old one), my_benv the per-car / per-env buffers — the entry's own view records, car polygons, tile flags and frames
  int my_benv = -1, my_term = 0;
  // List launches with `split_views`: a work slot is ONE VIEW (list entry s / N, agent s % N) instead of an env with its N views —
  // the few envs of a list are the tail of a chain on the step's critical path, and their views side by side take half the time
  // of one after the other (what an env's views share is fetched once per view then).
  const bool split_views = LIST && p.split_views != 0;
  {
    const int s = (int)blockIdx.x + (LIST ? lane * vgrid : 0);
    int e = -1;
    bool ojr = only_just_reset != 0;
    int term_i = -1;                                                       // position in the chain's list of terminal entries
    if (p.role == 5) {
      // the tail of the caller's stream: the deferred envs, then the envs the main dynamics re-spawned (their first observation), then the
      // terminal entries of the caller-side chains — one launch
      const int idx = split_views ? s / N : s, nd = p.dlist[0], nr = p.rlist[0];
      if (idx < nd) e = p.dlist[1 + idx]; else if (idx - nd < nr) { e = p.rlist[1 + idx - nd]; ojr = true; }
      else if (p.term_cnt != nullptr && idx - nd - nr < p.term_cnt[1]) term_i = idx - nd - nr;
      my_agent = split_views ? s % N : 0;
    }
    else if (p.role == 6) { const int idx = split_views ? s / N : s; if (p.term_cnt != nullptr && idx < p.term_cnt[1]) term_i = idx; my_agent = split_views ? s % N : 0; }   // single-stream step: the terminal entries alone
    else if (p.role >= 2) {
      const int idx = split_views ? s / N : s, nl = mcr_list_len(p);
      e = mcr_env_of_slot(p, idx); if (e >= p.env0 + p.nenv) e = -1; my_agent = split_views ? s % N : 0;
      if (p.role == 2 && p.term_cnt != nullptr && idx >= nl && idx - nl < p.term_cnt[2]) term_i = p.term_cap + idx - nl;      // the contact chain's entries
    }
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
    my_env = e; my_benv = e;
    if (LIST && term_i >= 0) {
      const int ti = p.term_list[term_i];
      const McrTermEnv te = p.term_env[ti];
      my_env = te.env; my_slot = te.slot; my_benv = ti; my_term = 1;
    }
  }
  unsigned long long todo = LIST ? __ballot(my_env >= 0) : (my_env >= 0 ? 1ull : 0ull);
  // (soft_sync: the step's join, see McrParams::await_tail)
  // (... and, with terminal observations, the step's last act: the entry count and the consumed-episode news, mcr_kernels.h: term_finish)
  auto join_tail = [&]() {
    if (LIST && p.await_tail && blockIdx.x == 0) {
      if (threadIdx.x == 0) { (void)mcr_await(p, W_SIDE); (void)mcr_await(p, W_MAIN); }
      if (p.term_cnt != nullptr) { __syncthreads(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); term_finish(p); }
    }
  };
  if (!todo) { join_tail(); return; }
  auto slot_of = [&](int k) -> const uint8_t* {
    return p.slots + ((size_t)__builtin_amdgcn_readlane(my_env, k) * 2 + __builtin_amdgcn_readlane(my_slot, k)) * MCR_SLOT_BYTES;
  };
  int env, P, a_lo;
  int term = 0;                                                             // (LIST) the item being drawn is a terminal entry
  const uint8_t* __restrict__ slot;
  { const int k = UNI(__builtin_ctzll(todo)); todo &= todo - 1; env = __builtin_amdgcn_readlane(my_benv, k); slot = slot_of(k); a_lo = __builtin_amdgcn_readlane(my_agent, k); if (LIST) term = __builtin_amdgcn_readlane(my_term, k); }
  // the per-car / per-env buffers of an item: the live ones, or the terminal entries' (same layouts, `env` = the entry index)
  auto vp_of = [&](int t) -> const float* { return (LIST && t) ? p.term_viewp : p.viewp; };
  auto cp_of = [&](int t) -> const float* { return (LIST && t) ? p.term_carpoly : p.carpoly; };
  auto tf_of = [&](int t) -> const uint16_t* { return (LIST && t) ? p.term_tflags : p.tile_flags; };
  auto ob_of = [&](int t) -> uint8_t* { return (LIST && t) ? p.term_obs : p.obs; };
  // what a candidate needs from HBM, requested one round ahead: a quad's 4 vertices + meta word, or 4 vertices of a Car.draw polygon +
  // its vertex count (the 8-gon's two slots fetch vertices 0..3 and 4..7)
  struct Raw { float4 a, b; uint32_t m; };
  Raw nxt; nxt.a = nxt.b = make_float4(0.0f, 0.0f, 0.0f, 0.0f); nxt.m = 0u;
  P = (!LIST && my_P >= 0) ? UNI(my_P) : ((const McrSlotHeader*)slot)->P;
  const int dbg = p.debug;
  unsigned long long pacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = PHASES ? __builtin_readcyclecounter() : 0ull;

  __shared__ __attribute__((aligned(16))) uint32_t keyb[ROWS * KS];         // draw key per scene pixel, GL rows from 12 up (row 0 = GL row 12)
  __shared__ __attribute__((aligned(16))) float4 rdat[RC][2];               // per record: slopes and constants of 4 edges
  __shared__ uint32_t rmeta[RC];
  __shared__ uint32_t rkey[RC];                                             // rank << 24 | RGB
  __shared__ uint16_t tasks4[TASK4_CAP];                                    // record slot << 5 | group of 4 lines
  __shared__ unsigned long long trec[(MCR_TILE_CAP / 128) * 2];             // "tile was touched" (:102-104) bit masks: [tile / 128][tile & 1], bit (tile >> 1) & 63
  __shared__ uint32_t palc[24];
  __shared__ float vrec[2][MCR_VIEWP_FLOATS];                               // view record of the view being drawn / the next one
  __shared__ float glo[20], ghi[20];
  __shared__ uint8_t glyphs[80];
  __shared__ int wsum[4];
  __shared__ uint8_t vblk[2][NBLK];                                         // visible blocks of the view being drawn / the next one
  __shared__ uint8_t vcar[2][MCR_MAX_AGENTS];                               // visible cars
  __shared__ int nvb[2], nvc[2];
  static_assert(MCR_TILE_CAP / 2 == VIEW_THREADS, "one tile-flag word per thread");
  static_assert(view::PAL_COUNT <= 24, "palette copy");
  // One wavefront lists, per view, the road_poly blocks (runs of MCR_QBLK consecutive entries; boxes from the track
  // generator) the scene rectangle (obs x 0..96, y 12..96) can see: only their entries become candidates.  Separating-
  // axis test both ways — the box in pixel space against the rectangle, the rectangle in world space against the box —
  // with a pixel of slack.  Lanes 48 .. 48 + N list the cars whose polygons can reach the scene rectangle.
  // In two parts, so that what runs between them (the HUD's preparation) covers the loads: lb_load requests a lane's box, a car's anchor and — lanes
  // 0..11, one float each: vector loads, which unlike scalar ones are not waited for by the first LDS access — the view's camera and its inverse.
  struct LbRaw { float4 bbox; float a0, a1, a8, a9, cam; };
  auto lb_load = [&](const uint8_t* __restrict__ sl, int e, int vw, const int lane, int t) -> LbRaw {
    LbRaw r; r.bbox = make_float4(1.0f, 1.0f, -1.0f, -1.0f); r.a0 = r.a1 = r.a8 = r.a9 = 0.0f;
    if (lane < NBLK) r.bbox = ((const float4*)(sl + MCR_OFF_QBLK))[lane];
    const int cc = lane - NBLK;
    if (cc >= 0 && cc < N) {                                                // two opposite vertices of the hull's 8-gon (Car.draw polygon 10)
      const float* __restrict__ cp = cp_of(t) + (size_t)(e * N + cc) * MCR_CARPOLY_FLOATS + 10 * 16;
      r.a0 = cp[0]; r.a1 = cp[1]; r.a8 = cp[8]; r.a9 = cp[9];
    }
    r.cam = vp_of(t)[(size_t)vw * MCR_VIEWP_FLOATS + min(lane, 11)];        // VP_CAM (6) then VP_INV (6)
    return r;
  };
  static_assert(VP_CAM == 0 && VP_INV == 6, "lb_load fetches the first 12 floats of a view record");
  auto lb_finish = [&](const LbRaw& r, int buf, const int lane) {
    const float4 bbox = r.bbox;
    const int cc = lane - NBLK;
    const float ax = 0.5f * (r.a0 + r.a8), ay = 0.5f * (r.a1 + r.a9);       // the middle of the hull
    const int cam = __float_as_int(r.cam);
    const float c0 = __int_as_float(__builtin_amdgcn_readlane(cam, 0)), c1 = __int_as_float(__builtin_amdgcn_readlane(cam, 1)), c2 = __int_as_float(__builtin_amdgcn_readlane(cam, 2));
    const float c3 = __int_as_float(__builtin_amdgcn_readlane(cam, 3)), c4 = __int_as_float(__builtin_amdgcn_readlane(cam, 4)), c5 = __int_as_float(__builtin_amdgcn_readlane(cam, 5));
    const float v0 = __int_as_float(__builtin_amdgcn_readlane(cam, 6)), v1 = __int_as_float(__builtin_amdgcn_readlane(cam, 7)), v2 = __int_as_float(__builtin_amdgcn_readlane(cam, 8));
    const float v3 = __int_as_float(__builtin_amdgcn_readlane(cam, 9)), v4 = __int_as_float(__builtin_amdgcn_readlane(cam, 10)), v5 = __int_as_float(__builtin_amdgcn_readlane(cam, 11));
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
    const float cpx = __builtin_fmaf(c0, ax, __builtin_fmaf(c1, ay, c4)), cpy = __builtin_fmaf(c2, ax, __builtin_fmaf(c3, ay, c5));
    const float rx = CAR_RADIUS * (fabsf(c0) + fabsf(c1)) + 1.0f, ry = CAR_RADIUS * (fabsf(c2) + fabsf(c3)) + 1.0f;
    const bool cvis = cc >= 0 && cc < N && cpx >= -rx && cpx <= 96.0f + rx && cpy >= 12.0f - ry && cpy <= 96.0f + ry;
    const unsigned long long mask = __ballot(vis), cmask = __ballot(cvis);
    if (vis) vblk[buf][__popcll(mask & ((1ull << lane) - 1ull))] = (uint8_t)lane;
    if (cvis) vcar[buf][__popcll(cmask & ((1ull << lane) - 1ull))] = (uint8_t)cc;
    if (lane == 0) { nvb[buf] = __popcll(mask); nvc[buf] = __popcll(cmask); }
  };
  // the env's "tile was touched" flags (tile_flags bit 8: recoloured, :102-104), one bit per tile: thread t holds tiles 2t and 2t + 1
  auto pack_tile_flags = [&](uint32_t tf) {
    const unsigned long long me = __ballot((tf & 0x100u) != 0u), mo = __ballot((tf & 0x1000000u) != 0u);
    if (lane == 0) { trec[wave * 2] = me; trec[wave * 2 + 1] = mo; }
  };

  // ---- once per workgroup: palette, glyphs, grass lattice; the first env's tile flags, view record, block list
  if (tid < 24) palc[tid] = PALETTE_RGB[tid];
  if (tid >= 128 && tid < 128 + 77) glyphs[tid - 128] = ((const uint8_t*)LABEL_GLYPHS)[tid - 128];
  pack_tile_flags(((const uint32_t*)(tf_of(term) + (size_t)env * MCR_TILE_CAP))[tid]);
  if (tid >= 64 && tid < 64 + MCR_VIEWP_FLOATS) vrec[0][tid - 64] = vp_of(term)[(size_t)(env * N + a_lo) * MCR_VIEWP_FLOATS + (tid - 64)];
  if (wave == 3) lb_finish(lb_load(slot, env, env * N + a_lo, lane, term), 0, lane);
  // grass lattice as the reference builds it (:620-627): f32(k*x) and f32(k*x + k) for x = -20, -18, .., 18
  if (tid >= 224 && tid < 244) { const double k = MCR_PLAYFIELD / 20.0, x = 2.0 * (double)(tid - 224 - 10); glo[tid - 224] = (float)(k * x + 0); ghi[tid - 224] = (float)(k * x + k); }
  // Candidates of a view (env e, episode slot sl with pe road_poly entries, visible blocks vblk[buf][0 .. pv / QBLK), visible
  // cars vcar[buf][0 .. cs / 14)): [0, pv) the road_poly entries of the visible blocks; from `sb` on the specials: 14 slots per
  // visible car (its 12 Car.draw polygons, world vertices from k_dynamics; the 8-gon hull polygon takes slots 10 + 11, slot 13
  // is a pad that keeps pairs aligned), the 5 vertical gauges when one of them is taller than the HUD bar, the playfield quad
  // (only when the view leaves it), the G light grass squares the viewport can see.
  auto quad_of = [&](int c, int buf) -> int { return (int)vblk[buf][c / MCR_QBLK] * MCR_QBLK + (c & (MCR_QBLK - 1)); };
  auto fetch_raw = [&](int c, int pv, int sb, int cs, int buf, const uint8_t* __restrict__ sl, int pe, int e, int t) -> Raw {
    Raw r; r.a = r.b = make_float4(0.0f, 0.0f, 0.0f, 0.0f); r.m = 0u;
    if (c < pv) {
      const int q = quad_of(c, buf);
      if (q < pe) { r.a = ((const float4*)(sl + MCR_OFF_QA))[q]; r.b = ((const float4*)(sl + MCR_OFF_QB))[q]; r.m = ((const uint32_t*)(sl + MCR_OFF_QMETA))[q]; }
    } else if (c >= sb && c - sb < cs) {
      const int vc = (c - sb) / 14, sl14 = (c - sb) - vc * 14;
      if (sl14 != 13) {
        const int j = sl14 < 11 ? sl14 : sl14 - 1;                          // slot 11 sets up the second half of polygon 10 (the 8-gon): its vertices 4..7
        const float* __restrict__ cp = cp_of(t) + (size_t)(e * N + (int)vcar[buf][vc]) * MCR_CARPOLY_FLOATS;
        const float4* cv = (const float4*)(cp + j * 16) + (sl14 == 11 ? 2 : 0);
        r.a = cv[0]; r.b = cv[1];
        r.m = (uint32_t)__float_as_int(cp[MCR_CARPOLY_NOFF + j]);
      }
    }
    return r;
  };
  // The slot layout of a view.  When the road quads and the cars fit the first two wavefronts and the rest the 40 slots of the
  // third (the rule once the camera has zoomed in) the cars END at slot 128 and gauges / playfield / grass start there — every
  // wavefront then runs one or two kinds of candidate, not all of them one after the other; otherwise the specials are contiguous
  // at the end of the last round (they land on the wavefronts the quads leave idle).
  struct Layout { int pv, cs, tg, f, g, nround, sb; };
  auto layout_of = [&](int buf) -> Layout {
    const float* __restrict__ v = vrec[buf];
    Layout L;
    L.pv = UNI(nvb[buf]) * MCR_QBLK;
    L.cs = (dbg & 4) ? 0 : UNI(nvc[buf]) * 14;
    L.tg = UNI(__float_as_int(v[VP_HUDTOP])) > UNI(__float_as_int(13.0f)) ? 5 : 0;      // (positive floats order like their bit patterns)
    L.f = L.cs + L.tg + (UNI(__float_as_int(v[VP_GRASS + 4])) ? 0 : 1);                 // + the playfield quad
    L.g = UNI(__float_as_int(v[VP_GRASS + 1])) * UNI(__float_as_int(v[VP_GRASS + 3]));
    const int nspec = (L.f + L.g + 1) & ~1;                                 // even: the 8-gon's slot pair never straddles two rounds
    L.nround = (L.pv + nspec + RC - 1) / RC;
    const bool spread = L.pv + L.cs <= 128 && L.f - L.cs + L.g <= RC - 128;
    L.sb = spread ? 128 - L.cs : L.nround * RC - nspec;
    if (spread) L.nround = 1;
    return L;
  };
  int vs = 0;                                                               // views this workgroup has drawn
#pragma nounroll
  for (;;) {
  // the workgroup's next env, if any
  const bool has_next = LIST && todo != 0ull;
  int env_n = env, a_lo_n = 0, term_n = term; const uint8_t* __restrict__ slot_n = slot;
  uint32_t tfl_n = 0u; int P_nv = 0;
  if (LIST && has_next) { const int k = UNI(__builtin_ctzll(todo)); todo &= todo - 1; env_n = __builtin_amdgcn_readlane(my_benv, k); slot_n = slot_of(k); a_lo_n = __builtin_amdgcn_readlane(my_agent, k); term_n = __builtin_amdgcn_readlane(my_term, k); }
  const int a_hi = split_views ? a_lo + 1 : N;
#pragma nounroll
  for (int agent = a_lo; agent < a_hi; ++agent, ++vs) {
    const int vw = env * N + agent;
    const int buf = vs & 1;
    const float* __restrict__ vr = vrec[buf];
    // the view after this one: the env's next agent, or agent 0 of the workgroup's next env
    const bool last = agent + 1 == a_hi;
    const bool nv_ok = !last || has_next;
    const int env_v = last ? env_n : env, vw_v = last ? env_n * N + a_lo_n : vw + 1, term_v = last ? term_n : term;
    const uint8_t* __restrict__ slot_v = last ? slot_n : slot;
    // the view's own copy of the thread index: what the phases derive from it (key-buffer addresses of clear and resolve, frame offsets,
    // the HUD's column masks) is computed where it is used instead of being hoisted out of the view loop and kept in 20 VGPRs across it
    int tl = tid; asm volatile("" : "+v"(tl));
    const int ll = tl & 63;
    PHASE_ACC(0);
    __syncthreads();                                                        // this view's record and block list are in LDS; the previous view's resolve is through with the key buffer
    PHASE_ACC(1);
    const unsigned long long tview = PHASES ? __builtin_readcyclecounter() : 0ull;
    if (!PERSIST && vs == 0 && tl == 0 && my_P >= 0) p.vorder[blockIdx.x] = -1;   // every wavefront has read the entry: free it for the step after next
    // the next view's record and (new env) tile flags / entry count travel while this one is drawn
    // (requested here, stored to LDS behind the candidates: nobody waits for it)
    const bool vrec_mine = nv_ok && tl >= 64 && tl < 64 + MCR_VIEWP_FLOATS;
    float vrec_n = 0.0f;
    if (vrec_mine) vrec_n = vp_of(term_v)[(size_t)vw_v * MCR_VIEWP_FLOATS + (tl - 64)];
    if (last && has_next) { tfl_n = ((const uint32_t*)(tf_of(term_n) + (size_t)env_n * MCR_TILE_CAP))[tl]; if (LIST) P_nv = ((const McrSlotHeader*)slot_n)->P; }
    // camera, shifted so that key-buffer row 0 is GL row 12
    const float m00 = vr[VP_CAM + 0], m01 = vr[VP_CAM + 1], m10 = vr[VP_CAM + 2], m11 = vr[VP_CAM + 3], ctx = vr[VP_CAM + 4], cty = vr[VP_CAM + 5] - (float)HUD_ROWS;
    // grass squares the viewport can see / "is the whole viewport inside the playfield" (k_dynamics, from the inverse camera)
    const int mu0 = UNI(__float_as_int(vr[VP_GRASS + 0])), nu = UNI(__float_as_int(vr[VP_GRASS + 1])), mv0 = UNI(__float_as_int(vr[VP_GRASS + 2]));
    const bool inside_field = UNI(__float_as_int(vr[VP_GRASS + 4])) != 0;
    // key-buffer clear: the scene's base colour
    {
      const uint32_t kbase = inside_field ? ((uint32_t)RANK_PLAYFIELD << 24) | palc[PAL_GRASS0] : 0u;   // black beyond the playfield
      uint4* k4 = (uint4*)keyb;
      for (int i = tl; i < KEYW4; i += VIEW_THREADS) k4[i] = make_uint4(kbase, kbase, kbase, kbase);
    }
    if (wave == 3) {
      // ---- the wavefront without candidate slots, while the other three set the first round's candidates up
      // The next view's block / car lists and this view's HUD rows, straight into the frame: loads first, their use after the HUD's
      // preparation, the HUD's stores last.  This block is one long dependency chain (16 ticks per instruction at equal priority, 10 for
      // the candidate code of the others) and the phase's critical path: it issues first.  (Measured with the lists in wavefront 0, whose
      // candidates are road quads only: that one then arrives last — 82 instead of 79 us.)
      __builtin_amdgcn_s_setprio(2);
      LbRaw lbr; lbr.bbox = make_float4(1.0f, 1.0f, -1.0f, -1.0f); lbr.a0 = lbr.a1 = lbr.a8 = lbr.a9 = lbr.cam = 0.0f;
      if (nv_ok) lbr = lb_load(slot_v, env_v, vw_v, ll, term_v);
      const bool hud_flag = (__float_as_uint(vr[VP_OLDFLAGS]) & 1u) != 0u && p.backwards_flag != 0;
      const HudState hud = hud_prep(vr, glyphs, ll, hud_flag);
      if (nv_ok) lb_finish(lbr, buf ^ 1, ll);
      if (!(dbg & 8)) hud_store(hud, (uint32_t*)(ob_of(term) + (size_t)vw * (96 * 96 * 3)), ll, hud_flag);
      __builtin_amdgcn_s_setprio(LIST ? 3 : 0);
    }
    const Layout L = layout_of(buf);
    const int Pv = L.pv, CS = L.cs, TG = L.tg, F = L.f, G = L.g, nround = L.nround, SB = L.sb;
    if (vs == 0 && tl < RC) nxt = fetch_raw(tl, Pv, SB, CS, buf, slot, P, env, term);   // later views: requested while the previous one was drawn
#pragma nounroll
    for (int rd = 0; rd < nround; ++rd) {
      const int c = rd * RC + tl;
      const Raw cur = nxt;
      if (rd > 0) {                                                         // what the earlier rounds drew keeps its colour, below everything to come (but above the grass)
        uint4* k4 = (uint4*)keyb;
        for (int i = tl; i < KEYW4; i += VIEW_THREADS) {
          uint4 k = k4[i];
          k.x = k.x > (((uint32_t)RANK_FLAT << 24) | 0xffffffu) ? (k.x & 0xffffffu) | ((uint32_t)RANK_FLAT << 24) : k.x;
          k.y = k.y > (((uint32_t)RANK_FLAT << 24) | 0xffffffu) ? (k.y & 0xffffffu) | ((uint32_t)RANK_FLAT << 24) : k.y;
          k.z = k.z > (((uint32_t)RANK_FLAT << 24) | 0xffffffu) ? (k.z & 0xffffffu) | ((uint32_t)RANK_FLAT << 24) : k.z;
          k.w = k.w > (((uint32_t)RANK_FLAT << 24) | 0xffffffu) ? (k.w & 0xffffffu) | ((uint32_t)RANK_FLAT << 24) : k.w;
          k4[i] = k;
        }
      }
      int incl = 0, g4 = 0;
      if (wave != 3) {
        const bool mine = tl < RC;
        uint32_t my_meta = 0u, my_key = 0u;                                 // a culled slot has no lines
        // Every candidate, whatever its kind, is "4 vertices in world or key-buffer space + a rank and a colour": a short kind-specific
        // head produces them, ONE common tail transforms, culls and sets the span record up — a wavefront that holds several kinds
        // of candidates runs the tail once, not once per kind.
        float wx[4], wy[4];
        int rank = 0, pal = 0, nn = 0;                                      // nn: 0 nothing; 4: vertices in world space; -4: in key-buffer space
        bool eight = false, half2 = false;                                  // this slot sets up one half of an 8-gon / the SECOND half (edges 4..7)
        const int q = (mine && c < Pv) ? quad_of(c, buf) : P;
        if (q < P) {
          // ---- road_poly entry
          const uint32_t meta = cur.m;
          const uint32_t tile1 = (meta >> 8) & 0x3ffu;
          if (!(dbg & 18)) {
            wx[0] = cur.a.x; wy[0] = cur.a.y; wx[1] = cur.a.z; wy[1] = cur.a.w; wx[2] = cur.b.x; wy[2] = cur.b.y; wx[3] = cur.b.z; wy[3] = cur.b.w;
            uint32_t col = meta & 0xffu;
            if (tile1) {                                                    // touched tile -> ROAD_COLOR (:102-104)
              const uint32_t t0 = tile1 - 1u;
              if ((trec[((t0 >> 7) << 1) | (t0 & 1u)] >> ((t0 >> 1) & 63u)) & 1ull) col = MCR_COL_ROAD0;
            }
            pal = col == MCR_COL_ROAD0 ? PAL_ROAD0 : col == MCR_COL_ROAD1 ? PAL_ROAD1 : col == MCR_COL_ROAD2 ? PAL_ROAD2 : col == MCR_COL_KERB_WHITE ? PAL_WHITE : PAL_RED255;
            rank = RANK_SLOT0 + tl; nn = 4;
          }
        } else if (const int sidx = (mine && c >= SB) ? c - SB : -1; sidx >= 0 && sidx < F + G) {
          // ---- specials
          if (sidx < CS) {                                                  // Car.draw polygon
            const int vc = sidx / 14, sl = sidx - vc * 14;
            const int j = sl < 11 ? sl : sl - 1;
            const int n = (int)cur.m;
            if (sl != 13 && n > 0 && (sl != 11 || n > 4)) {
              wx[0] = cur.a.x; wy[0] = cur.a.y; wx[1] = cur.a.z; wy[1] = cur.a.w; wx[2] = cur.b.x; wy[2] = cur.b.y; wx[3] = cur.b.z; wy[3] = cur.b.w;
              const int cc = (int)vcar[buf][vc];
              if (j < 8) pal = (j & 1) ? PAL_WHEELWHITE : PAL_BLACK;
              else { pal = PAL_CAR0 + (cc & 7); if (p.use_ego_color) pal = (cc == agent) ? PAL_CAR0 + 0 : PAL_CAR0 + 1; }   // :402, :560-563
              // only HULL_POLY3 (slots 10 + 11) has more than 4 vertices (mcr_create checks); k_dynamics pads to 8.  Both of its
              // slots carry the rank of the FIRST one: the fill reads the second record through the first one's meta word
              eight = n > 4 && sl >= 10 && sl <= 11;
              half2 = sl == 11;
              rank = RANK_SLOT0 + tl - (half2 ? 1 : 0);
              nn = 4;
            }
          } else if (sidx < CS + TG) {                                      // a vertical gauge (:643-648) taller than the HUD bar: its part in the scene rows
            const int h = sidx - CS;
            const float gx0 = vr[VP_IND + h * 4], gx1 = vr[VP_IND + h * 4 + 1], gy0 = vr[VP_IND + h * 4 + 2] - (float)HUD_ROWS, gy1 = vr[VP_IND + h * 4 + 3] - (float)HUD_ROWS;
            if (gx1 > gx0 && gy1 > gy0) {
              wx[0] = gx0; wy[0] = gy0; wx[1] = gx1; wy[1] = gy0; wx[2] = gx1; wy[2] = gy1; wx[3] = gx0; wy[3] = gy1;
              pal = h == 0 ? PAL_WHITE : h <= 2 ? PAL_BLUE255 : PAL_PURPLE;
              rank = RANK_SLOT0 + tl; nn = -4;
            }
          } else if (sidx < F) {                                            // playfield quad (:615-619), only when the view leaves it
            const float PF = (float)MCR_PLAYFIELD;
            wx[0] = -PF; wy[0] = PF; wx[1] = PF; wy[1] = PF; wx[2] = PF; wy[2] = -PF; wx[3] = -PF; wy[3] = -PF;
            rank = RANK_PLAYFIELD; pal = PAL_GRASS0; nn = 4;
          } else {                                                          // light grass square (:620-627)
            const int g = sidx - F, iv = g / nu, iu = g - iv * nu;
            const int tu = mu0 + iu + 10, tv = mv0 + iv + 10;
            wx[0] = ghi[tu]; wy[0] = glo[tv]; wx[1] = glo[tu]; wy[1] = glo[tv]; wx[2] = glo[tu]; wy[2] = ghi[tv]; wx[3] = ghi[tu]; wy[3] = ghi[tv];
            rank = RANK_GRASS; pal = PAL_GRASS1; nn = 4;
          }
        }
        // ---- common tail
        if (nn != 0) {
          my_key = ((uint32_t)rank << 24) | palc[pal];
          float px[4], py[4];
          if (nn > 0) {                                                     // camera transform (world -> key buffer)
#pragma unroll
            for (int i = 0; i < 4; ++i) { px[i] = __builtin_fmaf(m00, wx[i], __builtin_fmaf(m01, wy[i], ctx)); py[i] = __builtin_fmaf(m10, wx[i], __builtin_fmaf(m11, wy[i], cty)); }
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) { px[i] = wx[i]; py[i] = wy[i]; }
          }
          float x0 = fminf(fminf(px[0], px[1]), fminf(px[2], px[3])), x1 = fmaxf(fmaxf(px[0], px[1]), fmaxf(px[2], px[3]));
          float y0 = fminf(fminf(py[0], py[1]), fminf(py[2], py[3])), y1 = fmaxf(fmaxf(py[0], py[1]), fmaxf(py[2], py[3]));
          float area = (px[0] * py[1] - px[1] * py[0]) + (px[1] * py[2] - px[2] * py[1]) + (px[2] * py[3] - px[3] * py[2]);
          float cx = px[0], cy = py[0];                                     // the vertex that closes my four edges
          if (eight) {
            // the two halves of the 8-gon sit in an even / odd lane pair: the partner's first vertex closes my edges, and bounding box
            // and area are the whole polygon's (both lanes arrive at the same values: min, max and + commute)
            cx = pair_swap(px[0]); cy = pair_swap(py[0]);
            area += px[3] * cy - cx * py[3];
            x0 = fminf(x0, pair_swap(x0)); x1 = fmaxf(x1, pair_swap(x1)); y0 = fminf(y0, pair_swap(y0)); y1 = fmaxf(y1, pair_swap(y1));
            area += pair_swap(area);
          } else area += px[3] * py[0] - px[0] * py[3];
          int i0, i1, j0, j1;
          if (centre_range(x0, x1, 0, 95, i0, i1) && centre_range(y0, y1, 0, ROWS - 1, j0, j1)) {
            const uint32_t geo = setup_half(px, py, cx, cy, area, i0, i1, j0, j1, rdat, tl);
            // the second half keeps its bound bits (the fill reads them through the first half's chain bit) but has no lines of its own
            if (geo) my_meta = half2 ? (geo & 0xfu) : geo | (eight ? REC_CHAIN : 0u);
          }
        }
        if (mine) { rmeta[tl] = my_meta; rkey[tl] = my_key; }
        // line counts in slot order -> exclusive prefix sum of the 4-line task groups over the three wavefronts
        const int nl = (int)(my_meta >> REC_NL_SHIFT);
        g4 = (nl + 3) >> 2;
        incl = g4;
        // wave-wide inclusive scan by DPP: shifts within the rows of 16 lanes (zero fill), then the row totals broadcast to the rows behind
        // (six v_add with a DPP operand; __shfl_up is six trips through the LDS crossbar with a wait each)
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x111 /* row_shr:1 */, 0xf, 0xf, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x112 /* row_shr:2 */, 0xf, 0xf, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x114 /* row_shr:4 */, 0xf, 0xf, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x118 /* row_shr:8 */, 0xf, 0xf, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x142 /* row_bcast:15 */, 0xa, 0xf, false);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x143 /* row_bcast:31 */, 0xc, 0xf, false);
        if (ll == 63) wsum[wave] = incl;
      }
      if (rd == 0 && vrec_mine) vrec[buf ^ 1][tl - 64] = vrec_n;
      if constexpr (PHASES) { if (rd == 0 && wave > 0 && ll == 0 && stamps) stamps[(size_t)vw * 16 + 12 + wave] = __builtin_readcyclecounter() - tview; }   // when wavefronts 1..3 reach the barrier
      PHASE_ACC(2);
      __syncthreads();                                                      // records and line counts of the round are in LDS
      PHASE_ACC(3);
      // next round's candidates (or round 0 of the next agent's view: the same quads): their HBM / L2 data travels
      // while this round's spans are drawn
      if (tl < RC) {
        if (rd + 1 < nround) nxt = fetch_raw(c + RC, Pv, SB, CS, buf, slot, P, env, term);
        else if (nv_ok) {                                                   // round 0 of the next view (its record, block and car lists are in LDS)
          const Layout Ln = layout_of(buf ^ 1);
          nxt = fetch_raw(tl, Ln.pv, Ln.sb, Ln.cs, buf ^ 1, slot_v, last ? UNI(P_nv) : P, env_v, term_v);
        }
      }
      int off = incl - g4;
      if (wave >= 1) off += wsum[0];
      if (wave >= 2) off += wsum[1];
      const int total = wsum[0] + wsum[1] + wsum[2];
      int t0 = 0;
      do {
        for (int k4 = 0; k4 < g4; ++k4) {
          const int idx = off + k4 - t0;
          if ((unsigned)idx < (unsigned)TASK4_CAP) tasks4[idx] = (uint16_t)((tl << 5) | k4);
        }
        PHASE_ACC(4);
        __syncthreads();
        PHASE_ACC(5);
        const int nlines = min(TASK4_CAP, total - t0) * 4;
        // Two line tasks per ll and trip: the dependent LDS reads of a task (task entry -> record meta -> record) are
        // issued for both before either is used — the phase is bound by that latency chain, not by the arithmetic.  Lanes
        // without a task read entry 0 (harmless) and end up with an empty span.
        for (int i0 = 0; i0 < nlines; i0 += 2 * VIEW_THREADS) {
          const int iA = i0 + tl, iB = iA + VIEW_THREADS;
          const bool inA = iA < nlines, inB = iB < nlines;
          const uint32_t eA = tasks4[inA ? iA >> 2 : 0], eB = tasks4[inB ? iB >> 2 : 0];
          const int slA = min((int)(eA >> 5), RC - 1), slB = min((int)(eB >> 5), RC - 1);
          const int kA = (int)((eA & 31u) << 2) | (iA & 3), kB = (int)((eB & 31u) << 2) | (iB & 3);
          const uint32_t mA = rmeta[slA], mB = rmeta[slB];
          const uint32_t cA = rkey[slA], cB = rkey[slB];
          const float4 SA = rdat[slA][0], CA = rdat[slA][1];
          const float4 SB4 = rdat[slB][0], CB = rdat[slB][1];
          auto bounds = [&](uint32_t kinds, const float4 Sl, const float4 Cn, float v, float& lo, float& hi) {
            const float b0 = __builtin_fmaf(Sl.x, v, Cn.x), b1 = __builtin_fmaf(Sl.y, v, Cn.y), b2 = __builtin_fmaf(Sl.z, v, Cn.z), b3 = __builtin_fmaf(Sl.w, v, Cn.w);
            const bool l0 = (kinds & 1u) != 0u, l1 = (kinds & 2u) != 0u, l2 = (kinds & 4u) != 0u, l3 = (kinds & 8u) != 0u;
            lo = fmaxf(lo, fmaxf(fmaxf(l0 ? b0 : -BIG, l1 ? b1 : -BIG), fmaxf(l2 ? b2 : -BIG, l3 ? b3 : -BIG)));
            hi = fminf(hi, fminf(fminf(l0 ? BIG : b0, l1 ? BIG : b1), fminf(l2 ? BIG : b2, l3 ? BIG : b3)));
          };
          auto span = [&](bool in, uint32_t meta, int sl, int k, const float4 Sl, const float4 Cn, uint32_t col,
                          int& len, int& addr, int& stride, uint32_t& key) {
            len = 0; addr = 0; stride = 1; key = 0u;
            if (in && k < (int)(meta >> REC_NL_SHIFT)) {
              const int line = (int)((meta >> REC_L0_SHIFT) & 127u) + k;
              const float v = (float)line + 0.5f;
              const bool row = (meta & REC_ROW) != 0u;
              float lo = 0.0f, hi = row ? 96.0f : (float)ROWS;              // the scene rectangle
              bounds(meta, Sl, Cn, v, lo, hi);
              if (meta & REC_CHAIN) bounds(rmeta[sl + 1], rdat[sl + 1][0], rdat[sl + 1][1], v, lo, hi);
              key = col;
              const int a = (int)ceilf(lo - 0.5f), b = (int)floorf(hi - 0.5f);    // pixel centres a+.5 .. b+.5 lie in [lo, hi]
              len = b - a + 1;
              addr = (row ? line : a) * KS + (row ? a : line);
              stride = row ? 1 : KS;
            }
          };
          int lenA, addrA, strideA, lenB, addrB, strideB; uint32_t keyA, keyB;
          span(inA, mA, slA, kA, SA, CA, cA, lenA, addrA, strideA, keyA);
          span(inB, mB, slB, kB, SB4, CB, cB, lenB, addrB, strideB, keyB);
          // a ll leaves a loop when its span is drawn (the wavefront iterates to its longest span)
          // (two pixels per trip: the loop's bookkeeping — 3 scalar + 2 vector instructions — is what the phase issues most of;
          // the second ds_max of an odd span's last trip carries key 0, which changes nothing wherever it lands — a column scan's
          // can land one row behind the buffer, in the records: max(x, 0) = x there as well)
          if (lenA > 0) { int j = 0; do { atomicMax(&keyb[addrA], keyA); atomicMax(&keyb[addrA + strideA], j + 1 < lenA ? keyA : 0u); addrA += 2 * strideA; j += 2; } while (j < lenA); }
          if (lenB > 0) { int j = 0; do { atomicMax(&keyb[addrB], keyB); atomicMax(&keyb[addrB + strideB], j + 1 < lenB ? keyB : 0u); addrB += 2 * strideB; j += 2; } while (j < lenB); }
        }
        PHASE_ACC(6);
        __syncthreads();                                                    // the next chunk / round rewrites tasks and records
        PHASE_ACC(7);
        t0 += TASK4_CAP;
      } while (t0 < total);
    }
    // ---- resolve + packed RGB write-out of the scene rows: 4 pixels -> 12 bytes per ll, rows top-down (arr[::-1], :602)
    if (!(dbg & 8)) {
      uint32_t* __restrict__ out = (uint32_t*)(ob_of(term) + (size_t)vw * (96 * 96 * 3));
      // 240 threads x 9 trips of 10 rows: a thread keeps its 4-pixel group's column and walks down 10 rows per trip, so that key-buffer and
      // frame addresses are one division per view plus constants (256 threads x 8 trips: a division and a bounds check per trip); the
      // winners' RGB bytes are packed with three byte permutes
      const int r0 = tl / 24, c4 = (tl - r0 * 24) * 4;
      if (tl < 240) {
        const uint32_t* kp0 = &keyb[(ROWS - 1 - r0) * KS + c4];
        uint32_t* o0 = out + (size_t)tl * 3;
#pragma unroll
        for (int g0 = 0; g0 < (ROWS + 9) / 10; ++g0) {
          if (g0 * 10 + 10 <= ROWS || r0 + g0 * 10 < ROWS) {               // (the last trip: rows 80..83)
            const uint32_t* kp = kp0 - g0 * 10 * KS;
            const uint32_t c0 = kp[0], c1 = kp[1], c2 = kp[2], c3 = kp[3];
            // streaming store: the frames of a step (226 MB at B = 4096, N = 2) are written once and read by nobody on the
            // device — they must not evict the state the step's latency-bound chains live on from L2 / MALL
            typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
            u32x3 wv;
            wv.x = __builtin_amdgcn_perm(c1, c0, 0x04020100u); wv.y = __builtin_amdgcn_perm(c2, c1, 0x05040201u); wv.z = __builtin_amdgcn_perm(c3, c2, 0x06050402u);
            __builtin_nontemporal_store(wv, (u32x3*)(o0 + (size_t)g0 * 240 * 3));
          }
        }
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
  env = env_n; slot = slot_n; P = UNI(P_nv); pack_tile_flags(tfl_n); a_lo = a_lo_n; term = term_n;
  }
  join_tail();
}
