// k_collide.h — one wavefront per env: the sensor half of b2ContactManager::Collide plus the
// FrictionDetector callbacks (multi_car_racing.py:80-123) at the poses ENTERING the step.
//
//  * lanes 0..8N-1 first build the env's car fixtures in world space (4 hull polygons + 4 wheels per
//    car) into LDS, 8-lane groups reduce each car's AABB;
//  * the wave then streams the track's tile AABBs from HBM (one float4 per lane per 64 tiles, fully
//    coalesced); only tiles whose AABB meets a car's box load their hull and run the overlap test;
//  * begin events are replayed in the order Box2D's Collide fires them — contacts of a later FindNewContacts first, then
//    tile descending, then car / wheel descending (the broadphase model below) — so that "who visited the
//    tile first" and the f64 reward accumulation are the reference's (:113-120);
//  * the per-wheel "touches any tile" mask (friction_limit, Car.step) is a wave-wide OR;
//  * car<->car: proxies for all 8 fixtures of a car (b2DynamicTree::MoveProxy semantics) and a creation stamp per pair of fixtures of two
//    cars whose fat AABBs overlap (= Box2D's contact exists) — what orders the contact edges b2World::Solve's island search walks; the
//    touching pairs' manifolds (b2CollidePolygons), stored in the order of that search together with each car's joint order.
//
// Overlap predicate == b2TestOverlap itself (GJK b2Distance with radii, touching iff distance < 10*FLT_EPSILON: k_gjk.h),
// behind a SAT far-field filter that never changes its result (col::overlap).
#pragma once
#include "mcr_kernels.h"
#include "k_carcontacts.h"
#include "k_gjk.h"
#include "k_world.h"

namespace col {

// Tile sensor hull (3 or 4 vertices) kept in NAMED registers: every loop over it is fully unrolled with an
// `i < n` predicate so that nothing is indexed dynamically (dynamic indexing would put it in scratch memory).
struct TilePoly { int n; float vx[4], vy[4], nx[4], ny[4]; };

#define CAND_CAP 128                 // (tile, car) candidate pairs buffered before the overlap pass runs
#define EVQ_CAP 128                  // begin events replayed per env and pass (4N wheels x the few tiles a wheel can reach in a step)
// dynamic LDS of k_collide: 8N fixtures x (2 float4 + 4 x 8 floats + count)
__host__ __device__ inline size_t lds_bytes(int N) { return (size_t)8 * N * (2 * 16 + 4 * 8 * 4 + 4); }
#define TILE_FOR(i) _Pragma("unroll") for (int i = 0; i < 4; ++i)

// max over A's edge normals of the min projection of B's vertices (A = car fixture in LDS, B = tile)
__device__ __forceinline__ float sat_fixture_tile(const float* avx, const float* avy, const float* anx, const float* any_, int an,
                                                  const TilePoly& T) {
  float best = -MCR_MAXFLT;
  for (int i = 0; i < an; ++i) {
    const float nx = anx[i], ny = any_[i], ax = avx[i], ay = avy[i];
    float mn = MCR_MAXFLT;
    TILE_FOR(j) { if (j < T.n) { const float d = nx * (T.vx[j] - ax) + ny * (T.vy[j] - ay); mn = mcr_min(mn, d); } }
    best = mcr_max(best, mn);
  }
  return best;
}
__device__ __forceinline__ float sat_tile_fixture(const TilePoly& T, const float* bvx, const float* bvy, int bn) {
  float mn[4] = {MCR_MAXFLT, MCR_MAXFLT, MCR_MAXFLT, MCR_MAXFLT};
  for (int j = 0; j < bn; ++j) {
    const float bx = bvx[j], by = bvy[j];
    TILE_FOR(i) { const float d = T.nx[i] * (bx - T.vx[i]) + T.ny[i] * (by - T.vy[i]); mn[i] = mcr_min(mn[i], d); }
  }
  float best = -MCR_MAXFLT;
  TILE_FOR(i) { if (i < T.n) best = mcr_max(best, mn[i]); }
  return best;
}
// b2TestOverlap(tile, car fixture) — Box2D's GJK (k_gjk.h) behind a far-field filter.  The maximum SAT separation s over the
// edge normals of both polygons (f32, error ~1e-5 at 250-unit coordinates) is a lower bound of the core distance:
//   s > 0.02 + 1e-3  -> the cores are farther apart than the two radii by ~50x the f32 noise: GJK says "apart";
//   s <= 0           -> no edge normal separates the cores: they intersect or are within the noise of it (corners of >= 50
//                       degrees: distance <= 1.6 s): GJK returns a distance far below 0.02 and says "touching";
//   in between       -> GJK decides.  (oracle sweep, tests/test_oracle_pinning.py: an exact-distance predicate and GJK
//                       never differ beyond 5e-5 of the threshold, so the filter's 1e-3 changes no result.)
// (fixture_first: the car fixture holds the lower proxy id and is the pair's fixtureA — b2TestOverlap(fixture, tile) instead of (tile, fixture))
__device__ __forceinline__ bool overlap(const float* avx, const float* avy, const float* anx, const float* any_, int an, const TilePoly& T,
                                        const float4 va, const float4 vb, const McrPoly* __restrict__ PB, const float4 xfB, const bool fixture_first) {
  const float R = 2.0f * B2_POLYGON_RADIUS;
  const float s = mcr_max(sat_fixture_tile(avx, avy, anx, any_, an, T), sat_tile_fixture(T, avx, avy, an));
  if (s > R + 1e-3f) return false;
  if (s <= 0.0f) return true;
  return fixture_first ? gjk::touching_fixture_first(va, vb, T.n, PB, xfB) : gjk::touching(va, vb, T.n, PB, xfB);
}

// cross-lane reductions by DPP (one VALU operation per level, no LDS round trip).  Groups of 8 lanes: lane ^ 1, lane ^ 2 inside the quad,
// then the mirror image inside the 8 (the quads are uniform by then).
#define GRP8_REDUCE(op, x) do { x = op(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xb1, 0xf, 0xf, true))); \
                                x = op(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4e, 0xf, 0xf, true))); \
                                x = op(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141, 0xf, 0xf, true))); } while (0)
// OR over the wavefront, uniform result: rows of 16 by DPP (quad, half-row mirror, row mirror), then the four rows' lane 0
__device__ __forceinline__ uint32_t wave_or(uint32_t x) {
  x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xb1, 0xf, 0xf, true);
  x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4e, 0xf, 0xf, true);
  x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xf, 0xf, true);
  x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xf, 0xf, true);
  return (uint32_t)__builtin_amdgcn_readlane((int)x, 0) | (uint32_t)__builtin_amdgcn_readlane((int)x, 16) | (uint32_t)__builtin_amdgcn_readlane((int)x, 32) | (uint32_t)__builtin_amdgcn_readlane((int)x, 48);
}

}  // namespace col

// pass 0: every active env.  pass 1: only envs with `resetting` set (their per-tile state is cleared first).
// debug bit 15: clock of lane 0 at the phase boundaries of the step's contact pass (tools/collide_phases.py)
#define COL_STAMP(i) do { if ((p.debug & 32768) && pass == 0 && threadIdx.x == 0) p.dbg_stamps[(size_t)env * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
__device__ __forceinline__ void collide_block(const McrParams& p, const int pass, const int blk) {
  using namespace col;
  const int env = mcr_env_of_slot(p, blk), lane = threadIdx.x;
  if (env >= p.env0 + p.nenv) return;
  COL_STAMP(0);
  // Everything the pass reads that does not depend on the env's state word is asked for HERE, together with it: the wavefront is a
  // chain of dependent memory round trips (env state -> slot -> tiles), and loads issued one after the other cost one trip, not one each.
  const int pre_c = lane >> 3, pre_fi = lane & 7, pre_body = pre_fi < 4 ? 0 : pre_fi - 3;
  const bool pre_on = pre_c < p.N;
  float pre_cx = 0.0f, pre_cy = 0.0f, pre_a = 0.0f;
  float4 pre_fat = make_float4(0.0f, 0.0f, 0.0f, 0.0f), pre_prev = pre_fat, pre_prevp = pre_fat;
  float pvx[8], pvy[8], pnx[8], pny[8]; int pre_pn = 0;
  if (pre_on) {
    const int ci = env * p.N + pre_c, wi = ci * MCR_BP_FIX + pre_fi, WS = MCR_BP_FIX * p.BN;
    pre_cx = p.carf[(CF_CX + pre_body) * p.BN + ci]; pre_cy = p.carf[(CF_CY + pre_body) * p.BN + ci]; pre_a = p.carf[(CF_A + pre_body) * p.BN + ci];
    pre_fat = p.bpf[BP_FAT * WS + wi]; pre_prev = p.bpf[BP_PREV * WS + wi]; pre_prevp = p.bpf[BP_PREVP * WS + wi];
    const McrPoly& P = pre_fi < 4 ? p.shapes->hull[pre_fi] : p.shapes->wheel;
    pre_pn = P.n;
#pragma unroll
    for (int i = 0; i < 8; ++i) { pvx[i] = P.vx[i]; pvy[i] = P.vy[i]; pnx[i] = P.nx[i]; pny[i] = P.ny[i]; }      // (all eight: no load waits for the count)
  }
  const McrEnvState es = p.env[env];
  if (pass == 0 && p.fuse_collide && p.role < 2 && p.part[env]) return;      // a contact env: its chain's workgroup runs this pass itself (k_list_chain.h)
  if (!es.active) {
    // contact pass in front (single stream, N > 4, serialised kernels): this pass owns the contact chain's marks — an env
    // that froze while its cars were touching must not keep a stale one (it would be skipped by every main launch)
    if (pass == 0 && p.split && !p.cc_mode && lane == 0) p.part[env] = 0;
    return;
  }
  if (pass == 1 && !es.resetting) return;
  const int N = p.N, BN = p.BN;
  const uint8_t* slot = p.slots + ((size_t)env * 2 + es.slot) * MCR_SLOT_BYTES;
  const int T = ((const McrSlotHeader*)slot)->T;
  // broadphase proxy ids (k_world.h: mcr_pid_tables): the env's ONE world across its episodes keeps a table (the default); a fresh_world handle's
  // ids ascend in creation order (tile t -> t, car fixture pf -> TILE_CAP + pf sorts the same way) unless the episode slot brought tables along
  // (a caller's literal tree, mcr_world.cpp)
  const McrPidTables pidt = mcr_pid_tables(p, env, slot);
  const bool has_pid = pidt.has;
  const uint16_t* TPID = pidt.tile; const uint16_t* FPID = pidt.fix;
  auto tile_pid = [&](int t) -> uint32_t { return has_pid ? (uint32_t)TPID[t] : (uint32_t)t; };
  auto fix_pid = [&](int pf) -> uint32_t { return has_pid ? (uint32_t)FPID[pf] : (uint32_t)(MCR_TILE_CAP + pf); };
  // (asked for now, used after the fixtures are built: the boxes of the track's tile blocks; the cars' reward / visit-count accumulators)
  float4 kb_pre = make_float4(1.0f, 1.0f, -1.0f, -1.0f);
  if (lane < MCR_TILE_CAP / MCR_TBLK) kb_pre = ((const float4*)(slot + MCR_OFF_TBLK))[lane];
  double reward_pre = 0.0; uint32_t tvc_pre = 0u;
  if (lane < N) { reward_pre = p.card[CD_REWARD * BN + env * N + lane]; tvc_pre = p.caru[CU_TVC * BN + env * N + lane]; }
  uint32_t* touch = p.tile_touch + (size_t)env * MCR_TILE_CAP;
  uint16_t* tflags = p.tile_flags + (size_t)env * MCR_TILE_CAP;
  if (pass == 1) for (int t = lane; t < MCR_TILE_CAP; t += 64) { touch[t] = 0; tflags[t] = 0; }

  // per-fixture arrays for the env's 8N car fixtures live in DYNAMIC LDS (col::lds_bytes(N)): at N=2 that is 2.6 KB
  // instead of 10.5 KB, which lifts the LDS-limited occupancy from 11 to 16 wavefronts per CU — with one wavefront
  // per env, all 4096 envs of the bench are then resident in a single round.
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
  const int F = 8 * p.N;
  float4* fxf = (float4*)dyn_lds; float4* fbox = fxf + F;           // body transform (p.x p.y s c), world AABB +-0.05
  float (*fvx)[8] = (float (*)[8])(fbox + F); float (*fvy)[8] = fvx + F; float (*fnx)[8] = fvy + F; float (*fny)[8] = fnx + F;
  int* fcnt = (int*)(fny + F);
  __shared__ float cbox[MCR_MAX_AGENTS][4];
  __shared__ uint32_t newrec[MCR_CC_MAX][16];
  __shared__ uint32_t tres[MCR_TILE_CAP], tany[MCR_TILE_CAP / 32];
  __shared__ __attribute__((aligned(8))) uint32_t cand[CAND_CAP];
  // ---- broadphase model of the wheels (Box2D: b2Fixture::Synchronize / b2DynamicTree::MoveProxy at the END of the last
  // step, b2BroadPhase::UpdatePairs -> b2ContactManager::AddPair right after it; evaluated lazily here, on the next pass).
  // A tile<->wheel contact exists exactly while the two proxies' FAT AABBs overlap, and Collide serves the contacts
  // newest batch first (AddPair inserts at the list head, Collide walks from the head), inside a batch in descending
  // (tile proxy id, wheel proxy id) order (UpdatePairs sorts the pairs ascending before AddPair sees them): what has to be
  // remembered per (tile, wheel) is the pass whose FindNewContacts made the contact (bp_stamp), and that follows from the
  // wheel's fat AABB before and after each move.  First-episode proxy ids: ascending in creation order — tiles in track
  // order, then car by car the hull polygons and the wheels.
  __shared__ float4 wfat_new[MCR_MAX_AGENTS * 4], wfat_old[MCR_MAX_AGENTS * 4];
  __shared__ float cfat[MCR_MAX_AGENTS][4];                        // per car: union of its wheels' fat AABBs
  __shared__ __attribute__((aligned(16))) unsigned long long evq[EVQ_CAP];   // begin events of this pass: stamp << 14 | tile << 5 | car * 4 + wheel
  __shared__ __attribute__((aligned(16))) uint16_t tfl16[MCR_TILE_CAP];      // flags word of the tiles that take a begin event
  __shared__ int evn;
  // reset() on the env's one world (k_world.h): _destroy + the new episode's fixtures re-issue the proxy ids before this pass looks one up
  if (pass == 1 && p.pid_tab) mcr_world_reissue_ids(p, env, T, (uint16_t*)tres);
  // the fat AABB of EVERY car fixture (hull polygons as well): needed by the car<->car broadphase contacts, which are settled right
  // after the proxies — before the tile phases first write the two arrays these alias
  float (* const cfat8)[4] = (float (*)[4])cand;                  // per car: union of the fat AABBs of all 8 fixtures (which car pairs can hold contacts); dead before the candidate list is first written
  float4* const ffat_new = (float4*)evq;                           // [8N] (64 x 16 B = the event queue's 1 KB)
  static_assert(sizeof(unsigned long long) * EVQ_CAP >= sizeof(float4) * MCR_MAX_AGENTS * 8, "ffat_new aliases evq");
  const uint32_t label = pass == 1 ? 0u : es.bp_step;
  const bool fresh = pass == 1 || p.bp_fresh != 0;
  // car<->car broadphase contacts of the env (mcr_common.h: mcr_cc_stamp_words); the two words of car pairs that hold any are asked for
  // here, with the proxies' loads, and used after the first barrier
  uint32_t* const ccs = p.cc_stamp + (size_t)env * mcr_cc_stamp_words(N);
  unsigned long long live_old = 0ull;
  if (p.car_contacts && N > 1 && !fresh) live_old = (unsigned long long)ccs[0] | ((unsigned long long)ccs[1] << 32);
  unsigned long long moved_mask;                                    // bit car * 4 + wheel (in the low 32 bits)
  {
    const int c = lane >> 3, fi = lane & 7;
    float lox = MCR_MAXFLT, loy = MCR_MAXFLT, hix = -MCR_MAXFLT, hiy = -MCR_MAXFLT;
    if (c < N) {
      const int body = pre_body;
      const McrShapes& S = *p.shapes;
      V2 cc = v2(pre_cx, pre_cy);
      float a = pre_a;
      V2 lc = body == 0 ? v2(S.hull_lcx, S.hull_lcy) : v2(0.0f, 0.0f);
      Xf xf = xf_of(cc, a, lc);
      fcnt[lane] = pre_pn;
      fxf[lane] = make_float4(xf.p.x, xf.p.y, xf.q.s, xf.q.c);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (i < pre_pn) {
          V2 w = xmul(xf, v2(pvx[i], pvy[i]));
          V2 n = rmul(xf.q, v2(pnx[i], pny[i]));
          fvx[lane][i] = w.x; fvy[lane][i] = w.y; fnx[lane][i] = n.x; fny[lane][i] = n.y;
          lox = mcr_min(lox, w.x); loy = mcr_min(loy, w.y); hix = mcr_max(hix, w.x); hiy = mcr_max(hiy, w.y);
        }
      }
      fbox[lane] = make_float4(lox - 0.05f, loy - 0.05f, hix + 0.05f, hiy + 0.05f);
    }
    bool moved = false;
    float flx = MCR_MAXFLT, fly = MCR_MAXFLT, fhx = -MCR_MAXFLT, fhy = -MCR_MAXFLT;       // this WHEEL's fat AABB after the update
    float glx = MCR_MAXFLT, gly = MCR_MAXFLT, ghx = -MCR_MAXFLT, ghy = -MCR_MAXFLT;       // this fixture's (wheel or hull polygon)
    if (c < N) {
      const int wi = (env * N + c) * MCR_BP_FIX + fi, WS = MCR_BP_FIX * BN;
      const float4 xfq = fxf[lane];
      const float ax0 = lox - B2_POLYGON_RADIUS, ay0 = loy - B2_POLYGON_RADIUS, ax1 = hix + B2_POLYGON_RADIUS, ay1 = hiy + B2_POLYGON_RADIUS;   // b2PolygonShape::ComputeAABB
      float4 of = make_float4(MCR_MAXFLT, MCR_MAXFLT, -MCR_MAXFLT, -MCR_MAXFLT);           // "no old proxy": overlaps nothing
      float ux0 = ax0, uy0 = ay0, ux1 = ax1, uy1 = ay1, dx = 0.0f, dy = 0.0f;
      if (!fresh) {
        of = pre_fat;
        // b2Fixture::Synchronize: union of the AABBs at the transforms the last step was entered and left with
        const float4 pv = pre_prev, pp = pre_prevp;
        ux0 = mcr_min(pv.x, ax0); uy0 = mcr_min(pv.y, ay0); ux1 = mcr_max(pv.z, ax1); uy1 = mcr_max(pv.w, ay1);
        dx = 2.0f * (xfq.x - pp.x); dy = 2.0f * (xfq.y - pp.y);                              // b2_aabbMultiplier * displacement
      }
      float4 nf = of;
      bool mv = false;
      if (fresh || !(of.x <= ux0 && of.y <= uy0 && ux1 <= of.z && uy1 <= of.w)) {          // b2DynamicTree::MoveProxy (b2AABB::Contains)
        nf = make_float4(ux0 - 0.1f, uy0 - 0.1f, ux1 + 0.1f, uy1 + 0.1f);                   // b2_aabbExtension
        if (dx < 0.0f) nf.x += dx; else nf.z += dx;
        if (dy < 0.0f) nf.y += dy; else nf.w += dy;
        mv = true;
        p.bpf[BP_FAT * WS + wi] = nf;
      }
      p.bpf[BP_PREV * WS + wi] = make_float4(ax0, ay0, ax1, ay1);
      p.bpf[BP_PREVP * WS + wi] = make_float4(xfq.x, xfq.y, 0.0f, 0.0f);
      ffat_new[lane] = nf;
      glx = nf.x; gly = nf.y; ghx = nf.z; ghy = nf.w;
      if (fi >= 4) {
        moved = mv;
        wfat_new[c * 4 + (fi - 4)] = nf; wfat_old[c * 4 + (fi - 4)] = of;
        flx = nf.x; fly = nf.y; fhx = nf.z; fhy = nf.w;
      }
    }
    {
      const unsigned long long mm = __ballot(moved);                                         // lane = car * 8 + 4 + wheel -> bit car * 4 + wheel
      unsigned long long packed = 0ull;
      for (int cc_ = 0; cc_ < N; ++cc_) packed |= ((mm >> (cc_ * 8 + 4)) & 0xfull) << (cc_ * 4);
      moved_mask = packed;
    }
    GRP8_REDUCE(mcr_min, flx); GRP8_REDUCE(mcr_min, fly); GRP8_REDUCE(mcr_max, fhx); GRP8_REDUCE(mcr_max, fhy);
    if (fi == 0 && c < MCR_MAX_AGENTS) { cfat[c][0] = flx; cfat[c][1] = fly; cfat[c][2] = fhx; cfat[c][3] = fhy; }
    GRP8_REDUCE(mcr_min, glx); GRP8_REDUCE(mcr_min, gly); GRP8_REDUCE(mcr_max, ghx); GRP8_REDUCE(mcr_max, ghy);
    if (fi == 0 && c < MCR_MAX_AGENTS) { cfat8[c][0] = glx; cfat8[c][1] = gly; cfat8[c][2] = ghx; cfat8[c][3] = ghy; }
    if (lane == 0) evn = 0;
    GRP8_REDUCE(mcr_min, lox); GRP8_REDUCE(mcr_min, loy); GRP8_REDUCE(mcr_max, hix); GRP8_REDUCE(mcr_max, hiy);
    if (fi == 0 && c < MCR_MAX_AGENTS) { cbox[c][0] = lox - 0.05f; cbox[c][1] = loy - 0.05f; cbox[c][2] = hix + 0.05f; cbox[c][3] = hiy + 0.05f; }
  }
  __syncthreads();
  COL_STAMP(1);

  // ---- car<->car broadphase contacts (b2ContactManager: AddPair / Destroy for the dynamic fixture pairs).  A contact of fixtures
  // (pa, pb) of two different cars (not wheel vs wheel: the fixtures' filters) exists exactly while their fat AABBs overlap: made by
  // the FindNewContacts that first sees the overlap (label + 1 is remembered: contacts of a later batch, and inside a batch of a
  // higher (pa, pb), sit nearer the heads of the bodies' contact-edge lists — what b2World::Solve's island DFS walks, below),
  // destroyed by the Collide that sees it gone.  Only car pairs whose boxes of fat AABBs meet, or that held contacts, are looked at.
  if (p.car_contacts && N > 1) {
    unsigned long long live_new = 0ull;
    const int Fq = 8 * N;
    unsigned long long look;                                              // bit a * 8 + b: the pair's boxes meet
    {
      const int a = lane >> 3, b = lane & 7;
      const bool near = a < b && b < N && !(cfat8[a][0] > cfat8[b][2] || cfat8[a][2] < cfat8[b][0] || cfat8[a][1] > cfat8[b][3] || cfat8[a][3] < cfat8[b][1]);
      look = __ballot(near) | live_old;
    }
    while (look) {
      const int ab = __builtin_ctzll(look); look &= look - 1ull;
      const int a = ab >> 3, b = ab & 7;
      const bool had = ((live_old >> ab) & 1ull) != 0ull;
      const int fa = lane >> 3, fb = lane & 7, pa = a * 8 + fa, pb = b * 8 + fb;
      const bool pairs = !(fa >= 4 && fb >= 4);                             // wheel vs wheel: filtered (b2ContactFilter::ShouldCollide)
      uint32_t st = 0u, st0 = 0u;
      if (pairs) {
        if (had) st0 = ccs[2 + pa * Fq + pb];
        const float4 A = ffat_new[pa], Bq = ffat_new[pb];
        const bool ov = !(Bq.x - A.z > 0.0f || Bq.y - A.w > 0.0f || A.x - Bq.z > 0.0f || A.y - Bq.w > 0.0f);       // b2TestOverlap(b2AABB, b2AABB)
        st = ov ? (st0 ? st0 : label + 1u) : 0u;
      }
      const bool any = __ballot(st != 0u) != 0ull;
      // (a pair that held no contact has nothing valid in memory: its first contacts write all of its words)
      if (pairs && (had ? st != st0 : any)) ccs[2 + pa * Fq + pb] = st;
      if (any) live_new |= 1ull << (a * 8 + b);
    }
    if (lane == 0 && (fresh || live_new != live_old)) { ccs[0] = (uint32_t)live_new; ccs[1] = (uint32_t)(live_new >> 32); }
  }
  __syncthreads();                                                 // ffat_new (= evq) is free from here on
  COL_STAMP(2);

  // per-car accumulators: lane c keeps car c's (reward_pre, tvc_pre above)
  double reward_acc = reward_pre; int tvc_acc = (int)tvc_pre;

  const float4* TAABB = (const float4*)(slot + MCR_OFF_TAABB);
  const float4* TVA = (const float4*)(slot + MCR_OFF_TVA); const float4* TVB = (const float4*)(slot + MCR_OFF_TVB);
  const float4* TNA = (const float4*)(slot + MCR_OFF_TNA); const float4* TNB = (const float4*)(slot + MCR_OFF_TNB);
  const uint32_t* TCNT = (const uint32_t*)(slot + MCR_OFF_TCNT);
  // ---- phase A/B: candidate (tile, car) pairs by AABB, compacted into LDS; the expensive overlap predicate then
  // runs with one lane per (pair, fixture) item, so all 64 lanes work instead of the few whose tile is near a car.
  // Result per tile in LDS: tres bit 4c+w = wheel w of car c touches; tany bit = some fixture (hull or wheel) touches.
  for (int t = lane; t < T; t += 64) tres[t] = 0;
  if (lane < MCR_TILE_CAP / 32) tany[lane] = 0;
  __syncthreads();
  int ncand = 0;                                                    // wave-uniform
  auto flush = [&]() {
    __syncthreads();                                               // cand[] writes -> reads (one wave per block)
    const int items = ncand * 8;
    for (int i0 = 0; i0 < items; i0 += 64) {
      const int i = i0 + lane;
      if (i < items) {
        const uint32_t pc = cand[i >> 3]; const int fi = i & 7;
        const int t = (int)(pc & 0xffffu), c = (int)(pc >> 16), f = c * 8 + fi;
        const float4 bb = TAABB[t], fb = fbox[f];
        if (!(bb.x > fb.z || bb.z < fb.x || bb.y > fb.w || bb.w < fb.y)) {
          const float4 va = TVA[t], vb = TVB[t], na = TNA[t], nb = TNB[t];
          TilePoly TP;
          TP.n = (int)(TCNT[t] & 0xffu);
          TP.vx[0] = va.x; TP.vy[0] = va.y; TP.vx[1] = va.z; TP.vy[1] = va.w; TP.vx[2] = vb.x; TP.vy[2] = vb.y; TP.vx[3] = vb.z; TP.vy[3] = vb.w;
          TP.nx[0] = na.x; TP.ny[0] = na.y; TP.nx[1] = na.z; TP.ny[1] = na.w; TP.nx[2] = nb.x; TP.ny[2] = nb.y; TP.nx[3] = nb.z; TP.ny[3] = nb.w;
          const McrShapes& S = *p.shapes;
          if (overlap(fvx[f], fvy[f], fnx[f], fny[f], fcnt[f], TP, va, vb, fi < 4 ? &S.hull[fi] : &S.wheel, fxf[f], has_pid && fix_pid(f) < tile_pid(t)))
          { if (fi >= 4) atomicOr(&tres[t], 1u << (c * 4 + (fi - 4))); atomicOr(&tany[t >> 5], 1u << (t & 31)); }
        }
      }
    }
    ncand = 0;
    __syncthreads();
  };
  // Only the blocks of MCR_TBLK consecutive tiles (boxes from the track generator) that a car's box meets can hold a
  // new contact, and only those that held a touching tile after the last pass can lose one: the pass looks at the union
  // (typically 1-3 of ~18 blocks) instead of every tile of the track.
  uint32_t visit;
  {
    const float4 kb = kb_pre;
    bool near = false;
    for (int c = 0; c < N; ++c) near = near || !(kb.x > cbox[c][2] || kb.z < cbox[c][0] || kb.y > cbox[c][3] || kb.w < cbox[c][1]);
    // ... and, where a wheel's proxy moved, the blocks whose tiles' fat AABBs (tight -+ 0.11) can meet the car's fat AABBs:
    // that is where FindNewContacts makes contacts
    if (moved_mask) for (int c = 0; c < N; ++c)
      if ((moved_mask >> (4 * c)) & 0xfull) near = near || !(kb.x - 0.12f > cfat[c][2] || kb.z + 0.12f < cfat[c][0] || kb.y - 0.12f > cfat[c][3] || kb.w + 0.12f < cfat[c][1]);
    visit = (uint32_t)__ballot(near && kb.x <= kb.z) | (pass == 1 ? 0u : es.touch_blocks);
  }
  // the visited blocks' tiles, four blocks at a time: lane -> tile t (or -1)
  auto next_tiles = [&](uint32_t& m) -> int {
    int b[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) { b[g] = m ? (int)__builtin_ctz(m) : -1; m &= m - 1u; }
    const int g = lane >> 4;
    const int kb = g == 0 ? b[0] : g == 1 ? b[1] : g == 2 ? b[2] : b[3];
    const int t = kb * MCR_TBLK + (lane & (MCR_TBLK - 1));
    return (kb >= 0 && t < T) ? t : -1;
  };
  for (uint32_t vm = visit; vm;) {
    const int t = next_tiles(vm);
    float4 bb = make_float4(MCR_MAXFLT, MCR_MAXFLT, -MCR_MAXFLT, -MCR_MAXFLT);
    if (t >= 0) bb = TAABB[t];
    if (moved_mask) {
      // FindNewContacts: a moved wheel proxy gets a contact with every tile whose fat AABB its NEW fat AABB overlaps and its
      // old one did not (those already had one); all of them belong to this pass's batch
      const float tx0 = (bb.x - B2_POLYGON_RADIUS) - 0.1f, ty0 = (bb.y - B2_POLYGON_RADIUS) - 0.1f, tx1 = (bb.z + B2_POLYGON_RADIUS) + 0.1f, ty1 = (bb.w + B2_POLYGON_RADIUS) + 0.1f;
      for (int c = 0; c < N; ++c) {
        unsigned mw = (unsigned)(moved_mask >> (4 * c)) & 0xfu;
        if (!mw) continue;
        // (a new fat AABB lies inside the box of its car's four: cars none of whose boxes meets a tile of this batch are skipped whole)
        const bool nearc = t >= 0 && !(cfat[c][0] > tx1 || tx0 > cfat[c][2] || cfat[c][1] > ty1 || ty0 > cfat[c][3]);
        if (!__any(nearc)) continue;
        for (; mw; mw &= mw - 1u) {
          const int f = c * 4 + __builtin_ctz(mw);
          const float4 nf = wfat_new[f], of = wfat_old[f];
          const bool on = !(nf.x > tx1 || tx0 > nf.z || nf.y > ty1 || ty0 > nf.w);         // b2TestOverlap(aabb, aabb)
          const bool oo = !(of.x > tx1 || tx0 > of.z || of.y > ty1 || ty0 > of.w);
          if (t >= 0 && on && !oo) p.bp_stamp[((size_t)env * MCR_TILE_CAP + t) * (4 * N) + f] = label;
        }
      }
    }
    for (int c = 0; c < N; ++c) {
      const bool hit = !(bb.x > cbox[c][2] || bb.z < cbox[c][0] || bb.y > cbox[c][3] || bb.w < cbox[c][1]);
      const unsigned long long m = __ballot(hit);
      if (!m) continue;
      const int k = __popcll(m);
      if (ncand + k > CAND_CAP) flush();
      if (hit) cand[ncand + __popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)t | ((uint32_t)c << 16);
      ncand += k;
    }
  }
  flush();
  COL_STAMP(3);

  // ---- phase C: per-tile contact state (lane = tile); the begin events go to a queue ...
  uint32_t or_bits = 0, new_blocks = 0;
  for (uint32_t vm = visit; vm;) {
    const int t = next_tiles(vm);
    const bool valid = t >= 0;
    uint32_t newbits = 0;
    uint32_t old = 0; uint32_t fl = 0;
    if (valid) {
      old = touch[t]; fl = tflags[t];
      newbits = tres[t];
      if (((tany[t >> 5] >> (t & 31)) & 1u) || old != newbits) fl |= 0x100u;          // any Begin/End recolours the tile (:102-104)
    }
    uint32_t begins = newbits & ~old;
    if (valid) {
      touch[t] = newbits;
      if (begins) tfl16[t] = (uint16_t)fl; else tflags[t] = (uint16_t)fl;               // a tile with begin events is written back after the replay
    }
    for (; begins; begins &= begins - 1u) {
      const int f = __builtin_ctz(begins);                                              // car * 4 + wheel
      // key: batch, then the pair's (lower, higher) proxy id — b2PairLessThan — then, below the bits that order, the event itself
      const uint32_t pt = tile_pid(t), pw = fix_pid((f >> 2) * 8 + 4 + (f & 3));
      const unsigned long long key = (1ull << 62) | ((unsigned long long)p.bp_stamp[((size_t)env * MCR_TILE_CAP + t) * (4 * N) + f] << 38) |
                                     ((unsigned long long)(pt < pw ? pt : pw) << 26) | ((unsigned long long)(pt < pw ? pw : pt) << 14) | ((unsigned long long)t << 5) | (unsigned long long)f;
      const int slot = atomicAdd(&evn, 1);
      if (slot < EVQ_CAP) evq[slot] = key;
    }
    or_bits |= newbits;
    if (newbits) new_blocks |= 1u << (t / MCR_TBLK);
  }
  __syncthreads();
  // ... and are replayed in Box2D's order: highest (batch, lower proxy id, higher proxy id) first — (batch, tile, car * 4 + wheel) in a fresh world.  Every lane keeps the same reward /
  // count accumulators; the road_visited bits of the tiles involved live in LDS during the replay.
  {
    int n = evn;
    if (n > EVQ_CAP) { if (lane == 0) mcr_raise(p, ST_EVENT_OVERFLOW); n = EVQ_CAP; }
    unsigned long long k0 = lane < n ? evq[lane] : 0ull, k1 = lane + 64 < n ? evq[lane + 64] : 0ull;
    const unsigned long long m0 = k0, m1 = k1;
    for (int i = 0; i < n; ++i) {
      unsigned long long m = k0 > k1 ? k0 : k1;
      for (int o = 1; o < 64; o <<= 1) { const unsigned long long other = (unsigned long long)__shfl_xor((long long)m, o); m = other > m ? other : m; }
      if (k0 == m) k0 = 0ull; else if (k1 == m) k1 = 0ull;                              // keys are distinct ((tile, wheel) pairs) and never 0 (bit 62)
      const int t = (int)((m >> 5) & 511ull), c = (int)((m >> 2) & 7ull);
      uint32_t fl = tfl16[t];
      if (!(fl & (1u << c))) {                                                          // FrictionDetector._contact (:113-120)
        fl |= 1u << c;
        if (lane == c) {
          tvc_acc += 1;
          const int past = __popc(fl & 0xffu) - 1;
          const double factor = 1 - ((double)past / (double)N);
          reward_acc += factor * 1000.0 / (double)T;
        }
        __syncthreads();                                                                // every lane has read the word
        if (lane == 0) tfl16[t] = (uint16_t)fl;
      }
      __syncthreads();
    }
    if (m0) { const int t = (int)((m0 >> 5) & 511ull); tflags[t] = tfl16[t]; }          // (lanes sharing a tile store the same word)
    if (m1) { const int t = (int)((m1 >> 5) & 511ull); tflags[t] = tfl16[t]; }
  }
  COL_STAMP(4);
  new_blocks = wave_or(new_blocks);
  if (lane == 0) { p.env[env].touch_blocks = new_blocks; p.env[env].bp_step = label + 1u; }
  or_bits = wave_or(or_bits);
  if (lane < N) {
    const int ci = env * N + lane;
    const uint32_t onr = (or_bits >> (4 * lane)) & 0xFu;
    const double r = reward_acc; const int tv = tvc_acc;
    if (pass == 0 && p.cc_mode) {
      // the main dynamics, running beside this launch (possibly on another XCD, behind another L2), reads these three words
      // of the car: device-scope stores and loads (write-through / L2-bypassing) instead of a release fence per workgroup —
      // a fence writes the XCD's whole dirty L2 back (measured: 4096 of them take the launch from 24 to 146 us)
      __hip_atomic_store((unsigned long long*)&p.card[CD_REWARD * BN + ci], (unsigned long long)__double_as_longlong(r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&p.caru[CU_TVC * BN + ci], (uint32_t)tv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&p.caru[CU_ONROAD_NEW * BN + ci], onr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      p.card[CD_REWARD * BN + ci] = r;
      p.caru[CU_TVC * BN + ci] = (uint32_t)tv;
      p.caru[CU_ONROAD_NEW * BN + ci] = onr;
    }
  }

  COL_STAMP(5);
  // ---- car<->car manifolds (b2Contact::Update for the dynamic pairs) at the same entry poses.
  // Candidate fixture pairs are enumerated in ascending (carA, fixA, carB, fixB); touching ones are compacted in that
  // order, inherit the stored impulses of the previous step by feature id, and are STORED in island order (below).
  uint32_t* store = p.cc_store + (size_t)env * (MCR_CC_MAX * MCR_CC_WORDS + 4);
  // cheap exit: no pair of car boxes (tight AABB + 0.05) overlaps -> no fixture pair can touch
  unsigned long long pairs = 0ull;                             // bit a * 8 + b (a < b): the two cars' boxes overlap
  if (p.car_contacts && N > 1) {
    const int a = lane >> 3, b = lane & 7;
    pairs = __ballot(a < b && b < N && !(cbox[a][0] > cbox[b][2] || cbox[a][2] < cbox[b][0] || cbox[a][1] > cbox[b][3] || cbox[a][3] < cbox[b][1]));
  }
  const bool any_pair = pairs != 0ull;
  int nn_final = 0;                                            // touching car<->car pairs of this env (wave-uniform)
  if (p.car_contacts && N > 1 && !any_pair) {
    if (lane == 0) { store[0] = 0; store[1] = 0; }
  } else if (p.car_contacts && N > 1) {
    const int old_n = (pass == 1) ? 0 : (int)store[0];
    const McrShapes& S = *p.shapes;
    int base = 0;
    // one trip per pair of cars whose boxes overlap: its 8 x 8 fixture pairs on the 64 lanes (the records are sorted into island
    // order below: the order they are found in does not matter)
    for (unsigned long long pm = pairs; pm; pm &= pm - 1ull) {
      const int ab = __builtin_ctzll(pm), a = ab >> 3, b = ab & 7;
      const int fa = lane >> 3, fb = lane & 7;
      const bool valid = !(fa >= 4 && fb >= 4);                              // wheel (0x20, mask 1) vs wheel: filtered
      cc::Manifold M; M.n = 0; M.type = 0; M.pl[0] = M.pl[1] = v2(0.0f, 0.0f); M.id[0] = M.id[1] = 0; M.localNormal = M.localPoint = v2(0.0f, 0.0f);
      bool flipped = false;
      if (valid) {
        const int ia = a * 8 + fa, ib = b * 8 + fb;
        const float4 A = fbox[ia], Bb = fbox[ib];
        if (!(A.x > Bb.z || A.z < Bb.x || A.y > Bb.w || A.w < Bb.y)) {
          const McrPoly& pa = fa < 4 ? S.hull[fa] : S.wheel; const McrPoly& pb = fb < 4 ? S.hull[fb] : S.wheel;
          Xf xa, xb; const float4 ta = fxf[ia], tb = fxf[ib];
          xa.p = v2(ta.x, ta.y); xa.q.s = ta.z; xa.q.c = ta.w; xb.p = v2(tb.x, tb.y); xb.q.s = tb.z; xb.q.c = tb.w;
          flipped = has_pid && fix_pid(ib) < fix_pid(ia);                  // fixtureA = the fixture with the lower proxy id
          if (flipped) cc::collide_polygons(M, pb, xb, pa, xa); else cc::collide_polygons(M, pa, xa, pb, xb);
        }
      }
      const bool hit = valid && M.n > 0;
      const unsigned long long mask = __ballot(hit);
      if (hit) {
        const int slot = base + __popcll(mask & ((1ull << lane) - 1ull));
        if (slot < MCR_CC_MAX) {
          const uint32_t key = flipped ? ((uint32_t)b | ((uint32_t)fb << 4) | ((uint32_t)a << 8) | ((uint32_t)fa << 12)) : ((uint32_t)a | ((uint32_t)fa << 4) | ((uint32_t)b << 8) | ((uint32_t)fb << 12));
          float ni[2] = {0.0f, 0.0f}, ti[2] = {0.0f, 0.0f};
          for (int j = 0; j < old_n; ++j) {
            const uint32_t* o = store + 4 + j * MCR_CC_WORDS;
            if (o[0] != key) continue;
            const int on = (int)(o[1] >> 8);
#pragma unroll
            for (int i = 0; i < 2; ++i) {                    // fixed trip counts: ni/ti/M.id stay in registers
              if (i >= M.n) continue;
              bool found = false;
#pragma unroll
              for (int jj = 0; jj < 2; ++jj)
                if (!found && jj < on && o[6 + jj * 5 + 4] == M.id[i]) { ni[i] = __uint_as_float(o[6 + jj * 5 + 2]); ti[i] = __uint_as_float(o[6 + jj * 5 + 3]); found = true; }
            }
          }
          uint32_t* d = newrec[slot];
          d[0] = key; d[1] = (uint32_t)M.type | ((uint32_t)M.n << 8);
          d[2] = __float_as_uint(M.localNormal.x); d[3] = __float_as_uint(M.localNormal.y);
          d[4] = __float_as_uint(M.localPoint.x); d[5] = __float_as_uint(M.localPoint.y);
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            d[6 + i * 5 + 0] = __float_as_uint(M.pl[i].x); d[6 + i * 5 + 1] = __float_as_uint(M.pl[i].y);
            d[6 + i * 5 + 2] = __float_as_uint(ni[i]); d[6 + i * 5 + 3] = __float_as_uint(ti[i]); d[6 + i * 5 + 4] = M.id[i];
          }
        }
      }
      base += __popcll(mask);
    }
    __syncthreads();
    const int nn = base < MCR_CC_MAX ? base : MCR_CC_MAX;
    // ---- the order b2Island::Solve runs the touching contacts in (and, per car, its four joints): b2World::Solve builds an island by a
    // depth-first search with an explicit stack — seeds from m_bodyList (its head is the body created LAST: car N-1's wheels 3, 2, 1, 0,
    // its hull, car N-2's ..); for the body it pops it appends the body's touching contacts in contact-edge-list order (head = the
    // contact created last: descending (batch, pa, pb), cc_stamp above) and pushes their other bodies, then appends the body's joints
    // (hull: j3, j2, j1, j0; a wheel: its one joint) and pushes theirs.  Constraints of different islands, and joints of different
    // cars, touch disjoint bodies and commute: what is left is ONE order of the contacts — the records are stored in it, the contact
    // chain of k_dynamics takes them as they come — and an order of the 4 joints per car (store[2..3], 2 bits per position; a car
    // that the search enters through its wheel k solves joint k first).  One lane; the scratch aliases the candidate list.
    // (keys: creation batch, then b2PairLessThan's (lower, higher) proxy id — 64 bits: the batch label counts the episode's steps)
    unsigned long long* const skey = (unsigned long long*)cand;
    uint8_t* const sorted = (uint8_t*)(cand + 2 * MCR_CC_MAX); uint8_t* const dord = sorted + MCR_CC_MAX;     // island order: dord[i] = index of the i-th contact in newrec
    uint8_t* const stk = dord + MCR_CC_MAX;
    static_assert(2 * MCR_CC_MAX * 4 + 2 * MCR_CC_MAX + 5 * MCR_MAX_AGENTS <= CAND_CAP * 4, "the island search's scratch aliases the candidate list");
    if (lane < nn) {
      const uint32_t k = newrec[lane][0];
      const int pa = (int)(k & 15u) * 8 + (int)((k >> 4) & 15u), pb = (int)((k >> 8) & 15u) * 8 + (int)((k >> 12) & 15u);
      const int sa = pa < pb ? pa : pb, sb = pa < pb ? pb : pa;            // (the stamps are kept per pair, lower car first)
      const uint32_t ia = fix_pid(pa), ib = fix_pid(pb);
      skey[lane] = ((unsigned long long)ccs[2 + sa * 8 * N + sb] << 24) | ((unsigned long long)(ia < ib ? ia : ib) << 12) | (unsigned long long)(ia < ib ? ib : ia);
    }
    __syncthreads();
    uint32_t incars = 0u;                                                  // cars with a touching contact (the others: no contacts, joints 3,2,1,0)
    if (lane < nn) {
      const unsigned long long mine = skey[lane];
      int rank = 0;
      for (int j2 = 0; j2 < nn; ++j2) rank += skey[j2] > mine ? 1 : 0;     // (keys are distinct) newest first
      sorted[rank] = (uint8_t)lane;
    }
    for (int i = 0; i < nn; ++i) { const uint32_t k = newrec[i][0]; incars |= (1u << (k & 15u)) | (1u << ((k >> 8) & 15u)); }
    __syncthreads();
    if (lane == 0) {
      unsigned long long bflag = 0ull, jo = 0ull; uint32_t cflag = 0u, jflag = 0u, jn = 0u; int no = 0;
      for (int c = N - 1; c >= 0; --c) for (int kb = 4; kb >= 0; --kb) {
        if (!((incars >> c) & 1u)) { jo |= 0x1bull << (c * 8); break; }       // (seeded at wheel 3: j3, then the hull's j2, j1, j0)
        if ((bflag >> (c * 5 + kb)) & 1ull) continue;
        int sp = 0; stk[sp++] = (uint8_t)(c * 5 + kb); bflag |= 1ull << (c * 5 + kb);
        while (sp > 0) {
          const int b = stk[--sp], bc = b / 5, bk = b - bc * 5;
          for (int t = 0; t < nn; ++t) {
            const int i = sorted[t];
            if ((cflag >> i) & 1u) continue;
            const uint32_t k = newrec[i][0];
            const int fa = (int)((k >> 4) & 15u), fb = (int)((k >> 12) & 15u);
            const int ba = (int)(k & 15u) * 5 + (fa < 4 ? 0 : fa - 3), bb = (int)((k >> 8) & 15u) * 5 + (fb < 4 ? 0 : fb - 3);
            if (ba != b && bb != b) continue;
            dord[no++] = (uint8_t)i; cflag |= 1u << i;
            const int other = ba == b ? bb : ba;
            if (!((bflag >> other) & 1ull)) { stk[sp++] = (uint8_t)other; bflag |= 1ull << other; }
          }
          if (bk == 0) {
            for (int q = 3; q >= 0; --q) if (!((jflag >> (bc * 4 + q)) & 1u)) {
              const uint32_t pos = (jn >> (bc * 4)) & 15u;
              jo |= (unsigned long long)q << (bc * 8 + 2 * pos); jn += 1u << (bc * 4); jflag |= 1u << (bc * 4 + q);
              const int w = bc * 5 + q + 1;
              if (!((bflag >> w) & 1ull)) { stk[sp++] = (uint8_t)w; bflag |= 1ull << w; }
            }
          } else {
            const int q = bk - 1;
            if (!((jflag >> (bc * 4 + q)) & 1u)) {
              const uint32_t pos = (jn >> (bc * 4)) & 15u;
              jo |= (unsigned long long)q << (bc * 8 + 2 * pos); jn += 1u << (bc * 4); jflag |= 1u << (bc * 4 + q);
              const int hb = bc * 5;
              if (!((bflag >> hb) & 1ull)) { stk[sp++] = (uint8_t)hb; bflag |= 1ull << hb; }
            }
          }
        }
      }
      if (p.debug & 16384) {                                              // debug bit 14 (timing experiments): rounds 1-3's order — contacts ascending, joints 3,2,1,0
        for (int i = 0; i < nn; ++i) dord[i] = (uint8_t)i;
        jo = 0x1b1b1b1b1b1b1b1bull;
      }
      store[2] = (uint32_t)jo; store[3] = (uint32_t)(jo >> 32);
    }
    __syncthreads();
    for (int i = lane; i < nn * 16; i += 64) store[4 + (i >> 4) * MCR_CC_WORDS + (i & 15)] = newrec[dord[i >> 4]][i & 15];
    if (lane == 0) { store[0] = (uint32_t)nn; store[1] = base > MCR_CC_MAX ? 1u : 0u; if (base > MCR_CC_MAX) mcr_raise(p, ST_CC_OVERFLOW); }
    nn_final = nn;
  } else if (lane == 0 && pass == 1) store[0] = 0;
  COL_STAMP(6);
  // side-stream partition: envs whose dynamics chain is going to be long (a touching car<->car pair).  cc_mode: the verdict
  // the main launches go by (p.part: mcr_touch_verdict, evaluated by last step's bookkeeping on the same poses) must
  // agree — counters[4] counts disagreements (tests and bench assert 0).
  if (pass == 0 && lane == 0) MCR_TRACE(p, env, p.role == 2 ? 3 : 4, nn_final);                                  // (contact pass by the chain / by k_collide)
  if (pass == 0 && (p.split || p.fuse_collide) && lane == 0) {
    if (nn_final > 0) { if (!p.fuse_collide) p.clist[1 + atomicAdd(&p.clist[0], 1)] = env; atomicAdd(&p.counters[2], 1ull); }     // (fuse_collide: the list was made with the verdicts)
    if (!p.cc_mode) p.part[env] = nn_final > 0 ? 1 : 0;         // the contact pass runs first: it is the one that marks the contact chain's envs
    else if ((nn_final > 0) != (p.part[env] != 0)) {
      atomicAdd(&p.counters[4], 1ull); mcr_raise(p, ST_VERDICT);
      p.counters[6] = (unsigned long long)env | ((unsigned long long)nn_final << 20) | ((unsigned long long)p.part[env] << 28) | ((unsigned long long)p.role << 32) | ((unsigned long long)es.steps << 36);   // (diagnostics: the last mismatch)
      p.counters[7] = (unsigned long long)(uint32_t)mcr_epoch(p);
    }
  }
  if (pass == 0 && p.cc_mode && !((p.debug & 4096) && env == p.env0)) {     // the main dynamics, running beside this launch, may read this env's results now
    // (debug bit 12: env 0's word is withheld — what a starved contact pass looks like to the dynamics; tests)
    // The three result words above left as write-through device-scope stores; "s_waitcnt vmcnt(0)" (all a workgroup-scope release
    // fence costs) holds the epoch word back until they are acknowledged by the memory side, which for write-through stores is the
    // device's coherence point.  An agent-scope RELEASE would in addition write this XCD's whole dirty L2 back (4096 times per
    // launch: 24 -> 146 us, measured) to publish lines nobody reads across XCDs; debug bit 11 selects it for measurements.
    if (p.debug & 2048) {
      __syncthreads();
      if (lane == 0) __hip_atomic_store(&p.collide_epoch[env], mcr_epoch(p), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __syncthreads();
      if (lane == 0) __hip_atomic_store(&p.collide_epoch[env], mcr_epoch(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// one workgroup per env (the list launches of roles >= 2 call collide_block from k_list_chain.h)
// (4 wavefronts per SIMD = 16 per CU: with one wavefront per env the 4096 envs of the bench are resident in ONE round; LDS: see lds_bytes)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_collide(McrParams p, int pass) {
  if ((p.debug & (1 << 18)) && pass == 0 && p.cc_mode) {                  // debug bit 18 (tests): a contact pass that is held up — every workgroup idles ~400 us first
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 40000ull) __builtin_amdgcn_s_sleep(64);
  }
  collide_block(p, pass, (int)blockIdx.x);
}
