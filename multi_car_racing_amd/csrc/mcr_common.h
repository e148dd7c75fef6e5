// mcr_common.h — constants, HBM layouts and small math shared by host code and gfx950 kernels.
//
// Numeric contract: physics state is IEEE binary32 exactly as Box2D keeps it,
// Car.step arithmetic is binary64 exactly as CPython evaluates it; every translation unit is built with
// -ffp-contract=off so that host and device round identically.  sinf/cosf are DEFINED by mcr_sincosf
// below (f64 Cody-Waite reduction + fdlibm minimax kernels, rounded once to f32).
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MCR_HD __host__ __device__ __forceinline__
#else
#define MCR_HD inline
#endif

#define MCR_MAX_AGENTS 8
#define MCR_TILE_CAP 512
#define MCR_QUAD_CAP 768

// ------------------------------------------------------------------ reference constants
// multi_car_racing.py:43-78
#define MCR_SCALE 6.0
#define MCR_PLAYFIELD (2000 / 6.0)
#define MCR_FPS 50
#define MCR_ZOOM 2.7
#define MCR_WINDOW_W 1000
#define MCR_WINDOW_H 800
// gym car_dynamics.py
#define MCR_SIZE 0.02
#define MCR_WHEEL_R 27
#define MCR_WHEEL_W 14
// Box2D b2Settings.h
#define B2_PI 3.14159265359f
#define B2_EPSILON 1.1920928955078125e-07f
#define B2_LINEAR_SLOP 0.005f
#define B2_ANGULAR_SLOP (2.0f / 180.0f * B2_PI)
#define B2_POLYGON_RADIUS (2.0f * B2_LINEAR_SLOP)
#define B2_MAX_LINEAR_CORRECTION 0.2f
#define B2_MAX_ANGULAR_CORRECTION (8.0f / 180.0f * B2_PI)
#define B2_MAX_TRANSLATION 2.0f
#define B2_MAX_ROTATION (0.5f * B2_PI)
#define B2_BAUMGARTE 0.2f
#define B2_TIME_TO_SLEEP 0.5f
#define B2_LINEAR_SLEEP_TOL 0.01f
#define B2_ANGULAR_SLEEP_TOL (2.0f / 180.0f * B2_PI)
#define MCR_MAXFLT 3.402823466e+38f

// ------------------------------------------------------------------ episode slot (host blob == device image)
// One env owns two slots (current / staged).  Arrays are capacity-strided so that a wave reading
// consecutive tiles/quads issues fully coalesced 16-byte-per-lane loads.
struct McrSlotHeader {
  int32_t T, P, cw, pad0;            // pad0: 1 = the slot carries proxy-id tables (MCR_OFF_TPID / MCR_OFF_FPID)
  double spawn[MCR_MAX_AGENTS][3];   // (angle, x, y) per car id, as handed to Car(...)
  int32_t pad1[12];
};
#define MCR_OFF_HDR 0
#define MCR_OFF_TRACK_X 256                                   // f64 [TILE_CAP]
#define MCR_OFF_TRACK_Y (MCR_OFF_TRACK_X + 8 * MCR_TILE_CAP)
#define MCR_OFF_TRACK_B (MCR_OFF_TRACK_Y + 8 * MCR_TILE_CAP)
#define MCR_OFF_TRACK_A (MCR_OFF_TRACK_B + 8 * MCR_TILE_CAP)  // f64 [TILE_CAP] alpha (facade attribute env.track only)
#define MCR_OFF_TRACK_C (MCR_OFF_TRACK_A + 8 * MCR_TILE_CAP)  // f64 [TILE_CAP] cos(beta) as libm computed it on the host
#define MCR_OFF_TRACK_S (MCR_OFF_TRACK_C + 8 * MCR_TILE_CAP)  // f64 [TILE_CAP] sin(beta)  (on-grass test in f64, :470-472)
#define MCR_OFF_QA (MCR_OFF_TRACK_S + 8 * MCR_TILE_CAP)       // float4 [QUAD_CAP]  x0 y0 x1 y1
#define MCR_OFF_QB (MCR_OFF_QA + 16 * MCR_QUAD_CAP)           // float4 [QUAD_CAP]  x2 y2 x3 y3
#define MCR_OFF_QMETA (MCR_OFF_QB + 16 * MCR_QUAD_CAP)        // u32    [QUAD_CAP]  colour id | (tile+1)<<8 for tile quads | (owner tile+1)<<18 for kerb quads
#define MCR_OFF_TAABB (MCR_OFF_QMETA + 4 * MCR_QUAD_CAP)      // float4 [TILE_CAP]  lo.xy hi.xy
#define MCR_OFF_TVA (MCR_OFF_TAABB + 16 * MCR_TILE_CAP)       // float4 [TILE_CAP]  hull v0 v1 (CCW)
#define MCR_OFF_TVB (MCR_OFF_TVA + 16 * MCR_TILE_CAP)         // float4             v2 v3
#define MCR_OFF_TNA (MCR_OFF_TVB + 16 * MCR_TILE_CAP)         // float4             n0 n1
#define MCR_OFF_TNB (MCR_OFF_TNA + 16 * MCR_TILE_CAP)         // float4             n2 n3
#define MCR_OFF_TCNT (MCR_OFF_TNB + 16 * MCR_TILE_CAP)        // u32    [TILE_CAP]  hull vertex count (3|4) | kerb<<8
#define MCR_QBLK 16                                           // road_poly entries per culling block of the raster
#define MCR_OFF_QBLK (MCR_OFF_TCNT + 4 * MCR_TILE_CAP)        // float4 [QUAD_CAP / QBLK]  lo.xy hi.xy of each run of QBLK road_poly entries (empty: lo > hi)
#define MCR_TBLK 16                                           // tiles per culling block of the contact pass
#define MCR_OFF_TBLK (MCR_OFF_QBLK + 16 * (MCR_QUAD_CAP / MCR_QBLK))   // float4 [TILE_CAP / TBLK]  lo.xy hi.xy of each run of TBLK tile sensor AABBs (empty: lo > hi)
// broadphase proxy ids of the episode's fixtures (header.pad0 != 0: present — a caller's literal tree, mcr_world.cpp, honoured by fresh_world = 1 handles; 0: the ids of a
// fresh world, ascending in creation order): what orders the contact callbacks of a step and names fixtureA of a pair (k_collide.h)
#define MCR_OFF_TPID (MCR_OFF_TBLK + 16 * (MCR_TILE_CAP / MCR_TBLK))  // u16 [TILE_CAP]          tile t
#define MCR_OFF_FPID (MCR_OFF_TPID + 2 * MCR_TILE_CAP)                // u16 [MAX_AGENTS * 8]    car * 8 + fixture
#define MCR_PID_LIMIT 4096                                            // ids are node indices of the tree: < 2 * (TILE_CAP + 64) rounded up to its pool size
#define MCR_SLOT_BYTES (MCR_OFF_FPID + 2 * MCR_MAX_AGENTS * 8)
// One b2World per env for the env's life (k_world.h): the live episode's proxy ids [MCR_PID_TAB] u16 (laid out like the slot's TPID | FPID
// region), the LIFO of free LEAF ids of the world's tree [MCR_PID_STACK] u16 and {stack height, fresh leaves handed out, tiles of the live
// episode (0: no episode yet), spare} [4] i32
#define MCR_PID_TAB (MCR_TILE_CAP + MCR_MAX_AGENTS * 8)
#define MCR_PID_STACK MCR_PID_TAB

// quad colour ids (u8 RGB after the GL float->unorm8 conversion)
enum { MCR_COL_ROAD0 = 0, MCR_COL_ROAD1 = 1, MCR_COL_ROAD2 = 2, MCR_COL_KERB_WHITE = 3, MCR_COL_KERB_RED = 4 };

// ------------------------------------------------------------------ per-car SoA field indices
// f32 fields  carf[field][B*N]
enum {
  CF_CX = 0,   // +body (0 hull, 1..4 wheels)
  CF_CY = 5, CF_A = 10, CF_VX = 15, CF_VY = 20, CF_W = 25,
  CF_JIX = 30, // +joint
  CF_JIY = 34, CF_JIZ = 38, CF_JM = 42,
  CF_SLEEP = 46,  // +body
  CF_COUNT = 51
};
// f64 fields  card[field][B*N]
enum {
  CD_GAS = 0,    // +0,1 rear-left, rear-right (wheels 2,3)
  CD_STEER = 2,  // front wheels share the target
  CD_BRAKE = 3,
  CD_OMEGA = 4,  // +wheel
  CD_PHASE = 8,  // +wheel
  CD_REWARD = 12, CD_PREV_REWARD = 13,
  CD_EPRET = 14,                 // sum of the step rewards handed out in the running episode (episode statistics)
  CD_COUNT = 15
};
// u32 fields  caru[field][B*N]
enum {
  CU_LIMIT = 0,       // 2 bits per joint
  CU_ONROAD = 1,      // bits 0..3: wheel k has a tile under it, as of the Collide of the step that ENDED last = what Car.step reads
                      // (len(w.tiles) > 0: the listener's sets change inside world.Step, after Car.step)
  CU_TVC = 2,         // tile_visited_count
  CU_FLAGS = 3,       // bit0 driving_backward, bit1 driving_on_grass
  CU_ONROAD_NEW = 4,  // the same bits as this step's Collide found them (k_collide writes, the step's k_dynamics promotes them to CU_ONROAD)
  CU_COUNT = 5
};
// broadphase model of the car fixtures (k_collide.h): float4 planes  bpf[plane][MCR_BP_FIX * BN]  (car * 8 + fixture: 0..3 hull polygons, 4..7 wheels)
#define MCR_BP_FIX 8
enum {
  BP_FAT = 0,    // fat AABB of the fixture's proxy (lo.x lo.y hi.x hi.y)
  BP_PREV = 1,   // the fixture's AABB at the transform the last contact pass saw (b2PolygonShape::ComputeAABB)
  BP_PREVP = 2,  // .xy: that transform's p
  BP_COUNT = 3
};
// car<->car broadphase contacts of an env (k_collide.h): words [0,1] = which car pairs hold any (bit carA * 8 + carB), then per fixture
// pair (pa, pb), pa = carA * 8 + fixA < pb: 0 = no contact (the fat AABBs do not overlap), else 1 + the label of the FindNewContacts
// batch that made it
MCR_HD size_t mcr_cc_stamp_words(int N) { return 2 + (size_t)64 * N * N; }
// per-env state
struct McrEnvState {
  double t;                // self.t
  int32_t steps;           // TimeLimit counter
  int32_t slot;            // current episode slot (0/1); the other one is the staged slot
  int32_t staged_ready;    // staged slot holds a fresh episode
  int32_t consumed;        // staged episode was installed since the last poll
  int32_t active;          // 0 until the first reset
  int32_t resetting;       // episode installed; the action-less step of reset() (:408) is still pending
  int32_t just_reset;      // the obs being produced is a first observation (a7 bookkeeping is skipped, :435)
  int32_t frozen;          // auto-reset found no staged episode when this env finished: inactive until the host stages one (then it thaws)
  uint32_t touch_blocks;   // bit b: some tile of block b (MCR_TBLK tiles) had a wheel on it after the last contact pass (k_collide looks at those + the ones near a car)
  uint32_t bp_step;        // contact passes this episode has seen (the reset pass is 0): labels the batches of the broadphase model (k_collide.h), owned by k_collide
};

// fixtures of one car in body-local coordinates (host builds them with its b2PolygonShape::Set
// restatement; kernels receive a pointer)
struct McrPoly { int32_t n; int32_t pad; float vx[8], vy[8], nx[8], ny[8]; };
struct McrShapes {
  McrPoly hull[4];
  McrPoly wheel;
  float hull_invMass, hull_invI, hull_lcx, hull_lcy;
  float wheel_invMass, wheel_invI;
  float anchor_x[4], anchor_y[4];     // revolute joint localAnchorA
  float pad[2];                       // [0] hull radius around its centre of mass, [1] wheel radius (coarse car<->car test, k_touch.h)
};

// ------------------------------------------------------------------ synthetic action stream (bench / tests)
// Counter-based: the action of (global env g, agent a) at step t is a pure function of (seed, g, a, t) — the same
// on the device, on the host and whatever the batch or world size.  splitmix64 finaliser, 24-bit uniforms.
MCR_HD uint64_t mcr_mix64(uint64_t x) {
  x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31;
  return x;
}
MCR_HD float mcr_synth_uniform(uint64_t seed, uint32_t g, uint32_t agent, uint32_t t, uint32_t comp) {
  const uint64_t ctr = ((uint64_t)t << 32) | ((uint64_t)g * 8u + agent);
  const uint64_t x = mcr_mix64(mcr_mix64(seed + 0x9e3779b97f4a7c15ull * ctr) + comp);
  return (float)(x >> 40) * (1.0f / 16777216.0f);
}
// (steer ~ U(-1,1), gas ~ U(0,1), brake ~ U(0,1)): the action_space bounds of multi_car_racing.py:162-165
MCR_HD void mcr_synth_action(uint64_t seed, uint32_t g, uint32_t agent, uint32_t t, float* a3) {
  a3[0] = mcr_synth_uniform(seed, g, agent, t, 0) * 2.0f - 1.0f;
  a3[1] = mcr_synth_uniform(seed, g, agent, t, 1);
  a3[2] = mcr_synth_uniform(seed, g, agent, t, 2);
}

// ------------------------------------------------------------------ math
struct V2 { float x, y; };
MCR_HD V2 v2(float x, float y) { V2 r; r.x = x; r.y = y; return r; }
MCR_HD V2 operator+(V2 a, V2 b) { return v2(a.x + b.x, a.y + b.y); }
MCR_HD V2 operator-(V2 a, V2 b) { return v2(a.x - b.x, a.y - b.y); }
MCR_HD V2 operator-(V2 a) { return v2(-a.x, -a.y); }
MCR_HD V2 operator*(float s, V2 a) { return v2(s * a.x, s * a.y); }
MCR_HD float dot(V2 a, V2 b) { return a.x * b.x + a.y * b.y; }
MCR_HD float cross(V2 a, V2 b) { return a.x * b.y - a.y * b.x; }
MCR_HD V2 cross(V2 a, float s) { return v2(s * a.y, -s * a.x); }
MCR_HD V2 cross(float s, V2 a) { return v2(-s * a.y, s * a.x); }
MCR_HD float mcr_min(float a, float b) { return a < b ? a : b; }   // std::min semantics
MCR_HD float mcr_max(float a, float b) { return a < b ? b : a; }   // std::max semantics
MCR_HD float mcr_clamp(float a, float lo, float hi) { return mcr_max(lo, mcr_min(a, hi)); }
MCR_HD float length(V2 a) { return sqrtf(a.x * a.x + a.y * a.y); }

// The coefficients: literals on the host; on the device words of constant memory (not `const`: the compiler must not fold them back into
// literals) — scalar loads, hoisted out of the sweeps' loops, whose SGPR pairs v_fma_f64 takes as its addend directly.  A 64-bit literal
// cannot be an operand: each Horner step was a v_mov_b64 of its coefficient plus a v_fmac_f64 — 40 moves per position sweep.
#define MCR_SINCOS_COEFFS { 1.58969099521155010221e-10, -2.50507602534068634195e-08, 2.75573137070700676789e-06, -1.98412698298579493134e-04, \
  8.33333333332248946124e-03, -1.66666666666666324348e-01, -1.13596475577881948265e-11, 2.08757232129817482790e-09, -2.75573143513906633035e-07, \
  2.48015872894767294178e-05, -1.38888888888741095749e-03, 4.16666666666666019037e-02, 6.07710050650619224932e-11, 1.57079632673412561417e+00 }
#if defined(__HIP_DEVICE_COMPILE__)
static __constant__ double MCR_SINCOS_K[14] = MCR_SINCOS_COEFFS;
#else
static const double MCR_SINCOS_K[14] = MCR_SINCOS_COEFFS;
#endif
// f64 sin/cos core: 2-constant Cody-Waite reduction + fdlibm minimax kernels, Horner with fused multiply-adds (as mcr_sincosf below);
// absolute error ~1e-16.  Branch-free and ~25 f64 operations: also used where the device would otherwise call the
// general-purpose libm sin/cos (camera rotation, wheel stripe phases), whose results only reach pixels.
MCR_HD void mcr_sincos_core(double x, double* s, double* c) {
  const double* K = MCR_SINCOS_K;
  const double fn = rint(x * 6.36619772367581382433e-01);
  const int n = (int)fn;
  const double r = fma(-fn, K[12], fma(-fn, K[13], x));
  const double z = r * r;
  const double S = fma(z, fma(z, fma(z, fma(z, fma(z, K[0], K[1]), K[2]), K[3]), K[4]), K[5]);
  const double C = fma(z, fma(z, fma(z, fma(z, fma(z, K[6], K[7]), K[8]), K[9]), K[10]), K[11]);
  const double ps = fma(r * z, S, r);
  const double pc = fma(z * z, C, fma(-0.5, z, 1.0));
  switch (n & 3) {
    case 0: *s = ps; *c = pc; break;
    case 1: *s = pc; *c = -ps; break;
    case 2: *s = -ps; *c = -pc; break;
    default: *s = -pc; *c = ps; break;
  }
}
// sinf/cosf spec of the build: f64 Cody-Waite reduction + fdlibm's kernel polynomials in Horner form with fused multiply-adds, rounded
// once to f32 — bit-identical on host (x86-64: fma() is exact whether the machine has the instruction or libm emulates it) and gfx950.
// (the quadrant's swap and signs are applied AFTER the rounding — rounding is symmetric, so (float)(-x) == -(float)x —: two selects and
// two sign flips on f32 values instead of masked swaps of f64 pairs; this function sits on the serial chain of the position sweeps)
MCR_HD void mcr_sincosf(float a, float* s, float* c) {
  const double* K = MCR_SINCOS_K;
  const double x = (double)a;
  const double fn = rint(x * 6.36619772367581382433e-01);
  const int n = (int)fn;
  // fused multiply-adds throughout (explicit: contraction stays off for everything else): 20 f64 operations instead of 34
  const double r = fma(-fn, K[12], fma(-fn, K[13], x));
  const double z = r * r;
  const double S = fma(z, fma(z, fma(z, fma(z, fma(z, K[0], K[1]), K[2]), K[3]), K[4]), K[5]);
  const double C = fma(z, fma(z, fma(z, fma(z, fma(z, K[6], K[7]), K[8]), K[9]), K[10]), K[11]);
  const double ps = fma(r * z, S, r);
  const double pc = fma(z * z, C, fma(-0.5, z, 1.0));
  const float fs = (float)ps, fc = (float)pc;
  const bool odd = (n & 1) != 0;
  const float ms = odd ? fc : fs, mc = odd ? fs : fc;                 // |sin|-side and |cos|-side magnitudes with their own signs
  // quadrant 0: (s, c); 1: (c, -s); 2: (-s, -c); 3: (-c, s)
  *s = (n & 2) ? -ms : ms;
  *c = ((n + 1) & 2) ? -mc : mc;
}
struct Rot { float s, c; };
MCR_HD Rot rot_of(float a) { Rot q; mcr_sincosf(a, &q.s, &q.c); return q; }
MCR_HD V2 rmul(Rot q, V2 v) { return v2(q.c * v.x - q.s * v.y, q.s * v.x + q.c * v.y); }
MCR_HD V2 rmulT(Rot q, V2 v) { return v2(q.c * v.x + q.s * v.y, -q.s * v.x + q.c * v.y); }
struct Xf { V2 p; Rot q; };
MCR_HD V2 xmul(const Xf& t, V2 v) { return v2((t.q.c * v.x - t.q.s * v.y) + t.p.x, (t.q.s * v.x + t.q.c * v.y) + t.p.y); }
MCR_HD V2 xmulT(const Xf& t, V2 v) { float px = v.x - t.p.x, py = v.y - t.p.y; return v2(t.q.c * px + t.q.s * py, -t.q.s * px + t.q.c * py); }
// b2Body::SynchronizeTransform from (sweep.c, sweep.a, localCenter)
MCR_HD Xf xf_of(V2 c, float a, V2 lc) { Xf t; t.q = rot_of(a); t.p = c - rmul(t.q, lc); return t; }
