// mcr_host.cpp — host half of the product: numpy-compatible MT19937, episode setup (track walk,
// tiles, kerbs, spawn poses) and the car's fixture/mass tables.  No GPU needed for anything here.
//
// Follows  multi_car_racing.py:183-338 (_create_track), :349-406 (reset draws + spawn),
//          gym car_dynamics.py Car.__init__ (fixtures), Box2D b2PolygonShape::{Set,ComputeMass},
//          b2Body::ResetMassData, numpy legacy RandomState (mt19937.c, distributions.c).
// Double-precision trig goes through libm exactly like CPython's `math` module does.
#include "../../include/mcr.h"
#include "mcr_common.h"
#include <cmath>
#include <cstring>
#include <vector>
#include <thread>
#include <atomic>
#include <mutex>
#include <condition_variable>
#include <functional>
#include <sys/prctl.h>
#include <unistd.h>
#include <string>

// ============================================================================ MT19937 (numpy layout)
namespace {
const int MT_N = 624, MT_M = 397;

inline void mt_regen(uint32_t* key) {
  const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MATRIX_A = 0x9908b0dfu;
  int i; uint32_t y;
  for (i = 0; i < MT_N - MT_M; i++) {
    y = (key[i] & UPPER) | (key[i + 1] & LOWER);
    key[i] = key[i + MT_M] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
  }
  for (; i < MT_N - 1; i++) {
    y = (key[i] & UPPER) | (key[i + 1] & LOWER);
    key[i] = key[i + (MT_M - MT_N)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
  }
  y = (key[MT_N - 1] & UPPER) | (key[0] & LOWER);
  key[MT_N - 1] = key[MT_M - 1] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
}
inline uint32_t mt_next32(uint32_t* mt) {
  uint32_t& pos = mt[MT_N];
  if (pos >= (uint32_t)MT_N) { mt_regen(mt); pos = 0; }
  uint32_t y = mt[pos++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}
inline double mt_double(uint32_t* mt) {   // genrand_res53
  int32_t a = mt_next32(mt) >> 5, b = mt_next32(mt) >> 6;
  return (a * 67108864.0 + b) / 9007199254740992.0;
}
inline double mt_uniform(uint32_t* mt, double lo, double hi) { return lo + (hi - lo) * mt_double(mt); }
inline uint32_t mt_interval(uint32_t* mt, uint32_t max) {   // random_interval: masked rejection
  if (max == 0) return 0;
  uint32_t mask = max;
  mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
  uint32_t v;
  while ((v = (mt_next32(mt) & mask)) > max) {}
  return v;
}
}  // namespace

extern "C" void mcr_mt_seed(uint32_t* mt, uint32_t seed) {
  for (int pos = 0; pos < MT_N; pos++) {
    mt[pos] = seed;
    seed = (1812433253u * (seed ^ (seed >> 30)) + pos + 1);
  }
  mt[MT_N] = MT_N;
}
extern "C" void mcr_mt_seed_by_array(uint32_t* mt, const uint32_t* init_key, int key_length) {
  // init_by_array (mt19937ar.c) as used by RandomState.seed(sequence)
  mcr_mt_seed(mt, 19650218u);
  int i = 1, j = 0;
  int k = (MT_N > key_length ? MT_N : key_length);
  for (; k; k--) {
    mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + init_key[j] + j;
    i++; j++;
    if (i >= MT_N) { mt[0] = mt[MT_N - 1]; i = 1; }
    if (j >= key_length) j = 0;
  }
  for (k = MT_N - 1; k; k--) {
    mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - i;
    i++;
    if (i >= MT_N) { mt[0] = mt[MT_N - 1]; i = 1; }
  }
  mt[0] = 0x80000000u;
  mt[MT_N] = MT_N;
}
extern "C" double mcr_mt_random_sample(uint32_t* mt) { return mt_double(mt); }
// np.random.choice(['CW','CCW']) == a[randint(0,2)]: one masked 32-bit draw; index 0 is 'CW'.
extern "C" int mcr_mt_choice_cw(uint32_t* mt) { return mt_interval(mt, 1) == 0 ? 1 : 0; }
// np.random.choice(ids, size=N, replace=False) == permutation(N)[:N]: Fisher-Yates from the top.
extern "C" void mcr_mt_car_order(uint32_t* mt, int n, int32_t* out) {
  for (int i = 0; i < n; ++i) out[i] = i;
  for (int i = n - 1; i >= 1; --i) {
    uint32_t j = mt_interval(mt, (uint32_t)i);
    int32_t tmp = out[i]; out[i] = out[j]; out[j] = tmp;
  }
}

// ============================================================================ polygons / mass (Box2D f32)
namespace {
struct HostPoly { int n; float x[8], y[8], nx[8], ny[8]; };

// b2PolygonShape::Set: weld near-duplicates, gift-wrap CCW from the right-most vertex, edge normals.
bool hull_from_points(const float* px, const float* py, int count, HostPoly& out) {
  float qx[8], qy[8]; int n = 0;
  const float weld2 = (0.5f * B2_LINEAR_SLOP) * (0.5f * B2_LINEAR_SLOP);
  for (int i = 0; i < count && i < 8; ++i) {
    bool dup = false;
    for (int j = 0; j < n; ++j) {
      float dx = px[i] - qx[j], dy = py[i] - qy[j];
      if (dx * dx + dy * dy < weld2) { dup = true; break; }
    }
    if (!dup) { qx[n] = px[i]; qy[n] = py[i]; ++n; }
  }
  if (n < 3) return false;
  int start = 0; float best_x = qx[0];
  for (int i = 1; i < n; ++i)
    if (qx[i] > best_x || (qx[i] == best_x && qy[i] < qy[start])) { start = i; best_x = qx[i]; }
  int order[8]; int m = 0; int cur = start;
  while (true) {
    order[m] = cur;
    int cand = 0;
    for (int j = 1; j < n; ++j) {
      if (cand == cur) { cand = j; continue; }
      float rx = qx[cand] - qx[order[m]], ry = qy[cand] - qy[order[m]];
      float vx = qx[j] - qx[order[m]], vy = qy[j] - qy[order[m]];
      float c = rx * vy - ry * vx;
      if (c < 0.0f) cand = j;
      if (c == 0.0f && (vx * vx + vy * vy) > (rx * rx + ry * ry)) cand = j;
    }
    ++m; cur = cand;
    if (cand == start || m >= 8) break;
  }
  if (m < 3) return false;
  out.n = m;
  for (int i = 0; i < m; ++i) { out.x[i] = qx[order[i]]; out.y[i] = qy[order[i]]; }
  for (int i = 0; i < m; ++i) {
    int k = (i + 1 < m) ? i + 1 : 0;
    float ex = out.x[k] - out.x[i], ey = out.y[k] - out.y[i];
    float nx = 1.0f * ey, ny = -1.0f * ex;                  // b2Cross(edge, 1)
    float len = sqrtf(nx * nx + ny * ny);
    if (len >= B2_EPSILON) { float inv = 1.0f / len; nx *= inv; ny *= inv; }
    out.nx[i] = nx; out.ny[i] = ny;
  }
  return true;
}

// b2PolygonShape::ComputeMass
void polygon_mass(const HostPoly& P, float density, float& mass, float& cx, float& cy, float& inertia) {
  float sx = 0.0f, sy = 0.0f;
  for (int i = 0; i < P.n; ++i) { sx += P.x[i]; sy += P.y[i]; }
  float invn = 1.0f / P.n; sx *= invn; sy *= invn;
  const float third = 1.0f / 3.0f;
  float area = 0.0f, I = 0.0f, ccx = 0.0f, ccy = 0.0f;
  for (int i = 0; i < P.n; ++i) {
    int k = (i + 1 < P.n) ? i + 1 : 0;
    float e1x = P.x[i] - sx, e1y = P.y[i] - sy, e2x = P.x[k] - sx, e2y = P.y[k] - sy;
    float D = e1x * e2y - e1y * e2x;
    float tri = 0.5f * D;
    area += tri;
    float wgt = tri * third;
    ccx += wgt * (e1x + e2x); ccy += wgt * (e1y + e2y);
    float ix = e1x * e1x + e2x * e1x + e2x * e2x;
    float iy = e1y * e1y + e2y * e1y + e2y * e2y;
    I += (0.25f * third * D) * (ix + iy);
  }
  mass = density * area;
  float inva = 1.0f / area; ccx *= inva; ccy *= inva;
  cx = ccx + sx; cy = ccy + sy;
  inertia = density * I;
  inertia += mass * ((cx * cx + cy * cy) - (ccx * ccx + ccy * ccy));
}

const int kWheelPos[4][2] = {{-55, 80}, {55, 80}, {-55, -82}, {55, -82}};
const int kHullCount[4] = {4, 4, 8, 4};
const int kHull[4][8][2] = {
    {{-60, 130}, {60, 130}, {60, 110}, {-60, 110}},
    {{-15, 120}, {15, 120}, {20, 20}, {-20, 20}},
    {{25, 20}, {50, -10}, {50, -40}, {20, -90}, {-20, -90}, {-50, -40}, {-50, -10}, {-25, 20}},
    {{-50, -120}, {50, -120}, {50, -90}, {-50, -90}}};

void to_mcr_poly(const HostPoly& h, McrPoly& p) {
  memset(&p, 0, sizeof(p));
  p.n = h.n;
  for (int i = 0; i < h.n; ++i) { p.vx[i] = h.x[i]; p.vy[i] = h.y[i]; p.nx[i] = h.nx[i]; p.ny[i] = h.ny[i]; }
}
}  // namespace

// gym Car.__init__ fixtures + b2Body::ResetMassData (fixture list is LIFO: last created first).
void mcr_build_shapes(McrShapes* S) {
  memset(S, 0, sizeof(*S));
  HostPoly hp[4], wp;
  for (int k = 0; k < 4; ++k) {
    float x[8], y[8];
    for (int i = 0; i < kHullCount[k]; ++i) { x[i] = (float)(kHull[k][i][0] * MCR_SIZE); y[i] = (float)(kHull[k][i][1] * MCR_SIZE); }
    hull_from_points(x, y, kHullCount[k], hp[k]);
    to_mcr_poly(hp[k], S->hull[k]);
  }
  {
    float x[4] = {(float)(-MCR_WHEEL_W * MCR_SIZE), (float)(MCR_WHEEL_W * MCR_SIZE), (float)(MCR_WHEEL_W * MCR_SIZE), (float)(-MCR_WHEEL_W * MCR_SIZE)};
    float y[4] = {(float)(MCR_WHEEL_R * MCR_SIZE), (float)(MCR_WHEEL_R * MCR_SIZE), (float)(-MCR_WHEEL_R * MCR_SIZE), (float)(-MCR_WHEEL_R * MCR_SIZE)};
    hull_from_points(x, y, 4, wp);
    to_mcr_poly(wp, S->wheel);
  }
  float M = 0.0f, I = 0.0f, lx = 0.0f, ly = 0.0f;
  for (int k = 3; k >= 0; --k) {
    float m, cx, cy, in; polygon_mass(hp[k], 1.0f, m, cx, cy, in);
    M += m; lx += m * cx; ly += m * cy; I += in;
  }
  S->hull_invMass = 1.0f / M; lx *= S->hull_invMass; ly *= S->hull_invMass;
  I -= M * (lx * lx + ly * ly);
  S->hull_invI = 1.0f / I; S->hull_lcx = lx; S->hull_lcy = ly;
  {
    float m, cx, cy, in; polygon_mass(wp, 0.1f, m, cx, cy, in);
    float Mw = 0.0f + m; float wx = (1.0f / Mw) * (m * cx), wy = (1.0f / Mw) * (m * cy);
    float Iw = 0.0f + in; Iw -= Mw * (wx * wx + wy * wy);
    S->wheel_invMass = 1.0f / Mw; S->wheel_invI = 1.0f / Iw;
  }
  for (int k = 0; k < 4; ++k) { S->anchor_x[k] = (float)(kWheelPos[k][0] * MCR_SIZE); S->anchor_y[k] = (float)(kWheelPos[k][1] * MCR_SIZE); }
  // bounding radii for the coarse car<->car test of k_touch.h: hull vertices around the hull's centre of mass, wheel
  // vertices around the wheel's (its origin)
  float hr = 0.0f, wr = 0.0f;
  for (int k = 0; k < 4; ++k)
    for (int i = 0; i < S->hull[k].n; ++i) hr = fmaxf(hr, hypotf(S->hull[k].vx[i] - S->hull_lcx, S->hull[k].vy[i] - S->hull_lcy));
  for (int i = 0; i < S->wheel.n; ++i) wr = fmaxf(wr, hypotf(S->wheel.vx[i], S->wheel.vy[i]));
  S->pad[0] = hr; S->pad[1] = wr;
}

extern "C" void mcr_mass_props(float* out) {
  McrShapes S; mcr_build_shapes(&S);
  out[0] = S.hull_invMass; out[1] = S.hull_invI; out[2] = S.hull_lcx; out[3] = S.hull_lcy; out[4] = S.wheel_invMass; out[5] = S.wheel_invI;
}
extern "C" void mcr_sincos_host(float a, float* s, float* c) { mcr_sincosf(a, s, c); }

// ============================================================================ track
namespace {
const double kTrackRad = 900 / MCR_SCALE;
const double kDetailStep = 21 / MCR_SCALE;
const double kTurnRate = 0.31;
const double kTrackWidth = 40 / MCR_SCALE;
const double kBorder = 8 / MCR_SCALE;
const int kBorderMinCount = 4;
const int kCheckpoints = 12;

struct TrackPt { double alpha, beta, x, y; };
inline double sgn(double v) { return v > 0 ? 1.0 : (v < 0 ? -1.0 : 0.0); }

// One attempt (multi_car_racing.py:183-291).  false == the reference's `return False`.
bool track_attempt(uint32_t* mt, std::vector<TrackPt>& lap) {
  const double PI = M_PI;
  double gate_angle[kCheckpoints], gate_x[kCheckpoints], gate_y[kCheckpoints];
  for (int c = 0; c < kCheckpoints; ++c) {
    double noise = mt_uniform(mt, 0, 2 * PI * 1 / kCheckpoints);
    double rad = mt_uniform(mt, kTrackRad / 3, kTrackRad);
    double alpha = 2 * PI * c / kCheckpoints + noise;
    if (c == 0) { alpha = 0; rad = 1.5 * kTrackRad; }
    if (c == kCheckpoints - 1) { alpha = 2 * PI * c / kCheckpoints; rad = 1.5 * kTrackRad; }
    gate_angle[c] = alpha; gate_x[c] = rad * cos(alpha); gate_y[c] = rad * sin(alpha);
  }
  const double start_alpha = 2 * PI * (-0.5) / kCheckpoints;

  std::vector<TrackPt> path; path.reserve(2600);
  double x = 1.5 * kTrackRad, y = 0, beta = 0;
  long gate = 0; int laps = 0; int budget = 2500; bool below_axis = false;
  double trig_beta = -1.0, trig_c = 0.0, trig_s = 0.0;
  for (;;) {
    double alpha = atan2(y, x);
    if (below_axis && alpha > 0) { ++laps; below_axis = false; }
    if (alpha < 0) { below_axis = true; alpha += 2 * PI; }
    double gate_a, gate_px, gate_py;
    for (;;) {
      bool wrapped = true;
      for (;;) {
        int k = (int)(gate % kCheckpoints);
        gate_a = gate_angle[k]; gate_px = gate_x[k]; gate_py = gate_y[k];
        if (alpha <= gate_a) { wrapped = false; break; }
        ++gate;
        if (gate % kCheckpoints == 0) break;
      }
      if (!wrapped) break;
      alpha -= 2 * PI;
    }
    if (beta != trig_beta) { trig_c = cos(beta); trig_s = sin(beta); trig_beta = beta; }   // (the heading only turns while the walk is off the gate's line: same argument, same libm value)
    double head_x = trig_c, head_y = trig_s;
    double fwd_x = -head_y, fwd_y = head_x;
    double to_cp_x = gate_px - x, to_cp_y = gate_py - y;
    double lateral = head_x * to_cp_x + head_y * to_cp_y;
    while (beta - alpha > 1.5 * PI) beta -= 2 * PI;
    while (beta - alpha < -1.5 * PI) beta += 2 * PI;
    double heading_before = beta;
    lateral *= MCR_SCALE;
    if (lateral > 0.3) beta -= fmin(kTurnRate, fabs(0.001 * lateral));
    if (lateral < -0.3) beta += fmin(kTurnRate, fabs(0.001 * lateral));
    x += fwd_x * kDetailStep;
    y += fwd_y * kDetailStep;
    TrackPt tp = {alpha, heading_before * 0.5 + beta * 0.5, x, y};
    path.push_back(tp);
    if (laps > 4) break;
    if (--budget == 0) break;
  }
  // closed loop between the last two start-line crossings
  int cross_first = -1, cross_last = -1;
  int i = (int)path.size();
  for (;;) {
    --i;
    if (i == 0) return false;
    bool pass = path[i].alpha > start_alpha && path[i - 1].alpha <= start_alpha;
    if (pass && cross_last == -1) cross_last = i;
    else if (pass && cross_first == -1) { cross_first = i; break; }
  }
  if (cross_last - 1 <= cross_first) return false;   // python would produce an empty slice and raise on track[0]
  lap.assign(path.begin() + cross_first, path.begin() + (cross_last - 1));
  double first_heading = lap[0].beta;
  double seam_x = cos(first_heading) * (lap[0].x - lap.back().x), seam_y = sin(first_heading) * (lap[0].y - lap.back().y);
  double seam = sqrt(seam_x * seam_x + seam_y * seam_y);
  if (seam > kDetailStep) return false;
  return true;
}

inline int wrap(int i, int n) { i %= n; return i < 0 ? i + n : i; }
}  // namespace

// The sensor fixture of one tile as the device slot stores it: b2PolygonShape::Set of its 4 points -> tight AABB, hull
// vertices v0 v1 | v2 v3 (CCW), normals n0 n1 | n2 n3, vertex count (3 or 4; a triangle repeats vertex 0 / normal 2).
// (also used by mcr_debug_overlap, mcr_hip.hip)
void mcr_tile_hull(const float* fx, const float* fy, float* aabb4, float* va4, float* vb4, float* na4, float* nb4, int* count) {
  HostPoly hp;
  if (!hull_from_points(fx, fy, 4, hp)) { hp.n = 3; for (int k = 0; k < 3; ++k) { hp.x[k] = fx[k]; hp.y[k] = fy[k]; hp.nx[k] = hp.ny[k] = 0; } }
  float lox = MCR_MAXFLT, loy = MCR_MAXFLT, hix = -MCR_MAXFLT, hiy = -MCR_MAXFLT;
  for (int k = 0; k < hp.n; ++k) { lox = fminf(lox, hp.x[k]); loy = fminf(loy, hp.y[k]); hix = fmaxf(hix, hp.x[k]); hiy = fmaxf(hiy, hp.y[k]); }
  aabb4[0] = lox; aabb4[1] = loy; aabb4[2] = hix; aabb4[3] = hiy;
  if (hp.n == 3) { hp.x[3] = hp.x[0]; hp.y[3] = hp.y[0]; hp.nx[3] = hp.nx[2]; hp.ny[3] = hp.ny[2]; }
  va4[0] = hp.x[0]; va4[1] = hp.y[0]; va4[2] = hp.x[1]; va4[3] = hp.y[1];
  vb4[0] = hp.x[2]; vb4[1] = hp.y[2]; vb4[2] = hp.x[3]; vb4[3] = hp.y[3];
  na4[0] = hp.nx[0]; na4[1] = hp.ny[0]; na4[2] = hp.nx[1]; na4[3] = hp.ny[1];
  nb4[0] = hp.nx[2]; nb4[1] = hp.ny[2]; nb4[2] = hp.nx[3]; nb4[3] = hp.ny[3];
  *count = hp.n;
}

extern "C" size_t mcr_episode_bytes(void) { return MCR_SLOT_BYTES; }

extern "C" int mcr_episode_generate(uint32_t* mt_track, int num_agents, int cw, const int32_t* car_order,
                                    void* blob_out, int32_t* info_out) {
  if (!mt_track || !blob_out || !car_order || num_agents < 1 || num_agents > MCR_MAX_AGENTS) return MCR_ERR_ARG;
  std::vector<TrackPt> lap;
  int retries = 0;
  int T = 0, P = 0;
  std::vector<uint8_t> kerb;
  for (;;) {
    if (track_attempt(mt_track, lap)) {
      T = (int)lap.size();
      // kerb flags (:293-307)
      kerb.assign(T, 0);
      for (int i = 0; i < T; ++i) {
        bool good = true; double oneside = 0;
        for (int neg = 0; neg < kBorderMinCount; ++neg) {
          double b1 = lap[wrap(i - neg, T)].beta, b2 = lap[wrap(i - neg - 1, T)].beta;
          good = good && (fabs(b1 - b2) > kTurnRate * 0.2);
          oneside += sgn(b1 - b2);
        }
        good = good && (fabs(oneside) == kBorderMinCount);
        kerb[i] = good;
      }
      for (int i = 0; i < T; ++i) for (int neg = 0; neg < kBorderMinCount; ++neg) { int j = wrap(i - neg, T); kerb[j] = kerb[j] | kerb[i]; }
      P = T; for (int i = 0; i < T; ++i) P += kerb[i];
      if (T <= MCR_TILE_CAP && P <= MCR_QUAD_CAP && T >= 20) break;   // capacity overflow == failed attempt (never observed)
    }
    ++retries;
  }
  uint8_t* blob = (uint8_t*)blob_out;
  memset(blob, 0, MCR_SLOT_BYTES);
  McrSlotHeader* H = (McrSlotHeader*)(blob + MCR_OFF_HDR);
  H->T = T; H->P = P; H->cw = cw ? 1 : 0;
  double* TX = (double*)(blob + MCR_OFF_TRACK_X); double* TY = (double*)(blob + MCR_OFF_TRACK_Y); double* TB = (double*)(blob + MCR_OFF_TRACK_B);
  double* TAl = (double*)(blob + MCR_OFF_TRACK_A); double* TCs = (double*)(blob + MCR_OFF_TRACK_C); double* TSn = (double*)(blob + MCR_OFF_TRACK_S);
  float* QA = (float*)(blob + MCR_OFF_QA); float* QB = (float*)(blob + MCR_OFF_QB); uint32_t* QM = (uint32_t*)(blob + MCR_OFF_QMETA);
  float* TA = (float*)(blob + MCR_OFF_TAABB); float* VA = (float*)(blob + MCR_OFF_TVA); float* VB = (float*)(blob + MCR_OFF_TVB);
  float* NA = (float*)(blob + MCR_OFF_TNA); float* NB = (float*)(blob + MCR_OFF_TNB); uint32_t* TC = (uint32_t*)(blob + MCR_OFF_TCNT);
  int q = 0;
  // cos / sin of every tile's heading ONCE (the reference evaluates math.cos(beta1), math.cos(beta2) per tile, :300-307: tile i's beta2 is tile
  // i-1's beta1 — the same argument, the same libm value; half of this loop's libm calls, which are what the generator spends its time in)
  for (int i = 0; i < T; ++i) { TCs[i] = cos(lap[i].beta); TSn[i] = sin(lap[i].beta); }
  for (int i = 0; i < T; ++i) {
    const TrackPt& a = lap[i]; const TrackPt& b = lap[wrap(i - 1, T)];
    TX[i] = a.x; TY[i] = a.y; TB[i] = a.beta; TAl[i] = a.alpha;
    const double c1 = TCs[i], s1 = TSn[i], c2 = TCs[wrap(i - 1, T)], s2 = TSn[wrap(i - 1, T)];
    double vx[4] = {a.x - kTrackWidth * c1, a.x + kTrackWidth * c1, b.x + kTrackWidth * c2, b.x - kTrackWidth * c2};
    double vy[4] = {a.y - kTrackWidth * s1, a.y + kTrackWidth * s1, b.y + kTrackWidth * s2, b.y - kTrackWidth * s2};
    float fx[4], fy[4];
    for (int k = 0; k < 4; ++k) { fx[k] = (float)vx[k]; fy[k] = (float)vy[k]; }
    QA[q * 4 + 0] = fx[0]; QA[q * 4 + 1] = fy[0]; QA[q * 4 + 2] = fx[1]; QA[q * 4 + 3] = fy[1];
    QB[q * 4 + 0] = fx[2]; QB[q * 4 + 1] = fy[2]; QB[q * 4 + 2] = fx[3]; QB[q * 4 + 3] = fy[3];
    QM[q] = ((uint32_t)(i + 1) << 8) | (uint32_t)(MCR_COL_ROAD0 + (i % 3));
    ++q;
    // sensor fixture of the tile body (:317-325): b2PolygonShape::Set of the same 4 points
    HostPoly hp;
    mcr_tile_hull(fx, fy, TA + i * 4, VA + i * 4, VB + i * 4, NA + i * 4, NB + i * 4, &hp.n);
    TC[i] = (uint32_t)hp.n | (kerb[i] ? 0x100u : 0u);
    if (kerb[i]) {
      double side = sgn(b.beta - a.beta);
      double w0 = side * kTrackWidth, w1 = side * (kTrackWidth + kBorder);
      double kx[4] = {a.x + w0 * c1, a.x + w1 * c1, b.x + w1 * c2, b.x + w0 * c2};
      double ky[4] = {a.y + w0 * s1, a.y + w1 * s1, b.y + w1 * s2, b.y + w0 * s2};
      QA[q * 4 + 0] = (float)kx[0]; QA[q * 4 + 1] = (float)ky[0]; QA[q * 4 + 2] = (float)kx[1]; QA[q * 4 + 3] = (float)ky[1];
      QB[q * 4 + 0] = (float)kx[2]; QB[q * 4 + 1] = (float)ky[2]; QB[q * 4 + 2] = (float)kx[3]; QB[q * 4 + 3] = (float)ky[3];
      QM[q] = ((uint32_t)(i + 1) << 18) | (uint32_t)((i % 2 == 0) ? MCR_COL_KERB_WHITE : MCR_COL_KERB_RED);   // bits 18..27: owner tile + 1
      ++q;
    }
  }
  // bounding boxes of the runs of MCR_QBLK consecutive road_poly entries (the raster skips the runs a view cannot see)
  {
    float* QK = (float*)(blob + MCR_OFF_QBLK);
    for (int b = 0; b < MCR_QUAD_CAP / MCR_QBLK; ++b) {
      float lox = MCR_MAXFLT, loy = MCR_MAXFLT, hix = -MCR_MAXFLT, hiy = -MCR_MAXFLT;
      for (int e = b * MCR_QBLK; e < (b + 1) * MCR_QBLK && e < P; ++e)
        for (int k = 0; k < 2; ++k) {
          const float* v = (k ? QB : QA) + e * 4;
          lox = fminf(lox, fminf(v[0], v[2])); hix = fmaxf(hix, fmaxf(v[0], v[2]));
          loy = fminf(loy, fminf(v[1], v[3])); hiy = fmaxf(hiy, fmaxf(v[1], v[3]));
        }
      QK[b * 4 + 0] = lox; QK[b * 4 + 1] = loy; QK[b * 4 + 2] = hix; QK[b * 4 + 3] = hiy;
    }
  }
  // ... and of the runs of MCR_TBLK consecutive tile sensor AABBs (the contact pass skips the runs no car is near)
  {
    float* TK = (float*)(blob + MCR_OFF_TBLK);
    for (int b = 0; b < MCR_TILE_CAP / MCR_TBLK; ++b) {
      float lox = MCR_MAXFLT, loy = MCR_MAXFLT, hix = -MCR_MAXFLT, hiy = -MCR_MAXFLT;
      for (int t = b * MCR_TBLK; t < (b + 1) * MCR_TBLK && t < T; ++t) {
        lox = fminf(lox, TA[t * 4 + 0]); loy = fminf(loy, TA[t * 4 + 1]); hix = fmaxf(hix, TA[t * 4 + 2]); hiy = fmaxf(hiy, TA[t * 4 + 3]);
      }
      TK[b * 4 + 0] = lox; TK[b * 4 + 1] = loy; TK[b * 4 + 2] = hix; TK[b * 4 + 3] = hiy;
    }
  }
  // spawn poses (:366-406)
  const double pos_x = lap[0].x, pos_y = lap[0].y;
  for (int car = 0; car < num_agents; ++car) {
    int ord = car_order[car];
    int line = (int)floor(ord / 2.0);
    int side = 2 * (ord % 2) - 1;
    const TrackPt& ref = lap[wrap(-line * 5, T)];
    double dx = ref.x - pos_x, dy = ref.y - pos_y;
    double angle = ref.beta;
    if (cw) angle -= M_PI;
    double nt = angle - M_PI / 2;
    H->spawn[car][0] = angle;
    H->spawn[car][1] = pos_x + dx + (3 * sin(nt) * side);
    H->spawn[car][2] = pos_y + dy + (3 * cos(nt) * side);
  }
  if (info_out) { info_out[0] = T; info_out[1] = P; info_out[2] = retries; info_out[3] = cw ? 1 : 0; }
  return MCR_OK;
}

// Helper threads of mcr_episodes_generate, kept between calls: a batch is a handful of tracks (the refill thread asks for the
// ~4 episodes that ended in a step, every step) at 0.07 ms each, and starting a thread costs about as much as generating one —
// with threads started per call the second generator thread of a 2-core host share did a third of the work instead of half.
// One job at a time (a second caller that finds the pool busy generates on its own thread).
namespace {
struct GenPool {
  std::mutex m, run_m;
  std::condition_variable cv_work, cv_done;
  std::vector<std::thread> threads;
  const std::function<void()>* job = nullptr;
  uint64_t seq = 0;
  int want = 0, started = 0, finished = 0;
  bool stop = false;
  void worker() {
    (void)prctl(PR_SET_NAME, "mcr-gen", 0, 0, 0);
    uint64_t seen = 0;
    std::unique_lock<std::mutex> lk(m);
    for (;;) {
      cv_work.wait(lk, [&] { return stop || (seq != seen && job && started < want); });
      if (stop) return;
      seen = seq;
      ++started;
      const std::function<void()>* fn = job;
      lk.unlock();
      (*fn)();
      lk.lock();
      if (++finished == want) cv_done.notify_all();
    }
  }
  // runs fn on the caller and on `helpers` threads; returns when all of them are through (also when the caller's fn throws: the helpers
  // hold a pointer to it).  A caller that finds the pool busy — reset() of a whole batch while the refill thread holds it, several
  // VecEnvs in one process — starts threads of its own for the call, as round 2 did for every call.
  void run(int helpers, const std::function<void()>& fn) {
    if (helpers <= 0) { fn(); return; }
    std::unique_lock<std::mutex> one(run_m, std::try_to_lock);
    if (!one.owns_lock()) {
      std::vector<std::thread> own;
      for (int i = 0; i < helpers; ++i) own.emplace_back(fn);
      struct Join { std::vector<std::thread>& t; ~Join() { for (auto& x : t) if (x.joinable()) x.join(); } } join{own};
      fn();
      return;
    }
    {
      std::lock_guard<std::mutex> lk(m);
      while ((int)threads.size() < helpers) threads.emplace_back([this] { worker(); });
      job = &fn; ++seq; want = helpers; started = finished = 0;
    }
    cv_work.notify_all();
    struct Wait { GenPool& p; ~Wait() { std::unique_lock<std::mutex> lk(p.m); p.cv_done.wait(lk, [&] { return p.finished == p.want; }); p.job = nullptr; } } wait{*this};
    fn();
  }
  ~GenPool() {
    { std::lock_guard<std::mutex> lk(m); stop = true; }
    cv_work.notify_all();
    for (auto& t : threads) if (t.joinable()) t.join();
  }
};
// (never destroyed: its threads may outlive main() in an embedding process.  Not inherited either: in a fork()ed child the pool object
// says it has threads that do not exist there — run() would wait for them for ever —, so the child starts a pool of its own)
GenPool& gen_pool() {
  static std::mutex gm; static GenPool* p = nullptr; static pid_t owner = 0;
  std::lock_guard<std::mutex> lk(gm);
  if (!p || owner != getpid()) { p = new GenPool(); owner = getpid(); }     // (the parent's pool object is leaked in the child: its mutexes may be held by threads that are gone)
  return *p;
}
}  // namespace

extern "C" int mcr_episodes_generate(uint32_t* mt_track, uint32_t* mt_global, int n, int num_agents, int direction_mode,
                                     void* blobs_out, int32_t* info_out, int num_threads) {
  if (!mt_track || !mt_global || !blobs_out || n < 0) return MCR_ERR_ARG;
  if (num_threads < 1) num_threads = 1;
  if (num_threads > n) num_threads = n > 0 ? n : 1;
  std::atomic<int> next(0), err(0);
  auto work = [&]() {
    for (;;) {
      int i = next.fetch_add(1);
      if (i >= n) break;
      uint32_t* g = mt_global + (size_t)i * MCR_MT_WORDS;
      int cw = direction_mode == 1;
      if (direction_mode == 2) cw = mcr_mt_choice_cw(g);              // :351-352
      int32_t order[MCR_MAX_AGENTS];
      mcr_mt_car_order(g, num_agents, order);                          // :355-357
      int32_t info[4];
      int rc = mcr_episode_generate(mt_track + (size_t)i * MCR_MT_WORDS, num_agents, cw, order,
                                    (uint8_t*)blobs_out + (size_t)i * MCR_SLOT_BYTES, info);
      if (rc != MCR_OK) err.store(rc);
      if (info_out) { int32_t* o = info_out + (size_t)i * 12; for (int k = 0; k < 4; ++k) o[k] = info[k]; for (int k = 0; k < 8; ++k) o[4 + k] = k < num_agents ? order[k] : -1; }
    }
  };
  const std::function<void()> job = work;
  gen_pool().run(num_threads - 1, job);
  return err.load();
}

// the same for rows ids[0..n) of PER-ENV arrays, in place: what the refill service (mcr_hip.hip) asks for — no gather / scatter of RNG states
extern "C" int mcr_episodes_generate_rows(uint32_t* mt_track_all, uint32_t* mt_global_all, const int32_t* ids, int n, int num_agents,
                                          int direction_mode, void* blobs_all, int32_t* info_all, int num_threads) {
  if (!mt_track_all || !mt_global_all || !blobs_all || !ids || n < 0) return MCR_ERR_ARG;
  if (num_threads < 1) num_threads = 1;
  if (num_threads > n) num_threads = n > 0 ? n : 1;
  std::atomic<int> next(0), err(0);
  auto work = [&]() {
    for (;;) {
      const int i = next.fetch_add(1);
      if (i >= n) break;
      const size_t e = (size_t)ids[i];
      uint32_t* g = mt_global_all + e * MCR_MT_WORDS;
      int cw = direction_mode == 1;
      if (direction_mode == 2) cw = mcr_mt_choice_cw(g);              // :351-352
      int32_t order[MCR_MAX_AGENTS];
      mcr_mt_car_order(g, num_agents, order);                          // :355-357
      int32_t info[4];
      const int rc = mcr_episode_generate(mt_track_all + e * MCR_MT_WORDS, num_agents, cw, order, (uint8_t*)blobs_all + e * MCR_SLOT_BYTES, info);
      if (rc != MCR_OK) err.store(rc);
      if (info_all) { int32_t* o = info_all + e * 12; for (int k = 0; k < 4; ++k) o[k] = info[k]; for (int k = 0; k < 8; ++k) o[4 + k] = k < num_agents ? order[k] : -1; }
    }
  };
  const std::function<void()> job = work;
  gen_pool().run(num_threads - 1, job);
  return err.load();
}

extern "C" int mcr_episode_unpack(const void* blob_in, int32_t* T, int32_t* P, int32_t* cw, double* track_xyb,
                                  float* quads, uint32_t* quad_meta, double* spawn, double* track_alpha) {
  if (!blob_in) return MCR_ERR_ARG;
  const uint8_t* blob = (const uint8_t*)blob_in;
  const McrSlotHeader* H = (const McrSlotHeader*)blob;
  if (T) *T = H->T; if (P) *P = H->P; if (cw) *cw = H->cw;
  const double* TX = (const double*)(blob + MCR_OFF_TRACK_X); const double* TY = (const double*)(blob + MCR_OFF_TRACK_Y); const double* TB = (const double*)(blob + MCR_OFF_TRACK_B);
  if (track_alpha) { const double* TAl = (const double*)(blob + MCR_OFF_TRACK_A); for (int i = 0; i < H->T; ++i) track_alpha[i] = TAl[i]; }
  if (track_xyb) for (int i = 0; i < H->T; ++i) { track_xyb[i * 3] = TX[i]; track_xyb[i * 3 + 1] = TY[i]; track_xyb[i * 3 + 2] = TB[i]; }
  const float* QA = (const float*)(blob + MCR_OFF_QA); const float* QB = (const float*)(blob + MCR_OFF_QB); const uint32_t* QM = (const uint32_t*)(blob + MCR_OFF_QMETA);
  if (quads) for (int i = 0; i < H->P; ++i) { for (int k = 0; k < 4; ++k) { quads[i * 8 + k] = QA[i * 4 + k]; quads[i * 8 + 4 + k] = QB[i * 4 + k]; } }
  if (quad_meta) for (int i = 0; i < H->P; ++i) quad_meta[i] = QM[i];
  if (spawn) memcpy(spawn, H->spawn, sizeof(H->spawn));
  return MCR_OK;
}

// host twin of mcr_synth_actions (same counter-based stream; the CPU baseline and the tests consume it)
extern "C" void mcr_synth_actions_host(float* out, int num_envs, int num_agents, uint64_t seed, uint32_t t, uint32_t env_offset) {
  for (int e = 0; e < num_envs; ++e)
    for (int a = 0; a < num_agents; ++a) mcr_synth_action(seed, env_offset + (uint32_t)e, (uint32_t)a, t, out + ((size_t)e * num_agents + a) * 3);
}
