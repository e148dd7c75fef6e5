// k_raster_common.h — what the two rasterisers (k_view.h: the 96x96 observation path, k_render.h: render(mode) at
// any viewport) share: palette, draw-order keys, convex-polygon helpers, the score label and the f64 point-in-quad of
// the on-grass bookkeeping.  Replaces pyglet/OpenGL state of multi_car_racing.py:511-604, 613-674 and gym Car.draw.
#pragma once
#include "mcr_kernels.h"

namespace view {

#define VIEW_THREADS 256
#define CAR_KEY 1024                 // draw indices >= CAR_KEY are car polygons (drawn after every road_poly entry)
#define CARPOLY_CAP (MCR_MAX_AGENTS * 12)

// palette (one index per pixel until the write-out)
enum { PAL_BLACK = 0, PAL_GRASS0, PAL_GRASS1, PAL_ROAD0, PAL_ROAD1, PAL_ROAD2, PAL_WHITE, PAL_RED255, PAL_WHEELWHITE,
       PAL_CAR0, PAL_BLUE255 = PAL_CAR0 + 8, PAL_PURPLE, PAL_GREEN255, PAL_MUD, PAL_COUNT };

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
    case PAL_MUD: return rgb(c8(0.4), c8(0.4), 0);           // gym car_dynamics MUD_COLOR (skid particles on grass; on road: WHEEL_COLOR = black)
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

// the same palette as a table (GL float colour -> unorm8 evaluated at compile time: 0.4 -> 102, 0.8 -> 204, 0.9 -> 229 (the f32
// value of 0.9 is just below), 0.41 -> 105, 0.42 -> 107, 0.3 -> 77, 0.2 -> 51); tests/test_abi.py checks it against palette_rgb
static __device__ const uint32_t PALETTE_RGB[32] = {
    0x000000u, 0x66cc66u, 0x66e566u, 0x666666u, 0x696969u, 0x6b6b6bu, 0xffffffu, 0x0000ffu, 0x4d4d4du,
    0x0000ccu, 0xcc0000u, 0x00cc00u, 0xcccc00u, 0xccccccu, 0x000000u, 0xcc00ccu, 0x00ccccu,
    0xff0000u, 0xff0033u, 0x00ff00u, 0x006666u, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

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

// ---- score label (multi_car_racing.py:533-535, 665-666): pyglet.text.Label('0000', font_size=36, x=20,
// y=WINDOW_H*2.5/40, anchor_x='left', anchor_y='center', white), text "%04i" % reward[agent].  Real pyglet output is
// font/platform dependent, so the build DEFINES the glyphs: a 5x7 bitmap font laid out in WINDOW units (36 pt at 96 dpi
// = 48 px em: advance 26, glyph box 20 x 35 = 5 x 7 cells of 4 x 5 window units, vertically centred on y = 50),
// sampled at pixel centres like every other primitive — the same rule at 96x96 (a 2 x 4 px smudge per digit, as in the
// reference's observations) and at 600x400.  Rows top to bottom, 5 bits per row, bit 4 = leftmost column.
#define LABEL_X0 20.0f
#define LABEL_ADV 26.0f
#define LABEL_CELL_W 4.0f
#define LABEL_CELL_H 5.0f
#define LABEL_Y0 32.5f
#define LABEL_MAX_CHARS 6
static __device__ const uint8_t LABEL_GLYPHS[11][7] = {
    {0x0E, 0x11, 0x13, 0x15, 0x19, 0x11, 0x0E},   // 0
    {0x04, 0x0C, 0x04, 0x04, 0x04, 0x04, 0x0E},   // 1
    {0x0E, 0x11, 0x01, 0x02, 0x04, 0x08, 0x1F},   // 2
    {0x1F, 0x02, 0x04, 0x02, 0x01, 0x11, 0x0E},   // 3
    {0x02, 0x06, 0x0A, 0x12, 0x1F, 0x02, 0x02},   // 4
    {0x1F, 0x10, 0x1E, 0x01, 0x01, 0x11, 0x0E},   // 5
    {0x06, 0x08, 0x10, 0x1E, 0x11, 0x11, 0x0E},   // 6
    {0x1F, 0x01, 0x02, 0x04, 0x08, 0x08, 0x08},   // 7
    {0x0E, 0x11, 0x11, 0x0E, 0x11, 0x11, 0x0E},   // 8
    {0x0E, 0x11, 0x11, 0x0F, 0x01, 0x02, 0x0C},   // 9
    {0x00, 0x00, 0x00, 0x1F, 0x00, 0x00, 0x00}};  // '-'
// number of characters of "%04i" % v
__device__ __forceinline__ int label_chars(int v) {
  const int neg = v < 0 ? 1 : 0;
  unsigned a = (unsigned)(neg ? -(long long)v : v);
  int nd = 1; for (unsigned t = a; t >= 10u; t /= 10u) ++nd;
  const int width = 4 - neg;
  if (nd < width) nd = width;
  return neg + nd;
}
// is the window-space sample (wx, wy) lit by the label showing integer v?
// glyphs: the 11 x 7 table above (or a copy of it in LDS)
__device__ __forceinline__ bool label_on(int v, float wx, float wy, const uint8_t* glyphs = &LABEL_GLYPHS[0][0]) {
  const float fy = (wy - LABEL_Y0) * (1.0f / LABEL_CELL_H);
  if (!(fy >= 0.0f && fy < 7.0f)) return false;
  const float fx = wx - LABEL_X0;
  if (!(fx >= 0.0f)) return false;
  const int j = (int)floorf(fx * (1.0f / LABEL_ADV));
  const int nch = label_chars(v);
  if (j >= nch) return false;
  const float lx = fx - (float)j * LABEL_ADV;
  const int c = (int)floorf(lx * (1.0f / LABEL_CELL_W));
  if (c >= 5) return false;
  const int r = 6 - (int)floorf(fy);
  const int neg = v < 0 ? 1 : 0;
  int g;
  if (neg && j == 0) g = 10;
  else {
    unsigned a = (unsigned)(neg ? -(long long)v : v);
    const int nd = nch - neg, d = j - neg;              // digit d of nd, most significant first
    for (int t = 0; t < nd - 1 - d; ++t) a /= 10u;
    g = (int)(a % 10u);
  }
  return (((uint32_t)glyphs[g * 7 + r] >> (4 - c)) & 1u) != 0u;
}
}  // namespace view

// One 4-edge record vs one pixel centre: all oriented edge functions >= 0 (no short-circuit: one LDS burst).
__device__ __forceinline__ bool inside4(const float4 a, const float4 b, const float4 c, float cx, float cy) {
  const float e0 = a.x * cx + a.y * cy + a.z, e1 = a.w * cx + b.x * cy + b.y, e2 = b.z * cx + b.w * cy + c.x, e3 = c.y * cx + c.z * cy + c.w;
  return fminf(fminf(e0, e1), fminf(e2, e3)) >= 0.0f;
}
#define UNI(x) __builtin_amdgcn_readfirstlane(x)

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
    const double db = TB[u] - TB[t];
    const double side = db > 0.0 ? 1.0 : (db < 0.0 ? -1.0 : 0.0);          // np.sign
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
