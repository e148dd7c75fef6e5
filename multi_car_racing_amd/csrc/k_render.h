// k_render.h — render(mode) at an arbitrary viewport (multi_car_racing.py:573-597): 'rgb_array' is the same scene
// as 'state_pixels' drawn into glViewport(0,0,600,400).  Debug / video path of ONE env, not the training hot path:
// a straightforward tile rasteriser (16x16 pixels per workgroup, road quads culled per tile in chunks of 256 through
// LDS, painter's order == highest draw index wins) that reuses the camera, HUD rectangles and car polygons the last
// k_dynamics left in HBM.  Same sampling rule as k_view: pixel centres, inside <=> all oriented edge functions >= 0.
// The score label (:665-666) is the build's bitmap font (k_raster_common.h).  Skid particles (Car.draw(viewer, True), :564)
// are drawn when the handle tracks them (mcr_config.skid_particles): 5-pixel-wide rectangles around the segments.
#pragma once
#include "k_raster_common.h"

#define RENDER_TILE 16

// grid (ceil(W/16), ceil(H/16), N), 256 threads.  out: [N][H][W][3] u8, rows top-down (arr[::-1], :602)
__global__ __launch_bounds__(256) void k_render_frame(McrParams p, int env, int W, int H, uint8_t* __restrict__ out) {
  using namespace view;
  const int tid = threadIdx.x, lane = tid & 63;
  const int N = p.N, BN = p.BN, agent = blockIdx.z;
  const int ci = env * N + agent;
  const McrEnvState es = p.env[env];
  if (!es.active) return;
  const uint8_t* slot = p.slots + ((size_t)env * 2 + es.slot) * MCR_SLOT_BYTES;
  const McrSlotHeader* hd = (const McrSlotHeader*)slot;
  const int P = hd->P;
  const float4* __restrict__ QA = (const float4*)(slot + MCR_OFF_QA); const float4* __restrict__ QB = (const float4*)(slot + MCR_OFF_QB);
  const uint32_t* __restrict__ QM = (const uint32_t*)(slot + MCR_OFF_QMETA);
  const uint16_t* tflags = p.tile_flags + (size_t)env * MCR_TILE_CAP;
  const float* __restrict__ vp = p.viewp + (size_t)ci * MCR_VIEWP_FLOATS;
  // camera: viewp holds pixel = M * world + t for the 96x96 viewport; the window->viewport scale is linear
  const float rx = (float)W / 96.0f, ry = (float)H / 96.0f;
  const float m00 = vp[VP_CAM + 0] * rx, m01 = vp[VP_CAM + 1] * rx, ctx = vp[VP_CAM + 4] * rx;
  const float m10 = vp[VP_CAM + 2] * ry, m11 = vp[VP_CAM + 3] * ry, cty = vp[VP_CAM + 5] * ry;

  const int X0 = blockIdx.x * RENDER_TILE, Y0 = blockIdx.y * RENDER_TILE;      // GL pixel coords (origin bottom-left)
  const int px_i = X0 + (tid & 15), py_i = Y0 + (tid >> 4);
  const float cx = (float)px_i + 0.5f, cy = (float)py_i + 0.5f;
  const float bx0 = (float)X0 + 0.5f, bx1 = bx0 + 15.0f, by0 = (float)Y0 + 0.5f, by1 = by0 + 15.0f;   // pixel-centre box of the tile

  __shared__ float4 ent[256][3];
  __shared__ uint32_t ekey[256];
  __shared__ int ecount;
  __shared__ float4 car8[CARPOLY_CAP * 6];
  __shared__ uint32_t cinfo[CARPOLY_CAP];

  // background (:615-627): playfield + checker, evaluated through the inverse camera in 96-viewport units
  uint32_t col = PAL_BLACK;
  {
    const float hk = 0.5f / (float)(MCR_PLAYFIELD / 20.0);
    const float X = cx / rx, Y = cy / ry;
    const float U = (vp[VP_INV + 0] * X + vp[VP_INV + 1] * Y + vp[VP_INV + 2]) * hk, V = (vp[VP_INV + 3] * X + vp[VP_INV + 4] * Y + vp[VP_INV + 5]) * hk;
    if (fabsf(U) <= 10.0f && fabsf(V) <= 10.0f) col = ((U - floorf(U)) < 0.5f && (V - floorf(V)) < 0.5f) ? PAL_GRASS1 : PAL_GRASS0;
  }
  int best = -1;
  // road_poly in chunks of 256 quads: cull against the tile, survivors' edge equations through LDS
  for (int base = 0; base < P; base += 256) {
    if (tid == 0) ecount = 0;
    __syncthreads();
    const int q = base + tid;
    bool keep = false; float e[12]; uint32_t key = 0;
    if (q < P) {
      const float4 a = QA[q], b = QB[q];
      const float wx[4] = {a.x, a.z, b.x, b.z}, wy[4] = {a.y, a.w, b.y, b.w};
      float qx[4], qy[4], x0 = MCR_MAXFLT, x1 = -MCR_MAXFLT, y0 = MCR_MAXFLT, y1 = -MCR_MAXFLT;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        qx[i] = __builtin_fmaf(m00, wx[i], __builtin_fmaf(m01, wy[i], ctx)); qy[i] = __builtin_fmaf(m10, wx[i], __builtin_fmaf(m11, wy[i], cty));
        x0 = fminf(x0, qx[i]); x1 = fmaxf(x1, qx[i]); y0 = fminf(y0, qy[i]); y1 = fmaxf(y1, qy[i]);
      }
      if (!(x0 > bx1 || x1 < bx0 || y0 > by1 || y1 < by0) && edge_setup(qx, qy, 4, e)) {
        keep = true;
        const uint32_t meta = QM[q];
        uint32_t c = meta & 0xffu; const uint32_t tile1 = (meta >> 8) & 0x3ffu;
        if (tile1 && (tflags[tile1 - 1] & 0x100u)) c = MCR_COL_ROAD0;                   // touched tile -> ROAD_COLOR (:102-104)
        const uint32_t pal = c == MCR_COL_ROAD0 ? PAL_ROAD0 : c == MCR_COL_ROAD1 ? PAL_ROAD1 : c == MCR_COL_ROAD2 ? PAL_ROAD2 : c == MCR_COL_KERB_WHITE ? PAL_WHITE : PAL_RED255;
        key = ((uint32_t)q << 5) | pal;
      }
    }
    const unsigned long long mask = __ballot(keep);
    if (mask) {
      int at = 0;
      if (lane == 0) at = atomicAdd(&ecount, __popcll(mask));
      at = __shfl(at, 0) + __popcll(mask & ((1ull << lane) - 1ull));
      if (keep) { ent[at][0] = make_float4(e[0], e[1], e[2], e[3]); ent[at][1] = make_float4(e[4], e[5], e[6], e[7]); ent[at][2] = make_float4(e[8], e[9], e[10], e[11]); ekey[at] = key; }
    }
    __syncthreads();
    const int n = ecount;
    for (int s = 0; s < n; ++s)
      if (inside4(ent[s][0], ent[s][1], ent[s][2], cx, cy) && (int)ekey[s] > best) best = (int)ekey[s];
    __syncthreads();
  }
  // car polygons (Car.draw, particles off): padded 8-gons prepared by k_dynamics, every car, wheels then hull
  if (tid < N * 12) {
    const int k = tid, c = k / 12, j = k % 12;
    const float* __restrict__ cp = p.carpoly + (size_t)(env * N + c) * MCR_CARPOLY_FLOATS;
    const float* cv = cp + j * 16;
    const int n = __float_as_int(cp[MCR_CARPOLY_NOFF + j]);
    uint32_t info = 0;
    if (n > 0) {
      uint32_t colr;
      if (j < 8) colr = (j & 1) ? PAL_WHEELWHITE : PAL_BLACK;
      else { colr = PAL_CAR0 + (c & 7); if (p.use_ego_color) colr = (c == agent) ? PAL_CAR0 + 0 : PAL_CAR0 + 1; }
      float qx[8], qy[8], x0 = MCR_MAXFLT, x1 = -MCR_MAXFLT, y0 = MCR_MAXFLT, y1 = -MCR_MAXFLT;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float wx = cv[i * 2], wy = cv[i * 2 + 1];
        qx[i] = __builtin_fmaf(m00, wx, __builtin_fmaf(m01, wy, ctx)); qy[i] = __builtin_fmaf(m10, wx, __builtin_fmaf(m11, wy, cty));
        x0 = fminf(x0, qx[i]); x1 = fmaxf(x1, qx[i]); y0 = fminf(y0, qy[i]); y1 = fmaxf(y1, qy[i]);
      }
      float area = 0.0f;
#pragma unroll
      for (int i = 0; i < 8; ++i) { const int jj = (i + 1) & 7; area += qx[i] * qy[jj] - qx[jj] * qy[i]; }
      if (!(x0 > bx1 || x1 < bx0 || y0 > by1 || y1 < by0) && area != 0.0f) {
        const float sg = area > 0.0f ? 1.0f : -1.0f;
        float e[24];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int jj = (i + 1) & 7;
          const float ex = qx[jj] - qx[i], ey = qy[jj] - qy[i];
          const float A = -sg * ey, B = sg * ex;
          e[i * 3 + 0] = A; e[i * 3 + 1] = B; e[i * 3 + 2] = -(A * qx[i] + B * qy[i]);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) car8[k * 6 + i] = make_float4(e[i * 4], e[i * 4 + 1], e[i * 4 + 2], e[i * 4 + 3]);
        info = 0x100u | colr;
      }
    }
    cinfo[k] = info;
  }
  __syncthreads();
  if (best >= 0) col = (uint32_t)best & 31u;
  // Car.draw per car, in car order (:559-564): its skid particles first (draw_particles = mode != 'state_pixels'; gym draws
  // each as a GL_LINE_STRIP of width 5 — here every segment is the 5-pixel-wide rectangle around it, no caps or joins:
  // GL's own wide-line rasterisation is implementation-defined), then its wheels and hull
  for (int c = 0; c < N; ++c) {
    if (p.particles) {
      const uint32_t* __restrict__ pc = p.particles + (size_t)(env * N + c) * MCR_PART_WORDS;
      const int created = (int)pc[0], first = created > MCR_PART_MAX ? created - MCR_PART_MAX : 0;
      const int nseg = (created - first) * (MCR_PART_PTS - 1);
      int pbest = -1;
      for (int base = 0; base < nseg; base += 256) {
        if (tid == 0) ecount = 0;
        __syncthreads();
        const int sidx = base + tid;
        bool keep = false; float e[12]; uint32_t key = 0;
        if (sidx < nseg) {
          const int ord = sidx / (MCR_PART_PTS - 1), i = sidx - ord * (MCR_PART_PTS - 1);
          const int sl = (first + ord) % MCR_PART_MAX;
          const uint32_t info = pc[17 + sl];
          if (i + 1 < (int)(info & 255u)) {
            const uint32_t* pt = pc + MCR_PART_HDR + (sl * MCR_PART_PTS + i) * 2;
            const float ax = __uint_as_float(pt[0]), ay = __uint_as_float(pt[1]), bx = __uint_as_float(pt[2]), by = __uint_as_float(pt[3]);
            const float pax = __builtin_fmaf(m00, ax, __builtin_fmaf(m01, ay, ctx)), pay = __builtin_fmaf(m10, ax, __builtin_fmaf(m11, ay, cty));
            const float pbx = __builtin_fmaf(m00, bx, __builtin_fmaf(m01, by, ctx)), pby = __builtin_fmaf(m10, bx, __builtin_fmaf(m11, by, cty));
            const float dx = pbx - pax, dy = pby - pay, len = sqrtf(dx * dx + dy * dy);
            if (len > 0.0f) {
              const float nx = -dy / len * 2.5f, ny = dx / len * 2.5f;
              const float qx[4] = {pax + nx, pax - nx, pbx - nx, pbx + nx}, qy[4] = {pay + ny, pay - ny, pby - ny, pby + ny};
              const float x0 = fminf(fminf(qx[0], qx[1]), fminf(qx[2], qx[3])), x1 = fmaxf(fmaxf(qx[0], qx[1]), fmaxf(qx[2], qx[3]));
              const float y0 = fminf(fminf(qy[0], qy[1]), fminf(qy[2], qy[3])), y1 = fmaxf(fmaxf(qy[0], qy[1]), fmaxf(qy[2], qy[3]));
              if (!(x0 > bx1 || x1 < bx0 || y0 > by1 || y1 < by0) && edge_setup(qx, qy, 4, e)) {
                keep = true;
                key = ((uint32_t)sidx << 5) | (((info >> 8) & 1u) ? PAL_MUD : PAL_BLACK);
              }
            }
          }
        }
        const unsigned long long mask = __ballot(keep);
        if (mask) {
          int at = 0;
          if (lane == 0) at = atomicAdd(&ecount, __popcll(mask));
          at = __shfl(at, 0) + __popcll(mask & ((1ull << lane) - 1ull));
          if (keep) { ent[at][0] = make_float4(e[0], e[1], e[2], e[3]); ent[at][1] = make_float4(e[4], e[5], e[6], e[7]); ent[at][2] = make_float4(e[8], e[9], e[10], e[11]); ekey[at] = key; }
        }
        __syncthreads();
        const int n = ecount;
        for (int s2 = 0; s2 < n; ++s2)
          if (inside4(ent[s2][0], ent[s2][1], ent[s2][2], cx, cy) && (int)ekey[s2] > pbest) pbest = (int)ekey[s2];
        __syncthreads();
      }
      if (pbest >= 0) col = (uint32_t)pbest & 31u;
    }
    for (int k = c * 12; k < c * 12 + 12; ++k) {
      const uint32_t ci2 = cinfo[k];
      if (!ci2) continue;
      const float4* r = &car8[k * 6];
      if (inside4(r[0], r[1], r[2], cx, cy) && inside4(r[3], r[4], r[5], cx, cy)) col = ci2 & 31u;
    }
  }
  // render_indicators (:634-674) in window space: bar of 5h = 1/8 of the height, gauges, backwards flag
  {
    const float kx = (float)W / 1000.0f, ky = (float)H / 800.0f;
    if (cy < 100.0f * ky) col = PAL_BLACK;
    const uint32_t ind_col[7] = {PAL_WHITE, PAL_BLUE255, PAL_BLUE255, PAL_PURPLE, PAL_PURPLE, PAL_GREEN255, PAL_RED255};
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const float x0 = vp[VP_IND + i * 4] * rx, x1 = vp[VP_IND + i * 4 + 1] * rx, y0 = vp[VP_IND + i * 4 + 2] * ry, y1 = vp[VP_IND + i * 4 + 3] * ry;
      if (x1 > x0 && y1 > y0 && cx >= x0 && cx <= x1 && cy >= y0 && cy <= y1) col = ind_col[i];
    }
    // score label (:665-666) of the CURRENT reward (render() between steps shows self.reward after the step's -0.1)
    if (label_on(mcr_label_value(p.card[CD_REWARD * BN + ci]), cx * (1000.0f / (float)W), cy * (800.0f / (float)H))) col = PAL_WHITE;
    if ((p.caru[CU_FLAGS * BN + ci] & 1u) && p.backwards_flag) {
      const float fx[3] = {900.0f * kx, 925.0f * kx, 950.0f * kx}, fy[3] = {30.0f * ky, 70.0f * ky, 30.0f * ky};
      float fe[9];
      if (edge_setup(fx, fy, 3, fe) && (fe[0] * cx + fe[1] * cy + fe[2] >= 0.0f) && (fe[3] * cx + fe[4] * cy + fe[5] >= 0.0f) && (fe[6] * cx + fe[7] * cy + fe[8] >= 0.0f)) col = PAL_BLUE255;
    }
  }
  if (px_i < W && py_i < H) {
    const uint32_t c3 = palette_rgb((int)col);
    uint8_t* o = out + (((size_t)agent * H + (size_t)(H - 1 - py_i)) * W + px_i) * 3;
    o[0] = (uint8_t)(c3 & 255u); o[1] = (uint8_t)((c3 >> 8) & 255u); o[2] = (uint8_t)((c3 >> 16) & 255u);
  }
}
