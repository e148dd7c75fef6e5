// mcr_kernels.h — kernel parameter block shared by the three step kernels (gfx950 only).
#pragma once
#include "mcr_common.h"

struct McrParams {
  int32_t B, N, G;              // envs, agents, lanes per env in the dynamics kernel (pow2 >= N)
  int32_t BN;                   // B*N: stride of every per-car SoA field
  int32_t env0, nenv;           // env sub-range [env0, env0+nenv) covered by this launch (stream-level pipelining)
  float* carf;                  // [CF_COUNT][BN]
  double* card;                 // [CD_COUNT][BN]
  uint32_t* caru;               // [CU_COUNT][BN]
  McrEnvState* env;             // [B]
  uint8_t* slots;               // [B][2][MCR_SLOT_BYTES]
  uint32_t* tile_touch;         // [B][TILE_CAP]  bit (car*4+wheel): wheel currently in contact with the tile
  uint16_t* tile_flags;         // [B][TILE_CAP]  bits 0..7 road_visited[car], bit 8 recoloured
  uint32_t* cc_store;           // [B][...] car<->car manifold store (warm starting)
  const McrShapes* shapes;
  float* viewp;                 // [BN][MCR_VIEWP_FLOATS] per-car camera + HUD geometry, written by k_dynamics, read by k_view
  float* carpoly;               // [BN][MCR_CARPOLY_FLOATS] world-space vertices of the car's 12 draw polygons (Car.draw)
  int32_t* consumed_host;       // [B] mapped host memory: episode counter of the last install
  uint32_t* ready;              // [B] step serial published by k_dynamics once the env's post-step state is in HBM
  uint32_t serial;              // serial of this mcr_step call (k_view waits for ready[env] == serial when overlapped)
  int32_t wait_ready;           // 1: k_view runs concurrently with k_dynamics on disjoint CUs and gates each view on ready[env]
  // step I/O
  const float* actions;         // [B,N,3] or null
  uint8_t* obs;                 // [B,N,96,96,3] or null
  double* reward_out;           // [B,N]
  uint8_t* done_out;            // [B]
  uint8_t* trunc_out;           // [B] or null
  const uint8_t* reset_mask;    // [B] or null (k_install)
  int32_t auto_reset, max_steps, car_contacts, backwards_flag, use_ego_color;
  int32_t debug;                // ablation switches for profiling (0 in production)
  double h_ratio;
};

// per-car view parameters (f32): camera (:540-556) and HUD rectangles (:634-674) in pixel units
#define MCR_VIEWP_FLOATS 48
enum { VP_CAM = 0 /*m00 m01 m10 m11 tx ty*/, VP_INV = 6 /*ax bx cx0 ay by cy0*/, VP_IND = 12 /*7 x (x0 x1 y0 y1)*/, VP_HUDTOP = 40 };

// Car.draw polygons per car in draw order: 4 x (wheel box, white stripe) then the 4 hull polygons; each slot holds
// 8 vertices (x0 y0 .. x7 y7) and slot header words live in carpoly_n: vertex count (0 = not drawn)
#define MCR_CARPOLY_FLOATS (12 * 16 + 16)
#define MCR_CARPOLY_NOFF (12 * 16)
#define MCR_CC_MAX 24           // touching car<->car fixture pairs kept per env (warm start)
#define MCR_CC_WORDS 20         // u32 words per stored manifold
