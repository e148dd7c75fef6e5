// k_world.h — the ONE b2World an env keeps for its life (multi_car_racing.py:138; _destroy :173-181; reset :341), reduced to what outlives an
// episode and reaches a result: the PROXY IDS of the broadphase — b2DynamicTree node indices — that the next episode's fixtures receive.  They
// order the Begin callbacks of a step (FrictionDetector, :113-120: which of two cars that reach a tile in the same step is its first visitor)
// and name fixtureA of a pair (k_collide.h), so from its second episode on an env differs from "every episode is the first of a fresh world"
// (profiles/r06_world_reuse_effect.txt: tile counts of 4-17 % of the envs of a policy that drives).
//
// Rounds 4-5 carried a literal tree for this (the oracle's DynTree, mcr_world.cpp for the facade) and called a device version a ~1 ms
// serial rebuild per auto-reset.  It is not: the LEAF ids never depend on the tree's shape.
//   * b2DynamicTree::MoveProxy = RemoveLeaf (FreeNode(parent): pushed on the LIFO free list) + InsertLeaf (AllocateNode: pops the head —
//     the node just freed): the free list is the same before and after, the leaf keeps its id.  Only tiles and car fixtures exist; tiles
//     never move.
//   * DestroyProxy frees (parent, leaf) per proxy — the last proxy of the world only its leaf —, so after _destroy the list reads, from the
//     top: L_n, (L_n-1, P_n-1), .., (L_1, P_1), then whatever lay beneath, where L_k is the leaf destroyed k-th.  CreateProxy pops a leaf,
//     InsertLeaf a parent (none for the first proxy of an empty tree): the new episode's k-th proxy gets L_(n-k) and the P's — the only
//     entries whose ORDER depends on rotations — are only ever popped as parents again.  By induction (the pool's never-used tail is
//     0 | (1, 2) (3, 4) .. in the same pairing) the ids split into leaf ids {0, 1, 3, 5, ..} and parent ids for good.
// So a world is a STACK of free leaf ids plus a count of fresh ones: _destroy pushes the live ids in Box2D's destroy order (tiles in road
// order; per car the hull's fixture list from its head — polygons 3, 2, 1, 0 — then the wheels, gym Car.destroy), the new episode pops in
// creation order (tiles in track order, :318-327; per car hull polygons 0..3 then wheels 0..3, :366-406), fresh leaf j = 0 for j = 0, else
// 2j - 1.  O(T) independent copies for the wavefront that runs the env's reset pass.  tests: the oracle's literal b2DynamicTree (world mode 1)
// over consecutive episodes of shrinking and growing tracks (test_world_ids.py on the CPU, test_gpu_world.py through the kernels).
#pragma once
#include "mcr_kernels.h"

// the table k_collide / k_touch look the ids up in: the env's world (above), or the tables an episode blob brought along (mcr_world.cpp),
// or none (fresh world: ids ascend in creation order)
struct McrPidTables { bool has; const uint16_t* tile; const uint16_t* fix; };
__device__ __forceinline__ McrPidTables mcr_pid_tables(const McrParams& p, int env, const uint8_t* slot) {
  McrPidTables t;
  if (p.pid_tab) { t.has = true; t.tile = p.pid_tab + (size_t)env * MCR_PID_TAB; }
  else { t.has = ((const McrSlotHeader*)slot)->pad0 != 0; t.tile = (const uint16_t*)(slot + MCR_OFF_TPID); }
  t.fix = t.tile + MCR_TILE_CAP;                                    // (MCR_OFF_FPID = MCR_OFF_TPID + 2 * MCR_TILE_CAP: the same layout)
  return t;
}

// reset() on the env's world, by the ONE wavefront (64 lanes) that runs the env's reset pass, before it looks an id up.
// `scratch`: MCR_PID_TAB u16 of LDS.
__device__ __forceinline__ void mcr_world_reissue_ids(const McrParams& p, int env, int T_new, uint16_t* scratch) {
  const int lane = threadIdx.x & 63, N = p.N;
  int32_t* meta = p.pid_meta + (size_t)env * 4;
  uint16_t* tab = p.pid_tab + (size_t)env * MCR_PID_TAB;
  uint16_t* stk = p.pid_stack + (size_t)env * MCR_PID_STACK;
  const int sp_old = meta[0], fresh = meta[1], T_old = meta[2];
  const int n_old = T_old > 0 ? T_old + 8 * N : 0, n_new = T_new + 8 * N;
  // _destroy: the live ids in destroy order
  for (int i = lane; i < n_old; i += 64) {
    int src = i;
    if (i >= T_old) { const int j = i - T_old, c = j >> 3, r = j & 7; src = MCR_TILE_CAP + c * 8 + (r < 4 ? 3 - r : r); }
    scratch[i] = tab[src];
  }
  __syncthreads();
  // the new episode's fixtures in creation order: the stack from its top (the ids just pushed, then older ones), then fresh leaves
  const int sp_tot = sp_old + n_old;
  for (int k = lane; k < n_new; k += 64) {
    uint16_t id;
    if (k < n_old) id = scratch[n_old - 1 - k];
    else if (k < sp_tot) id = stk[sp_old - 1 - (k - n_old)];
    else { const int j = fresh + (k - sp_tot); id = (uint16_t)(j == 0 ? 0 : 2 * j - 1); }
    tab[k < T_new ? k : MCR_TILE_CAP + (k - T_new)] = id;
  }
  // what stays on the stack: entries below sp_old as they are, pushed ones the new episode did not reach
  const int sp_new = sp_tot > n_new ? sp_tot - n_new : 0;
  for (int idx = sp_old + lane; idx < sp_new; idx += 64) stk[idx] = scratch[idx - sp_old];
  if (lane == 0) { meta[0] = sp_new; meta[1] = fresh + (n_new > sp_tot ? n_new - sp_tot : 0); meta[2] = T_new; }
  __threadfence();                                                  // (the pass reads the table it just wrote: not from a stale L1 line)
  __syncthreads();
}
