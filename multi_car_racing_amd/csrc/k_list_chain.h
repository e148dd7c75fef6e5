// k_list_chain.h — the step of the LIST envs (roles 2 / 3 / 4: the contact envs, the envs the main dynamics deferred, the
// envs it re-spawned) in as few launches as possible.  These are a handful of envs per step whose work is a serial chain
// of solver iterations; every launch of such a chain that has to start beside the main raster — which fills every CU —
// queues for registers and LDS and then runs 2-4x slower than alone (measured), so the chain's kernels are fused and
// launched before the raster: a small grid (how long the lists are only the device knows) whose workgroups walk the
// list, wavefronts at issue priority 3.
#pragma once
#include "k_collide.h"
#include "k_dynamics.h"
#include "k_flags.h"
#include "k_viewprep.h"

// the reset pass (:408) of the list's envs that were re-spawned in this step: collide pass 1 (clear the per-tile state,
// re-detect), then the action-less dynamics step
__device__ __forceinline__ void list_reset_pass(const McrParams& p, const int blk) {
  __threadfence();                                             // E->resetting as this step's dynamics left it
  if (p.term_idx) for (int k = 0; k < p.list_envs_per_block; ++k) term_prepare(p, mcr_env_of_slot(p, blk * p.list_envs_per_block + k));   // (before the pass clears the tile flags)
  for (int k = 0; k < p.list_envs_per_block; ++k) { collide_block(p, 1, blk * p.list_envs_per_block + k); __syncthreads(); }
  __threadfence();                                             // the dynamics lanes read what the collide lanes stored
  __syncthreads();
  dynamics_block<true>(p, 1, blk);
  __syncthreads();
}

// roles 2 / 3: dynamics (for role 3: the rest of it) -> reset pass if the episode ended -> bookkeeping (k_flags.h).
// The launch can carry a SECOND list with its own parameter block: workgroups [ga, gridDim) run the reset pass of the envs the
// main dynamics re-spawned (role 4, what k_reset_list does) beside the chain of the first list — one launch, so that neither
// waits for the other in a stream (the re-spawned envs used to queue behind the contact chain, which is long in 1 step of 6).
// CC = false: the first list holds no env with touching car<->car contacts (role 3: the deferred envs) — its dynamics are the contact-free build
template <bool CC>
__global__ __launch_bounds__(64) void k_list_chain(McrParams pa, McrParams pb, const int with_flags, const int ga) {
  __builtin_amdgcn_s_setprio(3);
  // soft_sync (mcr_kernels.h): the contact chain follows the contact pass in its stream — its start IS the contact pass's completion
  if (pa.soft_sync && pa.role == 2 && pa.cc_mode && !pa.fuse_collide && blockIdx.x == 0 && threadIdx.x == 0) mcr_post(pa, W_COL);
  if (pa.role == 2 && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(&pa.host_counts[HC_CONTACT_ENVS], (uint32_t)pa.clist[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (the contact pass is complete: the list is)
  if (pa.soft_sync && pa.role == 3) {
    // the resume chain follows the main dynamics in its stream: its start IS the dynamics' completion (the third stream's kernels wait for
    // that); and its envs read what the contact pass left for them (a deferred env never got to the main dynamics' in-kernel wait)
    if (blockIdx.x == 0 && threadIdx.x == 0) mcr_post(pa, W_DYN);
    if (pa.cc_mode) {
      if (threadIdx.x == 0) (void)mcr_await(pa, W_COL);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __syncthreads();
    }
  }
  // (two branches instead of one body on `first ? pa : pb`: selecting between the two argument blocks makes the compiler copy
  // one into scratch memory, 1 KB per lane)
  if ((int)blockIdx.x < ga) {
    const McrParams& p = pa;
    const int nb = mcr_virtual_blocks(p, p.list_envs_per_block);
    for (int blk = blockIdx.x; blk < nb; blk += ga) {
      if (CC && p.fuse_collide) {                                  // the env's own contact pass (k_collide skipped it): manifolds, tile events, rewards at the entry poses
        for (int k = 0; k < p.list_envs_per_block; ++k) { collide_block(p, 0, blk * p.list_envs_per_block + k); __syncthreads(); }
        __threadfence();                                           // the dynamics lanes read what the collide lanes stored
        __syncthreads();
      }
      dynamics_block<CC, CC, true>(p, 0, blk);                    // (CC: the contact chain, one env per wavefront — the uniform contact sweeps)
      viewprep_list_block(p, blk);                                 // (the envs' view records and car polygons, five lanes per car)
      __syncthreads();
      if (p.auto_reset) list_reset_pass(p, blk);
      if (with_flags) {
        __threadfence();                                         // poses and env state as the dynamics left them
        for (int c = 0; c < p.list_envs_per_block * p.N; ++c) flags_block(p, blk * p.list_envs_per_block * p.N + c);
      }
      __syncthreads();
    }
  } else {
    const McrParams& p = pb;
    const int nb = mcr_virtual_blocks(p, p.list_envs_per_block), stride = (int)gridDim.x - ga;
    for (int blk = (int)blockIdx.x - ga; blk < nb; blk += stride) {
      list_reset_pass(p, blk);
      // the next step's touch verdict of the re-spawned envs: this launch is its only writer (the main dynamics left a 0, so the
      // main envs' bookkeeping kernel does not look at these envs; their cars take no bookkeeping in this step: first observation)
      if (p.part_next) {
        __threadfence();
        for (int k = 0; k < p.list_envs_per_block; ++k) {
          const int env = mcr_env_of_slot(p, blk * p.list_envs_per_block + k);
          if (env >= p.env0 + p.nenv) continue;
          const bool v = p.env[env].active ? mcr_touch_verdict(p, env) : false;
          if (threadIdx.x == 0) mcr_set_verdict(p, env, v);
        }
      }
    }
  }
}

// the bookkeeping of a list's cars as a launch of its own (one wavefront per car at a time): what the chains use when an
// env has more than two cars — inside k_list_chain the cars of a block take their turns one after the other
__global__ __launch_bounds__(64) void k_flags_list(McrParams p) {
  __builtin_amdgcn_s_setprio(3);
  const int nb = mcr_list_len(p) * p.N;
  for (int blk = blockIdx.x; blk < nb; blk += gridDim.x) flags_block(p, blk);
}

__global__ void k_term_finish(McrParams p) { term_finish(p); }
// single-stream step: the terminal entries of all re-spawned envs, in front of the reset pass (one wavefront per env)
__global__ __launch_bounds__(64) void k_term_prep(McrParams p) { term_prepare(p, p.env0 + (int)blockIdx.x); }

// soft_sync's one-thread kernels (see mcr_post / mcr_await)
// (debug bit 13: the side stream's completion is never posted — what a stalled stream looks like to the step's join; tests)
// (fuse_collide, side stream: this step's contact list has had its last reader — the chain and its raster — and is the list the NEXT step's
// verdict writers fill: emptied here, a kernel boundary and a whole step ahead of the first append)
__global__ void k_post(McrParams p, int w) {
  if (threadIdx.x == 0 && w == W_SIDE && p.fuse_collide) p.clist[0] = 0;
  if (threadIdx.x == 0 && !((p.debug & 8192) && w == W_SIDE)) mcr_post(p, w);
}
__global__ void k_await(McrParams p, int w0, int w1) {
  if (threadIdx.x == 0) { if (w0 >= 0) (void)mcr_await(p, w0); if (w1 >= 0) (void)mcr_await(p, w1); }
}

// The main envs' view records (k_viewprep.h, one lane per car: workgroups [0, vp_blocks)) and bookkeeping (k_flags.h, one wavefront
// per car: the rest) in ONE launch: neither reads what the other writes, and one after the other they were 18 + 23 us in front of
// the main raster.  Within k_flags' 64 VGPRs, so that its 8192 wavefronts still fit the chip in one round.
__global__ __launch_bounds__(64, 8) void k_flags_viewprep(McrParams p, const int vp_blocks) {
  if ((int)blockIdx.x < vp_blocks) viewprep_block(p, (int)blockIdx.x);
  else flags_block(p, (int)blockIdx.x - vp_blocks);
}
