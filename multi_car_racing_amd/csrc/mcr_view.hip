// mcr_view.hip — the raster kernel (k_view.h) as a translation unit of its own: it is built with -fno-slp-vectorize.
// The SLP vectoriser pairs independent f32 operations into v_pk_* instructions, which on gfx950 issue at 0.55x the rate of
// the scalar forms (profiles/r02_ubench_issue_rates.txt) and need their operands in adjacent registers: for the raster
// that costs 6 VGPRs it does not have (168 + 40 B of scratch vs 162 and none) and 5 us per launch.  (mcr_hip.hip is built the same way since
// round 5: its velocity sweeps are written on pairs of floats by hand, and the vectoriser's own pairs only cost there — build.py has the numbers.)
#include "mcr_kernels.h"
#define MCR_DEVICE_FUNCTIONS_ONLY          // k_flags.h / k_touch.h: the device functions, not the kernels (they live in mcr_hip.hip)
#include "k_flags.h"
#include "k_view.h"
#include <hip/hip_ext.h>

// variant: 0 main launch, 1 main launch with the per-phase clocks (debug bit 5), 2 list launch (persistent workgroups that walk a list)
// stop: an event the launch itself completes (hipExtLaunchKernelGGL: the dispatch packet's completion signal — no marker packet
// behind the kernel, which costs ~3 us in-stream and ~3.5 us more on a stream hop; tools/ubench/event_gap.hip), or nullptr
// start: an event that takes the dispatch's own BEGIN timestamp (with `stop`: the kernel's duration as the profiler sees it — mcr_timing), or nullptr
void mcr_view_launch(int variant, int grid, hipStream_t st, const McrParams& P, unsigned long long* stamps, int only_just_reset, hipEvent_t stop, hipEvent_t start) {
  if (variant == 2) hipExtLaunchKernelGGL((k_view<false, true>), dim3(grid), dim3(VIEW_THREADS), 0, st, start, stop, 0, P, stamps, only_just_reset);
  else if (variant == 1) hipExtLaunchKernelGGL((k_view<true, false>), dim3(grid), dim3(VIEW_THREADS), 0, st, start, stop, 0, P, stamps, only_just_reset);
  else hipExtLaunchKernelGGL((k_view<false, false>), dim3(grid), dim3(VIEW_THREADS), 0, st, start, stop, 0, P, stamps, only_just_reset);
}
