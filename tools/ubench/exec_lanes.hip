// exec_lanes.hip — gfx950: what a VALU instruction of a LONE wavefront costs as a function of how many of its lanes are active
// (is a 16-lane pass whose EXEC bits are all zero skipped?), for independent and for dependent instructions, f32 / packed f32 / f64,
// and what v_readlane + use costs.  Not a test, not product code.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/_bin/exec_lanes tools/ubench/exec_lanes.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER (1 << 17)
typedef float f2 __attribute__((ext_vector_type(2)));
template <int KIND, bool DEP>
__global__ void k(float* out, float a, float b, int lanes, long long* clk) {
  float r[16]; f2 p[8]; double d[8];
  for (int i = 0; i < 16; ++i) r[i] = threadIdx.x * 0.001f + i;
  for (int i = 0; i < 8; ++i) { p[i] = (f2){threadIdx.x * 0.001f + i, (float)i}; d[i] = threadIdx.x * 0.001 + i; }
  f2 aa = {a, a}, bb = {b, b}; double da = a, db = b;
  long long t0 = 0, t1 = 0;
  if ((int)threadIdx.x < lanes) {
    t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
      if (KIND == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { if (DEP) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[0]) : "v"(a), "v"(b)); else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b)); }
      } else if (KIND == 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { if (DEP) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[0]) : "v"(aa), "v"(bb)); else asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i & 7]) : "v"(aa), "v"(bb)); }
      } else if (KIND == 2) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { if (DEP) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[0]) : "v"(da), "v"(db)); else asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i & 7]) : "v"(da), "v"(db)); }
      } else if (KIND == 3) {   // readlane -> VALU use of the SGPR
#pragma unroll
        for (int i = 0; i < 8; ++i) { int s; asm volatile("v_readlane_b32 %0, %1, 1" : "=s"(s) : "v"(r[DEP ? 0 : i])); asm volatile("v_add_f32 %0, %1, %0" : "+v"(r[DEP ? 0 : i]) : "s"(s)); }
      } else if (KIND == 4) {   // v_cndmask with an SGPR-pair mask written by v_cmp
#pragma unroll
        for (int i = 0; i < 8; ++i) { unsigned long long m; asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(m) : "v"(r[DEP ? 0 : i]), "v"(b)); asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(r[DEP ? 0 : i]) : "v"(a), "s"(m)); }
      } else if (KIND == 5) {   // SALU
        int q = lanes;
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("s_add_u32 %0, %0, %1" : "+s"(q) : "s"(lanes) : "scc");
        r[0] += (float)q;
      }
    }
    t1 = __builtin_readcyclecounter();
  }
  float s = 0; for (int i = 0; i < 16; ++i) s += r[i]; for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y + (float)d[i];
  if (s == 12345.678f) out[0] = s;
  if (threadIdx.x == 0) clk[0] = t1 - t0;
}
template <int KIND, bool DEP> void run(const char* name, float* out, long long* clk) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int lanes : {1, 16, 64}) {
    hipLaunchKernelGGL((k<KIND, DEP>), dim3(1), dim3(64), 0, 0, out, 1.0001f, 0.5f, lanes, clk);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<KIND, DEP>), dim3(1), dim3(64), 0, 0, out, 1.0001f, 0.5f, lanes, clk);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    printf("%-28s %-4s lanes %2d: %6.2f ticks, %6.2f ns per instruction (counter at %.0f MHz)\n", name, DEP ? "dep" : "ind", lanes, (double)c / (ITER * 16.0), ms * 1e6 / (ITER * 16.0), (double)c / (ms * 1e3));
  }
}
int main() {
  float* out; long long* clk; hipMalloc(&out, 64); hipMalloc(&clk, 64);
  run<0, false>("v_fma_f32", out, clk); run<0, true>("v_fma_f32", out, clk);
  run<1, false>("v_pk_fma_f32", out, clk); run<1, true>("v_pk_fma_f32", out, clk);
  run<2, false>("v_fma_f64", out, clk); run<2, true>("v_fma_f64", out, clk);
  run<3, false>("readlane+add (pairs)", out, clk); run<3, true>("readlane+add (pairs)", out, clk);
  run<4, false>("cmp+cndmask (pairs)", out, clk); run<4, true>("cmp+cndmask (pairs)", out, clk);
  run<5, true>("s_add_u32", out, clk);
  return 0;
}
