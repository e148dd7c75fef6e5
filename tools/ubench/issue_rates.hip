// issue_rates.hip — gfx950 micro-benchmarks behind the raster design decisions (not a test, not product code).
// Measures wave-instruction throughput per CU for the instruction kinds k_view's shade loop is made of, at 1..8 waves per
// SIMD: v_fma_f32, v_pk_fma_f32, SALU, VALU+SALU interleaved, uniform (broadcast) ds_read_b128, v_readlane, ds_max_u32.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/_bin/issue_rates tools/ubench/issue_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define ITER 2048
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void k_fma(float* out, float a, float b) {
  float r[16];
  for (int i = 0; i < 16; ++i) r[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
  }
  float s = 0; for (int i = 0; i < 16; ++i) s += r[i];
  if (s == 12345.678f) out[0] = s;
}
__global__ void k_pkfma(float* out, float a, float b) {
  f2 r[8]; f2 aa = {a, a}, bb = {b, b};
  for (int i = 0; i < 8; ++i) r[i] = (f2){threadIdx.x * 0.001f + i, (float)i};
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int rep = 0; rep < 2; ++rep)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(aa), "v"(bb));
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += r[i].x + r[i].y;
  if (s == 12345.678f) out[0] = s;
}
__global__ void k_salu(float* out, int a) {
  int r[16];
  for (int i = 0; i < 16; ++i) r[i] = __builtin_amdgcn_readfirstlane(a + i);
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("s_add_u32 %0, %0, %1" : "+s"(r[i]) : "s"(a) : "scc");
  }
  int s = 0; for (int i = 0; i < 16; ++i) s += r[i];
  if (s == 123456789) out[0] = (float)s;
}
// 8 VALU + 8 SALU interleaved
__global__ void k_mix(float* out, float a, float b, int c) {
  float r[8]; int q[8];
  for (int i = 0; i < 8; ++i) { r[i] = threadIdx.x * 0.001f + i; q[i] = __builtin_amdgcn_readfirstlane(c + i); }
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
      asm volatile("s_add_u32 %0, %0, %1" : "+s"(q[i]) : "s"(c) : "scc");
    }
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += r[i] + (float)q[i];
  if (s == 12345.678f) out[0] = s;
}
// uniform-address ds_read_b128 (what the shade loop does per list entry: 3 of them) — 16 per iteration
__global__ void k_ldsbcast(float* out, int sel) {
  __shared__ f4 tab[256];
  tab[threadIdx.x] = (f4){(float)threadIdx.x, 1, 2, 3};
  __syncthreads();
  f4 acc = {0, 0, 0, 0};
  unsigned base = (unsigned)(size_t)tab + (unsigned)__builtin_amdgcn_readfirstlane(sel) * 16u;
  for (int it = 0; it < ITER; ++it) {
    f4 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[i]) : "v"(base), "i"(i * 112));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("" ::"v"(v[i]));
    acc += v[3];
  }
  if (acc.x == 12345.678f) out[0] = acc.x;
}
// per-lane ds_read_b128, consecutive lanes consecutive 16 B (conflict-free pattern)
__global__ void k_ldsvec(float* out, int sel) {
  __shared__ f4 tab[512];
  tab[threadIdx.x] = (f4){(float)threadIdx.x, 1, 2, 3}; tab[threadIdx.x + 256] = tab[threadIdx.x];
  __syncthreads();
  f4 acc = {0, 0, 0, 0};
  unsigned base = (unsigned)(size_t)tab + ((threadIdx.x & 63) + sel) * 16u;
  for (int it = 0; it < ITER; ++it) {
    f4 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[i]) : "v"(base), "i"((i & 3) * 1024));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("" ::"v"(v[i]));
    acc += v[3];
  }
  if (acc.x == 12345.678f) out[0] = acc.x;
}
__global__ void k_readlane(float* out, int a) {
  int v = threadIdx.x * 3 + a; int s[16];
  for (int i = 0; i < 16; ++i) s[i] = 0;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_readlane_b32 %0, %1, %2" : "=s"(s[i]) : "v"(v), "i"(i));
    asm volatile("" ::"s"(s[0]), "s"(s[5]), "s"(s[15]));
  }
  int t = 0; for (int i = 0; i < 16; ++i) t += s[i];
  if (t == 123456789) out[0] = (float)t;
}
// ds_max_u32 without return, one address per lane (conflict-free) — the key-buffer write of a span rasteriser
__global__ void k_ldsmax(float* out, int a) {
  __shared__ unsigned fb[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) fb[i] = 0;
  __syncthreads();
  unsigned key = a + threadIdx.x;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) atomicMax(&fb[(threadIdx.x + i * 256 + it) & 4095], key + i);
  }
  __syncthreads();
  if (fb[threadIdx.x] == 0xdeadbeefu) out[0] = 1.0f;
}
// ds_write_b8 per lane (LDS framebuffer byte stores)
__global__ void k_ldsw8(float* out, int a) {
  __shared__ unsigned char fb[16384];
  unsigned key = a + threadIdx.x;
  unsigned base = (unsigned)(size_t)fb + threadIdx.x;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("ds_write_b8 %0, %1 offset:%2" ::"v"(base), "v"(key), "i"(i * 256) : "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (fb[threadIdx.x] == 0xee && a == 77777) out[0] = 1.0f;
}
// v_cmp + v_cndmask + v_min3 chain typical of the coverage test tail
__global__ void k_cmpsel(float* out, float a, float b) {
  float r[8]; int best[8];
  for (int i = 0; i < 8; ++i) { r[i] = threadIdx.x * 0.001f + i; best[i] = -1; }
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
      best[i] = (r[i] >= 0.0f && it > best[i]) ? it : best[i];
    }
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += r[i] + best[i];
  if (s == 12345.678f) out[0] = s;
}

template <typename F> static double run(const char* name, F launch, int wps, double inst_per_wave) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  launch(256 * wps);
  hipDeviceSynchronize();
  hipEventRecord(a);
  launch(256 * wps);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  const double waves = 256.0 * wps * 4;
  const double per_cu_per_s = waves * inst_per_wave / 256.0 / (ms * 1e-3);
  printf("%-12s waves/SIMD %d  %8.3f ms  %7.3f wave-instr/ns/CU  -> %6.2f cycles@2.4GHz per wave-instr per SIMD\n", name, wps, ms,
         per_cu_per_s * 1e-9, 2.4 / (per_cu_per_s * 1e-9 / 4));
  return ms;
}

int main() {
  float* out; hipMalloc(&out, 64);
  const double n16 = 16.0 * ITER;
  for (int wps : {1, 2, 4, 6, 8}) {
    run("v_fma_f32", [&](int g) { hipLaunchKernelGGL(k_fma, dim3(g), dim3(256), 0, 0, out, 1.0001f, 0.5f); }, wps, n16);
    run("v_pk_fma_f32", [&](int g) { hipLaunchKernelGGL(k_pkfma, dim3(g), dim3(256), 0, 0, out, 1.0001f, 0.5f); }, wps, n16);
    run("s_add_u32", [&](int g) { hipLaunchKernelGGL(k_salu, dim3(g), dim3(256), 0, 0, out, 3); }, wps, n16);
    run("fma+salu", [&](int g) { hipLaunchKernelGGL(k_mix, dim3(g), dim3(256), 0, 0, out, 1.0001f, 0.5f, 3); }, wps, n16);
    run("lds_bcast128", [&](int g) { hipLaunchKernelGGL(k_ldsbcast, dim3(g), dim3(256), 0, 0, out, 3); }, wps, n16);
    run("lds_vec128", [&](int g) { hipLaunchKernelGGL(k_ldsvec, dim3(g), dim3(256), 0, 0, out, 3); }, wps, n16);
    run("v_readlane", [&](int g) { hipLaunchKernelGGL(k_readlane, dim3(g), dim3(256), 0, 0, out, 3); }, wps, n16);
    run("ds_max_u32", [&](int g) { hipLaunchKernelGGL(k_ldsmax, dim3(g), dim3(256), 0, 0, out, 3); }, wps, n16);
    run("ds_write_b8", [&](int g) { hipLaunchKernelGGL(k_ldsw8, dim3(g), dim3(256), 0, 0, out, 3); }, wps, n16);
    run("fma+cmp+sel", [&](int g) { hipLaunchKernelGGL(k_cmpsel, dim3(g), dim3(256), 0, 0, out, 1.0001f, 0.5f); }, wps, 8.0 * ITER);
    printf("\n");
  }
  return 0;
}
