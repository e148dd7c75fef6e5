// false_sharing.hip — gfx950: two kernels on two streams, running at the same time on different XCDs, write INTERLEAVED elements of one array
// with plain stores and keep running for a while (their dirty lines sit in two L2s).  Are both kernels' elements there afterwards?
// For element sizes 1, 2 and 4 bytes.  (The step's verdict bytes, mcr_kernels.h: part / part_next, are written like this.)  Not product code.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/_bin/false_sharing tools/ubench/false_sharing.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <typename T>
__global__ void k_write(T* a, int n, int who, T value, int* flag, int spin) {
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) * 2 + who;
  if (i < n) a[i] = value;
  // stay resident: the line stays dirty in this XCD's L2 while the other kernel writes its elements
  if (threadIdx.x == 0) { __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); for (int k = 0; k < spin; ++k) __builtin_amdgcn_s_sleep(64); }
}
template <typename T> int run(const char* name) {
  const int n = 1 << 16;
  T* d; int* flag; hipMalloc(&d, n * sizeof(T)); hipMalloc(&flag, 4);
  hipStream_t s0, s1; hipStreamCreateWithFlags(&s0, hipStreamNonBlocking); hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  int lost_total = 0;
  for (int rep = 0; rep < 50; ++rep) {
    hipMemset(d, 0, n * sizeof(T)); hipMemset(flag, 0, 4); hipDeviceSynchronize();
    hipLaunchKernelGGL(k_write<T>, dim3(n / 2 / 64), dim3(64), 0, s0, d, n, 0, (T)1, flag, 2000);
    hipLaunchKernelGGL(k_write<T>, dim3(n / 2 / 64), dim3(64), 0, s1, d, n, 1, (T)2, flag, 2000);
    hipDeviceSynchronize();
    std::vector<T> h(n); hipMemcpy(h.data(), d, n * sizeof(T), hipMemcpyDeviceToHost);
    int lost = 0; for (int i = 0; i < n; ++i) if (h[i] != (T)(1 + (i & 1))) ++lost;
    lost_total += lost;
  }
  printf("%-8s elements written by two concurrent kernels, interleaved: %d of %d lost over 50 runs\n", name, lost_total, 50 * n);
  hipFree(d); hipFree(flag); return lost_total;
}
int main() { run<unsigned char>("1-byte"); run<unsigned short>("2-byte"); run<unsigned int>("4-byte"); return 0; }
