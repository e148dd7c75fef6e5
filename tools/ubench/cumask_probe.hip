// Where do the workgroups of a stream created with hipExtStreamCreateWithCUMask land?  Prints, per mask, the set of
// (XCC, SE, CU) ids seen by 2048 probe workgroups.  hipcc --offload-arch=gfx950 -O2 cumask_probe.hip -o cumask_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <vector>
__global__ void probe(uint32_t* out) {
  if (threadIdx.x == 0) {
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);
    out[blockIdx.x] = (xcc << 16) | ((hw >> 8) & 0xff) | (((hw >> 13) & 7) << 8);   // xcc | se | sh:cu
    for (volatile int i = 0; i < 20000; ++i) {}
  }
}
static void run(const char* name, const std::vector<uint32_t>& mask) {
  hipStream_t s;
  if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: create failed\n", name); return; }
  const int n = 2048;
  uint32_t* d; hipMalloc(&d, n * 4);
  hipLaunchKernelGGL(probe, dim3(n), dim3(64), 0, s, d);
  hipStreamSynchronize(s);
  std::vector<uint32_t> h(n); hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
  std::set<uint32_t> u(h.begin(), h.end());
  printf("%s: %zu distinct CUs:", name, u.size());
  int k = 0; for (uint32_t v : u) { if (k++ < 24) printf(" x%u.se%u.cu%02x", v >> 16, (v >> 8) & 7, v & 0xff); }
  printf("\n");
  hipFree(d); hipStreamDestroy(s);
}
int main() {
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0); printf("CUs %d\n", pr.multiProcessorCount);
  std::vector<uint32_t> m(8, 0);
  m[0] = 0xff; run("bits 0-7", m);
  m.assign(8, 0); m[0] = 0x1; run("bit 0", m);
  m.assign(8, 0); m[0] = 0x100; run("bit 8", m);
  m.assign(8, 0); m[1] = 0x1; run("bit 32", m);
  m.assign(8, 0); m[7] = 0xff000000u; run("bits 248-255", m);
  m.assign(8, 0xffffffffu); m[0] = 0xffffff00u; run("all but 0-7", m);
  m.assign(8, 0xffffffffu); run("all", m);
  return 0;
}
