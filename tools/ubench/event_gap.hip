// event_gap.hip — what a dependency between two kernels costs on gfx950 / ROCm 7, by idiom.  Kernel A writes the constant-rate
// clock (s_memrealtime, 100 MHz) when it ends, kernel B when it starts; every case is queued behind a 300 us spin kernel so the
// host is out of the picture.  build: hipcc --offload-arch=gfx950 -O2 -o /tmp/event_gap tools/ubench/event_gap.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void k_spin(long long ticks) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8); }
__global__ void k_a(long long* out, long long ticks) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2); if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = wall_clock64(); }
__global__ void k_a_store(long long* out, long long ticks, unsigned* word, unsigned val) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2); if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = wall_clock64(); __hip_atomic_store(word, val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); } }
__global__ void k_postw(unsigned* word, unsigned val) { __hip_atomic_store(word, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void k_awaitw(unsigned* word, unsigned val) { for (int i = 0; i < (1 << 22) && __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != val; ++i) __builtin_amdgcn_s_sleep(8); }
__global__ void k_b(long long* out) { if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = wall_clock64(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
  hipStream_t s1, s2, s3; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s3, hipStreamNonBlocking));
  long long* d; CK(hipMalloc(&d, 64)); long long hst[2];
  unsigned* sig = nullptr; const bool have_sig = hipExtMallocWithFlags((void**)&sig, 64, hipMallocSignalMemory) == hipSuccess;
  unsigned* dw = nullptr; CK(hipMalloc(&dw, 64)); unsigned token = 0;
  if (!have_sig) { (void)hipGetLastError(); printf("(no signal memory: the stream memory op cases are skipped)\n"); }
  const char* names[] = {"same stream, nothing between", "same stream: record(ev) between", "same stream: record(ev) + wait(done event of s2) between",
                         "hop: s1 A, record; s2 wait, B", "hop: s1 A with stopEvent (hipExtLaunchKernelGGL); s2 wait, B", "same stream: 2 waits on done events between",
                         "same stream: 3 records between", "hop, ReleaseToDevice events", "two hops: s1 A,record; s2 wait,record; s3 wait,B",
                         "same stream: A with stopEvent, then B", "join of two: s1 A; s2 A'; s3 waits both, B (gap after the later)",
                         "same stream: wait on s2's event, complete long before A ends but not at enqueue", "same stream: two such waits (s2, s3)",
                         "same stream: two such waits + record", "same stream: wait on s2's kernel ending 5 us BEFORE A ends", "same stream: wait on s2's kernel ending 5 us AFTER A ends (gap after it)",
                         "hop by stream memory ops: s1 A, hipStreamWriteValue32; s2 hipStreamWaitValue32, B", "hop by kernels: s1 A, post kernel; s2 await kernel, B",
                         "hop: A's last thread stores the word; s2 hipStreamWaitValue32, B", "same stream: hipStreamWaitValue32 on a word set long ago between A and B",
                         "hop: s1 A, hipStreamWriteValue32; s2 await kernel (spinning since before A), B"};
  for (int variant = 0; variant < 21; ++variant) {
    std::vector<double> gaps;
    for (int rep = 0; rep < 40; ++rep) {
      const unsigned fl = hipEventDisableTiming | (variant == 7 ? hipEventReleaseToDevice : 0u);
      hipEvent_t e0, e1, e2, e3; CK(hipEventCreateWithFlags(&e0, fl)); CK(hipEventCreateWithFlags(&e1, fl)); CK(hipEventCreateWithFlags(&e2, fl)); CK(hipEventCreateWithFlags(&e3, fl));
      // a done event on s2 / s3
      hipLaunchKernelGGL(k_b, dim3(1), dim3(64), 0, s2, d + 4); CK(hipEventRecord(e2, s2)); CK(hipStreamSynchronize(s2));
      hipLaunchKernelGGL(k_b, dim3(1), dim3(64), 0, s3, d + 4); CK(hipEventRecord(e3, s3)); CK(hipStreamSynchronize(s3));
      // block all streams behind the spin on s1
      hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s1, 30000LL); CK(hipEventRecord(e0, s1)); CK(hipStreamWaitEvent(s2, e0, 0)); CK(hipStreamWaitEvent(s3, e0, 0));
      const long long a_ticks = 2000;   // 20 us
      switch (variant) {
        case 0: hipLaunchKernelGGL(k_a, dim3(256), dim3(64), 0, s1, d, a_ticks); hipLaunchKernelGGL(k_b, dim3(256), dim3(64), 0, s1, d); break;
        case 1: hipLaunchKernelGGL(k_a, dim3(256), dim3(64), 0, s1, d, a_ticks); CK(hipEventRecord(e1, s1)); hipLaunchKernelGGL(k_b, dim3(256), dim3(64), 0, s1, d); break;
        case 2: hipLaunchKernelGGL(k_a, dim3(256), dim3(64), 0, s1, d, a_ticks); CK(hipEventRecord(e1, s1)); CK(hipStreamWaitEvent(s1, e2, 0)); hipLaunchKernelGGL(k_b, dim3(256), dim3(64), 0, s1, d); break;
        case 3: case 7: hipLaunchKernelGGL(k_a, dim3(256), dim3(64), 0, s1, d, a_ticks); CK(hipEventRecord(e1, s1)); CK(hipStreamWaitEvent(s2, e1, 0)); hipLaunchKernelGGL(k_b, dim3(256), dim3(64), 0, s2, d); break;
        case 4: hipExtLaunchKernelGGL(k_a, dim3(256), dim3(64), 0, s1, nullptr, e1, 0, d, a_ticks); CK(hipStreamWaitEvent(s2, e1, 0)); hipLaunchKernelGGL(k_b, dim3(256), dim3(64), 0, s2, d); break;
        case 5: hipLaunchKernelGGL(k_a, dim3(256), dim3(64), 0, s1, d, a_ticks); CK(hipStreamWaitEvent(s1, e2, 0)); CK(hipStreamWaitEvent(s1, e3, 0)); hipLaunchKernelGGL(k_b, dim3(256), dim3(64), 0, s1, d); break;
        case 6: { hipEvent_t x, y; CK(hipEventCreateWithFlags(&x, fl)); CK(hipEventCreateWithFlags(&y, fl)); hipLaunchKernelGGL(k_a, dim3(256), dim3(64), 0, s1, d, a_ticks); CK(hipEventRecord(e1, s1)); CK(hipEventRecord(x, s1)); CK(hipEventRecord(y, s1)); hipLaunchKernelGGL(k_b, dim3(256), dim3(64), 0, s1, d); CK(hipDeviceSynchronize()); CK(hipEventDestroy(x)); CK(hipEventDestroy(y)); break; }
        case 8: hipLaunchKernelGGL(k_a, dim3(256), dim3(64), 0, s1, d, a_ticks); CK(hipEventRecord(e1, s1)); CK(hipStreamWaitEvent(s2, e1, 0)); CK(hipEventRecord(e2, s2)); CK(hipStreamWaitEvent(s3, e2, 0)); hipLaunchKernelGGL(k_b, dim3(256), dim3(64), 0, s3, d); break;
        case 9: hipExtLaunchKernelGGL(k_a, dim3(256), dim3(64), 0, s1, nullptr, e1, 0, d, a_ticks); hipLaunchKernelGGL(k_b, dim3(256), dim3(64), 0, s1, d); break;
        case 10: hipLaunchKernelGGL(k_a, dim3(256), dim3(64), 0, s1, d, a_ticks); CK(hipEventRecord(e1, s1)); hipLaunchKernelGGL(k_a, dim3(256), dim3(64), 0, s2, d + 2, a_ticks - 500); CK(hipEventRecord(e2, s2));
                 CK(hipStreamWaitEvent(s3, e1, 0)); CK(hipStreamWaitEvent(s3, e2, 0)); hipLaunchKernelGGL(k_b, dim3(256), dim3(64), 0, s3, d); break;
        case 11: hipExtLaunchKernelGGL(k_b, dim3(1), dim3(64), 0, s2, nullptr, e2, 0, d + 4); hipLaunchKernelGGL(k_a, dim3(256), dim3(64), 0, s1, d, a_ticks); CK(hipStreamWaitEvent(s1, e2, 0)); hipLaunchKernelGGL(k_b, dim3(256), dim3(64), 0, s1, d); break;
        case 12: case 13: hipExtLaunchKernelGGL(k_b, dim3(1), dim3(64), 0, s2, nullptr, e2, 0, d + 4); hipExtLaunchKernelGGL(k_b, dim3(1), dim3(64), 0, s3, nullptr, e3, 0, d + 4);
                 hipLaunchKernelGGL(k_a, dim3(256), dim3(64), 0, s1, d, a_ticks); CK(hipStreamWaitEvent(s1, e2, 0)); CK(hipStreamWaitEvent(s1, e3, 0)); if (variant == 13) CK(hipEventRecord(e1, s1)); hipLaunchKernelGGL(k_b, dim3(256), dim3(64), 0, s1, d); break;
        case 14: hipExtLaunchKernelGGL(k_a, dim3(64), dim3(64), 0, s2, nullptr, e2, 0, d + 2, a_ticks - 500); hipLaunchKernelGGL(k_a, dim3(256), dim3(64), 0, s1, d, a_ticks); CK(hipStreamWaitEvent(s1, e2, 0)); hipLaunchKernelGGL(k_b, dim3(256), dim3(64), 0, s1, d); break;
        case 15: hipExtLaunchKernelGGL(k_a, dim3(64), dim3(64), 0, s2, nullptr, e2, 0, d, a_ticks); hipLaunchKernelGGL(k_a, dim3(256), dim3(64), 0, s1, d + 2, a_ticks - 500); CK(hipStreamWaitEvent(s1, e2, 0)); hipLaunchKernelGGL(k_b, dim3(256), dim3(64), 0, s1, d); break;
        case 16: case 18: case 19: case 20:
          if (!have_sig) break;
          ++token;
          if (variant == 19) { CK(hipStreamWriteValue32(s2, sig, token, 0)); CK(hipStreamSynchronize(s2)); hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s1, 1000LL); }
          if (variant == 20) hipLaunchKernelGGL(k_awaitw, dim3(1), dim3(64), 0, s2, sig, token);
          if (variant == 18) hipLaunchKernelGGL(k_a_store, dim3(256), dim3(64), 0, s1, d, a_ticks, sig, token);
          else hipLaunchKernelGGL(k_a, dim3(256), dim3(64), 0, s1, d, a_ticks);
          if (variant == 16 || variant == 20) CK(hipStreamWriteValue32(s1, sig, token, 0));
          if (variant == 19) { CK(hipStreamWaitValue32(s1, sig, token, hipStreamWaitValueEq, 0xffffffffu)); hipLaunchKernelGGL(k_b, dim3(256), dim3(64), 0, s1, d); }
          else { if (variant != 20) CK(hipStreamWaitValue32(s2, sig, token, hipStreamWaitValueEq, 0xffffffffu)); hipLaunchKernelGGL(k_b, dim3(256), dim3(64), 0, s2, d); }
          break;
        case 17: ++token; hipLaunchKernelGGL(k_awaitw, dim3(1), dim3(64), 0, s2, dw, token); hipLaunchKernelGGL(k_a, dim3(256), dim3(64), 0, s1, d, a_ticks); hipLaunchKernelGGL(k_postw, dim3(1), dim3(1), 0, s1, dw, token);
                 hipLaunchKernelGGL(k_b, dim3(256), dim3(64), 0, s2, d); break;
      }
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(hst, d, 16, hipMemcpyDeviceToHost));
      if (rep >= 5 && (have_sig || !(variant == 16 || variant >= 18))) gaps.push_back((hst[1] - hst[0]) / 100.0);
      CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1)); CK(hipEventDestroy(e2)); CK(hipEventDestroy(e3));
    }
    if (gaps.empty()) continue;
    std::sort(gaps.begin(), gaps.end());
    printf("%-75s gap median %6.2f us  min %6.2f  p90 %6.2f\n", names[variant], gaps[gaps.size() / 2], gaps[0], gaps[gaps.size() * 9 / 10]);
  }
  return 0;
}
