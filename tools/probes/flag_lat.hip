// How long does a system-scope flag store to pinned host memory take to reach the host while the
// rest of the GPU streams stores to HBM?  Block 0 waits `wait_us` from its start, then stores the
// flag; blocks 1.. stream `mb` MB (or nothing).  Host: time from launch call to flag seen.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1;} } while (0)
template <int MODE>  // 0: flag only, 1: plain stores, 2: nontemporal stores, 3: loads instead of stores
__global__ void K(unsigned long long* flag, unsigned long long seq, int wait_us, double* buf, size_t n_per_block, double* sink) {
  const unsigned long long t0 = wall_clock64();
  if (blockIdx.x == 0) {
    while (wall_clock64() - t0 < (unsigned long long)wait_us * 100) __builtin_amdgcn_s_sleep(4);
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  double* p = buf + (size_t)(blockIdx.x - 1) * n_per_block;
  double acc = 0;
  for (int rep = 0; rep < 4; ++rep)
    for (size_t i = threadIdx.x; i < n_per_block; i += blockDim.x) {
      if (MODE == 1) p[i] = (double)(i + rep);
      else if (MODE == 2) __builtin_nontemporal_store((double)(i + rep), p + i);
      else if (MODE == 3) acc += p[i];
    }
  if (MODE == 3 && acc == 12345.678) sink[0] = acc;
}
int main() {
  double *buf, *sink; CHECK(hipMalloc(&buf, 64ull << 20)); CHECK(hipMalloc(&sink, 8));
  unsigned long long* flag; CHECK(hipHostMalloc((void**)&flag, 64, hipHostMallocDefault));
  unsigned long long* dflag; CHECK(hipHostGetDevicePointer((void**)&dflag, flag, 0));
  hipStream_t st; CHECK(hipStreamCreate(&st));
  unsigned long long seq = 0;
  using clk = std::chrono::steady_clock;
  const int NB = 2048; const size_t npb = (40ull << 20) / 8 / NB;
  const char* names[4] = {"nothing else", "40 MB x4 plain stores", "40 MB x4 nontemporal stores", "40 MB x4 loads"};
  for (int mode = 0; mode < 4; ++mode) {
    std::vector<double> lat;
    for (int rep = 0; rep < 20; ++rep) {
      CHECK(hipDeviceSynchronize());
      ++seq;
      const auto t0 = clk::now();
      if (mode == 0) hipLaunchKernelGGL(K<0>, dim3(1 + NB), dim3(256), 0, st, dflag, seq, 15, buf, npb, sink);
      if (mode == 1) hipLaunchKernelGGL(K<1>, dim3(1 + NB), dim3(256), 0, st, dflag, seq, 15, buf, npb, sink);
      if (mode == 2) hipLaunchKernelGGL(K<2>, dim3(1 + NB), dim3(256), 0, st, dflag, seq, 15, buf, npb, sink);
      if (mode == 3) hipLaunchKernelGGL(K<3>, dim3(1 + NB), dim3(256), 0, st, dflag, seq, 15, buf, npb, sink);
      while (*(volatile unsigned long long*)flag != seq) __builtin_ia32_pause();
      const auto t1 = clk::now();
      if (rep >= 4) lat.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
    }
    std::sort(lat.begin(), lat.end());
    printf("%-30s launch call -> flag seen by the host: median %.1f us (min %.1f max %.1f)   [flag stored 15 us after block 0 starts]\n", names[mode], lat[lat.size() / 2], lat.front(), lat.back());
  }
  return 0;
}
