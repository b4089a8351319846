// When must the host issue a launch so that the kernel starts right behind a running one?
// A: 1024 blocks, block 0 raises a pinned flag at its start, every block spins `run_us`.  The host
// sees the flag, waits d microseconds, launches B (same stream).  Gap = first WG of B - last WG of A.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1;} } while (0)
__device__ unsigned long long g_t[4];
__device__ unsigned long long g_end[4096];
__global__ void A(volatile unsigned long long* flag, unsigned long long seq, int run_us) {
  const unsigned long long t0 = wall_clock64();
  if (blockIdx.x == 0 && threadIdx.x == 0) { g_t[0] = t0; __hip_atomic_store((unsigned long long*)flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
  while (wall_clock64() - t0 < (unsigned long long)run_us * 100) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) g_end[blockIdx.x] = wall_clock64();
}
__global__ void B(double* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) g_t[2] = wall_clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = 1.0;
}
int main() {
  double* out; const int NB = 1024;
  CHECK(hipMalloc(&out, 1 << 20));
  unsigned long long* flag; CHECK(hipHostMalloc((void**)&flag, 64, hipHostMallocDefault));
  unsigned long long* dflag; CHECK(hipHostGetDevicePointer((void**)&dflag, flag, 0));
  hipStream_t st; CHECK(hipStreamCreate(&st));
  unsigned long long seq = 0;
  using clk = std::chrono::steady_clock;
  for (int d : {0, 10, 20, 25, 30, 35, 40, 50}) {
    std::vector<double> gaps, lts;
    for (int rep = 0; rep < 14; ++rep) {
      CHECK(hipDeviceSynchronize());
      ++seq;
      hipLaunchKernelGGL(A, dim3(NB), dim3(256), 0, st, dflag, seq, 40);
      while (*(volatile unsigned long long*)flag != seq) __builtin_ia32_pause();
      const auto t0 = clk::now();
      while (std::chrono::duration<double, std::micro>(clk::now() - t0).count() < d) __builtin_ia32_pause();
      const auto t1 = clk::now();
      hipLaunchKernelGGL(B, dim3(64), dim3(256), 0, st, out);
      const auto t2 = clk::now();
      CHECK(hipStreamSynchronize(st));
      unsigned long long t[4]; CHECK(hipMemcpyFromSymbol(t, HIP_SYMBOL(g_t), sizeof(t)));
      std::vector<unsigned long long> ev(NB); CHECK(hipMemcpyFromSymbol(ev.data(), HIP_SYMBOL(g_end), sizeof(unsigned long long) * NB));
      const unsigned long long aend = *std::max_element(ev.begin(), ev.end());
      if (rep >= 2) { gaps.push_back(((double)t[2] - (double)aend) / 100.0); lts.push_back(std::chrono::duration<double, std::micro>(t2 - t1).count()); }
    }
    std::sort(gaps.begin(), gaps.end()); std::sort(lts.begin(), lts.end());
    printf("host waits %2d us after A's start flag, then launches B: B starts %.1f us after A's last workgroup (min %.1f max %.1f); hipLaunchKernel call %.1f us\n",
           d, gaps[gaps.size() / 2], gaps.front(), gaps.back(), lts[lts.size() / 2]);
  }
  return 0;
}
