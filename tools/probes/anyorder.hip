#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1;} } while (0)
__device__ unsigned long long g_t[4];
__device__ unsigned long long g_end[4096];
__global__ void A(int run_us) {
  const unsigned long long t0 = wall_clock64();
  if (blockIdx.x == 0 && threadIdx.x == 0) g_t[0] = t0;
  while (wall_clock64() - t0 < (unsigned long long)run_us * 100) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) g_end[blockIdx.x] = wall_clock64();
}
__global__ void B(double* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) g_t[2] = wall_clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = 1.0;
}
int main() {
  double* out; CHECK(hipMalloc(&out, 1 << 20));
  hipStream_t st; CHECK(hipStreamCreate(&st));
  for (int flags : {0, (int)hipExtAnyOrderLaunch}) {
    for (int rep = 0; rep < 4; ++rep) {
      CHECK(hipDeviceSynchronize());
      int us = 40;
      hipLaunchKernelGGL(A, dim3(256), dim3(256), 0, st, us);
      void* args[] = {&out};
      hipError_t e = hipExtLaunchKernel((const void*)B, dim3(16), dim3(256), args, 0, st, nullptr, nullptr, flags);
      if (e != hipSuccess) { printf("hipExtLaunchKernel flags=%d: %s\n", flags, hipGetErrorString(e)); break; }
      CHECK(hipStreamSynchronize(st));
      unsigned long long t[4]; CHECK(hipMemcpyFromSymbol(t, HIP_SYMBOL(g_t), sizeof(t)));
      std::vector<unsigned long long> ev(256); CHECK(hipMemcpyFromSymbol(ev.data(), HIP_SYMBOL(g_end), sizeof(unsigned long long) * 256));
      const unsigned long long aend = *std::max_element(ev.begin(), ev.end());
      printf("flags=%d: B's first workgroup starts %.1f us after A's first, A's last ends at %.1f us\n", flags, ((double)t[2] - (double)t[0]) / 100.0, ((double)aend - (double)t[0]) / 100.0);
    }
  }
  return 0;
}
