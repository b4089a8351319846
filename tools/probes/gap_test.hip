// gap between the last workgroup of kernel A and the first of kernel B (same stream), B queued while A runs;
// A either only spins or also writes `mb` megabytes (plain or nontemporal stores)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1;} } while (0)
__device__ unsigned long long g_t[4];
__device__ unsigned long long g_end[4096];
template <int NT>
__global__ void A(double* buf, size_t n_per_block, int spin_us) {
  const unsigned long long t0 = wall_clock64();
  if (blockIdx.x == 0 && threadIdx.x == 0) g_t[0] = t0;
  double* p = buf + (size_t)blockIdx.x * n_per_block;
  for (size_t i = threadIdx.x; i < n_per_block; i += blockDim.x) {
    if (NT) __builtin_nontemporal_store((double)i, p + i); else p[i] = (double)i;
  }
  while (wall_clock64() - t0 < (unsigned long long)spin_us * 100) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) g_end[blockIdx.x] = wall_clock64();
}
__global__ void B(const double* buf, double* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) g_t[2] = wall_clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = buf[threadIdx.x];
}
int main() {
  double *buf, *out; const size_t NB = 1024;
  CHECK(hipMalloc(&buf, 64ull << 20)); CHECK(hipMalloc(&out, 1 << 20));
  hipStream_t st; CHECK(hipStreamCreate(&st));
  for (int mb : {0, 4, 40}) for (int nt = 0; nt < 2; ++nt) {
    std::vector<double> gaps;
    for (int rep = 0; rep < 12; ++rep) {
      unsigned long long init[4] = {0, 0, ~0ull, 0};
      CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_t), init, sizeof(init)));
      CHECK(hipDeviceSynchronize());
      const size_t npb = ((size_t)mb << 20) / 8 / NB;
      if (nt) hipLaunchKernelGGL(A<1>, dim3(NB), dim3(256), 0, st, buf, npb, 40);
      else hipLaunchKernelGGL(A<0>, dim3(NB), dim3(256), 0, st, buf, npb, 40);
      hipLaunchKernelGGL(B, dim3(64), dim3(256), 0, st, buf, out);
      CHECK(hipStreamSynchronize(st));
      unsigned long long t[4]; CHECK(hipMemcpyFromSymbol(t, HIP_SYMBOL(g_t), sizeof(t)));
      std::vector<unsigned long long> ev(NB); CHECK(hipMemcpyFromSymbol(ev.data(), HIP_SYMBOL(g_end), sizeof(unsigned long long) * NB));
      t[1] = *std::max_element(ev.begin(), ev.end());
      if (rep >= 2) gaps.push_back(((double)t[2] - (double)t[1]) / 100.0);
    }
    std::sort(gaps.begin(), gaps.end());
    printf("A writes %2d MB (%s stores): gap last-WG-of-A -> first-WG-of-B median %.2f us (min %.2f max %.2f)\n", mb, nt ? "nontemporal" : "plain", gaps[gaps.size() / 2], gaps.front(), gaps.back());
  }
  return 0;
}
