// Issue cost (cycles per wave-instruction per SIMD) of the individual float64/int ops the
// exp2 sequence uses.  hipcc --offload-arch=gfx950 -O3 ubench_ops.hip -o ubench_ops
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int ITERS = 4096;
#define KERNEL(name, body)                                                        \
  __global__ void name(double* out, double a, int s) {                           \
    double x0 = threadIdx.x + 0.25, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;      \
    int n0 = s, n1 = s + 1, n2 = s + 2, n3 = s + 3;                              \
    for (int i = 0; i < ITERS; ++i) { body }                                      \
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + n0 + n1 + n2 + n3; \
  }
KERNEL(k_fma, x0 = fma(x0, a, a); x1 = fma(x1, a, a); x2 = fma(x2, a, a); x3 = fma(x3, a, a);)
KERNEL(k_add, x0 += a; x1 += a; x2 += a; x3 += a;)
KERNEL(k_mul, x0 *= a; x1 *= a; x2 *= a; x3 *= a;)
KERNEL(k_rndne, x0 = __builtin_rint(x0 * a); x1 = __builtin_rint(x1 * a); x2 = __builtin_rint(x2 * a); x3 = __builtin_rint(x3 * a);)
KERNEL(k_ldexp, x0 = __builtin_amdgcn_ldexp(x0, n0); x1 = __builtin_amdgcn_ldexp(x1, n1); x2 = __builtin_amdgcn_ldexp(x2, n2); x3 = __builtin_amdgcn_ldexp(x3, n3);)
KERNEL(k_cvt, n0 += (int)x0; n1 += (int)x1; n2 += (int)x2; n3 += (int)x3; x0 += a; x1 += a; x2 += a; x3 += a;)
KERNEL(k_rcp, x0 = __builtin_amdgcn_rcp(x0); x1 = __builtin_amdgcn_rcp(x1); x2 = __builtin_amdgcn_rcp(x2); x3 = __builtin_amdgcn_rcp(x3);)
KERNEL(k_iadd, n0 += n1; n1 += n2; n2 += n3; n3 += n0;)
KERNEL(k_fmadep, x0 = fma(x0, a, a); x0 = fma(x0, a, a); x0 = fma(x0, a, a); x0 = fma(x0, a, a);)
KERNEL(k_fmadep2, x0 = fma(x0, a, a); x1 = fma(x1, a, a); x0 = fma(x0, a, a); x1 = fma(x1, a, a);)
template <typename F> double time_ms(F launch) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  launch(); CHECK(hipDeviceSynchronize()); float best = 1e30f;
  for (int r = 0; r < 5; ++r) { CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
  return best;
}
int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  const int CU = p.multiProcessorCount; double* out; CHECK(hipMalloc(&out, sizeof(double) * 256 * CU * 8));
  for (int wps : {1, 2, 4}) {  // waves per SIMD
    const int blocks = CU * wps;
#define RUN(name, nops) { double t = time_ms([&] { hipLaunchKernelGGL(name, dim3(blocks), dim3(256), 0, 0, out, 1.0000001, 3); }); \
      printf("waves/SIMD=%d %-10s %7.2f cyc/instr (assuming 2.4 GHz)\n", wps, #name, t * 1e-3 * 2.4e9 / (ITERS * (double)(nops) * wps)); }
    RUN(k_fma, 4) RUN(k_add, 4) RUN(k_mul, 4) RUN(k_rndne, 8) RUN(k_ldexp, 4) RUN(k_cvt, 12) RUN(k_rcp, 4) RUN(k_iadd, 4) RUN(k_fmadep, 4) RUN(k_fmadep2, 4)
  }
  return 0;
}
