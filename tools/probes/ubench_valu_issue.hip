// How fast can ONE or TWO waves per SIMD issue v_fma_f64 on gfx950?  (The entropy kernel runs two waves per SIMD; its
// VALU issue is busy 71-78 % of the time.)  Long unrolled bodies, so loop overhead is nothing; C independent chains;
// operands all-VGPR or with one SGPR pair.   hipcc --offload-arch=gfx950 -O3 ubench_valu_issue.hip -o ubench_valu_issue
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ITERS = 512, UNR = 16;

template <int C, bool SG>
__global__ __launch_bounds__(256) void k_fma(double* out, double a, double b) {
  double x[C];
#pragma unroll
  for (int c = 0; c < C; ++c) x[c] = threadIdx.x + c;
  double av = a;
  if (!SG) asm volatile("" : "+v"(av));  // a per-lane copy: all operands VGPRs
  double bv = b + threadIdx.x;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNR; ++u)
#pragma unroll
      for (int c = 0; c < C; ++c) x[c] = fma(x[c], av, bv);
  }
  double s = 0;
#pragma unroll
  for (int c = 0; c < C; ++c) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int C, bool SG>
void run(int waves_per_simd, double* d_out) {
  // one workgroup of 256 threads = one wave per SIMD; `waves_per_simd` workgroups per CU via the grid (256 CUs)
  const int grid = 256 * waves_per_simd;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) k_fma<C, SG><<<grid, 256>>>(d_out, 1.0000001, 1e-9);
  CHECK(hipEventRecord(e0));
  const int reps = 10;
  for (int rep = 0; rep < reps; ++rep) k_fma<C, SG><<<grid, 256>>>(d_out, 1.0000001, 1e-9);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double per_launch_s = ms * 1e-3 / reps;
  const double instr_per_simd = (double)ITERS * UNR * C * waves_per_simd;
  const double cyc = per_launch_s * 2.4e9 / instr_per_simd;
  const double tflops = 2.0 * 64 * instr_per_simd * 1024 / per_launch_s / 1e12;
  printf("chains %2d  %s  waves/SIMD %d : %7.1f us  %5.2f cycles per instruction and SIMD (2.4 GHz)  %5.1f TFLOP/s\n", C,
         SG ? "sgpr" : "vgpr", waves_per_simd, per_launch_s * 1e6, cyc, tflops);
}

int main() {
  double* d_out;
  CHECK(hipMalloc(&d_out, sizeof(double) * 256 * 256 * 8));
  for (int w : {1, 2, 3, 4}) {
    run<1, false>(w, d_out);
    run<2, false>(w, d_out);
    run<4, false>(w, d_out);
    run<8, false>(w, d_out);
    run<8, true>(w, d_out);
    run<2, true>(w, d_out);
  }
  return 0;
}
