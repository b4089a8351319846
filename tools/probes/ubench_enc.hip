// Does the ENCODING of a float64 vector instruction change how fast a wave can issue it on gfx950?
// tools/probes/ubench_valu_issue.hip found one v_fma_f64 (VOP3, 8 bytes) per 5.9 / 5.12 / 4.65 / 4.4 cycles at 1 / 2 / 3 / 4
// waves per SIMD.  Here the same chains as
//   0: v_fma_f64 d, a, b, d        (VOP3, 8 bytes, three VGPR-pair sources)
//   1: v_fmac_f64_e32 d, a, b      (VOP2, 4 bytes, two VGPR-pair sources + the destination as addend)
//   2: v_fmac_f64_e32 d, s, b      (VOP2, SGPR pair as src0)
//   3: v_fma_f64 d, s, b, d        (VOP3, SGPR pair)
//   4: v_fma_f64 d, d, b, c        (VOP3, Horner shape: the running value is a multiplicand)
//   5: v_mul_f64 / v_add_f64 alternating (VOP3)
// hipcc --offload-arch=gfx950 -O3 ubench_enc.hip -o ubench_enc
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ITERS = 256, UNR = 8, C = 8;

template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, double a, double b) {
  double x[C];
#pragma unroll
  for (int c = 0; c < C; ++c) x[c] = threadIdx.x + c;
  double av = a + 1e-12 * threadIdx.x, bv = b + threadIdx.x, cv = 0.5 * b;
  asm volatile("" : "+v"(av), "+v"(bv), "+v"(cv));
  double as = a;  // uniform: SGPR pair
  asm volatile("" : "+s"(as));
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNR; ++u)
#pragma unroll
      for (int c = 0; c < C; ++c) {
        if (MODE == 0) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(x[c]) : "v"(av), "v"(bv));
        if (MODE == 1) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(x[c]) : "v"(av), "v"(bv));
        if (MODE == 2) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(x[c]) : "s"(as), "v"(bv));
        if (MODE == 3) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(x[c]) : "s"(as), "v"(bv));
        if (MODE == 4) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[c]) : "v"(av), "v"(cv));
        if (MODE == 5) {
          if (u & 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[c]) : "v"(av));
          else asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[c]) : "v"(cv));
        }
      }
  }
  double s = 0;
#pragma unroll
  for (int c = 0; c < C; ++c) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(int waves_per_simd, double* d_out, const char* what) {
  const int grid = 256 * waves_per_simd;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) k<MODE><<<grid, 256>>>(d_out, 1.0000001, 1e-9);
  CHECK(hipEventRecord(e0));
  const int reps = 10;
  for (int rep = 0; rep < reps; ++rep) k<MODE><<<grid, 256>>>(d_out, 1.0000001, 1e-9);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double per_launch_s = ms * 1e-3 / reps;
  const double instr_per_simd = (double)ITERS * UNR * C * waves_per_simd;
  printf("%-44s waves/SIMD %d : %7.1f us  %5.2f cycles per instruction and SIMD (2.4 GHz)\n", what, waves_per_simd,
         per_launch_s * 1e6, per_launch_s * 2.4e9 / instr_per_simd);
}

int main() {
  double* d_out;
  CHECK(hipMalloc(&d_out, sizeof(double) * 256 * 256 * 8));
  for (int w : {1, 2, 3, 4}) {
    run<0>(w, d_out, "v_fma_f64 d,a,b,d (VOP3, vgpr)");
    run<1>(w, d_out, "v_fmac_f64_e32 d,a,b (VOP2, vgpr)");
    run<2>(w, d_out, "v_fmac_f64_e32 d,s,b (VOP2, sgpr)");
    run<3>(w, d_out, "v_fma_f64 d,s,b,d (VOP3, sgpr)");
    run<4>(w, d_out, "v_fma_f64 d,d,b,c (VOP3, Horner)");
    run<5>(w, d_out, "v_mul_f64 / v_add_f64 (VOP3)");
  }
  return 0;
}
