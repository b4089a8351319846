// Micro-benchmarks that fix the float64 roofline numbers DESIGN.md quotes for gfx950:
// v_fma_f64 rate, v_mfma_f64_16x16x4_f64 rate, whether the two pipes overlap, and the
// cost of exp().   hipcc --offload-arch=gfx950 -O3 ubench_fp64.hip -o ubench_fp64
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                   \
  do {                                                                             \
    hipError_t e = (x);                                                            \
    if (e != hipSuccess) {                                                         \
      printf("%s: %s\n", #x, hipGetErrorString(e));                                \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

constexpr int ITERS = 2048;

__global__ void k_fma(double* out, double a, double b) {
  double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5,
         x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < ITERS; ++i) {
    x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b);
    x4 = fma(x4, a, b); x5 = fma(x5, a, b); x6 = fma(x6, a, b); x7 = fma(x7, a, b);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

__global__ void k_mfma(double* out, double a, double b) {
  double4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  double av = a + threadIdx.x, bv = b;
  for (int i = 0; i < ITERS; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c3, 0, 0, 0);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

__global__ void k_mfma4(double* out, double a, double b) {
  double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  double av = a + threadIdx.x, bv = b;
  for (int i = 0; i < ITERS; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, c3, 0, 0, 0);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3;
}

// waves alternate: even waves MFMA, odd waves FMA (same instruction counts as above)
__global__ void k_mixed(double* out, double a, double b) {
  const int wave = threadIdx.x >> 6;
  if (wave & 1) {
    double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5,
           x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < ITERS; ++i) {
      x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b);
      x4 = fma(x4, a, b); x5 = fma(x5, a, b); x6 = fma(x6, a, b); x7 = fma(x7, a, b);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  } else {
    double4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double av = a + threadIdx.x, bv = b;
    for (int i = 0; i < ITERS; ++i) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  }
}

// one wave interleaving both kinds of instruction (independent streams)
__global__ void k_interleave(double* out, double a, double b) {
  double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5,
         x6 = x0 + 6, x7 = x0 + 7;
  double4_t c0 = {0, 0, 0, 0}, c1 = c0;
  double av = a + threadIdx.x, bv = b;
  for (int i = 0; i < ITERS; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c0, 0, 0, 0);
    x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b);
    x4 = fma(x4, a, b); x5 = fma(x5, a, b); x6 = fma(x6, a, b); x7 = fma(x7, a, b);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c1, 0, 0, 0);
    x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b);
    x4 = fma(x4, a, b); x5 = fma(x5, a, b); x6 = fma(x6, a, b); x7 = fma(x7, a, b);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + c0[0] + c1[1];
}

__global__ void k_exp(double* out, double a, double b) {
  double x0 = -1e-3 * threadIdx.x, x1 = x0 - 1, x2 = x0 - 2, x3 = x0 - 3;
  double s = 0;
  for (int i = 0; i < ITERS / 4; ++i) {
    s += exp(x0) + exp(x1) + exp(x2) + exp(x3);
    x0 -= a; x1 -= a; x2 -= a; x3 -= a;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_log(double* out, double a, double b) {
  double x0 = 1.0 + 1e-3 * threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  double s = 0;
  for (int i = 0; i < ITERS / 4; ++i) {
    s += log(x0) + log(x1) + log(x2) + log(x3);
    x0 += a; x1 += a; x2 += a; x3 += a;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
double time_ms(F launch, int reps = 5) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  launch();
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    CHECK(hipEventRecord(e0));
    launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  printf("device %s %s CUs=%d clock=%d kHz\n", p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate);
  const int CU = p.multiProcessorCount;
  double* out;
  CHECK(hipMalloc(&out, sizeof(double) * 1024 * CU * 16));
  for (int wpc : {4, 8, 16}) {  // waves per CU
    const int threads = 256, blocks = CU * wpc / 4;
    const double nwaves = (double)blocks * 4;
    double t;
    t = time_ms([&] { hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(threads), 0, 0, out, 1.0000001, 1e-9); });
    double fl = nwaves * 64.0 * ITERS * 8 * 2;
    printf("waves/CU=%2d  v_fma_f64      : %8.3f ms  %7.2f TFLOP/s  (%.2f cyc/wave-instr @2.4GHz/SIMD)\n", wpc, t,
           fl / t / 1e9, t * 1e-3 * 2.4e9 / (ITERS * 8.0 * wpc / 4));
    t = time_ms([&] { hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(threads), 0, 0, out, 1.0000001, 1e-9); });
    fl = nwaves * ITERS * 4 * 2.0 * 16 * 16 * 4;
    printf("waves/CU=%2d  mfma_f64_16x16x4: %8.3f ms  %7.2f TFLOP/s  (%.2f cyc/mfma @2.4GHz/SIMD)\n", wpc, t,
           fl / t / 1e9, t * 1e-3 * 2.4e9 / (ITERS * 4.0 * wpc / 4));
    t = time_ms([&] { hipLaunchKernelGGL(k_mfma4, dim3(blocks), dim3(threads), 0, 0, out, 1.0000001, 1e-9); });
    fl = nwaves * ITERS * 4 * 2.0 * 4 * 4 * 4 * 4;
    printf("waves/CU=%2d  mfma_f64_4x4x4  : %8.3f ms  %7.2f TFLOP/s  (%.2f cyc/mfma @2.4GHz/SIMD)\n", wpc, t,
           fl / t / 1e9, t * 1e-3 * 2.4e9 / (ITERS * 4.0 * wpc / 4));
    t = time_ms([&] { hipLaunchKernelGGL(k_mixed, dim3(blocks), dim3(threads), 0, 0, out, 1.0000001, 1e-9); });
    fl = nwaves / 2 * (64.0 * ITERS * 8 * 2 + ITERS * 4 * 2.0 * 16 * 16 * 4);
    printf("waves/CU=%2d  mixed waves    : %8.3f ms  %7.2f TFLOP/s (sum of both)\n", wpc, t, fl / t / 1e9);
    t = time_ms([&] { hipLaunchKernelGGL(k_interleave, dim3(blocks), dim3(threads), 0, 0, out, 1.0000001, 1e-9); });
    fl = nwaves * (64.0 * ITERS * 16 * 2 + ITERS * 2 * 2.0 * 16 * 16 * 4);
    printf("waves/CU=%2d  interleaved    : %8.3f ms  %7.2f TFLOP/s (sum of both)\n", wpc, t, fl / t / 1e9);
    t = time_ms([&] { hipLaunchKernelGGL(k_exp, dim3(blocks), dim3(threads), 0, 0, out, 1e-3, 0.0); });
    printf("waves/CU=%2d  exp(double)    : %8.3f ms  %7.2f Gexp/s  (%.1f cyc/wave-exp/SIMD)\n", wpc, t,
           nwaves * 64.0 * ITERS / t / 1e6, t * 1e-3 * 2.4e9 / (ITERS * 1.0 * wpc / 4));
    t = time_ms([&] { hipLaunchKernelGGL(k_log, dim3(blocks), dim3(threads), 0, 0, out, 1e-3, 0.0); });
    printf("waves/CU=%2d  log(double)    : %8.3f ms  %7.2f Glog/s  (%.1f cyc/wave-log/SIMD)\n", wpc, t,
           nwaves * 64.0 * ITERS / t / 1e6, t * 1e-3 * 2.4e9 / (ITERS * 1.0 * wpc / 4));
  }
  return 0;
}
