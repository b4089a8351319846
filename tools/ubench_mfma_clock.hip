// FP64 matrix-instruction stream as predict_var_dma_kernel issues it (4 accumulators, operands
// from 8 + 8 register pairs), 1 or 2 waves per SIMD on every CU: cycles per instruction from the
// shader clock (s_memtime) and the shader clock's rate against the 100 MHz wall clock.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_clock.hip -o /tmp/ubench_mfma_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, unsigned long long* tm, int R, double s) {
  double4_t acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = (double4_t){0, 0, 0, 0};
  double a[16], b[16];
  for (int i = 0; i < 16; ++i) { a[i] = s + threadIdx.x + i; b[i] = s * 0.5 + i; }
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int r = 0; r < R; ++r) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int ia = MODE == 0 ? 0 : 2 * t, ib = MODE == 0 ? 0 : 2 * t;
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ia], b[ib], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ia], b[ib + 1], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ia + 1], b[ib], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ia + 1], b[ib + 1], acc[1][1], 0, 0, 0);
    }
    if (MODE == 2) __syncthreads();
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  double t = 0;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = t;
  if (threadIdx.x == 0) { tm[2 * blockIdx.x] = c1 - c0; tm[2 * blockIdx.x + 1] = w1 - w0; }
}
template <int MODE>
void run(const char* label, int blocks, int R, double* out, unsigned long long* tm) {
  std::vector<unsigned long long> h(2 * blocks);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, tm, R, 1.0);
    CHECK(hipDeviceSynchronize());
  }
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, tm, R, 1.0);
  CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  CHECK(hipMemcpy(h.data(), tm, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost));
  std::vector<double> cyc, wall;
  for (int i = 0; i < blocks; ++i) { cyc.push_back((double)h[2 * i]); wall.push_back((double)h[2 * i + 1]); }
  std::sort(cyc.begin(), cyc.end()); std::sort(wall.begin(), wall.end());
  const double cm = cyc[blocks / 2], wm = wall[blocks / 2];
  printf("%-44s blocks=%4d R=%5d  event %.1f us | in-kernel median: %.0f s_memtime ticks, %.2f us wall -> %.1f ticks/MFMA/wave, %.1f ns/MFMA/wave, memtime rate %.0f MHz\n",
         label, blocks, R, ms * 1e3, cm, wm / 100.0, cm / (32.0 * R), wm * 10.0 / (32.0 * R), cm / (wm / 100.0));
}
int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  const int CU = p.multiProcessorCount;
  double* out; unsigned long long* tm;
  CHECK(hipMalloc(&out, sizeof(double) * 256 * 4 * CU)); CHECK(hipMalloc(&tm, 16 * 4 * CU));
  for (int R : {14, 64, 512}) {
    run<0>("same operands, 1 wave/SIMD", CU, R, out, tm);
    run<0>("same operands, 2 waves/SIMD", 2 * CU, R, out, tm);
    run<1>("8+8 operand pairs, 2 waves/SIMD", 2 * CU, R, out, tm);
    run<2>("8+8 pairs + barrier per 32, 2 waves/SIMD", 2 * CU, R, out, tm);
  }
  return 0;
}
