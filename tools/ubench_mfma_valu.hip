// Does other vector-ALU work overlap with FP64 matrix instructions on gfx950?
// One to four waves per SIMD, each runs R rounds of {1 v_mfma_f64_16x16x4_f64, NV filler VALU ops};
// fillers: 32-bit integer adds / float64 FMAs / LDS reads.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef double double4_t __attribute__((ext_vector_type(4)));
constexpr int R = 2048;
template <int NV, int KIND, bool MFMA>
__global__ void k(double* out, int s) {
  __shared__ double lds[1024];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  double4_t acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  double a = threadIdx.x * 0.5, b = 1.0000001;
  int n[4] = {s, s + 1, s + 2, s + 3};
  double x[4] = {a, a + 1, a + 2, a + 3};
  for (int r = 0; r < R; ++r) {
    if (MFMA) acc[r & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[r & 3], 0, 0, 0);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      if (KIND == 0) n[v & 3] += n[(v + 1) & 3] ^ r;
      else if (KIND == 1) x[v & 3] = fma(x[v & 3], b, a);
      else x[v & 3] += lds[(threadIdx.x + v * 64 + r) & 1023];
    }
  }
  double t = 0;
  for (int i = 0; i < 4; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + n[i] + x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}
template <typename F> double time_ms(F launch) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  launch(); CHECK(hipDeviceSynchronize()); float best = 1e30f;
  for (int r = 0; r < 5; ++r) { CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
  return best;
}
int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  const int CU = p.multiProcessorCount; double* out; CHECK(hipMalloc(&out, sizeof(double) * 256 * CU * 8));
  for (int wps : {1, 2}) {
    const int blocks = CU * wps;
#define RUN(NV, KIND, MF, label) { double t = time_ms([&] { hipLaunchKernelGGL((k<NV, KIND, MF>), dim3(blocks), dim3(256), 0, 0, out, 3); }); \
      printf("waves/SIMD=%d %-34s %8.1f cycles per round per wave (2.4 GHz)\n", wps, label, t * 1e-3 * 2.4e9 / R / wps); }
    RUN(0, 0, true, "mfma only")
    RUN(8, 0, false, "8 int adds only")
    RUN(8, 0, true, "mfma + 8 int adds")
    RUN(16, 0, false, "16 int adds only")
    RUN(16, 0, true, "mfma + 16 int adds")
    RUN(8, 1, false, "8 f64 fma only")
    RUN(8, 1, true, "mfma + 8 f64 fma")
    RUN(8, 2, false, "8 lds reads+adds only")
    RUN(8, 2, true, "mfma + 8 lds reads+adds")
  }
  return 0;
}
