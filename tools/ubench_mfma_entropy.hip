// would the two D-deep contractions of the entropy kernel pay on the FP64 matrix
// pipe at BASELINE config 5's shape (D = 20, K = 100)?  (At config 3 -- D = 10, K = 50 -- padding to
// the 16 x 16 x 4 tile costs 1.54x / 1.66x and the answer was no: profiles/r02_entropy_ablation.md.)
//
// The entropy kernel's work per 64-row batch and component pair (j fixed, k = 1..K), both signs:
//   pass 1:  c[row][k]  = sum_d e[row][d] Delta_jk[d]                 (D FMAs per (row, k))
//            ~26 more vector instructions: two exponents, two exp2, two q accumulations
//   pass 2:  Td[row][d] = sum_k gd[row][k] Delta_jk[d]                (D FMAs per (row, k))
//            ~10 more: the normalised densities, W sums, the sigma sums
// Form V (what entmc_ws_kernel<20,25> does): lane = row, a wave owns KT = 25 components whose table
//   rows arrive as scalar operands; 2 D = 40 FMAs per (row, k).  At the 1 wave / SIMD its 512-register
//   build runs at, a wave issues one float64 instruction every ~6.9 cycles (tools/ubench_ops.hip).
// Form M (proposed): a wave owns 16 rows and ALL components; the products are
//   C'[k][row] = Delta_j[k][:] . e[row][:]        v_mfma_f64_16x16x4: M = k (7 tiles of 16 for K = 100: x1.12),
//                                                 N = 16 rows, 5 steps over D = 20 (exact): 35 per batch
//   Td[row][d] = gd[row][:] . Delta_j[:][d]       M = 16 rows, N = d (2 tiles of 16 for D = 20: x1.6),
//                                                 28 steps over the lane's k's: 56 per batch
//   so that C' comes out with lane = (row = lane & 15, four k's of every k-tile): the 28 (row, k) pairs of
//   a lane feed the second product as its A operand straight from registers -- no transpose -- and
//   the sums over k are 28 in-lane additions plus two cross-lane steps per row.  Delta_j sits in LDS.
// Each kernel below runs BATCHES batches of its form with the ~36 other instructions per (row, k)
// represented by dependent FMA chains of that length ("rest"), one workgroup per CU, and reports cycles
// per 64-row batch; a form's contraction alone (REST = 0) and with the rest (REST = 36).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off ubench_mfma_entropy.hip -o ubench_mfma_entropy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef double double4_t __attribute__((ext_vector_type(4)));
constexpr int D = 20, K = 100, KT = 25, BATCHES = 256;

// the "rest" of a (row, k) pair: REST dependent float64 instructions in two interleaved chains
template <int REST>
__device__ __forceinline__ double rest_work(double x, double y) {
  double a = x, b = y;
#pragma unroll
  for (int i = 0; i < REST / 2; ++i) {
    a = fma(a, 0.999999, b);
    b = fma(b, 1.000001, a);
  }
  return a + b;
}

// ---- form V: lane = row, wave = 25 components, table rows through scalar loads --------------------
template <int REST>
__global__ __launch_bounds__(256, 1) void form_v(const double* __restrict__ T, const double* __restrict__ eps, double* out) {
  extern __shared__ double pad[];  // 100 KB: one workgroup per CU, as the 512-register build
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double Td[D], acc = 0.0;
#pragma unroll
  for (int d = 0; d < D; ++d) Td[d] = 0.0;
  for (int it = 0; it < BATCHES; ++it) {
    double e[D];
#pragma unroll
    for (int d = 0; d < D; ++d) e[d] = eps[((size_t)(blockIdx.x * BATCHES + it) * 64 + lane) * D + d];
    int zoff;
    asm volatile("s_mov_b32 %0, 0" : "=s"(zoff));
    const double* Tw = T + (size_t)wave * 26 + zoff;
    double r[KT];
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) {  // pass 1
      const double* row = Tw + (size_t)(4 * kk) * 26;
      double c = 0.0;
#pragma unroll
      for (int d = 0; d < D; ++d) c = fma(row[d], e[d], c);
      r[kk] = REST ? rest_work<(REST * 26) / 36>(c, row[D]) : c;
    }
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) {  // pass 2
      const double* row = Tw + (size_t)(4 * kk) * 26;
      const double gd = REST ? rest_work<(REST * 10) / 36>(r[kk], row[D + 1]) : r[kk];
#pragma unroll
      for (int d = 0; d < D; ++d) Td[d] = fma(gd, row[d], Td[d]);
    }
#pragma unroll
    for (int d = 0; d < D; ++d) acc = fma(e[d], Td[d], acc);
  }
  if (pad == nullptr) pad[0] = acc;
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// ---- form M: wave = 16 rows x all components on v_mfma_f64_16x16x4_f64 ------------------------------
template <int REST>
__global__ __launch_bounds__(256, 1) void form_m(const double* __restrict__ T, const double* __restrict__ eps, double* out) {
  extern __shared__ double sT[];  // Delta_j: [112 k][20 d] (rows 100..111 zero) + constants [112][2]; padded to 100 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  for (int i = tid; i < 112 * 22; i += 256) {
    const int k = i / 22, c = i - k * 22;
    sT[i] = k < K ? T[(size_t)k * 26 + (c < D ? c : c)] : 0.0;
  }
  __syncthreads();
  double acc = 0.0;
  double4_t td[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  for (int it = 0; it < BATCHES; ++it) {
    // B operand of product 1: e[row = li][d = 4 s + lk], five values per lane (this wave's 16 rows)
    double eb[5];
    const size_t row = (size_t)(blockIdx.x * BATCHES + it) * 64 + wave * 16 + li;
#pragma unroll
    for (int s = 0; s < 5; ++s) eb[s] = eps[row * D + 4 * s + lk];
    // product 1: C'[k tile][row]; A = Delta[k = 16 kt + li][d = 4 s + lk] from LDS
    double4_t c[7];
#pragma unroll
    for (int kt = 0; kt < 7; ++kt) {
      c[kt] = (double4_t){0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < 5; ++s)
        c[kt] = __builtin_amdgcn_mfma_f64_16x16x4f64(sT[(16 * kt + li) * 22 + 4 * s + lk], eb[s], c[kt], 0, 0, 0);
    }
    // the rest of pass 1 and pass 2 on the lane's 28 pairs: lane = (row li, k = 16 kt + lk + 4 r)
    double gd[7][4];
#pragma unroll
    for (int kt = 0; kt < 7; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double cst = sT[(16 * kt + lk + 4 * r) * 22 + D];
        gd[kt][r] = REST ? rest_work<REST>(c[kt][r], cst) : c[kt][r];
      }
    // product 2: Td[row][d tile]: A = gd[row = li][k = 16 kt + lk + 4 r] (registers), B = Delta[k][d = 16 nt + li]
#pragma unroll
    for (int kt = 0; kt < 7; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const int d = 16 * nt + li;
          const double b = d < D ? sT[(16 * kt + lk + 4 * r) * 22 + d] : 0.0;
          td[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(gd[kt][r], b, td[nt], 0, 0, 0);
        }
    acc += eb[0] + eb[4];
  }
  acc += td[0][0] + td[0][1] + td[0][2] + td[0][3] + td[1][0] + td[1][1] + td[1][2] + td[1][3];
  out[blockIdx.x * 256 + tid] = acc;
}

template <typename F> double time_ms(F launch) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  launch(); CHECK(hipDeviceSynchronize()); float best = 1e30f;
  for (int r = 0; r < 5; ++r) { CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
  return best;
}
int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  const int CU = p.multiProcessorCount;
  double *T, *eps, *out;
  CHECK(hipMalloc(&T, sizeof(double) * 112 * 26)); CHECK(hipMemset(T, 0, sizeof(double) * 112 * 26));
  const size_t ne = (size_t)CU * BATCHES * 64 * D;
  CHECK(hipMalloc(&eps, sizeof(double) * ne)); CHECK(hipMemset(eps, 0, sizeof(double) * ne));
  CHECK(hipMalloc(&out, sizeof(double) * 256 * CU));
  const int lds = 100 * 1024;
#define SETLDS(kern) CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds))
  SETLDS((form_v<0>)); SETLDS((form_v<36>)); SETLDS((form_m<0>)); SETLDS((form_m<36>));
#define RUN(kern, label) { const double t = time_ms([&] { hipLaunchKernelGGL(kern, dim3(CU), dim3(256), lds, 0, T, eps, out); }); \
    printf("%-44s %9.0f cycles per 64-row batch (2.4 GHz) = %6.1f per (row, k) on a lane's share\n", label, t * 1e-3 * 2.4e9 / BATCHES, t * 1e-3 * 2.4e9 / BATCHES / 25.0); }
  RUN((form_v<0>), "V: lane = row, 40 FMAs per pair, alone")
  RUN((form_v<36>), "V: with the other ~36 instructions per pair")
  RUN((form_m<0>), "M: 35 + 56 matrix instructions, alone")
  RUN((form_m<36>), "M: with the other ~36 instructions per pair")
  printf("(the shipped kernel: 149.8 us for 2 rounds of 8 batches = %.0f cycles per batch)\n", 149.8e-6 / 16 * 2.4e9);
  return 0;
}
