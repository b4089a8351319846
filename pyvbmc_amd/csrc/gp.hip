// GP side of the ELBO path on gfx950:
//   * expected log joint of the mixture under the GP surrogate (Bayesian quadrature),
//     reference vbmc/variational_optimization.py:1374-1514 (_gp_log_joint);
//   * its variance: the reference's K(K+1)/2 pairs of triangular solves (:1489-1501)
//     restructured as ONE product V = Z L^-1 (K x N x N, triangular) + a K x K Gram
//     matrix, with L^-1 formed once per GP update (the GP is fixed during the
//     thousands of ELBO evaluations of one variational optimisation);
//   * GP.predict of gpyreg (third party): K* block, mean,
//     variance via the same L^-1.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "common.h"
#include <type_traits>
#include "fastmath.h"

namespace {
#if defined(DMA_ABL_TIMES) || defined(KSTAR_ABL_TIMES)
__device__ unsigned long long g_dma_times[8192 * 4];
#endif

constexpr int WAVES = 4;
typedef double double4_t __attribute__((ext_vector_type(4)));

__device__ inline double wave_sum(double v) {
  return fm::wave_sum_dpp(v);
}

// ---------------------------------------------------------------------------
// Inverse of the upper-triangular Cholesky factor, once per vbmc_set_gp (it stands in for the
// reference's pairs of triangular solves, variational_optimization.py:1489-1501).  Blocked, 64 x 64
// blocks, U padded with the identity up to a multiple of 64:
//   1. trinv_diag_kernel: every diagonal block is inverted in LDS by back substitution, one
//      thread per column (64^2/2 multiply-adds each);
//   2. trinv_strip_kernel: X = U^-1 solves U X = I one strip of 16 columns at a time; a workgroup
//      owns a strip of block column j and walks the block rows i = j .. 0,
//          X_i = Dinv_i (E_i - sum_{m = i+1..j} U_im X_m),
//      the sum and the product with the inverted diagonal block on the FP64 matrix cores
//      (v_mfma_f64_16x16x4_f64: wave w owns rows 16 w .. 16 w + 15 of the block row), the strip
//      of X in LDS.  ceil(N/16) workgroups per GP sample: 25 at N = 400.
// N <= 1088 (the strip must fit the 160 KB of LDS); larger N keep the one-thread-per-column kernel.
constexpr int TRB = 64, TRS = 16;

__global__ __launch_bounds__(64) void trinv_diag_kernel(const double* __restrict__ L, int N,
                                                        const double* __restrict__ smeta,
                                                        double* __restrict__ Dinv, int nb) {
  const int s = blockIdx.y, b = blockIdx.x, c = threadIdx.x;
  if (smeta[3 * s] == 0.0) return;  // not a Cholesky sample: nothing to invert
  __shared__ double sUT[TRB][TRB];  // sUT[r][m] = u_mr: the column above the diagonal element r, contiguous
  __shared__ double sRinv[TRB];     // 1 / u_rr
  const double* A = L + (size_t)s * N * N;
  const int r0 = b * TRB;
  {
    // all 64 row loads in flight at once (clamped addresses, the padding is patched in afterwards)
    double u[TRB];
    const int gc = r0 + c, gcc = min(gc, N - 1);
#pragma unroll
    for (int r = 0; r < TRB; ++r) u[r] = A[(size_t)min(r0 + r, N - 1) * N + gcc];
#pragma unroll
    for (int r = 0; r < TRB; ++r) {
      const int gr = r0 + r;
      const double v = (gr < N && gc < N) ? (gc >= gr ? u[r] : 0.0) : (gr == gc ? 1.0 : 0.0);
      sUT[c][r] = v;  // element (r, c) of the block
      if (r == c) sRinv[c] = 1.0 / v;
    }
  }
  __syncthreads();
  // Column c of the inverse by back substitution, column-oriented so that the multiply-adds of a
  // step are independent of each other: with s_m = delta_mc to start,
  //   for r = 63 .. 0:   x_r = s_r / u_rr;   s_m -= u_mr x_r  for all m < r.
  // Everything is unrolled and lives in registers (x_r = 0 for r > c comes out by itself); the
  // u_mr of a step are contiguous broadcast LDS reads that do not depend on the arithmetic.
  double x[TRB];
#pragma unroll
  for (int m = 0; m < TRB; ++m) x[m] = (m == c) ? 1.0 : 0.0;
#pragma unroll
  for (int r = TRB - 1; r >= 0; --r) {
    const double xr = x[r] * sRinv[r];
    x[r] = xr;
#pragma unroll
    for (int m = 0; m < r; ++m) x[m] = fma(-sUT[r][m], xr, x[m]);
  }
  double* out = Dinv + ((size_t)s * nb + b) * TRB * TRB;
#pragma unroll
  for (int r = 0; r < TRB; ++r) out[r * TRB + c] = x[r];
}

__global__ __launch_bounds__(256) void trinv_strip_kernel(const double* __restrict__ L, int N,
                                                          const double* __restrict__ smeta,
                                                          const double* __restrict__ Dinv, int nb,
                                                          double* __restrict__ Li) {
  const int s = blockIdx.y;
  if (smeta[3 * s] == 0.0) return;
  extern __shared__ double lds[];
  const int c0 = blockIdx.x * TRS, j = c0 / TRB;
  double* sX = lds;                          // [(j+1)*64][16]: the strip of X, rows of block rows 0..j
  double* sR = lds + (size_t)(j + 1) * TRB * TRS;  // [64][16]: right-hand side of the current block row
  const double* A = L + (size_t)s * N * N;
  double* B = Li + (size_t)s * N * N;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  // rows below block row j of this strip are zero
  for (int idx = tid; idx < (N - min(N, (j + 1) * TRB)) * TRS; idx += 256) {
    const int r = (j + 1) * TRB + idx / TRS, c = c0 + idx % TRS;
    if (c < N) B[(size_t)r * N + c] = 0.0;
  }
  for (int i = j; i >= 0; --i) {
    const int rw = i * TRB + wave * 16;  // first global row of this wave's 16 rows
    // ---- R = E_i - sum_{m>i} U_im X_m on this wave's 16 rows ----
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
    const int k_begin = (i + 1) * TRB, k_end = (j + 1) * TRB;  // contraction over rows of X already known
    const int ar = rw + li;                                    // row of U this lane feeds
    const bool ar_ok = ar < N;
    const double* Arow = A + (size_t)ar * N;
    // 16 columns of U per round.  The contraction order inside a round is free, so lane group lk
    // takes the four CONSECUTIVE columns k0 + 4 lk + q (q = step): one 32-byte load per lane and
    // 128 contiguous bytes per matrix row, instead of four 8-byte loads 32 bytes apart.  The next
    // round's columns are requested before the matrix instructions of the current one.
    auto loadA = [&](int k0, double (&a)[4]) {
      const int kc = k0 + 4 * lk;
      if (ar_ok && kc + 3 < N) {
        const double2 lo = *(const double2*)(Arow + kc), hi = *(const double2*)(Arow + kc + 2);
        a[0] = lo.x; a[1] = lo.y; a[2] = hi.x; a[3] = hi.y;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] = (ar_ok && kc + q < N) ? Arow[kc + q] : 0.0;  // (padding of U: off-diagonal 0)
      }
    };
    double a_cur[4], a_nxt[4];
    if (k_begin < k_end) loadA(k_begin, a_cur);
    for (int k0 = k_begin; k0 < k_end; k0 += 16) {
      if (k0 + 16 < k_end) loadA(k0 + 16, a_nxt);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double bq = sX[(size_t)(k0 + 4 * lk + q) * TRS + li];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[q], bq, acc, 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) a_cur[q] = a_nxt[q];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wave * 16 + lk + 4 * r;                   // row within block row i
      const double e = (i * TRB + row == c0 + li) ? 1.0 : 0.0;  // E: identity columns of this strip
      sR[row * TRS + li] = e - acc[r];
    }
    __syncthreads();
    // ---- X_i = Dinv_i R (Dinv_i upper triangular: k >= 16 wave) ----
    double4_t x = {0.0, 0.0, 0.0, 0.0};
    const double* Di = Dinv + ((size_t)s * nb + i) * TRB * TRB + (size_t)(wave * 16 + li) * TRB;
    for (int k0 = wave * 16; k0 < TRB; k0 += 4) {
      const double av = Di[k0 + lk];
      const double bv = sR[(k0 + lk) * TRS + li];
      x = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, x, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = i * TRB + wave * 16 + lk + 4 * r, c = c0 + li;
      sX[(size_t)row * TRS + li] = x[r];
      if (row < N && c < N) B[(size_t)row * N + c] = x[r];
    }
    __syncthreads();
  }
}

// Fallback for N beyond the strip kernel's LDS: one thread per column (back substitution).
__global__ void trinv_upper_kernel(const double* __restrict__ L, int N, double* __restrict__ Li) {
  const int s = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const double* A = L + (size_t)s * N * N;
  double* B = Li + (size_t)s * N * N;
  for (int r = N - 1; r > i; --r) B[(size_t)r * N + i] = 0.0;
  B[(size_t)i * N + i] = 1.0 / A[(size_t)i * N + i];
  for (int r = i - 1; r >= 0; --r) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int m = r + 1;
    for (; m + 3 <= i; m += 4) {
      a0 = fma(A[(size_t)r * N + m], B[(size_t)m * N + i], a0);
      a1 = fma(A[(size_t)r * N + m + 1], B[(size_t)(m + 1) * N + i], a1);
      a2 = fma(A[(size_t)r * N + m + 2], B[(size_t)(m + 2) * N + i], a2);
      a3 = fma(A[(size_t)r * N + m + 3], B[(size_t)(m + 3) * N + i], a3);
    }
    for (; m <= i; ++m) a0 = fma(A[(size_t)r * N + m], B[(size_t)m * N + i], a0);
    B[(size_t)r * N + i] = -((a0 + a1) + (a2 + a3)) / A[(size_t)r * N + r];
  }
}

// J[j][k] of one GP sample (variational_optimization.py:1473-1503), one wave per pair j<=k:
//   J_jk = exp(lnnf_jk - 1/2 sum_d delta_jk_d^2)  -/+  sum_c U[j][c] V[k][c] (/ sn2_eff),
//   tau_jk_d = sqrt((sigma_j^2+sigma_k^2) lambda_d^2 + ell_d^2), delta = (mu_j - mu_k)/tau,
//   lnnf_jk = 2 hyp[D] + sum_d hyp[d] - sum_d log tau_jk_d.   Writes both triangles.
// blockIdx.y = GP sample: U = V (L_chol sample: rows of Z L^-1) or Z (rows of z, V = Z L).
__global__ __launch_bounds__(256) void gram_kernel(const double* __restrict__ Z,
                                                   const double* __restrict__ V,
                                                   const double* __restrict__ mix, MixLayout ml,
                                                   const double* __restrict__ hyp, int P, int N,
                                                   const double* __restrict__ smeta,
                                                   double* __restrict__ J) {
  const int D = ml.D, K = ml.K;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pair = blockIdx.x * WAVES + wave;
  if (pair >= K * K) return;
  const int j = pair / K, k = pair - j * K;
  if (j > k) return;
  const int smp = blockIdx.y;
  const bool chol = smeta[3 * smp] != 0.0;
  const double inv_sn2 = smeta[3 * smp + 2];
  hyp += (size_t)smp * P;
  J += (size_t)smp * K * K;
  V += (size_t)smp * K * N;
  const double* u = (chol ? V : Z + (size_t)smp * K * N) + (size_t)j * N;
  const double* v = V + (size_t)k * N;
  double acc = 0.0;
  for (int c = lane; c < N; c += 64) acc = fma(u[c], v[c], acc);
  // closed-form part: lanes run over d
  double term = 0.0;
  const double sj = mix[ml.o_sig + j], sk = mix[ml.o_sig + k];
  const double ss = sj * sj + sk * sk;
  for (int d = lane; d < D; d += 64) {
    const double lam = mix[ml.o_lam + d];
    const double t2 = ss * lam * lam + exp(2.0 * hyp[d]);
    const double dm = mix[ml.o_mu + j * D + d] - mix[ml.o_mu + k * D + d];
    term += hyp[d] - 0.5 * log(t2) - 0.5 * dm * dm / t2;
  }
  acc = wave_sum(acc);
  term = wave_sum(term);
  if (lane == 0) {
    double Jv = exp(2.0 * hyp[D] + term);
    Jv = chol ? Jv - acc * inv_sn2 : Jv + acc;
    J[(size_t)j * K + k] = Jv;
    J[(size_t)k * K + j] = Jv;
  }
}

// ---------------------------------------------------------------------------
// predict, stage 2: T = A (M x N) * B (N x N) on the FP64 matrix cores, fused row epilogue
//   mode 0 (L_chol): B = L^-1 upper triangular; part[ct][m] = sum_{c in tile} T[m][c]^2
//   mode 1         : B = L (full, symmetric);   part[ct][m] = sum_{c in tile} A[m][c] T[m][c]
//   Cout != null   : also/only store T (used by the log-joint variance: V = Z L^-1 or Z L)
// v_mfma_f64_16x16x4_f64: lane l holds A[i=l&15][k=l>>4], B[k=l>>4][j=l&15] and 4 results
// C[row=(l>>4)+4r][col=l&15]  (cdna_hip_programming.md section 3, f64 layout).
// Workgroup = 64 x 64 output tile, 4 waves x (2 x 2) MFMA tiles, 16-deep LDS panels:
//   sA[64][17]  (row-major, +1 pad: the 16 rows of an A fragment hit distinct banks)
//   sB[16][80]  (row stride = 32 banks mod 64: the 4 k-rows of a B fragment do not collide)
// On gfx950 the FP64 MFMA peak equals the FP64 vector peak (78.6 TFLOP/s); what the
// matrix instruction buys here is issue efficiency: 1024 FMAs per instruction and no
// per-FMA operand traffic.
constexpr int TS = 64, TKD = 16, LDA = TKD + 1, LDB = TS + 16;
// blockIdx.z (predict only): GP sample -- A, part advance by their strides, B is the sample's
// L^-1 or L according to smeta[3z] (smeta == null: single problem, B and mode as given).
__global__ __launch_bounds__(256) void predict_var_mfma_kernel(const double* __restrict__ A,
                                                               const double* __restrict__ B,
                                                               int64_t M, int N, int mode,
                                                               double* __restrict__ part,
                                                               double* __restrict__ Cout,
                                                               const double* __restrict__ Bfull,
                                                               const double* __restrict__ smeta,
                                                               int64_t part_stride) {
  // XCD-aware tile order: the dispatcher places workgroup b (x fastest, then y, z) on XCD b % 8
  // (MI355X_MICROARCH.md, workgroup dispatch), each XCD has its own 4 MiB L2, and the nct column
  // tiles of one row tile all re-read the same rows of A.  Remap the linear id (bijectively) so
  // that one XCD runs all column tiles of a row tile back to back: A then comes from that XCD's L2
  // instead of nct times from the Infinity Cache.
  const int nct = gridDim.x;
  int tile_c, tile_z;
  int64_t tile_r;
  {
    const int64_t per_z = (int64_t)gridDim.x * gridDim.y;
    const int64_t nwg = per_z * gridDim.z;
    const int64_t orig = ((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const int64_t q = nwg / 8, r = nwg % 8;
    const int64_t xcd = orig % 8;
    int64_t wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
    tile_z = (int)(wgid / per_z);
    wgid -= (int64_t)tile_z * per_z;
    tile_r = wgid / nct;
    tile_c = (int)(wgid - tile_r * nct);
  }
  if (smeta) {
    const int z = tile_z;
    const bool chol = smeta[3 * z] != 0.0;
    A += (size_t)z * M * N;
    if (Cout) Cout += (size_t)z * M * N;
    B = (chol ? B : Bfull) + (size_t)z * N * N;
    mode = chol ? 0 : 1;
    part += (size_t)z * part_stride;
  }
  __shared__ double sA[TS * LDA];
  __shared__ double sB[TKD * LDB];
  __shared__ double sRow[TS][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wc = wave & 1;
  const int li = lane & 15, lk = lane >> 4;
  const int64_t m0 = tile_r * TS;
  const int c0 = tile_c * TS;
  double4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
  const int nmax = (mode == 0) ? min(N, c0 + TS) : N;  // upper-triangular B: n <= c
  // software pipeline: the next panel's global loads are issued into registers before the
  // MFMAs of the current panel and written to LDS afterwards, so HBM/L2 latency hides
  // behind the matrix work
  constexpr int PER = TS * TKD / 256;  // 4 elements of A and of B per thread per panel
  double ra[PER], rb[PER];
  auto fetch = [&](int n0) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int idx = tid + 256 * i;
      const int r = idx / TKD, kk = idx - r * TKD;
      const int64_t m = m0 + r;
      const int n = n0 + kk;
      ra[i] = (m < M && n < N) ? A[(size_t)m * N + n] : 0.0;
      const int kb = idx / TS, cc = idx - kb * TS;
      const int nb = n0 + kb, c = c0 + cc;
      rb[i] = (nb < N && c < N) ? B[(size_t)nb * N + c] : 0.0;
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int idx = tid + 256 * i;
      const int r = idx / TKD, kk = idx - r * TKD;
      sA[r * LDA + kk] = ra[i];
      const int kb = idx / TS, cc = idx - kb * TS;
      sB[kb * LDB + cc] = rb[i];
    }
  };
  fetch(0);
  for (int n0 = 0; n0 < nmax; n0 += TKD) {
    stash();
    __syncthreads();
    if (n0 + TKD < nmax) fetch(n0 + TKD);
#pragma unroll
    for (int kq = 0; kq < TKD / 4; ++kq) {
      const double a0 = sA[(wm * 32 + li) * LDA + kq * 4 + lk];
      const double a1 = sA[(wm * 32 + 16 + li) * LDA + kq * 4 + lk];
      const double b0 = sB[(kq * 4 + lk) * LDB + wc * 32 + li];
      const double b1 = sB[(kq * 4 + lk) * LDB + wc * 32 + 16 + li];
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
  }
  if (Cout) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t m = m0 + wm * 32 + mt * 16 + lk + 4 * r;
          const int c = c0 + wc * 32 + ct * 16 + li;
          if (m < M && c < N) Cout[(size_t)m * N + c] = acc[mt][ct][r];
        }
  }
  if (!part) return;
  // epilogue: per-row reduction over this wave's 32 columns, then over the two column waves
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wm * 32 + mt * 16 + lk + 4 * r;
      const int64_t m = m0 + row;
      double v = 0.0;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const int c = c0 + wc * 32 + ct * 16 + li;
        if (c < N && m < M) {
          const double t = acc[mt][ct][r];
          v += (mode == 0) ? t * t : A[(size_t)m * N + c] * t;
        }
      }
      v = fm::row16_sum_dpp(v);
      if (li == 0) sRow[row][wc] = v;
    }
  __syncthreads();
  if (tid < TS && m0 + tid < M) part[(size_t)tile_c * M + m0 + tid] = sRow[tid][0] + sRow[tid][1];
}

// ---------------------------------------------------------------------------
// predict, stage 2 for batches of points on L_chol samples: the same partial sums
//   part[ct][m] = sum_{c in tile ct} T[m][c]^2,  T = A (M x N) * B (N x N upper triangular),
// with a main loop that holds no vector-ALU work.  On gfx950 v_mfma_f64_16x16x4_f64 runs on the
// double-precision vector units and nothing else the SIMD issues overlaps it
// (tools/ubench_mfma_valu.hip, profiles/r02_predict_gemm_ablation.md): the address arithmetic,
// staging registers and ds_writes of predict_var_mfma_kernel above are paid on top of the matrix
// time.  Here
//   * panels (32 deep: A 64 x 32, B 32 x 64 = 32 KB per stage, two stages) come in by
//     global_load_lds_dwordx4 -- scalar base + per-lane offsets fixed before the loop, no staging
//     registers, no LDS stores;
//   * operands are padded (leading dimension ld = N rounded up to 64, zero filled, see
//     pad_square_kernel and predict_kstar_mfma_kernel's lda), so the loop has no bounds tests;
//   * the LDS image is lane-linear per DMA instruction; the A rows are XOR-swizzled on the SOURCE
//     address (16-byte chunk q of row r sits in slot q ^ (r & 15)), which makes the ds_read_b128
//     of an A fragment conflict-free in each of the instruction's 16-lane groups
//     (MI355X_MICROARCH.md, LDS); B needs none;
//   * one ds_read_b128 feeds two matrix instructions: the contraction index is permuted inside a
//     panel (MFMA step 2t+h, lane group lk  <->  k = 8t + 2 lk + h: an A chunk holds h = 0, 1)
//     and a wave's two column tiles are the even / odd columns of its 32 (a B chunk holds both);
//   * column tiles are folded so that every workgroup has (nearly) the same number of panels
//     (see the kernel); a workgroup's items are one panel sequence through the pipeline.
// One barrier per panel; the DMA of panel i+1 is in flight during the matrix work of panel i.
// predict, stage 3 for one point (also predict_finish_kernel's body): fmu = mean(x*) + the stage-1 partial means,
// fs2 = max(0, sf^2 -/+ the stage-2 partial row sums) (+ noise); s and f are those sums.
struct PredFin {
  int* tick = nullptr;  // non-null: predict_var_dma_kernel finishes the points itself (arrival ticket per row tile)
  const double* hyp_all = nullptr;
  const double* smeta = nullptr;
  const double* xs = nullptr;
  double* fmu = nullptr;
  double* fs2 = nullptr;
  int64_t ld = 0;
  int D = 0, P = 0, mean_kind = 0, add_noise = 0;
};
__device__ __forceinline__ void predict_point_finish(const PredFin& fin, int smp, int64_t m, double s, double f) {
  const double* hyp = fin.hyp_all + (size_t)smp * fin.P;
  const int D = fin.D;
  const bool chol = fin.smeta[3 * smp] != 0.0;
  const double sf2 = exp(2.0 * hyp[D]);
  const double add = fin.add_noise ? exp(2.0 * hyp[D + 1]) * fin.smeta[3 * smp + 1] : 0.0;
  fin.fs2[(size_t)smp * fin.ld + m] = fmax(chol ? sf2 - s : sf2 + s, 0.0) + add;
  // mean function at x* (variational_optimization.py:1383-1392 layout)
  double mean = 0.0;
  const double* hm = hyp + D + 2;
  if (fin.mean_kind == VBMC_MEAN_CONST) mean = hm[0];
  if (fin.mean_kind == VBMC_MEAN_NEGQUAD) {
    mean = hm[0];
    for (int d = 0; d < D; ++d) {
      const double t = (fin.xs[m * D + d] - hm[1 + d]) * exp(-hm[1 + D + d]);
      mean -= 0.5 * t * t;
    }
  }
  fin.fmu[(size_t)smp * fin.ld + m] = mean + f;
}

constexpr int DK = 32;                        // panel depth
constexpr int DMA_STAGE = 2 * TS * DK * 8;    // bytes of one stage: A panel, then B panel
constexpr int DMA_LDS = 2 * DMA_STAGE + TS * 2 * 8;
typedef double double2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) const void* glds_src_t;
typedef __attribute__((address_space(3))) void* glds_dst_t;

__device__ __forceinline__ const char* uniform_ptr(const void* p) {
  const uint64_t v = (uint64_t)p;
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32));
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
  return (const char*)(((uint64_t)hi << 32) | (uint64_t)lo);
}

__global__ __launch_bounds__(256, 2) void predict_var_dma_kernel(const double* __restrict__ A,
                                                                 const double* __restrict__ B,
                                                                 int64_t M, int N, int ld,
                                                                 int64_t a_stride,
                                                                 double* __restrict__ part,
                                                                 int64_t part_stride, int nrt, int G, PredFin fin) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  __shared__ int sLast;
  const unsigned bid = blockIdx.x;
  const int nct = (N + TS - 1) / TS;
  // Work assignment.  The triangular skip makes column tile c cost min(N, 64 c + 64) / 32 panels,
  // two workgroups are resident per CU and a CU's matrix pipes are shared by whatever runs there,
  // so the kernel ends when the CU with the largest panel total does.  Every workgroup therefore
  // gets (close to) the same total, 2 nct panels: per row tile the top column tile alone, the
  // others folded (nct-1-j, j-1), j = 1 .. (nct-1)/2; with nct even the middle tile nct/2 - 1 is
  // left over and shares a workgroup with the middle tile of the next row tile.  Row tiles are
  // taken in groups of two for that: nct + 1 workgroups per group; groups are dealt round robin
  // to the XCDs (block b runs on XCD b % 8), so that every workgroup of a group finds the
  // group's A rows in the same L2.  The grid is padded to 8 * ceil(ngroups / 8) * (nct + 1).
  // (M = 8192, N = 400: 512 workgroups of 13 or 14 panels, two per CU.)
  // up to two (row tile, column tile) items (four scalars, selected by ?: -- as arrays indexed at run time they
  // lived in 20 bytes of scratch)
  int it_c0 = 0, it_c1 = 0, it_g0 = 0, it_g1 = 0, nitem = 0;
  {
    const int x = bid & 7, o = bid >> 3;
    const int ngr = (G + 1) / 2;
    const int cnt = x < ngr ? (ngr - x + 7) / 8 : 0;  // groups of this XCD
    if (o >= cnt * (nct + 1)) return;
    const int grp = x + 8 * (o / (nct + 1)), w = o % (nct + 1);
    const int h = (nct - 1) / 2, per_row = 1 + h;
    if (w < 2 * per_row) {
      const int g = 2 * grp + w / per_row, j = w % per_row;
      if (g < G) {
        it_g0 = it_g1 = g;
        it_c0 = j == 0 ? nct - 1 : nct - 1 - j;
        it_c1 = j - 1;
        nitem = j == 0 ? 1 : 2;
      }
    } else {  // nct even: the two middle tiles of the group
      it_c0 = it_c1 = nct / 2 - 1;
      it_g0 = 2 * grp;
      it_g1 = 2 * grp + 1;
      nitem = it_g1 < G ? 2 : 1;
    }
    if (nitem == 0) return;
  }
#ifdef DMA_ABL_TIMES
  const unsigned tb_ = bid;
  if (threadIdx.x == 0 && tb_ < 8192) {
    g_dma_times[tb_ * 4] = wall_clock64();
    g_dma_times[tb_ * 4 + 2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_ID
  }
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wc = wave & 1;
  const int li = lane & 15, lk = lane >> 4;
  // panels of an item: n < min(N, 64 c + 64) (upper-triangular B: n <= c)
  const int P1 = (min(N, it_c0 * TS + TS) + DK - 1) / DK;
  const int P = P1 + (nitem > 1 ? (min(N, it_c1 * TS + TS) + DK - 1) / DK : 0);
  const char* Arow[2];
  const char* Bcol[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const bool second = nitem > 1 && q == 1;
    const int g = second ? it_g1 : it_g0, c = second ? it_c1 : it_c0;
    const int z = g / nrt;
    const int64_t m0 = (int64_t)(g - z * nrt) * TS;
    Arow[q] = uniform_ptr(A + (size_t)z * a_stride + (size_t)m0 * ld);
    Bcol[q] = uniform_ptr(B + (size_t)z * ld * ld + c * TS);
  }
  // DMA source offsets (bytes).  A: instruction j of wave w fills rows 16 w + 4 j + (lane >> 4),
  // slot lane & 15;  B: rows 2 (4 w + j) + (lane >> 5), chunk lane & 31.
  unsigned voffA[4], voffB[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = 16 * wave + 4 * j + (lane >> 4);
    voffA[j] = (unsigned)r * (unsigned)ld * 8u + (unsigned)(((lane & 15) ^ (r & 15)) * 16);
    const int kb = 2 * (4 * wave + j) + (lane >> 5);
    voffB[j] = (unsigned)kb * (unsigned)ld * 8u + (unsigned)((lane & 31) * 16);
  }
  // fragment addresses (bytes inside a stage)
  unsigned offA[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) offA[t] = (unsigned)((wm * 32 + li) * 256 + 64 * (t ^ (li >> 2)) + 16 * (lk ^ (li & 3)));
  const unsigned offB = (unsigned)(TS * DK * 8 + 2 * lk * 512 + (16 * wc + li) * 16);
  // DMA instruction j (0..3: A, 4..7: B) of a panel (scalar bases ag / bg) into `stage`.  The eight
  // instructions of the next panel are spread over the matrix instructions of the current one
  // (one behind every group of four), where their issue costs nothing.
  // (inline assembly: the scalar-base form `global_load_lds_dwordx4 voff, s[base]` -- hipcc's builtin
  // re-materialises a 64-bit vector address per instruction, and every vector-ALU instruction in
  // this loop is time the matrix pipe stands still.  The compiler does not count these loads:
  // DMA_BARRIER waits for them explicitly.)
  const unsigned lds0 = (unsigned)(uintptr_t)dsm + (unsigned)wave * 4096u;
  auto dma = [&](int j, int stage, const char* ag, const char* bg) {
    const unsigned dst = lds0 + (unsigned)(stage * DMA_STAGE + (j < 4 ? j * 1024 : TS * DK * 8 + (j - 4) * 1024));
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :
                 : "v"(j < 4 ? voffA[j] : voffB[j - 4]), "s"(j < 4 ? ag : bg), "s"(dst)
                 : "memory");
  };
  auto base_a = [&](int i) {
#ifdef DMA_ABL_SAMEPANEL
    i = 0;
#endif
    const bool second = i >= P1;
    return uniform_ptr((second ? Arow[1] : Arow[0]) + (size_t)((second ? i - P1 : i) * DK) * 8);
  };
  auto base_b = [&](int i) {
#ifdef DMA_ABL_SAMEPANEL
    i = 0;
#endif
    const bool second = i >= P1;
    return uniform_ptr((second ? Bcol[1] : Bcol[0]) + (size_t)((second ? i - P1 : i) * DK) * ld * 8);
  };
  double4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
  // fragments of k-step pair t+1 are read before the matrix instructions of pair t are issued
  auto compute = [&](int stage, auto prefetch, const char* ag, const char* bg) {
    const unsigned char* sa = dsm + stage * DMA_STAGE;
    double2_t fa[2][2], fb[2][2];  // [buffer][mt] / [buffer][h]
    auto load = [&](int t, int buf) {
#ifdef DMA_ABL_NOLDS
      if (t > 0) return;
#endif
      fa[buf][0] = *(const double2_t*)(sa + offA[t]);
      fa[buf][1] = *(const double2_t*)(sa + offA[t] + 16 * 256);
      fb[buf][0] = *(const double2_t*)(sa + offB + (8 * t) * 512);
      fb[buf][1] = *(const double2_t*)(sa + offB + (8 * t + 1) * 512);
    };
    load(0, 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int cur = t & 1;
      if (t < 3) load(t + 1, cur ^ 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[cur][0][h], fb[cur][h][0], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[cur][0][h], fb[cur][h][1], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[cur][1][h], fb[cur][h][0], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[cur][1][h], fb[cur][h][1], acc[1][1], 0, 0, 0);
#ifndef DMA_ABL_NODMA
        if constexpr (decltype(prefetch)::value) dma(2 * t + h, stage ^ 1, ag, bg);
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
#ifdef DMA_ABL_NOBAR
#define DMA_BARRIER() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define DMA_BARRIER()                                  \
  do {                                                 \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   \
    __syncthreads();                                   \
  } while (0)
#endif
  // row sums of T^2 over an item's 64 columns (the padding columns of B are zero)
  double* sRow = (double*)(dsm + 2 * DMA_STAGE);  // [64][2]
  auto finish_item = [&](int q) __attribute__((always_inline)) {
    const int g = q ? it_g1 : it_g0, c = q ? it_c1 : it_c0;
    const int z = g / nrt;
    const int64_t m0 = (int64_t)(g - z * nrt) * TS;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wm * 32 + mt * 16 + lk + 4 * r;
        double v = acc[mt][0][r] * acc[mt][0][r];
        v = fma(acc[mt][1][r], acc[mt][1][r], v);
        v = fm::row16_sum_dpp(v);
        if (li == 0) sRow[row * 2 + wc] = v;
        acc[mt][0][r] = 0.0;
        acc[mt][1][r] = 0.0;
      }
    __syncthreads();
    if (fin.tick == nullptr) {
      if (tid < TS && m0 + tid < M)
        part[(size_t)z * part_stride + (size_t)c * M + m0 + tid] = sRow[tid * 2] + sRow[tid * 2 + 1];
      return;
    }
    // Fused finish (round 5): the partial row sums are stored write-through; the workgroup that completes a row
    // tile's ticket -- every workgroup holding items of row tile g arrives once, after its last item of g -- adds
    // the tile's nct slots up in slot order (bit-identical to predict_finish_kernel) and writes fmu / fs2.  No
    // fence: write-through stores, drained before the relaxed ticket; the reader's loads bypass its L1 / L2 lines
    // the same way (adam_fused.hip's exchange).
    if (tid < TS && m0 + tid < M)
      __hip_atomic_store(part + (size_t)z * part_stride + (size_t)c * M + m0 + tid, sRow[tid * 2] + sRow[tid * 2 + 1],
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last_of_g = q == nitem - 1 || it_g0 != it_g1;
    if (!last_of_g) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const int expect = 1 + (nct - 1) / 2 + ((nct & 1) ? 0 : 1);
      sLast = __hip_atomic_fetch_add(fin.tick + g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == expect - 1;
    }
    __syncthreads();
    if (sLast && tid < TS && m0 + tid < M) {
      const double* pz = part + (size_t)z * part_stride;
      double sv = 0.0, fv = 0.0;
      for (int t = 0; t < nct; ++t) {
        sv += __hip_atomic_load(pz + (size_t)t * M + m0 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        fv += __hip_atomic_load(pz + (size_t)(nct + t) * M + m0 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // stage 1's partial means
      }
      predict_point_finish(fin, z, m0 + tid, sv, fv);
    }
  };
  {
    const char* ag = base_a(0);
    const char* bg = base_b(0);
#pragma unroll
    for (int j = 0; j < 8; ++j) dma(j, 0, ag, bg);
  }
  // one panel: wait for it, run it (with the next panel's DMA inside when there is one)
  auto panel = [&](int i, int stage, auto prefetch) {
    DMA_BARRIER();  // this wave's part of panel i has landed; behind the barrier, everyone's
    const char* ag = nullptr;
    const char* bg = nullptr;
    if constexpr (decltype(prefetch)::value) {
      ag = base_a(i + 1);
      bg = base_b(i + 1);
    }
    compute(stage, prefetch, ag, bg);
    if (i == P1 - 1 && P > P1) finish_item(0);
  };
  const std::true_type more;
  const std::false_type last;
  int i = 0;
  for (; i + 2 < P; i += 2) {
    panel(i, 0, more);
    panel(i + 1, 1, more);
  }
  if (P - i == 2) {
    panel(i, 0, more);
    panel(i + 1, 1, last);
  } else {
    panel(i, 0, last);
  }
  finish_item(nitem - 1);
#ifdef DMA_ABL_TIMES
  if (threadIdx.x == 0 && tb_ < 8192) {
    g_dma_times[tb_ * 4 + 1] = wall_clock64();
    g_dma_times[tb_ * 4 + 3] = P;
  }
#endif
}

// B[s] (N x N, row stride N) -> P[s] (ld x ld), zero filled beyond N: the operand layout of the
// kernel above.
__global__ __launch_bounds__(256) void pad_square_kernel(const double* __restrict__ B, int N, int ld,
                                                         double* __restrict__ P) {
  const int s = blockIdx.z, r = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= ld) return;
  P[((size_t)s * ld + r) * ld + c] = (r < N && c < N) ? B[((size_t)s * N + r) * N + c] : 0.0;
}

// The same partial sums for a handful of points (a CMA-ES population, a single candidate): with
// M <= 32 the 64-row MFMA tile above is mostly padding and its ~25 sequential LDS panels are pure
// latency (30 us for one point).  Here a workgroup owns 64 columns of one sample and a group of
// MT points (blockIdx.y); lane = column, its NW waves split the rows n of the triangular product
// (eight loads in flight each), the group's A rows sit in LDS and every B row is one coalesced
// 512-byte load.  Same outputs: part[tile_c * M + m].
template <int MT, int NW>
__global__ __launch_bounds__(64 * NW) void predict_var_small_kernel(const double* __restrict__ A,
                                                                const double* __restrict__ B,
                                                                int M, int N, double* __restrict__ part,
                                                                const double* __restrict__ Bfull,
                                                                const double* __restrict__ smeta,
                                                                int64_t part_stride) {
  extern __shared__ double sm[];  // the group's A rows [<= MT][N], then the cross-wave partials [NW][MT][64]
  const int z = blockIdx.z, tile_c = blockIdx.x;
  const int mg0 = blockIdx.y * MT;          // first point of this group
  const int Mtot = M;                       // part rows are indexed by the global point number
  const bool chol = smeta[3 * z] != 0.0;
  A += ((size_t)z * Mtot + mg0) * N;
  M = min(MT, Mtot - mg0);                  // points in this group
  B = (chol ? B : Bfull) + (size_t)z * N * N;
  part += (size_t)z * part_stride;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0 = tile_c * TS, c = c0 + lane;
  double* sA = sm;
  double* sT = sm + (size_t)M * N;
  for (int i = tid; i < M * N; i += 64 * NW) sA[i] = A[i];
  __syncthreads();
  const int nmax = chol ? min(N, c0 + TS) : N;  // upper-triangular L^-1: rows n <= c only
  double acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) acc[m] = 0.0;
  if (c < N) {
    constexpr int UB = 8;
    for (int n0 = wave; n0 < nmax; n0 += NW * UB) {
      double b[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int n = n0 + u * NW;
        b[u] = n < nmax ? B[(size_t)n * N + c] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int n = n0 + u * NW;
        if (n < nmax) {
#pragma unroll
          for (int m = 0; m < MT; ++m)
            if (m < M) acc[m] = fma(sA[m * N + n], b[u], acc[m]);
        }
      }
    }
  }
#pragma unroll
  for (int m = 0; m < MT; ++m) sT[(wave * MT + m) * 64 + lane] = acc[m];
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      if (m < M) {  // uniform
        double t = 0.0;
#pragma unroll
        for (int wv = 0; wv < NW; ++wv) t += sT[(wv * MT + m) * 64 + lane];
        double v = 0.0;
        if (c < N) v = chol ? t * t : sA[m * N + c] * t;
        v = fm::wave_sum_dpp(v);
        if (lane == 0) part[(size_t)tile_c * Mtot + mg0 + m] = v;
      }
    }
  }
}

// predict, stage 1 on the FP64 matrix cores: the dense pairwise-squared-distance block.
//   d2[m][n] = |a_m|^2 + |b_n|^2 - 2 a_m . b_n,   a = x*/ell, b = X/ell
// (the reference's own centred form, acquisition_functions/abstract_acq_fcn.py:195-222);
// the cross term a . b is a 64 x 64 x D product on v_mfma_f64_16x16x4_f64 (D padded to a
// multiple of 4), then Ks = sf^2 exp(-d2/2) (exp2 with folded constants), Ks * sW is stored
// for stage 2 and the partial means sum_{n in tile} Ks alpha_n go to fpart[ntile][m].
// Both sets are shifted by the column means of X first, as the reference's _sq_dist shifts by
// a common mean.  Cancellation: |d2 error| <= ~1e-16 (|a|^2+|b|^2), i.e. a relative error of
// the same size in Ks -- far inside the 1e-10 budget of the predictive variance.
// LDS row stride of the staged coordinate tiles.  The MFMA fragments are read as ds_read_b64 with lane =
// (row li = lane & 15, k-index lk = lane >> 4): a 32-lane group covers li = 0..15, lk = 0..1, dword bank
// (2 (stride li + lk)) mod 64.  An ODD stride (33, rounds 1-2) always puts some (li, lk = 1) on the
// bank of another (li', lk = 0) -- 38 % of the kernel's LDS cycles were conflict cycles (PMC, r02);
// stride = 2 mod 4 gives banks 4 li + 2 lk: 32 distinct ones.
constexpr int KDP = 32 + 2;
__global__ __launch_bounds__(256) void predict_kstar_mfma_kernel(
    const double* __restrict__ X, const double* __restrict__ xs, const double* __restrict__ alpha,
    const double* __restrict__ sW, const double* __restrict__ hyp, const double* __restrict__ cen,
    const double* __restrict__ smeta, int P, int N, int D, int64_t M, double* __restrict__ Ks,
    int lda, int64_t ks_stride, double* __restrict__ fpart, int64_t part_stride, int* __restrict__ tick) {
  // tick (may be null): arrival tickets of predict_var_dma_kernel's fused finish, one per row tile and sample, zeroed here
  if (tick != nullptr && blockIdx.x == 0 && threadIdx.x == 0) tick[(size_t)blockIdx.z * gridDim.y + blockIdx.y] = 0;
  // blockIdx.z = GP hyper-parameter sample: all S samples in one launch.  Ks: row stride lda,
  // sample stride ks_stride; lda > N (predict_var_dma_kernel's layout): columns N..lda-1 are zeroed.
  {
    const int s = blockIdx.z;
    alpha += (size_t)s * N;
    sW += (size_t)s * N;
    hyp += (size_t)s * P;
    Ks += (size_t)s * ks_stride;
    fpart += (size_t)s * part_stride;
  }
  const int scale_sw = smeta[3 * blockIdx.z] != 0.0;  // L_chol sample: stage 2 wants sW o K*
#ifdef KSTAR_ABL_TIMES
  const int kb_ = blockIdx.y * gridDim.x + blockIdx.x;
  if (threadIdx.x == 0 && kb_ < 2048) g_dma_times[kb_ * 4] = wall_clock64();
#endif
  __shared__ double sAm[TS * KDP];  // [64 m][d]
  __shared__ double sBn[TS * KDP];  // [64 n][d]
  __shared__ double sA2[TS], sB2[TS], sAl[TS], sSc[TS];
  __shared__ double sF[TS][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 15, lk = lane >> 4;
  const int64_t m0 = (int64_t)blockIdx.y * TS;
  const int n0 = blockIdx.x * TS;
  const int DQ = (D + 3) / 4;  // k-steps of 4
  // Staging: thread = (row r = tid / 4, quarter q = tid % 4) takes the coordinates d = q, q + 4, ...
  // of row r of both sets -- every global load of the workgroup is issued before the first is
  // used (one memory latency, not one per pass), and the squared norms fall out of the same
  // registers (quad sum by DPP), so a single barrier separates staging from the product.
  {
    const int r = tid >> 2, q = tid & 3;
    const int64_t m = m0 + r;
    const int n = n0 + r;
    double va[8], vb[8], ie[8], ce[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int d = q + 4 * i;
      const bool on = d < D;  // (D <= 32)
      va[i] = (on && m < M) ? xs[m * D + d] : 0.0;
      vb[i] = (on && n < N) ? X[(size_t)n * D + d] : 0.0;
      ie[i] = on ? hyp[d] : 0.0;
      ce[i] = on ? cen[d] : 0.0;
    }
    double a2 = 0.0, b2 = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int d = q + 4 * i;
      if (d < DQ * 4) {
        const bool on = d < D;
#ifdef KSTAR_ABL_NOEXP
        const double iell = on ? 1.0 - ie[i] : 0.0;
#else
        const double iell = on ? exp(-ie[i]) : 0.0;
#endif
        const double a = (on && m < M) ? (va[i] - ce[i]) * iell : 0.0;
        const double b = (on && n < N) ? (vb[i] - ce[i]) * iell : 0.0;
        sAm[r * KDP + d] = a;
        sBn[r * KDP + d] = b;
        a2 = fma(a, a, a2);
        b2 = fma(b, b, b2);
      }
    }
    a2 += fm::dpp_get<0xB1, 0xf>(a2);  // quad_perm [1,0,3,2]
    a2 += fm::dpp_get<0x4E, 0xf>(a2);  // quad_perm [2,3,0,1]
    b2 += fm::dpp_get<0xB1, 0xf>(b2);
    b2 += fm::dpp_get<0x4E, 0xf>(b2);
    if (q == 0) {
      sA2[r] = a2;
      sB2[r] = b2;
      sAl[r] = (n < N) ? alpha[n] : 0.0;
      sSc[r] = (n < N) ? (scale_sw ? sW[n] : 1.0) : 0.0;
    }
  }
  __syncthreads();
#ifdef KSTAR_ABL_TIMES
  const unsigned long long ks_t1 = wall_clock64();
#endif
  double4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
  for (int kq = 0; kq < DQ; ++kq) {
    const double a0 = sAm[(wm * 32 + li) * KDP + kq * 4 + lk];
    const double a1 = sAm[(wm * 32 + 16 + li) * KDP + kq * 4 + lk];
    const double b0 = sBn[(wn * 32 + li) * KDP + kq * 4 + lk];
    const double b1 = sBn[(wn * 32 + 16 + li) * KDP + kq * 4 + lk];
    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
  }
#ifdef KSTAR_ABL_TIMES
  const unsigned long long ks_t2 = wall_clock64();
#endif
  // epilogue: kernel values, store, partial means
  const double l2sf2 = 2.0 * hyp[D] * 0x1.71547652b82fep+0;  // log2(sf^2)
  const double c = -0.5 * 0x1.71547652b82fep+0;              // -log2(e)/2
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wm * 32 + mt * 16 + lk + 4 * r;
      const int64_t m = m0 + row;
      double f = 0.0;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int col = wn * 32 + nt * 16 + li;
        const int n = n0 + col;
        const double d2 = fmax(fma(-2.0, acc[mt][nt][r], sA2[row] + sB2[col]), 0.0);
        const double kv = fm::exp2_fast(fma(c, d2, l2sf2));
#ifndef KSTAR_ABL_NOSTORE
        // (sSc is 0 beyond N.)  Streaming store: the 8 N M bytes of K* are read next by another
        // kernel on other XCDs; written through now they overlap with this kernel's arithmetic
        // instead of being flushed from the L2s at its end.
        if (m < M && n < lda) __builtin_nontemporal_store(kv * sSc[col], Ks + (size_t)m * lda + n);
#endif
        f = fma(kv, sAl[col], f);  // alpha is 0 beyond N
      }
      f = fm::row16_sum_dpp(f);
      if (li == 0) sF[row][wn] = f;
    }
  __syncthreads();
  if (tid < TS && m0 + tid < M) fpart[(size_t)blockIdx.x * M + m0 + tid] = sF[tid][0] + sF[tid][1];
#ifdef KSTAR_ABL_TIMES
  if (threadIdx.x == 0 && kb_ < 2048) {
    g_dma_times[kb_ * 4 + 1] = wall_clock64();
    g_dma_times[kb_ * 4 + 2] = ks_t1;
    g_dma_times[kb_ * 4 + 3] = ks_t2;
  }
#endif
}

#if defined(DMA_ABL_TIMES) || defined(KSTAR_ABL_TIMES)
}  // namespace
extern "C" int vbmc_debug_dma_times(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dma_times), sizeof(unsigned long long) * n);
}
namespace {
#endif
// a13: AbstractAcqFcn._sq_dist (acquisition_functions/abstract_acq_fcn.py:195-222):
//   c[i][j] = max(|a_i - mu|^2 + |b_j - mu|^2 - 2 (a_i - mu).(b_j - mu), 0),
// mu = the size-weighted mean of both sets (computed by the caller, `cen`).  Same tiling as
// the kernel above; optional outputs: the matrix, and per row the (min, first argmin) of
// each 64-column tile for the nearest-neighbour lookup (:244-252).
__global__ __launch_bounds__(256) void sq_dist_mfma_kernel(
    const double* __restrict__ A, const double* __restrict__ B, const double* __restrict__ cen,
    int64_t NA, int NB, int D, double* __restrict__ C, double* __restrict__ pmin,
    int* __restrict__ pidx) {
  __shared__ double sAm[TS * KDP];
  __shared__ double sBn[TS * KDP];
  __shared__ double sA2[TS], sB2[TS];
  __shared__ double sMin[TS][2];
  __shared__ int sIdx[TS][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 15, lk = lane >> 4;
  const int64_t m0 = (int64_t)blockIdx.y * TS;
  const int n0 = blockIdx.x * TS;
  const int DQ = (D + 3) / 4;
  for (int idx = tid; idx < TS * DQ * 4; idx += 256) {
    const int r = idx / (DQ * 4), d = idx - r * (DQ * 4);
    const double c0 = (d < D) ? cen[d] : 0.0;
    const int64_t m = m0 + r;
    const int n = n0 + r;
    sAm[r * KDP + d] = (d < D && m < NA) ? A[m * D + d] - c0 : 0.0;
    sBn[r * KDP + d] = (d < D && n < NB) ? B[(size_t)n * D + d] - c0 : 0.0;
  }
  __syncthreads();
  if (tid < TS) {
    double a2 = 0.0, b2 = 0.0;
    for (int d = 0; d < D; ++d) {
      a2 = fma(sAm[tid * KDP + d], sAm[tid * KDP + d], a2);
      b2 = fma(sBn[tid * KDP + d], sBn[tid * KDP + d], b2);
    }
    sA2[tid] = a2;
    sB2[tid] = b2;
  }
  __syncthreads();
  double4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
  for (int kq = 0; kq < DQ; ++kq) {
    const double a0 = sAm[(wm * 32 + li) * KDP + kq * 4 + lk];
    const double a1 = sAm[(wm * 32 + 16 + li) * KDP + kq * 4 + lk];
    const double b0 = sBn[(wn * 32 + li) * KDP + kq * 4 + lk];
    const double b1 = sBn[(wn * 32 + 16 + li) * KDP + kq * 4 + lk];
    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wm * 32 + mt * 16 + lk + 4 * r;
      const int64_t m = m0 + row;
      double best = INFINITY;
      int bidx = 0x7fffffff;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int col = wn * 32 + nt * 16 + li;
        const int n = n0 + col;
        // the reference's association: |a|^2 + (|b|^2 - 2 a.b)
        const double d2 = fmax(sA2[row] + fma(-2.0, acc[mt][nt][r], sB2[col]), 0.0);
        if (m < NA && n < NB) {
          if (C) C[(size_t)m * NB + n] = d2;
          if (d2 < best) {  // nt ascending: ties keep the smaller column
            best = d2;
            bidx = n;
          }
        }
      }
      // first minimum over the 16 lanes that share this row
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) {
        const double ob = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(bidx, off, 64);
        if (ob < best || (ob == best && oi < bidx)) {
          best = ob;
          bidx = oi;
        }
      }
      if (li == 0) {
        sMin[row][wn] = best;
        sIdx[row][wn] = bidx;
      }
    }
  if (!pmin) return;
  __syncthreads();
  if (tid < TS && m0 + tid < NA) {
    double best = sMin[tid][0];
    int bidx = sIdx[tid][0];
    if (sMin[tid][1] < best) {  // columns of wn = 1 are larger: ties keep wn = 0
      best = sMin[tid][1];
      bidx = sIdx[tid][1];
    }
    pmin[(size_t)blockIdx.x * NA + m0 + tid] = best;
    pidx[(size_t)blockIdx.x * NA + m0 + tid] = bidx;
  }
}

__global__ void sq_dist_argmin_kernel(const double* __restrict__ pmin, const int* __restrict__ pidx,
                                      int ntiles, int64_t NA, int64_t* __restrict__ out) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= NA) return;
  double best = pmin[m];
  int bidx = pidx[m];
  for (int t = 1; t < ntiles; ++t) {  // tiles ascending: ties keep the first (np.argmin)
    const double v = pmin[(size_t)t * NA + m];
    if (v < best) {
      best = v;
      bidx = pidx[(size_t)t * NA + m];
    }
  }
  out[m] = bidx;
}

// predict, stage 3: fmu[m] = mean(x*_m) + sum of the stage-1 partial means,
// fs2[m] = max(0, sf^2 -/+ sum of the stage-2 partial row sums) (+ noise).
__global__ void predict_finish_kernel(const double* __restrict__ part_all, int64_t part_stride, int ntiles,
                                      int64_t M, PredFin fin) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const int smp = blockIdx.y;  // GP sample
  const double* part = part_all + (size_t)smp * part_stride;
  const double* fpart = part + (size_t)ntiles * M;
  double s = 0.0, f = 0.0;
  for (int t = 0; t < ntiles; ++t) {
    s += part[(size_t)t * M + m];
    f += fpart[(size_t)t * M + m];
  }
  predict_point_finish(fin, smp, m, s, f);
}

}  // namespace

// ---------------------------------------------------------------------------
void glj_fill_prep(const vbmc_ctx* ctx, int want_grad, double* res, double* Z, PrepArgs& a) {
  const GpState& g = ctx->gp;
  a.mix = ctx->d_mix;
  a.ml = ctx->ml;
  a.n_glj = g.S * ctx->K;
  a.N = g.N;
  a.P = g.P;
  a.want_grad = want_grad;
  a.X = g.d_X;
  a.XT = g.d_XT;
  a.alpha = g.d_alpha;
  a.hyp = g.d_hyp;
  a.res = res;
  a.Z = Z;
}

int launch_gp_log_joint(vbmc_ctx* ctx, int want_grad, double* d_res, double* d_Z) {
  PrepArgs a;
  glj_fill_prep(ctx, want_grad, d_res, d_Z, a);
  if (ctx->timing) HIP_TRY(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
  int rc = launch_prep(ctx, a);
  if (rc) return rc;
  if (ctx->timing) {
    HIP_TRY(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
    ctx->ev_valid[1] = true;
  }
  return 0;
}

// Q[s][j][k] = z_j^T (L^T L)^-1 z_k (L_chol)  or  z_j^T L z_k (otherwise), from Z.
int launch_gp_var(vbmc_ctx* ctx, const double* d_Z, double* d_V, double* d_Q) {
  const GpState& g = ctx->gp;
  const int K = ctx->K, N = g.N, S = g.S;
  // V = Z L^-1 (upper-triangular skip) or Z L on the FP64 matrix cores, then the K x K Gram
  // matrix; every GP sample in the same two launches
  hipLaunchKernelGGL(predict_var_mfma_kernel, dim3((N + TS - 1) / TS, (K + TS - 1) / TS, S), dim3(256), 0,
                     ctx->stream, d_Z, (const double*)g.d_Linv, (int64_t)K, N, 0, (double*)nullptr, d_V,
                     (const double*)g.d_L, (const double*)g.d_smeta, (int64_t)0);
  hipLaunchKernelGGL(gram_kernel, dim3((K * K + WAVES - 1) / WAVES, S), dim3(256), 0, ctx->stream, d_Z,
                     (const double*)d_V, (const double*)ctx->d_mix, ctx->ml, (const double*)g.d_hyp, g.P, N,
                     (const double*)g.d_smeta, d_Q);
  HIP_TRY(ctx, hipGetLastError());
  return 0;
}

// the zero-padded copy of L^-1 predict_var_dma_kernel reads (leading dimension N rounded up to 64)
static int launch_pad_linv(vbmc_ctx* ctx) {
  GpState& g = ctx->gp;
  if (!g.d_LinvP) return 0;
  const int ld = predict_ld(g.N);
  hipLaunchKernelGGL(pad_square_kernel, dim3((ld + 255) / 256, ld, g.S), dim3(256), 0, ctx->stream,
                     (const double*)g.d_Linv, g.N, ld, g.d_LinvP);
  HIP_TRY(ctx, hipGetLastError());
  return 0;
}

int launch_trinv(vbmc_ctx* ctx) {
  GpState& g = ctx->gp;
  const int N = g.N, S = g.S;
  const int nb = (N + TRB - 1) / TRB;
  const size_t lds = sizeof(double) * ((size_t)nb * TRB * TRS + TRB * TRS);
  if (lds > 156 * 1024) {  // N > 1088
    hipLaunchKernelGGL(trinv_upper_kernel, dim3((N + 63) / 64, S), dim3(64), 0, ctx->stream, g.d_L, N, g.d_Linv);
    HIP_TRY(ctx, hipGetLastError());
    return launch_pad_linv(ctx);
  }
  int rc = ensure_dev(ctx, &ctx->d_scratch, &ctx->d_scratch_cap, (size_t)S * nb * TRB * TRB);
  if (rc) return rc;
  double* Dinv = ctx->d_scratch;
  hipLaunchKernelGGL(trinv_diag_kernel, dim3(nb, S), dim3(64), 0, ctx->stream, (const double*)g.d_L, N,
                     (const double*)g.d_smeta, Dinv, nb);
  static size_t lds_set[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (lds > 48 * 1024 && lds > lds_set[dev & 63]) {
    HIP_TRY(ctx, hipFuncSetAttribute((const void*)trinv_strip_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    lds_set[dev & 63] = lds;
  }
  hipLaunchKernelGGL(trinv_strip_kernel, dim3((N + TRS - 1) / TRS, S), dim3(256), lds, ctx->stream,
                     (const double*)g.d_L, N, (const double*)g.d_smeta, (const double*)Dinv, nb, g.d_Linv);
  HIP_TRY(ctx, hipGetLastError());
  return launch_pad_linv(ctx);
}

// (grid.z / grid.y = sample).  d_Ks: S * M * N doubles; d_part: S * 2 * ntiles * M doubles;
// d_fmu / d_fs2: [S][ld].
// predict, stages 1 and 2 for all GP samples: K* (+ the partial means) and the variance product's partial row sums
// `fin` (may be null): the caller wants fmu / fs2 themselves; where the LDS-direct product kernel runs it then finishes the
// points in its epilogue and *fin_done = true; otherwise the caller launches predict_finish_kernel on the partial sums.
int launch_gp_predict_products(vbmc_ctx* ctx, int64_t M, const double* d_xs, double* d_Ks, double* d_part,
                               const void* fin_v, bool* fin_done) {
  const GpState& g = ctx->gp;
  const int N = g.N, D = g.D, S = g.S;
  const int ntiles = (N + TS - 1) / TS;
  const int64_t pstride = 2 * (int64_t)ntiles * M;
  const dim3 grid(ntiles, (unsigned)((M + TS - 1) / TS), S);
  if (fin_done) *fin_done = false;
  // M > 32 on Cholesky samples: the LDS-direct kernel on padded operands; otherwise the plain layout
  bool dma = ctx->opt_predict_dma && !(M <= 32 && N <= 3000) && g.d_LinvP;
  for (int s = 0; s < S && dma; ++s) dma = g.L_chol[s] != 0;
  const int lda = dma ? ntiles * TS : N;
  const int64_t ks_stride = dma ? (int64_t)((M + TS - 1) / TS) * TS * lda : M * N;
  const int nrt = (int)((M + TS - 1) / TS), G = nrt * S;
  PredFin fin;
  // The finish in the product's epilogue (two launches instead of three) pays where the product grid is ONE round of
  // workgroups (two per CU): the arrival costs every workgroup an atomic's round trip at its end, once per round --
  // M = 8192, S = 1: 61.1 -> 59.6 us between events; S = 4 (four rounds): 187 -> 205 us, so those keep the finish
  // launch.  Not while the product alone is being timed (vbmc_set_timing(2)).
  const int fuse_mode = ctx->opt_predict_fused;
  const int cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
  const int nprod = 8 * (((G + 1) / 2 + 7) / 8) * (ntiles + 1);
  if (dma && fin_v != nullptr && ctx->timing < 2 && (fuse_mode == 2 || (fuse_mode == 1 && nprod <= 2 * cus))) {
    const int rc = ensure_dev(ctx, &ctx->d_ptick, &ctx->d_ptick_cap, (size_t)(G + 1) / 2 + 1);
    if (rc) return rc;
    fin = *(const PredFin*)fin_v;
    fin.tick = (int*)ctx->d_ptick;
    *fin_done = true;
  }
  hipLaunchKernelGGL(predict_kstar_mfma_kernel, grid, dim3(256), 0, ctx->stream, g.d_X, d_xs,
                     (const double*)g.d_alpha, (const double*)g.d_sW, (const double*)g.d_hyp,
                     (const double*)g.d_xc, (const double*)g.d_smeta, g.P, N, D, M, d_Ks, lda, ks_stride,
                     d_part + (size_t)ntiles * M, pstride, fin.tick);
  if (ctx->timing >= 2) HIP_TRY(ctx, hipEventRecord(ctx->ev[10], ctx->stream));
  if (dma) {
    static bool lds_set[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!lds_set[dev & 63]) {
      HIP_TRY(ctx, hipFuncSetAttribute((const void*)predict_var_dma_kernel,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, DMA_LDS));
      lds_set[dev & 63] = true;
    }
    const int ngr = (G + 1) / 2;
    const dim3 pgrid((unsigned)(8 * ((ngr + 7) / 8) * (ntiles + 1)));
    hipLaunchKernelGGL(predict_var_dma_kernel, pgrid, dim3(256), DMA_LDS, ctx->stream, (const double*)d_Ks,
                       (const double*)g.d_LinvP, M, N, lda, ks_stride, d_part, pstride, nrt, G, fin);
  } else if (M <= 32 && N <= 3000) {  // a handful of points: see predict_var_small_kernel (LDS <= 128 KB)
    const dim3 sgrid(ntiles, (unsigned)((M + 3) / 4), S);
    const size_t lds = sizeof(double) * ((size_t)4 * N + 16 * 4 * 64);
    if (lds > 64 * 1024)
      HIP_TRY(ctx, hipFuncSetAttribute((const void*)predict_var_small_kernel<4, 16>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
    hipLaunchKernelGGL((predict_var_small_kernel<4, 16>), sgrid, dim3(1024), lds, ctx->stream,
                       (const double*)d_Ks, (const double*)g.d_Linv, (int)M, N, d_part, (const double*)g.d_L,
                       (const double*)g.d_smeta, pstride);
  } else {
    hipLaunchKernelGGL(predict_var_mfma_kernel, grid, dim3(256), 0, ctx->stream, (const double*)d_Ks,
                       (const double*)g.d_Linv, M, N, 0, d_part, (double*)nullptr, (const double*)g.d_L,
                       (const double*)g.d_smeta, pstride);
  }
  if (ctx->timing >= 2) {
    HIP_TRY(ctx, hipEventRecord(ctx->ev[11], ctx->stream));
    ctx->ev_valid[5] = true;
  } else {
    ctx->ev_valid[5] = false;  // (no pair around this product: an interval left by an earlier level-2 call is not this call's)
  }
  HIP_TRY(ctx, hipGetLastError());
  return 0;
}

// All S GP samples, one batch of M points already on the device: K*, the variance product with the finish in its epilogue
// (batches on Cholesky samples: two launches) or K*, product, finish (three).
int launch_gp_predict_all(vbmc_ctx* ctx, int64_t M, const double* d_xs, double* d_Ks, double* d_part,
                          int add_noise, double* d_fmu, double* d_fs2, int64_t ld) {
  const GpState& g = ctx->gp;
  const int N = g.N, D = g.D, S = g.S;
  const int ntiles = (N + TS - 1) / TS;
  const int64_t pstride = 2 * (int64_t)ntiles * M;
  PredFin fin;
  fin.hyp_all = g.d_hyp;
  fin.smeta = g.d_smeta;
  fin.xs = d_xs;
  fin.fmu = d_fmu;
  fin.fs2 = d_fs2;
  fin.ld = ld;
  fin.D = D;
  fin.P = g.P;
  fin.mean_kind = g.mean_kind;
  fin.add_noise = add_noise;
  bool done = false;
  const int rc = launch_gp_predict_products(ctx, M, d_xs, d_Ks, d_part, &fin, &done);
  if (rc) return rc;
  if (!done)
    hipLaunchKernelGGL(predict_finish_kernel, dim3((unsigned)((M + 255) / 256), S), dim3(256), 0, ctx->stream,
                       (const double*)d_part, pstride, ntiles, M, fin);
  HIP_TRY(ctx, hipGetLastError());
  return 0;
}

// d_pmin: scratch of 2 * ntiles * n doubles (tile minima, then their int indices) when the
// argmin is wanted.
int launch_sq_dist(vbmc_ctx* ctx, const double* d_a, int64_t n, const double* d_b, int m, int D,
                   const double* d_cen, double* d_c, double* d_pmin, int64_t* d_argmin) {
  const int ntiles = (m + TS - 1) / TS;
  int* d_pidx = d_argmin ? (int*)(d_pmin + (size_t)ntiles * n) : nullptr;
  const dim3 grid(ntiles, (unsigned)((n + TS - 1) / TS));
  hipLaunchKernelGGL(sq_dist_mfma_kernel, grid, dim3(256), 0, ctx->stream, d_a, d_b, d_cen, n, m, D,
                     d_c, d_argmin ? d_pmin : (double*)nullptr, d_pidx);
  if (d_argmin)
    hipLaunchKernelGGL(sq_dist_argmin_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const double*)d_pmin, (const int*)d_pidx, ntiles, n, d_argmin);
  HIP_TRY(ctx, hipGetLastError());
  return 0;
}
