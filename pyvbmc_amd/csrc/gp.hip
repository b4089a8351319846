// GP side of the ELBO path on gfx950:
//   * expected log joint of the mixture under the GP surrogate (Bayesian quadrature),
//     reference vbmc/variational_optimization.py:1374-1514 (_gp_log_joint);
//   * its variance: the reference's K(K+1)/2 pairs of triangular solves (:1489-1501)
//     restructured as ONE product V = Z L^-1 (K x N x N, triangular) + a K x K Gram
//     matrix, with L^-1 formed once per GP update (the GP is fixed during the
//     thousands of ELBO evaluations of one variational optimisation);
//   * GP.predict of gpyreg (third party): K* block, mean,
//     variance via the same L^-1.
#include <cmath>
#include <cstring>

#include "common.h"

namespace {

constexpr int WAVES = 4;

__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Inverse of an upper-triangular matrix, one thread per column (back substitution);
// run once per vbmc_set_gp.
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
__global__ __launch_bounds__(256) void gram_kernel(const double* __restrict__ U,
                                                   const double* __restrict__ V,
                                                   const double* __restrict__ mix, MixLayout ml,
                                                   const double* __restrict__ hyp, int N, int chol,
                                                   double inv_sn2, double* __restrict__ J) {
  const int D = ml.D, K = ml.K;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pair = blockIdx.x * WAVES + wave;
  if (pair >= K * K) return;
  const int j = pair / K, k = pair - j * K;
  if (j > k) return;
  const double* u = U + (size_t)j * N;
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
// predict, stage 1: Ks[m][n] = sf^2 exp(-1/2 sum_d ((X_nd - x*_md)/ell_d)^2) (times
// sW[n] if scale), and fmu[m] = m(x*_m) + sum_n Ks[m][n] alpha_n.   16 points / block.
constexpr int TMP = 16;
__global__ __launch_bounds__(256) void predict_kstar_kernel(
    const double* __restrict__ X, const double* __restrict__ xs, const double* __restrict__ alpha,
    const double* __restrict__ sW, const double* __restrict__ hyp, int N, int D, int mean_kind,
    int64_t M, int scale_sw, double* __restrict__ Ks, double* __restrict__ fmu) {
  extern __shared__ double lds[];
  double* sXs = lds;              // [TMP][D]  x* / ell
  double* sIell = sXs + TMP * D;  // [D]
  double* sRed = sIell + D;       // [WAVES][TMP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t m0 = (int64_t)blockIdx.x * TMP;
  if (tid < D) sIell[tid] = exp(-hyp[tid]);
  __syncthreads();
  for (int idx = tid; idx < TMP * D; idx += 256) {
    const int mm = idx / D, d = idx - mm * D;
    const int64_t m = m0 + mm;
    sXs[idx] = (m < M) ? xs[m * D + d] * sIell[d] : 0.0;
  }
  __syncthreads();
  const double sf2 = exp(2.0 * hyp[D]);
  double part[TMP];
#pragma unroll
  for (int mm = 0; mm < TMP; ++mm) part[mm] = 0.0;
  for (int n = tid; n < N; n += 256) {
    const double an = alpha[n];
    const double sc = scale_sw ? sW[n] : 1.0;
#pragma unroll
    for (int mm = 0; mm < TMP; ++mm) {
      double d2 = 0.0;
      for (int d = 0; d < D; ++d) {
        const double t = X[(size_t)n * D + d] * sIell[d] - sXs[mm * D + d];
        d2 = fma(t, t, d2);
      }
      const double kv = sf2 * exp(-0.5 * d2);
      part[mm] = fma(kv, an, part[mm]);
      if (m0 + mm < M) Ks[(size_t)(m0 + mm) * N + n] = kv * sc;
    }
  }
#pragma unroll
  for (int mm = 0; mm < TMP; ++mm) {
    const double v = wave_sum(part[mm]);
    if (lane == 0) sRed[wave * TMP + mm] = v;
  }
  __syncthreads();
  if (tid < TMP && m0 + tid < M) {
    double v = 0.0;
    for (int wv = 0; wv < WAVES; ++wv) v += sRed[wv * TMP + tid];
    // mean function at x* (variational_optimization.py:1383-1392 layout)
    double mean = 0.0;
    const double* hm = hyp + D + 2;
    if (mean_kind == VBMC_MEAN_CONST) mean = hm[0];
    if (mean_kind == VBMC_MEAN_NEGQUAD) {
      mean = hm[0];
      for (int d = 0; d < D; ++d) {
        const double t = (xs[(m0 + tid) * D + d] - hm[1 + d]) * exp(-hm[1 + D + d]);
        mean -= 0.5 * t * t;
      }
    }
    fmu[m0 + tid] = mean + v;
  }
}

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
typedef double double4_t __attribute__((ext_vector_type(4)));
constexpr int TS = 64, TKD = 16, LDA = TKD + 1, LDB = TS + 16;
__global__ __launch_bounds__(256) void predict_var_mfma_kernel(const double* __restrict__ A,
                                                               const double* __restrict__ B,
                                                               int64_t M, int N, int mode,
                                                               double* __restrict__ part,
                                                               double* __restrict__ Cout) {
  __shared__ double sA[TS * LDA];
  __shared__ double sB[TKD * LDB];
  __shared__ double sRow[TS][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wc = wave & 1;
  const int li = lane & 15, lk = lane >> 4;
  const int64_t m0 = (int64_t)blockIdx.y * TS;
  const int c0 = blockIdx.x * TS;
  double4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
  const int nmax = (mode == 0) ? min(N, c0 + TS) : N;  // upper-triangular B: n <= c
  for (int n0 = 0; n0 < nmax; n0 += TKD) {
    for (int idx = tid; idx < TS * TKD; idx += 256) {
      const int r = idx / TKD, kk = idx - r * TKD;
      const int64_t m = m0 + r;
      const int n = n0 + kk;
      sA[r * LDA + kk] = (m < M && n < N) ? A[(size_t)m * N + n] : 0.0;
      const int kb = idx / TS, cc = idx - kb * TS;
      const int nb = n0 + kb, c = c0 + cc;
      sB[kb * LDB + cc] = (nb < N && c < N) ? B[(size_t)nb * N + c] : 0.0;
    }
    __syncthreads();
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
      v += __shfl_xor(v, 1, 64);
      v += __shfl_xor(v, 2, 64);
      v += __shfl_xor(v, 4, 64);
      v += __shfl_xor(v, 8, 64);
      if (li == 0) sRow[row][wc] = v;
    }
  __syncthreads();
  if (tid < TS && m0 + tid < M) part[(size_t)blockIdx.x * M + m0 + tid] = sRow[tid][0] + sRow[tid][1];
}

__global__ void predict_var_finish_kernel(const double* __restrict__ part, int ntiles, int64_t M,
                                          double sf2, double sign, double add,
                                          double* __restrict__ fs2) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  double s = 0.0;
  for (int t = 0; t < ntiles; ++t) s += part[(size_t)t * M + m];
  fs2[m] = fmax(sf2 + sign * s, 0.0) + add;
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
  const int K = ctx->K, N = g.N;
  bool all_chol = true, none_chol = true;
  for (int s = 0; s < g.S; ++s) {
    all_chol = all_chol && g.L_chol[s];
    none_chol = none_chol && !g.L_chol[s];
  }
  for (int s = 0; s < g.S; ++s) {
    const int chol = g.L_chol[s];
    const double* Bm = (chol ? g.d_Linv : g.d_L) + (size_t)s * N * N;
    // V = Z L^-1 (upper-triangular skip) or Z L, on the FP64 matrix cores
    hipLaunchKernelGGL(predict_var_mfma_kernel, dim3((N + TS - 1) / TS, (K + TS - 1) / TS), dim3(256),
                       0, ctx->stream, d_Z + (size_t)s * K * N, Bm, (int64_t)K, N, chol ? 0 : 1,
                       (double*)nullptr, d_V + (size_t)s * K * N);
    const double* U = chol ? d_V + (size_t)s * K * N : d_Z + (size_t)s * K * N;
    hipLaunchKernelGGL(gram_kernel, dim3((K * K + WAVES - 1) / WAVES, 1), dim3(256), 0, ctx->stream,
                       U, (const double*)(d_V + (size_t)s * K * N), (const double*)ctx->d_mix,
                       ctx->ml, (const double*)(g.d_hyp + (size_t)s * g.P), N, chol,
                       1.0 / g.sn2_eff[s], d_Q + (size_t)s * K * K);
  }
  (void)all_chol;
  (void)none_chol;
  HIP_TRY(ctx, hipGetLastError());
  return 0;
}

int launch_trinv(vbmc_ctx* ctx) {
  GpState& g = ctx->gp;
  hipLaunchKernelGGL(trinv_upper_kernel, dim3((g.N + 63) / 64, g.S), dim3(64), 0, ctx->stream,
                     g.d_L, g.N, g.d_Linv);
  HIP_TRY(ctx, hipGetLastError());
  return 0;
}

// One GP sample s, one batch of M points already on the device.
int launch_gp_predict_sample(vbmc_ctx* ctx, int s, int64_t M, const double* d_xs, double* d_Ks,
                             double* d_part, int add_noise, double* d_fmu, double* d_fs2) {
  const GpState& g = ctx->gp;
  const int N = g.N, D = g.D;
  const double* h = g.hyp.data() + (size_t)s * g.P;
  const int chol = g.L_chol[s];
  size_t lds = sizeof(double) * ((size_t)TMP * D + D + WAVES * TMP);
  hipLaunchKernelGGL(predict_kstar_kernel, dim3((unsigned)((M + TMP - 1) / TMP)), dim3(256), lds,
                     ctx->stream, g.d_X, d_xs, g.d_alpha + (size_t)s * N, g.d_sW + (size_t)s * N,
                     g.d_hyp + (size_t)s * g.P, N, D, g.mean_kind, M, chol, d_Ks, d_fmu);
  const int ntiles = (N + TS - 1) / TS;
  const double* Bm = (chol ? g.d_Linv : g.d_L) + (size_t)s * N * N;
  hipLaunchKernelGGL(predict_var_mfma_kernel, dim3(ntiles, (unsigned)((M + TS - 1) / TS)), dim3(256),
                     0, ctx->stream, d_Ks, Bm, M, N, chol ? 0 : 1, d_part, (double*)nullptr);
  const double sf2 = std::exp(2.0 * h[D]);
  const double add = add_noise ? std::exp(2.0 * h[D + 1]) * g.sn2_mult[s] : 0.0;
  hipLaunchKernelGGL(predict_var_finish_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0,
                     ctx->stream, d_part, ntiles, M, sf2, chol ? -1.0 : 1.0, add, d_fs2);
  HIP_TRY(ctx, hipGetLastError());
  return 0;
}
