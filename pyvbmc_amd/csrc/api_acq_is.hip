// The two importance-sampled acquisition functions for noisy targets, AcqFcnVIQR and AcqFcnIMIQR:
//   acquisition_functions/acq_fcn_viqr.py:30-160, acq_fcn_imiqr.py:29-177
// (the classes AbstractAcqFcn.__call__ dispatches to when the target's evaluations are noisy --
// BASELINE config 5).  Per GP hyper-parameter sample s, with Xa the importance points the
// reference prepared (vbmc/active_importance_sampling.py) and C_tmp[s] = (K+Sigma)^-1 K(X, Xa)
// (or L K(X, Xa) for a non-Cholesky sample) formed there once (:262-306):
//     C      = K(Xs, Xa) -/+ K(Xs, X) C_tmp[s]                 posterior cross-covariance
//     tau2   = C^2 / (f_s2(Xs) + sn2(Xs))
//     s_pred = sqrt(max(f_s2(Xa) - tau2, 0))
//     zz     = ln_w[s] + u s_pred + log1p(-exp(-2 u s_pred))     (ln_w = 0 for VIQR)
//     acq_s  = logsumexp_a zz ,     acq = logsumexp_s acq_s - log S
// The importance state (Xa, C_tmp, f_s2(Xa), ln_w) is uploaded once per active-sampling round
// (vbmc_acq_is_set) and stays in HBM while CMA-ES calls vbmc_acq_is_eval thousands of times.
// Kernels: the predictive variance at Xs through the predict launches of gp.hip; K(Xs, X) by direct
// differences; the M x N x Na product on the FP64 matrix cores; one wave per (point, sample) for
// the cross-kernel K(Xs, Xa), the integrand and its log-sum-exp.
#include <cmath>
#include <cstring>

#include "common.h"
#include "fastmath.h"

namespace {

typedef double double4_t __attribute__((ext_vector_type(4)));

struct IsState {
  int64_t Na = 0;
  int S = 0, N = 0, D = 0, per_sample = 0, has_lnw = 0;
  double* d = nullptr;  // [Xa (S or 1) x Na x D | Ctmp S x N x Na | fs2a S x Na | lnw S x Na]
  size_t cap = 0;
  size_t o_C = 0, o_f = 0, o_w = 0;
};

// K[m][n] = sf2 exp(-1/2 sum_d ((a_md - b_nd)/ell_d)^2), direct differences
__global__ __launch_bounds__(256) void se_cross_kernel(const double* __restrict__ A, int64_t M,
                                                       const double* __restrict__ B, int NB, int D,
                                                       const double* __restrict__ hyp, double* __restrict__ K) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * NB) return;
  const int64_t m = idx / NB;
  const int n = (int)(idx - m * NB);
  double d2 = 0.0;
  for (int d = 0; d < D; ++d) {
    const double t = (A[m * D + d] - B[(size_t)n * D + d]) * exp(-hyp[d]);
    d2 = fma(t, t, d2);
  }
  K[idx] = exp(2.0 * hyp[D] - 0.5 * d2);
}

// C[M x NC] = A[M x KD] B[KD x NC], row-major, FP64 matrix cores; 64 x 64 tiles, 16-deep LDS panels
// (the layout of predict_var_mfma_kernel in gp.hip without its triangular skip and row epilogue).
constexpr int GT = 64, GK = 16, GLA = GK + 1, GLB = GT + 16;
__global__ __launch_bounds__(256) void gemm_nn_mfma_kernel(const double* __restrict__ A,
                                                           const double* __restrict__ B,
                                                           double* __restrict__ C, int64_t M, int KD,
                                                           int NC) {
  __shared__ double sA[GT * GLA];
  __shared__ double sB[GK * GLB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wc = wave & 1, li = lane & 15, lk = lane >> 4;
  const int64_t m0 = (int64_t)blockIdx.y * GT;
  const int c0 = blockIdx.x * GT;
  double4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < KD; k0 += GK) {
    for (int idx = tid; idx < GT * GK; idx += 256) {
      const int r = idx / GK, kk = idx - r * GK;
      const int64_t m = m0 + r;
      sA[r * GLA + kk] = (m < M && k0 + kk < KD) ? A[(size_t)m * KD + k0 + kk] : 0.0;
      const int kb = idx / GT, cc = idx - kb * GT;
      sB[kb * GLB + cc] = (k0 + kb < KD && c0 + cc < NC) ? B[(size_t)(k0 + kb) * NC + c0 + cc] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int kq = 0; kq < GK / 4; ++kq) {
      const double a0 = sA[(wm * 32 + li) * GLA + kq * 4 + lk];
      const double a1 = sA[(wm * 32 + 16 + li) * GLA + kq * 4 + lk];
      const double b0 = sB[(kq * 4 + lk) * GLB + wc * 32 + li];
      const double b1 = sB[(kq * 4 + lk) * GLB + wc * 32 + 16 + li];
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t m = m0 + wm * 32 + mt * 16 + lk + 4 * r;
        const int c = c0 + wc * 32 + ct * 16 + li;
        if (m < M && c < NC) C[(size_t)m * NC + c] = acc[mt][ct][r];
      }
}

// One wave per point m: acq_s[m] = logsumexp_a zz(m, a) for GP sample s.
__global__ __launch_bounds__(256) void is_reduce_kernel(const double* __restrict__ xs, int64_t M, int D,
                                                        const double* __restrict__ Xa, int64_t Na,
                                                        const double* __restrict__ hyp,
                                                        const double* __restrict__ T,      // M x Na
                                                        const double* __restrict__ fs2,    // M   (f_s2 at Xs, this sample)
                                                        const double* __restrict__ sn2,    // M
                                                        const double* __restrict__ fs2a,   // Na  (f_s2 at Xa, this sample)
                                                        const double* __restrict__ lnw,    // Na or null
                                                        int chol, double u, double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t m = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (m >= M) return;
  const double iy = 1.0 / (fs2[m] + sn2[m]);
  const double lsf2 = 2.0 * hyp[D];
  double mx = -INFINITY, sm = 0.0;  // per-lane running log-sum-exp
  bool bad = false;
  for (int64_t a = lane; a < Na; a += 64) {
    double d2 = 0.0;
    for (int d = 0; d < D; ++d) {
      const double t = (xs[m * D + d] - Xa[a * D + d]) * exp(-hyp[d]);
      d2 = fma(t, t, d2);
    }
    const double kxa = exp(lsf2 - 0.5 * d2);
    const double t = T[(size_t)m * Na + a];
    const double c = chol ? kxa - t : kxa + t;
    const double tau2 = c * c * iy;
    const double sp = sqrt(fmax(fs2a[a] - tau2, 0.0));
    double zz = u * sp + log1p(-exp(-2.0 * u * sp));
    if (lnw) zz += lnw[a];
    bad |= zz != zz;  // a NaN integrand (NaN f_s2, C_tmp, sn2 ...) makes the reference's logsumexp NaN: so here
    if (zz > -INFINITY) {
      if (zz > mx) {
        sm = sm * exp(mx - zz) + 1.0;
        mx = zz;
      } else {
        sm += exp(zz - mx);
      }
    }
  }
  // combine the 64 lanes
  double gm = mx;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) gm = fmax(gm, __shfl_xor(gm, off, 64));
  double part = (mx > -INFINITY && gm < INFINITY) ? sm * exp(mx - gm) : 0.0;
  part = fm::wave_sum_dpp(part);
  const bool any_bad = __any(bad ? 1 : 0) != 0;
  if (lane == 0)
    out[m] = any_bad ? (double)NAN : gm == INFINITY ? (double)INFINITY : (gm > -INFINITY) ? gm + log(part) : -INFINITY;
}

// acq[m] = logsumexp_s acq_s[s][m] - log S   (acq_fcn_viqr.py:152-158)
// var_tot[m] = mean_s f_s2 + var_s(f_mu, ddof = 1) (abstract_acq_fcn.py:82-97), for the caller's
// variance regularisation
__global__ void is_combine_kernel(const double* __restrict__ acq_s, int S, int64_t M, int64_t ld,
                                  double* __restrict__ acq, const double* __restrict__ fmu,
                                  const double* __restrict__ fs2, double* __restrict__ var_tot) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  {
    double fsum = 0.0, vsum = 0.0;
    for (int s = 0; s < S; ++s) {
      fsum += fmu[(size_t)s * ld + m];
      vsum += fs2[(size_t)s * ld + m];
    }
    const double f_bar = fsum / S;
    double q = 0.0;
    for (int s = 0; s < S; ++s) {
      const double t = fmu[(size_t)s * ld + m] - f_bar;
      q += t * t;
    }
    var_tot[m] = (S > 1 ? q / (S - 1) : 0.0) + vsum / S;
  }
  if (S == 1) {
    acq[m] = acq_s[m];
    return;
  }
  double mx = -INFINITY;
  for (int s = 0; s < S; ++s) mx = fmax(mx, acq_s[(size_t)s * ld + m]);
  if (!(mx > -INFINITY)) mx = 0.0;  // avoid -inf + inf
  double sum = 0.0;
  for (int s = 0; s < S; ++s) sum += exp(acq_s[(size_t)s * ld + m] - mx);
  acq[m] = mx + log(sum / S);
}

IsState* is_of(vbmc_ctx* ctx) {
  if (!ctx->acq_is) ctx->acq_is = new IsState();
  return (IsState*)ctx->acq_is;
}

}  // namespace

void acq_is_free(vbmc_ctx* ctx) {
  IsState* st = (IsState*)ctx->acq_is;
  if (!st) return;
  if (st->d) (void)hipFree(st->d);
  delete st;
  ctx->acq_is = nullptr;
}

extern "C" int vbmc_acq_is_set(vbmc_ctx* ctx, int64_t Na, const double* Xa, int per_sample_xa,
                               const double* Ctmp_SxNxNa, const double* fs2a_NaxS,
                               const double* lnw_SxNa) {
  if (!ctx || Na < 1 || !Xa || !Ctmp_SxNxNa || !fs2a_NaxS) return VBMC_E_ARG;
  NEED_DEVICE(ctx);
  if (!ctx->gp.set) return vbmc_fail(ctx, VBMC_E_ARG, "acq_is_set: GP not set");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const GpState& g = ctx->gp;
  IsState* st = is_of(ctx);
  const int S = g.S, N = g.N, D = g.D;
  const size_t n_xa = (size_t)(per_sample_xa ? S : 1) * Na * D, n_C = (size_t)S * N * Na, n_f = (size_t)S * Na;
  const size_t need = n_xa + n_C + 2 * n_f;
  HIP_TRY(ctx, stream_wait(ctx));
  if (st->cap < need) {
    if (st->d) HIP_TRY(ctx, hipFree(st->d));
    st->d = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&st->d, sizeof(double) * (need + need / 8)));
    st->cap = need + need / 8;
  }
  st->Na = Na; st->S = S; st->N = N; st->D = D; st->per_sample = per_sample_xa ? 1 : 0;
  st->has_lnw = lnw_SxNa ? 1 : 0;
  st->o_C = n_xa; st->o_f = n_xa + n_C; st->o_w = st->o_f + n_f;
  HIP_TRY(ctx, hipMemcpyAsync(st->d, Xa, sizeof(double) * n_xa, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(st->d + st->o_C, Ctmp_SxNxNa, sizeof(double) * n_C, hipMemcpyHostToDevice, ctx->stream));
  // f_s2 at Xa arrives (Na, S) as the reference stores it; keep it [S][Na]
  std::vector<double> ft(n_f);
  for (int64_t a = 0; a < Na; ++a)
    for (int s = 0; s < S; ++s) ft[(size_t)s * Na + a] = fs2a_NaxS[(size_t)a * S + s];
  HIP_TRY(ctx, hipMemcpyAsync(st->d + st->o_f, ft.data(), sizeof(double) * n_f, hipMemcpyHostToDevice, ctx->stream));
  if (lnw_SxNa)
    HIP_TRY(ctx, hipMemcpyAsync(st->d + st->o_w, lnw_SxNa, sizeof(double) * n_f, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, stream_wait(ctx));
  return VBMC_OK;
}

extern "C" int vbmc_acq_is_eval(vbmc_ctx* ctx, int64_t M, const double* xs_MxD, const double* sn2_M,
                                double u, double* acq_M, double* var_tot_M) {
  if (!ctx || (M > 0 && (!xs_MxD || !sn2_M || !acq_M))) return VBMC_E_ARG;
  NEED_DEVICE(ctx);
  if (!ctx->gp.set) return vbmc_fail(ctx, VBMC_E_ARG, "acq_is_eval: GP not set");
  IsState* st = (IsState*)ctx->acq_is;
  const GpState& g = ctx->gp;
  if (!st || !st->d || st->S != g.S || st->N != g.N || st->D != g.D)
    return vbmc_fail(ctx, VBMC_E_ARG, "acq_is_eval: importance state not set for this GP (vbmc_acq_is_set)");
  if (M == 0) return VBMC_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int N = g.N, D = g.D, S = g.S;
  const int64_t Na = st->Na;
  if (D > 32) return vbmc_fail(ctx, VBMC_E_UNSUP, "acq_is_eval: D=%d > 32 not supported", D);
  const int ntiles = (N + 63) / 64;
  int64_t mb = ((int64_t)1 << 26) / ((int64_t)S * N + N + Na);
  mb = mb > 16384 ? 16384 : (mb < 64 ? 64 : (mb / 64) * 64);
  if (M < mb) mb = M;
  // scratch: xs | Ks [S] | part [S] | fmu [S] | fs2 [S] | sn2 | Kx (mb x N) | T (mb x Na) | acq_s [S] | acq
  const size_t ks_n = predict_ks_elems(S, mb, N);
  const size_t need = align32((size_t)mb * D) + ks_n + 2 * (size_t)S * ntiles * mb + 2 * (size_t)S * mb +
                      (size_t)mb + (size_t)mb * N + (size_t)mb * Na + (size_t)S * mb + 2 * (size_t)mb;
  int rc = ensure_dev(ctx, &ctx->d_scratch, &ctx->d_scratch_cap, need);
  if (rc) return rc;
  rc = ensure_pinned(ctx, 2 * (size_t)mb);
  if (rc) return rc;
  double* d_xs = ctx->d_scratch;
  double* d_Ks = d_xs + align32((size_t)mb * D);  // 256-byte aligned: read by 16-byte LDS-direct loads
  double* d_part = d_Ks + ks_n;
  double* d_fmu = d_part + 2 * (size_t)S * ntiles * mb;
  double* d_fs2 = d_fmu + (size_t)S * mb;
  double* d_sn2 = d_fs2 + (size_t)S * mb;
  double* d_Kx = d_sn2 + mb;
  double* d_T = d_Kx + (size_t)mb * N;
  double* d_as = d_T + (size_t)mb * Na;
  double* d_acq = d_as + (size_t)S * mb;
  for (int64_t o = 0; o < M; o += mb) {
    const int64_t m = (M - o) < mb ? (M - o) : mb;
    HIP_TRY(ctx, hipMemcpyAsync(d_xs, xs_MxD + o * D, sizeof(double) * m * D, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d_sn2, sn2_M + o, sizeof(double) * m, hipMemcpyHostToDevice, ctx->stream));
    rc = launch_gp_predict_all(ctx, m, d_xs, d_Ks, d_part, 0, d_fmu, d_fs2, mb);  // f_s2 at Xs, every sample
    if (rc) return rc;
    for (int s = 0; s < S; ++s) {
      const double* hyp = g.d_hyp + (size_t)s * g.P;
      hipLaunchKernelGGL(se_cross_kernel, dim3((unsigned)((m * N + 255) / 256)), dim3(256), 0, ctx->stream,
                         (const double*)d_xs, m, (const double*)g.d_X, N, D, hyp, d_Kx);
      hipLaunchKernelGGL(gemm_nn_mfma_kernel, dim3((unsigned)((Na + GT - 1) / GT), (unsigned)((m + GT - 1) / GT)),
                         dim3(256), 0, ctx->stream, (const double*)d_Kx,
                         (const double*)(st->d + st->o_C + (size_t)s * N * Na), d_T, m, N, (int)Na);
      const double* Xa = st->d + (st->per_sample ? (size_t)s * Na * D : 0);
      hipLaunchKernelGGL(is_reduce_kernel, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, ctx->stream,
                         (const double*)d_xs, m, D, Xa, Na, hyp, (const double*)d_T,
                         (const double*)(d_fs2 + (size_t)s * mb), (const double*)d_sn2,
                         (const double*)(st->d + st->o_f + (size_t)s * Na),
                         st->has_lnw ? (const double*)(st->d + st->o_w + (size_t)s * Na) : (const double*)nullptr,
                         g.L_chol[s] ? 1 : 0, u, d_as + (size_t)s * mb);
    }
    hipLaunchKernelGGL(is_combine_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const double*)d_as, S, m, mb, d_acq, (const double*)d_fmu, (const double*)d_fs2,
                       d_acq + mb);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_pinned, d_acq, sizeof(double) * m, hipMemcpyDeviceToHost, ctx->stream));
    if (var_tot_M)
      HIP_TRY(ctx, hipMemcpyAsync(ctx->h_pinned + mb, d_acq + mb, sizeof(double) * m, hipMemcpyDeviceToHost,
                                  ctx->stream));
    HIP_TRY(ctx, stream_wait(ctx));
    memcpy(acq_M + o, ctx->h_pinned, sizeof(double) * m);
    if (var_tot_M) memcpy(var_tot_M + o, ctx->h_pinned + mb, sizeof(double) * m);
  }
  return VBMC_OK;
}
