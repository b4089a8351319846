// sampling from the variational posterior and the Monte-Carlo consumers
// that only need the samples on the device.
//
//   VariationalPosterior.sample (variational_posterior/variational_posterior.py:241-363),
//   transformed space, Gaussian components:  x = mu_i + (lambda o z) sigma_i  (:321-327)
//   VariationalPosterior.kl_div, gauss_flag=False branch (:1107-1123)
//
// The reference draws from NumPy's global MT19937 stream; here the draws come from the
// counter-based generator of philox.h (no state, any sample can be regenerated from its
// index), with counter word 3 separating the streams:
//     c = (n_lo, n_hi, pair, 2)        normals of dimensions 2 pair, 2 pair + 1 of sample n
//     c = (n_lo, n_hi, 0,    3)        the uniform that picks the component of sample n
//     c = (n_lo, n_hi, r,    4)        round r of the gamma variate of sample n (Student-t tails)
// oracle/sample_ref.py restates both.  Component choice: inverse CDF of the weights
// (np.random.choice(p=w), :316-319); with balance_flag the first sum_k floor(w_k N) samples
// are split exactly according to the weights and the remaining ones are drawn from the
// reference's corrected remainder weights (:296-313).  The reference then shuffles the
// component labels; the device keeps the samples grouped by component (callers that need a
// random order permute on the host -- the Monte-Carlo consumers below do not care).
#include <cmath>
#include <cstring>

#include "common.h"
#include "fastmath.h"
#include "philox.h"

namespace {

struct SampleArgs {
  const double* mix;
  MixLayout ml;
  int64_t N;
  int64_t n_exact;        // balance: samples [0, n_exact) are assigned by count
  const int64_t* cum_cnt; // balance: [K+1] cumulative exact counts
  const double* cdf;      // [K] cumulative selection probabilities (plain weights, or the
                          //     remainder weights under balance)
  uint64_t seed;
  double* x;              // N x D or null
  int32_t* comp;          // N or null
  double df;              // degrees of freedom of the multivariate-t tails; <= 0 or inf: Gaussian
};

__device__ __forceinline__ double philox_uniform(uint64_t n, uint32_t c3, uint64_t seed) {
  Philox4 r = philox4x32_10((uint32_t)n, (uint32_t)(n >> 32), 0u, c3, (uint32_t)seed, (uint32_t)(seed >> 32));
  const uint64_t a = (((uint64_t)r.x[0] << 32) | r.x[1]) >> 11;
  return (double)a * 0x1.0p-53;  // [0, 1)
}

// Gamma(shape, 1) variate of sample n by Marsaglia & Tsang's method (ACM TOMS 26, 2000): with
// d = shape' - 1/3, c = 1/sqrt(9 d): z ~ N(0,1), v = (1 + c z)^3, accept when v > 0 and
// ln U < z^2/2 + d - d v + d ln v; shape < 1 goes through shape' = shape + 1 and a factor U'^(1/shape).
// Round r takes its normal from Philox(n, 2r, 4) and its uniforms U, U' from Philox(n, 2r+1, 4):
// a pure function of (seed, n), restated by oracle/sample_ref.py.  64 rounds at >= 95 %
// acceptance each: the fall-through (value d) has probability < 1e-80.
__device__ inline double philox_gamma(uint64_t n, double shape, uint64_t seed) {
  const bool small = shape < 1.0;
  const double sh = small ? shape + 1.0 : shape;
  const double d = sh - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
  double g = d, up = 1.0;
  for (uint32_t r = 0; r < 64; ++r) {
    Philox4 q = philox4x32_10((uint32_t)n, (uint32_t)(n >> 32), 2 * r, 4u, (uint32_t)seed, (uint32_t)(seed >> 32));
    const uint64_t ua = (((uint64_t)q.x[0] << 32) | q.x[1]) >> 11, ub = (((uint64_t)q.x[2] << 32) | q.x[3]) >> 11;
    const double u1 = (double)(ua + 1) * 0x1.0p-53, u2 = (double)ub * 0x1.0p-53;
    double sn, cs;
    fm::sincospi_fast(2.0 * u2, sn, cs);
    const double z = sqrt(-2.0 * fm::log_fast(u1)) * cs;
    const double t = 1.0 + c * z;
    if (t <= 0.0) continue;
    const double v = t * t * t;
    q = philox4x32_10((uint32_t)n, (uint32_t)(n >> 32), 2 * r + 1, 4u, (uint32_t)seed, (uint32_t)(seed >> 32));
    const uint64_t uc = (((uint64_t)q.x[0] << 32) | q.x[1]) >> 11, ud = (((uint64_t)q.x[2] << 32) | q.x[3]) >> 11;
    const double U = (double)(uc + 1) * 0x1.0p-53;
    if (fm::log_fast(U) < 0.5 * z * z + d - d * v + d * fm::log_fast(v)) {
      g = d * v;
      up = (double)(ud + 1) * 0x1.0p-53;
      break;
    }
  }
  if (small) g *= exp(fm::log_fast(up) / shape);
  return g;
}

__global__ __launch_bounds__(256) void mixture_sample_kernel(SampleArgs a) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= a.N) return;
  const int D = a.ml.D, K = a.ml.K;
  int k = 0;
  if (K > 1) {
    if (n < a.n_exact) {
      while (k + 1 < K && n >= a.cum_cnt[k + 1]) ++k;
    } else {
      const double u = philox_uniform((uint64_t)n, 3u, a.seed);
      while (k + 1 < K && u >= a.cdf[k]) ++k;
    }
  }
  if (a.comp) a.comp[n] = k;
  if (!a.x) return;
  const double* mu = a.mix + a.ml.o_mu + (size_t)k * D;
  const double* lam = a.mix + a.ml.o_lam;
  const double sg = a.mix[a.ml.o_sig + k];
  // multivariate-t tails (:329-340): t = df/2 / sqrt(G), G ~ Gamma(df/2, scale df/2)
  const bool heavy = a.df > 0.0 && a.df < INFINITY;
  double tf = 1.0;
  if (heavy) {
    const double G = philox_gamma((uint64_t)n, 0.5 * a.df, a.seed) * (0.5 * a.df);
    tf = 0.5 * a.df / sqrt(G);
  }
  for (int p = 0; 2 * p < D; ++p) {
    double z0, z1;
    Philox4 r = philox4x32_10((uint32_t)n, (uint32_t)((uint64_t)n >> 32), (uint32_t)p, 2u,
                              (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
    const uint64_t ua = (((uint64_t)r.x[0] << 32) | r.x[1]) >> 11;
    const uint64_t ub = (((uint64_t)r.x[2] << 32) | r.x[3]) >> 11;
    const double u1 = (double)(ua + 1) * 0x1.0p-53, u2 = (double)ub * 0x1.0p-53;
    const double rad = sqrt(-2.0 * fm::log_fast(u1));
    double s, c;
    fm::sincospi_fast(2.0 * u2, s, c);
    z0 = rad * c;
    z1 = rad * s;
    const int d0 = 2 * p, d1 = 2 * p + 1;
    if (!heavy) {
      a.x[n * D + d0] = mu[d0] + (lam[d0] * z0) * sg;  // the reference's association (:323-326)
      if (d1 < D) a.x[n * D + d1] = mu[d1] + (lam[d1] * z1) * sg;
    } else if (K > 1) {  // mu + lam * z * t * sigma (:332-336)
      a.x[n * D + d0] = mu[d0] + ((lam[d0] * z0) * tf) * sg;
      if (d1 < D) a.x[n * D + d1] = mu[d1] + ((lam[d1] * z1) * tf) * sg;
    } else {  // mu + lam * t * z * sigma (:349-353)
      a.x[n * D + d0] = mu[d0] + ((lam[d0] * tf) * z0) * sg;
      if (d1 < D) a.x[n * D + d1] = mu[d1] + ((lam[d1] * tf) * z1) * sg;
    }
  }
}

// per-block partial sums of -(log qb - log qa) with the reference's replacement rules
// (:1113-1115 / :1120-1122): the density of the sampling mixture -> 1 where it is 0, the
// other one -> realmin where it is 0.  ya/yb are linear-domain densities.
__global__ __launch_bounds__(256) void kl_terms_kernel(const double* __restrict__ y_own,
                                                       const double* __restrict__ y_other, int64_t n,
                                                       double* __restrict__ part) {
  __shared__ double red[4];
  const int tid = threadIdx.x;
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < n; i += (int64_t)gridDim.x * 256) {
    double qo = y_own[i], qx = y_other[i];
    if (qo == 0.0) qo = 1.0;
    if (qx == 0.0) qx = 2.2250738585072014e-308;
    acc += log(qx) - log(qo);
  }
  acc = fm::wave_sum_dpp(acc);
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// host: counts / cdf of one mixture's component selection
struct Selector {
  int64_t n_exact = 0;
  std::vector<int64_t> cum;  // K+1
  std::vector<double> cdf;   // K
};

void make_selector(const double* w, int K, int64_t N, int balance, Selector& s) {
  s.cum.assign((size_t)K + 1, 0);
  s.cdf.assign((size_t)K, 1.0);
  std::vector<double> p(w, w + K);
  s.n_exact = 0;
  if (balance && K > 1) {
    // repeats = floor(w N); w_extra = w N - repeats; repeats_extra = ceil(sum w_extra);
    // w_extra += w (repeats_extra - sum w_extra); w_extra /= sum w_extra   (:298-306)
    double sum_extra = 0.0;
    std::vector<double> extra(K);
    for (int k = 0; k < K; ++k) {
      const double wn = w[k] * (double)N;
      const double rep = std::floor(wn);
      s.cum[k + 1] = s.cum[k] + (int64_t)rep;
      extra[k] = wn - rep;
      sum_extra += extra[k];
    }
    s.n_exact = s.cum[K] < N ? s.cum[K] : N;
    const double rep_extra = std::ceil(sum_extra);
    double tot = 0.0;
    for (int k = 0; k < K; ++k) {
      extra[k] += w[k] * (rep_extra - sum_extra);
      tot += extra[k];
    }
    for (int k = 0; k < K; ++k) p[k] = tot > 0.0 ? extra[k] / tot : w[k];
  }
  double c = 0.0;
  for (int k = 0; k < K; ++k) {
    c += p[k];
    s.cdf[k] = c;
  }
  s.cdf[K - 1] = 2.0;  // the last component catches rounding in the cumulative sum
}

// Draw N samples of `d_pack` into d_x (device), optionally labels into d_comp.
// d_sel: device scratch of (K+1) int64 + K doubles.
int launch_sample(vbmc_ctx* ctx, const double* d_pack, const MixLayout& ml, const double* w_host,
                  int64_t N, uint64_t seed, int balance, void* d_sel, double* d_x, int32_t* d_comp,
                  double df = INFINITY) {
  const int K = ml.K;
  Selector s;
  make_selector(w_host, K, N, balance, s);
  int64_t* d_cum = (int64_t*)d_sel;
  double* d_cdf = (double*)(d_cum + K + 1);
  HIP_TRY(ctx, hipMemcpyAsync(d_cum, s.cum.data(), sizeof(int64_t) * (K + 1), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_cdf, s.cdf.data(), sizeof(double) * K, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, stream_wait(ctx));  // the selector vectors are locals
  SampleArgs a;
  a.mix = d_pack;
  a.ml = ml;
  a.N = N;
  a.n_exact = s.n_exact;
  a.cum_cnt = d_cum;
  a.cdf = d_cdf;
  a.seed = seed;
  a.x = d_x;
  a.comp = d_comp;
  a.df = df;
  hipLaunchKernelGGL(mixture_sample_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, a);
  HIP_TRY(ctx, hipGetLastError());
  return 0;
}

}  // namespace

extern "C" int vbmc_mixture_sample(vbmc_ctx* ctx, int64_t N, uint64_t seed, int balance_flag,
                                   double* x_NxD, int32_t* comp_N) {
  return vbmc_mixture_sample_t(ctx, N, seed, balance_flag, INFINITY, x_NxD, comp_N);
}

extern "C" int vbmc_mixture_sample_t(vbmc_ctx* ctx, int64_t N, uint64_t seed, int balance_flag, double df,
                                     double* x_NxD, int32_t* comp_N) {
  if (!ctx || N < 0) return VBMC_E_ARG;
  if (df < 0.0 || df != df)
    return vbmc_fail(ctx, VBMC_E_ARG, "mixture_sample: df=%g (the reference's gamma draw needs df > 0)", df);
  if (!ctx->mix_set) return vbmc_fail(ctx, VBMC_E_ARG, "mixture_sample: mixture not set");
  if (N == 0) return VBMC_OK;
  NEED_DEVICE(ctx);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int D = ctx->D, K = ctx->K;
  const int64_t BATCH = (int64_t)1 << 22;
  // one launch covers all N indices so that sample n does not depend on the batching;
  // x is produced in full and copied back in one piece (N x D doubles of scratch)
  const size_t n_x = x_NxD ? (size_t)N * D : 0;
  const size_t n_c = comp_N ? ((size_t)N + 1) / 2 : 0;  // int32 labels, in doubles
  const size_t n_sel = (size_t)2 * K + 2;
  (void)BATCH;
  int rc = ensure_dev(ctx, &ctx->d_scratch, &ctx->d_scratch_cap, n_x + n_c + n_sel);
  if (rc) return rc;
  double* d_x = x_NxD ? ctx->d_scratch : nullptr;
  int32_t* d_c = comp_N ? (int32_t*)(ctx->d_scratch + n_x) : nullptr;
  void* d_sel = (void*)(ctx->d_scratch + n_x + n_c);
  rc = launch_sample(ctx, ctx->d_mix, ctx->ml, ctx->w.data(), N, seed, balance_flag, d_sel, d_x, d_c, df);
  if (rc) return rc;
  if (x_NxD)
    HIP_TRY(ctx, hipMemcpyAsync(x_NxD, d_x, sizeof(double) * n_x, hipMemcpyDeviceToHost, ctx->stream));
  if (comp_N)
    HIP_TRY(ctx, hipMemcpyAsync(comp_N, d_c, sizeof(int32_t) * N, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, stream_wait(ctx));
  return VBMC_OK;
}

extern "C" int vbmc_kl_div_mc(vbmc_ctx* ctx, int64_t N, uint64_t seed, int K2, const double* mu2_KxD,
                              const double* sigma2_K, const double* lambd2_D, const double* w2_K,
                              double kl_out[2]) {
  if (!ctx || N < 1 || K2 < 1 || !mu2_KxD || !sigma2_K || !lambd2_D || !w2_K || !kl_out) return VBMC_E_ARG;
  if (!ctx->mix_set) return vbmc_fail(ctx, VBMC_E_ARG, "kl_div: mixture not set");
  NEED_DEVICE(ctx);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int D = ctx->D, K1 = ctx->K;
  for (int k = 0; k < K2; ++k)
    if (!(sigma2_K[k] > 0.0) || !std::isfinite(sigma2_K[k]))
      return vbmc_fail(ctx, VBMC_E_NONFINITE, "kl_div: sigma2[%d] must be finite and > 0", k);
  MixLayout ml2;
  ml2.plan(D, K2);
  std::vector<double> pack2((size_t)ml2.total);
  write_mixture_pack(ml2, mu2_KxD, sigma2_K, lambd2_D, w2_K, pack2.data());
  const int nblk = 512;
  const int Kmax = K1 > K2 ? K1 : K2;
  const size_t need = (size_t)ml2.total + (size_t)N * D + 2 * (size_t)N + (size_t)2 * Kmax + 2 + 2 * (size_t)nblk;
  int rc = ensure_dev(ctx, &ctx->d_scratch, &ctx->d_scratch_cap, need);
  if (rc) return rc;
  double* d_pack2 = ctx->d_scratch;
  double* d_x = d_pack2 + ml2.total;
  double* d_y1 = d_x + (size_t)N * D;
  double* d_y2 = d_y1 + N;
  void* d_sel = (void*)(d_y2 + N);
  double* d_part = (double*)d_sel + 2 * Kmax + 2;
  HIP_TRY(ctx, hipMemcpyAsync(d_pack2, pack2.data(), sizeof(double) * ml2.total, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, stream_wait(ctx));
  std::vector<double> part(2 * (size_t)nblk);
  for (int dir = 0; dir < 2; ++dir) {
    // dir 0: samples of this mixture (xx1, :1110); dir 1: samples of vp2 (xx2, :1117)
    const double* d_own = dir == 0 ? ctx->d_mix : d_pack2;
    const MixLayout& ml_own = dir == 0 ? ctx->ml : ml2;
    const double* w_own = dir == 0 ? ctx->w.data() : w2_K;
    rc = launch_sample(ctx, d_own, ml_own, w_own, N, seed + (uint64_t)dir, 1, d_sel, d_x, nullptr);
    if (rc) return rc;
    rc = launch_mixture_pdf_on(ctx, ctx->d_mix, ctx->ml, N, d_x, 0, d_y1);
    if (rc) return rc;
    rc = launch_mixture_pdf_on(ctx, d_pack2, ml2, N, d_x, 0, d_y2);
    if (rc) return rc;
    hipLaunchKernelGGL(kl_terms_kernel, dim3(nblk), dim3(256), 0, ctx->stream,
                       (const double*)(dir == 0 ? d_y1 : d_y2), (const double*)(dir == 0 ? d_y2 : d_y1), N,
                       d_part + (size_t)dir * nblk);
    HIP_TRY(ctx, hipGetLastError());
  }
  HIP_TRY(ctx, hipMemcpyAsync(part.data(), d_part, sizeof(double) * 2 * nblk, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, stream_wait(ctx));
  for (int dir = 0; dir < 2; ++dir) {
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += part[(size_t)dir * nblk + b];
    const double kl = -s / (double)N;       // -mean(log q_other - log q_own)
    kl_out[dir] = kl > 0.0 ? kl : 0.0;      // np.maximum(0, kls)  (:1126)
  }
  return VBMC_OK;
}
