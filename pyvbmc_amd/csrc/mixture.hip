// Mixture density on a batch of points: VariationalPosterior.pdf / log_pdf in the
// transformed space (reference: variational_posterior/variational_posterior.py:441-541).
//
// One thread = one point (its D coordinates, lambda-scaled, live in registers); the
// K x D scaled means and per-component constants are wave-uniform reads.  Linear-
// domain accumulation over components then log with 0 -> -inf, exactly the
// reference's order of operations (:451-463, :531-541).
#include <cmath>
#include <cstring>

#include "common.h"
#include "fastmath.h"

namespace {

struct PdfArgs {
  const double* mix;
  MixLayout ml;
  const double* x;  // n x D
  int64_t n;
  int log_flag, grad_flag;
  int mode;     // 0 gaussian, 1 multivariate t (df>0), 2 product of univariate t (df<0)
  double df;    // |df|
  double nf;    // tail-specific normalisation replacing nconst (t variants)
  double* y;    // n
  double* dy;   // n x D or null
};

template <int DP, int MODE, bool GRAD>
__global__ __launch_bounds__(256) void mixture_pdf_kernel(PdfArgs a) {
  const int D = a.ml.D, K = a.ml.K;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const double* mup = a.mix + a.ml.o_mup;
  const double* is2 = a.mix + a.ml.o_is2;
  const double* wc = a.mix + a.ml.o_wc;
  const double* rc = a.mix + a.ml.o_rc;
  const double* w = a.mix + a.ml.o_w;
  const double* sig = a.mix + a.ml.o_sig;
  const double* ilam = a.mix + a.ml.o_ilam;

  double xs[DP], g[DP];
#pragma unroll
  for (int d = 0; d < DP; ++d) {
    xs[d] = (d < D) ? a.x[i * D + d] * ilam[d] : 0.0;
    g[d] = 0.0;
  }
  double y = 0.0;
  for (int k = 0; k < K; ++k) {
    const double* mk = mup + k * D;
    double nn;
    if (MODE == 0) {
      double d2 = 0.0;
#pragma unroll
      for (int d = 0; d < DP; ++d)
        if (d < D) {
          const double u = xs[d] - mk[d];
          d2 = fma(u, u, d2);
        }
      // exp(-d2 / (2 sigma_k^2)) as exp2 with log2(e) folded into the scale (fastmath.h, <= 1 ulp)
      nn = wc[k] * fm::exp2_fast((-0.5 * 0x1.71547652b82fep+0 * is2[k]) * d2);
      if (GRAD) {
        const double c = nn * is2[k];
#pragma unroll
        for (int d = 0; d < DP; ++d)
          if (d < D) g[d] = fma(c, xs[d] - mk[d], g[d]);
      }
    } else if (MODE == 1) {
      double d2 = 0.0;
#pragma unroll
      for (int d = 0; d < DP; ++d)
        if (d < D) {
          const double u = xs[d] - mk[d];
          d2 = fma(u, u, d2);
        }
      // nf w_k / sigma_k^D (1 + d2/df)^(-(df+D)/2)   (:484-496)
      nn = a.nf * w[k] * pow(sig[k], -(double)D) * pow(1.0 + d2 * is2[k] / a.df, -0.5 * (a.df + D));
    } else {
      // prod_d (1 + z_d^2/|df|)^(-(|df|+1)/2)   (:513-524)
      double lg = 0.0;
#pragma unroll
      for (int d = 0; d < DP; ++d)
        if (d < D) {
          const double u = xs[d] - mk[d];
          lg += log1p(u * u * is2[k] / a.df);
        }
      nn = a.nf * w[k] * pow(sig[k], -(double)D) * exp(-0.5 * (a.df + 1.0) * lg);
    }
    y += nn;
  }
  (void)rc;
  if (GRAD) {
    // dy = -sum_k nn (x - mu_k)/(lambda^2 sigma_k^2); log: dy / y taken before the log (:464-469,532)
    const double s = a.log_flag ? -1.0 / y : -1.0;
#pragma unroll
    for (int d = 0; d < DP; ++d)
      if (d < D) a.dy[i * D + d] = s * g[d] * ilam[d];
  }
  if (a.log_flag) y = (y == 0.0) ? -INFINITY : log(y);
  a.y[i] = y;
}

// Small batches (acquisition populations, single points): one WAVE per point, lanes over the
// components, so the K terms are evaluated side by side and summed by a shuffle tree instead of
// one thread walking K dependent load -> exp steps (24 us for one point at K = 50, against 5).
// The sum over components is a tree here and a running sum in the kernel above: results agree to
// rounding (~1e-16 relative), not bit for bit.
template <int DP, bool GRAD>
__global__ __launch_bounds__(256) void mixture_pdf_wave_kernel(PdfArgs a) {
  const int D = a.ml.D, K = a.ml.K;
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= a.n) return;
  const double* mup = a.mix + a.ml.o_mup;
  const double* is2 = a.mix + a.ml.o_is2;
  const double* wc = a.mix + a.ml.o_wc;
  const double* ilam = a.mix + a.ml.o_ilam;
  double xs[DP], g[DP];
#pragma unroll
  for (int d = 0; d < DP; ++d) {
    xs[d] = (d < D) ? a.x[i * D + d] * ilam[d] : 0.0;
    g[d] = 0.0;
  }
  double y = 0.0;
  for (int k = lane; k < K; k += 64) {
    const double* mk = mup + k * D;
    const double s2 = is2[k];
    double d2 = 0.0;
#pragma unroll
    for (int d = 0; d < DP; ++d)
      if (d < D) {
        const double u = xs[d] - mk[d];
        d2 = fma(u, u, d2);
      }
    const double nn = wc[k] * fm::exp2_fast((-0.5 * 0x1.71547652b82fep+0 * s2) * d2);
    y += nn;
    if (GRAD) {
      const double c = nn * s2;
#pragma unroll
      for (int d = 0; d < DP; ++d)
        if (d < D) g[d] = fma(c, xs[d] - mk[d], g[d]);
    }
  }
  y = fm::wave_sum_dpp(y);
  if (GRAD) {
    const double s = a.log_flag ? -1.0 / y : -1.0;
#pragma unroll
    for (int d = 0; d < DP; ++d)
      if (d < D) {
        const double gd = fm::wave_sum_dpp(g[d]);
        if (lane == 0) a.dy[i * D + d] = s * gd * ilam[d];
      }
  }
  if (lane == 0) a.y[i] = a.log_flag ? ((y == 0.0) ? -INFINITY : log(y)) : y;
}

// below this many points the thread-per-point kernel cannot fill the GPU (256 CUs x 2048 lanes)
constexpr int64_t kWavePerPointMax = 1 << 16;

template <int DP>
void launch_dp(vbmc_ctx* ctx, const PdfArgs& a) {
  if (a.mode == 0 && a.n <= kWavePerPointMax) {
    const dim3 grid((unsigned)((a.n + 3) / 4)), block(256);
    if (a.grad_flag) hipLaunchKernelGGL((mixture_pdf_wave_kernel<DP, true>), grid, block, 0, ctx->stream, a);
    else hipLaunchKernelGGL((mixture_pdf_wave_kernel<DP, false>), grid, block, 0, ctx->stream, a);
    return;
  }
  const dim3 grid((unsigned)((a.n + 255) / 256)), block(256);
  if (a.mode == 0) {
    if (a.grad_flag) hipLaunchKernelGGL((mixture_pdf_kernel<DP, 0, true>), grid, block, 0, ctx->stream, a);
    else hipLaunchKernelGGL((mixture_pdf_kernel<DP, 0, false>), grid, block, 0, ctx->stream, a);
  } else if (a.mode == 1) {
    hipLaunchKernelGGL((mixture_pdf_kernel<DP, 1, false>), grid, block, 0, ctx->stream, a);
  } else {
    hipLaunchKernelGGL((mixture_pdf_kernel<DP, 2, false>), grid, block, 0, ctx->stream, a);
  }
}

}  // namespace

int launch_mixture_pdf(vbmc_ctx* ctx, int64_t n, const double* d_x, int log_flag, int grad_flag,
                       double df, double* d_y, double* d_dy) {
  const int D = ctx->D;
  if (D > 32) return vbmc_fail(ctx, VBMC_E_UNSUP, "mixture_pdf: D=%d > 32 not supported", D);
  PdfArgs a;
  a.mix = ctx->d_mix;
  a.ml = ctx->ml;
  a.x = d_x;
  a.n = n;
  a.log_flag = log_flag;
  a.grad_flag = grad_flag;
  a.y = d_y;
  a.dy = d_dy;
  a.df = std::fabs(df);
  a.nf = 0.0;
  double prod_lam = 1.0;
  for (int d = 0; d < D; ++d) prod_lam *= ctx->lambd[d];
  if (!std::isfinite(df) || df == 0.0) {
    a.mode = 0;
  } else if (df > 0.0) {
    a.mode = 1;
    a.nf = std::exp(std::lgamma(0.5 * (df + D)) - std::lgamma(0.5 * df)) /
           std::pow(df * M_PI, 0.5 * D) / prod_lam;
  } else {
    a.mode = 2;
    const double ad = -df;
    a.nf = std::pow(std::exp(std::lgamma(0.5 * (ad + 1.0)) - std::lgamma(0.5 * ad)) /
                        std::sqrt(ad * M_PI),
                    (double)D) /
           prod_lam;
  }
  if (ctx->timing) HIP_TRY(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
  if (D <= 2) launch_dp<2>(ctx, a);
  else if (D <= 4) launch_dp<4>(ctx, a);
  else if (D <= 6) launch_dp<6>(ctx, a);
  else if (D <= 8) launch_dp<8>(ctx, a);
  else if (D <= 10) launch_dp<10>(ctx, a);
  else if (D <= 12) launch_dp<12>(ctx, a);
  else if (D <= 16) launch_dp<16>(ctx, a);
  else if (D <= 20) launch_dp<20>(ctx, a);
  else if (D <= 24) launch_dp<24>(ctx, a);
  else launch_dp<32>(ctx, a);
  if (ctx->timing) {
    HIP_TRY(ctx, hipEventRecord(ctx->ev[5], ctx->stream));
    ctx->ev_valid[2] = true;
  }
  HIP_TRY(ctx, hipGetLastError());
  return 0;
}

// Gaussian-mode density of an arbitrary mixture pack (not necessarily the ctx mixture) at
// device points: used by the Monte-Carlo KL divergence, which needs two mixtures at once.
int launch_mixture_pdf_on(vbmc_ctx* ctx, const double* d_pack, const MixLayout& ml, int64_t n,
                          const double* d_x, int log_flag, double* d_y) {
  const int D = ml.D;
  if (D > 32) return vbmc_fail(ctx, VBMC_E_UNSUP, "mixture_pdf: D=%d > 32 not supported", D);
  PdfArgs a;
  a.mix = d_pack;
  a.ml = ml;
  a.x = d_x;
  a.n = n;
  a.log_flag = log_flag;
  a.grad_flag = 0;
  a.y = d_y;
  a.dy = nullptr;
  a.df = 0.0;
  a.nf = 0.0;
  a.mode = 0;
  if (D <= 2) launch_dp<2>(ctx, a);
  else if (D <= 4) launch_dp<4>(ctx, a);
  else if (D <= 6) launch_dp<6>(ctx, a);
  else if (D <= 8) launch_dp<8>(ctx, a);
  else if (D <= 10) launch_dp<10>(ctx, a);
  else if (D <= 12) launch_dp<12>(ctx, a);
  else if (D <= 16) launch_dp<16>(ctx, a);
  else if (D <= 20) launch_dp<20>(ctx, a);
  else if (D <= 24) launch_dp<24>(ctx, a);
  else launch_dp<32>(ctx, a);
  HIP_TRY(ctx, hipGetLastError());
  return 0;
}

extern "C" int vbmc_mixture_pdf(vbmc_ctx* ctx, int64_t n, const double* x_nxD, int log_flag,
                                int grad_flag, double df, double* y_n, double* dy_nxD) {
  if (!ctx || (n > 0 && (!x_nxD || !y_n))) return VBMC_E_ARG;
  if (!ctx->mix_set) return vbmc_fail(ctx, VBMC_E_ARG, "mixture_pdf: mixture not set");
  if (grad_flag && !dy_nxD) return vbmc_fail(ctx, VBMC_E_ARG, "mixture_pdf: grad_flag without dy");
  if (grad_flag && std::isfinite(df) && df != 0.0)
    return vbmc_fail(ctx, VBMC_E_UNSUP, "Gradient of heavy-tailed pdf not supported yet.");
  if (n == 0) return VBMC_OK;
  NEED_DEVICE(ctx);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int D = ctx->D;
  // batch so that scratch stays bounded (x, y, dy)
  const int64_t BATCH = 1 << 22;
  const size_t per = (size_t)D + 1 + (grad_flag ? D : 0);
  const int64_t nb = n < BATCH ? n : BATCH;
  int rc = ensure_dev(ctx, &ctx->d_scratch, &ctx->d_scratch_cap, per * (size_t)nb);
  if (rc) return rc;
  double* d_x = ctx->d_scratch;
  double* d_y = d_x + (size_t)nb * D;
  double* d_dy = d_y + nb;
  for (int64_t o = 0; o < n; o += nb) {
    const int64_t m = (n - o) < nb ? (n - o) : nb;
    HIP_TRY(ctx, hipMemcpyAsync(d_x, x_nxD + o * D, sizeof(double) * m * D, hipMemcpyHostToDevice,
                                ctx->stream));
    rc = launch_mixture_pdf(ctx, m, d_x, log_flag, grad_flag, df, d_y, grad_flag ? d_dy : nullptr);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(y_n + o, d_y, sizeof(double) * m, hipMemcpyDeviceToHost, ctx->stream));
    if (grad_flag)
      HIP_TRY(ctx, hipMemcpyAsync(dy_nxD + o * D, d_dy, sizeof(double) * m * D,
                                  hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, stream_wait(ctx));
  }
  return VBMC_OK;
}
