// Monte-Carlo entropy of the Gaussian mixture (reference: entropy/entmc_vbmc.py:6-134)
// and the Jensen lower bound (entropy/entlb_vbmc.py:6-180), as HIP kernels for gfx950.
//
// Work decomposition of entmc: one workgroup = (component j, chunk of 256
// antithetic-pair rows).  A thread owns one row eps (D normals) and evaluates BOTH
// samples x+- = mu_j +- sigma_j lambda o eps, which share the dot product
// Delta_k . eps of the squared distance to every component k:
//     |x'+- - mu'_k|^2 = |Delta_k|^2 + sigma_j^2 |eps|^2 +- 2 sigma_j Delta_k . eps,
//     Delta_k = (mu_j - mu_k) / lambda    (per-j centred, so the expansion is benign).
// Per-workgroup partial sums go to HBM; two tiny kernels reduce them in a fixed
// order (bit-reproducible) into the raw accumulator vector
//     [H | mu (K blocks of D) | sigma (K) | lambda (D) | w (K)]
// that the host finalises (Jacobians) and that the multi-GPU path all-reduces.
#include <cstdlib>

#include <algorithm>

#include "common.h"
#include "fastmath.h"
#include "entropy_args.h"
#include "philox.h"
#ifdef FIN_TIMES
__device__ unsigned long long g_glj_stamp;
#define GLJ_STAMP() (g_glj_stamp = wall_clock64())
#endif
#include "glj_block.h"

namespace {

constexpr int WG = 256;
constexpr int WAVES = WG / 64;

__device__ inline double wave_sum(double v) {
  return fm::wave_sum_dpp(v);
}


// Partial row layout: [Slog | mu(D) | sig | lam(D) | W(K)]
template <int DP>
__global__ __launch_bounds__(WG) void entmc_valu_kernel(EntArgs a) {
  extern __shared__ double lds[];
  const int D = a.ml.D, K = a.ml.K;
  const int j = blockIdx.y;
  const int chunk = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;

  double* sDelta = lds;                 // [K][DP]
  double* sA = sDelta + K * DP;         // [K] |Delta_k|^2
  double* sIs2 = sA + K;                // [K]
  double* sWc = sIs2 + K;               // [K]
  double* sRc = sWc + K;                // [K]
  double* sRed = sRc + K;               // [WAVES][2*DP+1]
  double* sW = sRed + WAVES * (2 * DP + 1);  // [WAVES][K]

  const double* mup = a.mix + a.ml.o_mup;
  for (int idx = tid; idx < K * DP; idx += WG) {
    int k = idx / DP, d = idx - k * DP;
    sDelta[idx] = (d < D) ? (mup[j * D + d] - mup[k * D + d]) : 0.0;
  }
  for (int k = tid; k < K; k += WG) {
    sIs2[k] = a.mix[a.ml.o_is2 + k];
    sWc[k] = a.mix[a.ml.o_wc + k];
    sRc[k] = a.mix[a.ml.o_rc + k];
  }
  __syncthreads();
  for (int k = tid; k < K; k += WG) {
    double s = 0.0;
    for (int d = 0; d < DP; ++d) s = fma(sDelta[k * DP + d], sDelta[k * DP + d], s);
    sA[k] = s;
  }
  __syncthreads();

  const double sig_j = a.mix[a.ml.o_sig + j];
  const int64_t i_loc = (int64_t)chunk * WG + tid;  // row within this ctx's slice
  const bool valid = i_loc < a.row_count;

  double e[DP];
#pragma unroll
  for (int d = 0; d < DP; ++d) e[d] = 0.0;
  if (valid) {
    if (a.eps_mode == VBMC_EPS_RESIDENT) {
      const double* row = a.eps + ((int64_t)j * a.eps_rows + i_loc) * D;
#pragma unroll
      for (int d = 0; d < DP; ++d)
        if (d < D) e[d] = row[d];
    } else {
      const uint64_t grow = (uint64_t)j * (uint64_t)a.n_half + (uint64_t)(a.row_begin + i_loc);
#pragma unroll
      for (int b = 0; b < (DP + 3) / 4; ++b) {
        if (4 * b < D) {
          double z[4];
          philox_normal_quad(grow, (uint32_t)b, a.seed, z);
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (4 * b + q < DP && 4 * b + q < D) e[4 * b + q] = z[q];
        }
      }
    }
  }
  double e2 = 0.0;
#pragma unroll
  for (int d = 0; d < DP; ++d) e2 = fma(e[d], e[d], e2);
  const double b = sig_j * sig_j * e2;
  const double two_sj = 2.0 * sig_j;

  double qp = 0.0, qm = 0.0, gsp = 0.0, gsm = 0.0;
  double Ap[DP], Am[DP];
#pragma unroll
  for (int d = 0; d < DP; ++d) Ap[d] = Am[d] = 0.0;

  for (int k = 0; k < K; ++k) {
    const double* dk = sDelta + k * DP;
    double c = 0.0;
#pragma unroll
    for (int d = 0; d < DP; ++d) c = fma(dk[d], e[d], c);
    const double ab = sA[k] + b;
    const double is2 = sIs2[k];
    const double sp = fmax(fma(two_sj, c, ab), 0.0);
    const double sm = fmax(fma(-two_sj, c, ab), 0.0);
    const double wrp = sWc[k] * exp(-0.5 * sp * is2);
    const double wrm = sWc[k] * exp(-0.5 * sm * is2);
    qp += wrp;
    qm += wrm;
    if (a.want_grad) {
      const double gp = wrp * is2, gm = wrm * is2;
      gsp += gp;
      gsm += gm;
#pragma unroll
      for (int d = 0; d < DP; ++d) {
        Ap[d] = fma(gp, dk[d], Ap[d]);
        Am[d] = fma(gm, dk[d], Am[d]);
      }
    }
  }

  double slog = valid ? (log(qp) + log(qm)) : 0.0;
  const double ip = valid ? 1.0 / qp : 0.0;
  const double im = valid ? 1.0 / qm : 0.0;

  // ---- block reduction of [Slog, mu(DP), lam(DP)] ----
  {
    double v = wave_sum(slog);
    if (lane == 0) sRed[wave * (2 * DP + 1)] = v;
  }
  if (a.want_grad) {
#pragma unroll
    for (int d = 0; d < DP; ++d) {
      const double se = sig_j * e[d];
      const double lp = fma(se, gsp, Ap[d]) * ip;
      const double lm = fma(-se, gsm, Am[d]) * im;
      double vmu = wave_sum(lp + lm);
      double vlam = wave_sum((lp - lm) * e[d]);
      if (lane == 0) {
        sRed[wave * (2 * DP + 1) + 1 + d] = vmu;
        sRed[wave * (2 * DP + 1) + 1 + DP + d] = vlam;
      }
    }
    // ---- second pass: sum_n exp_k(x_n) / q(x_n) for every k ----
    for (int k = 0; k < K; ++k) {
      const double* dk = sDelta + k * DP;
      double c = 0.0;
#pragma unroll
      for (int d = 0; d < DP; ++d) c = fma(dk[d], e[d], c);
      const double ab = sA[k] + b;
      const double is2 = sIs2[k];
      const double sp = fmax(fma(two_sj, c, ab), 0.0);
      const double sm = fmax(fma(-two_sj, c, ab), 0.0);
      double v = exp(-0.5 * sp * is2) * ip + exp(-0.5 * sm * is2) * im;
      v = wave_sum(v);
      if (lane == 0) sW[wave * K + k] = v * sRc[k];  // norm_j1 / q of the reference
    }
  }
  __syncthreads();

  double* out = a.partial + ((int64_t)j * a.chunks + chunk) * a.stride;
  for (int t = tid; t < a.stride; t += WG) {
    double v = 0.0;
    if (t == 0) {
      for (int wv = 0; wv < WAVES; ++wv) v += sRed[wv * (2 * DP + 1)];
    } else if (a.want_grad) {
      if (t <= D) {
        for (int wv = 0; wv < WAVES; ++wv) v += sRed[wv * (2 * DP + 1) + t];
      } else if (t == D + 1) {
        for (int d = 0; d < D; ++d)
          for (int wv = 0; wv < WAVES; ++wv) v += sRed[wv * (2 * DP + 1) + 1 + DP + d];
      } else if (t < 2 * D + 2) {
        const int d = t - (D + 2);
        for (int wv = 0; wv < WAVES; ++wv) v += sRed[wv * (2 * DP + 1) + 1 + DP + d];
      } else {
        const int k = t - (2 * D + 2);
        for (int wv = 0; wv < WAVES; ++wv) v += sW[wv * K + k];
      }
    }
    out[t] = v;
  }
}

#ifdef FIN_TIMES
#define FIN_TIMES_HERE
#endif
#include "finish_body.h"
__global__ __launch_bounds__(256) void entmc_finish_kernel(const double* __restrict__ partial,
                                                           int chunks, int stride,
                                                           const double* __restrict__ mix,
                                                           MixLayout ml, double inv_ns,
                                                           int want_grad, int mu_from_w,
                                                           double* __restrict__ raw, GenSlice gen,
                                                           DoneSignal done) {
  entmc_finish_body(partial, chunks, stride, mix, ml, inv_ns, want_grad, mu_from_w, raw, gen, done);
}
#ifdef FIN_TIMES
}  // namespace
extern "C" int vbmc_debug_fin_times(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fin_times), sizeof(unsigned long long) * (4 + 3 * 64));
}
extern "C" int vbmc_debug_fin_phases(unsigned long long* out, int reset) {
  const int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fin_x), sizeof(unsigned long long) * 8);
  if (reset) {
    unsigned long long z[8] = {~0ull, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fin_x), z, sizeof(z));
  }
  return rc;
}
namespace {
#endif

// ---------------------------------------------------------------------------
// entlb (entropy/entlb_vbmc.py:80-159): workgroup j owns row j of the K x K table
//   gamma_ij = nconst / (s_i^2+s_j^2)^(D/2) exp(-1/2 |mu'_i-mu'_j|^2 / (s_i^2+s_j^2)).
// Every workgroup first forms all gsum_i = sum_i' w_i' gamma_ii' (K^2 cheap terms, one
// wave per row, fixed order), then the terms of its own j.  Raw output layout:
// [H | mu (K x D) | sigma (K) | lambda (D) | w (K)] (pre-Jacobian, sigma already carrying
// the reference's explicit sigma_j factor, entlb_vbmc.py:133); H and lambda are summed
// over j by entlb_finish_kernel from the per-j partials in `part` ([K] | [K][D]).
__device__ inline double entlb_gamma(const double* __restrict__ mup, const double* __restrict__ sig,
                                     int D, int i, int j, double lnc, double& s2_out,
                                     double& d2_out) {
  const double s2 = sig[i] * sig[i] + sig[j] * sig[j];
  double d2 = 0.0;
  for (int d = 0; d < D; ++d) {
    const double t = mup[i * D + d] - mup[j * D + d];
    d2 = fma(t, t, d2);
  }
  s2_out = s2;
  d2_out = d2;
  return exp(lnc - 0.5 * D * log(s2) - 0.5 * d2 / s2);
}

// stage A: workgroup i -> row i of gamma (stored) and gsum_i = sum_i' w_i' gamma_ii'
__global__ __launch_bounds__(256) void entlb_rows_kernel(const double* __restrict__ mix,
                                                         MixLayout ml, double* __restrict__ Gm,
                                                         double* __restrict__ gsum) {
  __shared__ double sRed[WAVES];
  const int D = ml.D, K = ml.K;
  const int i = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const double* mup = mix + ml.o_mup;
  const double* sig = mix + ml.o_sig;
  const double* w = mix + ml.o_w;
  const double* lam = mix + ml.o_lam;
  double lnc = -0.5 * D * log(2.0 * M_PI);
  for (int d = 0; d < D; ++d) lnc -= log(lam[d]);
  double acc = 0.0;
  for (int i2 = tid; i2 < K; i2 += 256) {
    double s2, d2;
    const double g = entlb_gamma(mup, sig, D, i, i2, lnc, s2, d2);
    Gm[(size_t)i * K + i2] = g;
    acc += w[i2] * g;
  }
  acc = wave_sum(acc);
  if (lane == 0) sRed[wave] = acc;
  __syncthreads();
  if (tid == 0) gsum[i] = (sRed[0] + sRed[1]) + (sRed[2] + sRed[3]);
}

// stage B: workgroup j -> gradient terms of component j (entlb_vbmc.py:99-159)
__global__ __launch_bounds__(256) void entlb_kernel(const double* __restrict__ mix, MixLayout ml,
                                                    int want_grad, const double* __restrict__ Gm,
                                                    const double* __restrict__ gsum,
                                                    double* __restrict__ res,
                                                    double* __restrict__ part) {
  extern __shared__ double lds[];
  const int D = ml.D, K = ml.K;
  const int j = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const double* mup = mix + ml.o_mup;  // mu / lambda
  const double* sig = mix + ml.o_sig;
  const double* w = mix + ml.o_w;
  const double* lam = mix + ml.o_lam;
  double* sRed = lds;  // [WAVES][2D+2]

  const double gj = gsum[j];
  if (tid == 0) part[j] = -w[j] * log(gj);
  if (!want_grad) return;
  const int NS = 2 * D + 2;
  double a_sig = 0.0, a_w = 0.0;
  for (int d = 0; d < D; ++d) {
    double a_mu = 0.0, a_lam = 0.0;
    for (int i = tid; i < K; i += 256) {
      const double s2 = sig[i] * sig[i] + sig[j] * sig[j];
      const double wg = w[i] * Gm[(size_t)j * K + i];  // gamma is symmetric
      const double gi = gsum[i];
      const double coef = wg * (1.0 / gi + 1.0 / gj);
      const double t = mup[i * D + d] - mup[j * D + d];  // (mu_i - mu_j)_d / lambda_d
      a_mu += coef * t / s2;
      a_lam += wg * (t * t / s2 - 1.0);
      if (d == 0) {
        double d2 = 0.0;
        for (int dd = 0; dd < D; ++dd) {
          const double u = mup[i * D + dd] - mup[j * D + dd];
          d2 = fma(u, u, d2);
        }
        a_sig += coef * (-(double)D / s2 + d2 / (s2 * s2));
        a_w += wg / gi;
      }
    }
    a_mu = wave_sum(a_mu);
    a_lam = wave_sum(a_lam);
    if (lane == 0) {
      sRed[wave * NS + d] = a_mu;
      sRed[wave * NS + D + d] = a_lam;
    }
  }
  a_sig = wave_sum(a_sig);
  a_w = wave_sum(a_w);
  if (lane == 0) {
    sRed[wave * NS + 2 * D] = a_sig;
    sRed[wave * NS + 2 * D + 1] = a_w;
  }
  __syncthreads();
  for (int it = tid; it < NS; it += 256) {
    const double v = (sRed[it] + sRed[NS + it]) + (sRed[2 * NS + it] + sRed[3 * NS + it]);
    if (it < D) {
      res[1 + j * D + it] = -w[j] * v / lam[it];
    } else if (it < 2 * D) {
      part[K + j * D + (it - D)] = (w[j] / gj) * v;
    } else if (it == 2 * D) {
      res[1 + K * D + j] = -w[j] * sig[j] * v;
    } else {
      res[1 + K * D + K + D + j] = -log(gj) - v;
    }
  }
}

// stage C: H and the lambda gradient are sums over j of the per-j partials (one wave each)
__global__ __launch_bounds__(256) void entlb_finish_kernel(const double* __restrict__ mix,
                                                           MixLayout ml, int want_grad,
                                                           const double* __restrict__ part,
                                                           double* __restrict__ res) {
  const int D = ml.D, K = ml.K;
  const double* lam = mix + ml.o_lam;
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (t > D || (t > 0 && !want_grad)) return;
  double s = 0.0;
  if (t == 0) {
    for (int j = lane; j < K; j += 64) s += part[j];
    s = wave_sum(s);
    if (lane == 0) res[0] = s;
  } else {
    const int d = t - 1;
    for (int j = lane; j < K; j += 64) s += part[K + j * D + d];
    s = wave_sum(s);
    if (lane == 0) res[1 + K * D + K + d] = -s / lam[d];
  }
}

template <int DP>
int launch_entmc_dp(vbmc_ctx* ctx, const EntArgs& a) {
  const int K = a.ml.K;
  size_t lds = sizeof(double) * ((size_t)K * DP + 4 * K + WAVES * (2 * DP + 1) + (size_t)WAVES * K);
  dim3 grid(a.chunks, K);
  hipLaunchKernelGGL(entmc_valu_kernel<DP>, grid, dim3(WG), lds, ctx->stream, a);
  return 0;
}

}  // namespace

static bool use_ws_kernel(const vbmc_ctx* ctx, int D, int K) {
  if (ctx->opt_entmc_valu) return false;  // vbmc_set_option("entmc_kernel", 1) / VBMC_ENTMC_KERNEL=valu
  return D <= 32 && K <= 128;
}

static int padded_d(int D) {
  const int dps[] = {2, 4, 6, 8, 10, 12, 16, 20, 24, 32};
  for (int dp : dps)
    if (D <= dp) return dp;
  return -1;
}

// Span mode: the front workgroup's share of a CU's batches, per mille (entropy_args.h WsSpan) -- the fraction of the
// issue slots one wave per SIMD of this instantiation uses when it runs alone.  Read off tools/ws_front_probe.py
// (kernel time over the share, every padded D x register-array size of the 2-waves/SIMD builds, value and value +
// gradient): padded D <= 12: 620-700 is the flat optimum everywhere (3-15 % under the equal chunks); padded D = 16 (the
// rows' loads weigh more against the arithmetic of a batch, the waves already alternate): 540-580, and with at most 8
// components per wave the equal chunks stay ahead by 1-4 % -- 0 = keep them.
static int ws_front_default(int DP, int KT) {
  if (DP <= 12) return 660;
  if (DP == 16) return KT <= 8 ? 0 : 570;
  return 660;  // (one wave per SIMD: no filler parts, the value is not used)
}

// Decide the launch geometry and carve the scratch buffer.
int entmc_plan(vbmc_ctx* ctx, int64_t ns_per_comp, int eps_mode, uint64_t seed, int64_t row_begin,
               int64_t row_count, int want_grad, EntPlan& p, int gp_items, bool allow_span, int gp_per_slot) {
  const int D = ctx->D, K = ctx->K;
  p.DP = padded_d(D);
  if (p.DP < 0) return vbmc_fail(ctx, VBMC_E_UNSUP, "entmc: D=%d > 32 not supported", D);
  EntArgs& a = p.a;
  a.mix = ctx->d_mix;
  a.ml = ctx->ml;
  a.eps = ctx->d_eps;
  a.eps_rows = ctx->eps_rows;
  a.n_half = ns_per_comp / 2;
  a.row_begin = row_begin;
  a.row_count = row_count;
  a.seed = seed;
  a.eps_mode = eps_mode;
  a.want_grad = want_grad;
  p.ws = use_ws_kernel(ctx, D, K);
  p.inv_ns = 1.0 / (double)ns_per_comp;
  int rows_per_wg = WG;
  a.rg = 1;
  size_t n_table = 0;
  if (p.ws) {
    // 64*rg rows per workgroup.  Two workgroups are resident per CU (2 waves/SIMD): size
    // the grid to about one full round of 2*CUs workgroups, which also amortises the
    // end-of-workgroup reductions over many batches.
    const int64_t total_rows = row_count * K;
    const int64_t cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
    int64_t rg = (total_rows + 64 * 2 * cus - 1) / (64 * 2 * cus);
    if (rg < 1) rg = 1;
    if (rg > 16) rg = 16;
    a.rg = (int)rg;
    rows_per_wg = 64 * a.rg;
    n_table = (size_t)K * (size_t)ws_table_rows(K) * (size_t)(p.DP + 6);
  }
  a.chunks = (int)((row_count + rows_per_wg - 1) / rows_per_wg);
  if (a.chunks < 1) a.chunks = 1;
  if (p.ws) {
    // chunks are per component: rounding them up can push K * chunks just past one round of resident
    // workgroups (K = 52, 10 000 rows: 10 chunks of 1 024 rows = 520 workgroups for 512 slots, a second
    // round of 8).  A few more batches per workgroup keep the grid in one round.
    const int64_t cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
    const int64_t slots = cus * ws_min_waves(p.DP, ws_ktmax_for(K), want_grad != 0);
    const int64_t fit = slots / K;  // chunks per component that fit one round
    if (fit >= 1 && a.chunks > fit && (int64_t)K * (a.chunks - 1) <= slots) {
      const int64_t rg2 = (row_count + 64 * fit - 1) / (64 * fit);
      if (rg2 <= 24) {
        a.rg = (int)rg2;
        rows_per_wg = 64 * a.rg;
        a.chunks = (int)((row_count + rows_per_wg - 1) / rows_per_wg);
      }
    }
  }
  a.pair_cus = 0;
  if (p.ws) {
    // one round of the 2-waves/SIMD build: workgroups b and b + CUs share a CU (entropy_ws.hip)
    const int cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
    const int64_t total = (int64_t)K * a.chunks;
    if (ws_min_waves(p.DP, ws_ktmax_for(K), want_grad != 0) == 2 && total > cus && total <= 2 * (int64_t)cus)
      a.pair_cus = cus;
  }
  // Span mode (entropy_args.h WsSpan): every CU gets a front part, all but the GP row's a filler part, sized so that
  // they end together.  Used when the launch is the plain wave-split kernel (not its matrix-pipe or small-count forms)
  // and a part is at least a few batches long.
  a.sp = WsSpan();
  a.gp_wgs = 0;
  p.gp_in_ws = false;
  EntArgs at_launch = a;  // (Philox draws generated ahead of the kernel reach it as resident draws: entmc_pregen)
  {
    const size_t n_eps = (size_t)K * (size_t)row_count * D;
    if (eps_mode == VBMC_EPS_PHILOX && ctx->opt_elbo_pregen && n_eps > 0 && n_eps <= ((size_t)1 << 28)) {
      at_launch.eps_mode = VBMC_EPS_RESIDENT;
      at_launch.eps = (const double*)ctx;  // any non-null address: only tested
    }
    // (resident draws whose buffer the caller fills in after planning: the optimiser loop's own)
    if (at_launch.eps_mode == VBMC_EPS_RESIDENT && at_launch.eps == nullptr) at_launch.eps = (const double*)ctx;
  }
  if (p.ws && allow_span && ctx->opt_ws_span && !entmc_small_applies(at_launch, p.DP) &&
      !(ctx->opt_entmc_mfma && entmc_mfma_applies(at_launch, p.DP))) {
    const int cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
    const int waves = ws_min_waves(p.DP, ws_ktmax_for(K), want_grad != 0);
    WsSpan sp;
    sp.cus = cus;
    sp.nb = (int)((row_count + 63) >> 6);
    sp.pad = ctx->opt_ws_pad >= 0 ? ctx->opt_ws_pad : 3;
    sp.T = (int64_t)K * sp.nbv();
    sp.front = ctx->opt_ws_front > 0 ? ctx->opt_ws_front : ws_front_default(p.DP, ws_ktmax_for(K));
    // The GP sums of the host-driven step ride in this launch when they fit WS_GP_SLOTS workgroups of at most six
    // items each (they are then done inside the first half of the kernel).  The slots are left free whether or not
    // anything rides in them: how the batches are cut -- hence the order of every sum -- depends on the job alone,
    // never on where the GP sums run or on any option (the step's launch plans stay bit-identical).
    // How many: five (s, k) blocks per slot are done well inside the kernel (six is the measured limit), so a
    // step with S hyper-parameter samples reserves ceil(S K / 5) slots -- ten at S = 1, K = 50, never fewer than ten
    // (other launches' jobs keep the round-4 partition) nor more than 100; each costs the kernel the filler part of one
    // CU (about a third of a CU's throughput), which is less than the 7-12 us the sums' latency chain costs in front of the
    // kernel.  Beyond 100 slots' worth the sums stay in the prep launch and ten slots are left, as without GP items.
    static const int slots_env = [] {
      const char* e = getenv("VBMC_WS_GP_SLOTS");  // measurement aid: a fixed slot count
      return e ? atoi(e) : 0;
    }();
    int WS_GP_SLOTS = 10;
    constexpr int WS_GP_PER = 5;  // GP items per rider slot
    int gp_wgs = 0;
    bool gp_here = false;
    if (gp_per_slot > 0) {
      // the optimiser loop's two-launch iteration (adam.hip): the pre workgroup of the same launch waits for these sums
      // and must itself be done before the entropy parts are, so the sums get gp_per_slot items per workgroup and the
      // pre workgroup a slot of its own
      gp_wgs = (gp_items + gp_per_slot - 1) / gp_per_slot;
      WS_GP_SLOTS = std::max(10, gp_wgs + 1);
      gp_here = waves == 2 && gp_wgs > 0 && WS_GP_SLOTS <= cus / 2;
      if (!gp_here) {
        WS_GP_SLOTS = 10;
        gp_wgs = 0;
      }
    } else {
      static const int per_env = [] {
        const char* e = getenv("VBMC_WS_GP_PER");  // measurement aid: items per slot
        return e ? atoi(e) : 0;
      }();
      const int per = per_env > 0 ? per_env : WS_GP_PER;
      if (slots_env > 0) WS_GP_SLOTS = slots_env;
      else if (gp_items > 10 * per && gp_items <= 100 * per) WS_GP_SLOTS = (gp_items + per - 1) / per;
      if (WS_GP_SLOTS > cus / 2) WS_GP_SLOTS = cus / 2;
      gp_wgs = gp_items > 0 ? std::min(WS_GP_SLOTS, (gp_items + per - 1) / per) : 0;
      gp_here = waves == 2 && gp_wgs > 0 && gp_items <= (per + 1) * WS_GP_SLOTS;
    }
    sp.pb = waves == 2 ? cus - WS_GP_SLOTS : 0;
    if (sp.pb == 0) sp.front = 1000;
    const int64_t min_part = sp.front > 0 ? (sp.pb > 0 ? 1000 - sp.front : sp.front) * sp.T / sp.W() : 0;
    // (the optimiser loop's two-launch iteration exists in span mode only and gains more than short parts cost: from 2 batches)
    if (min_part >= (gp_per_slot > 0 ? 2 : 3) && sp.T < ((int64_t)1 << 40)) {
      int R = 1;
      int longest = 0;
      for (int j = 0; j < K; ++j) {
        const int n = sp.part_of((int64_t)j * sp.nbv() + sp.nb - 1) - sp.part_of((int64_t)j * sp.nbv()) + 1;
        R = n > R ? n : R;
      }
      for (int u = 0; u < sp.n_parts(); ++u) longest = std::max(longest, (int)(sp.lo(u + 1) - sp.lo(u)));
      sp.R = R;
      a.sp = sp;
      a.chunks = R;
      a.rg = longest;
      a.pair_cus = 0;
      if (gp_here) {
        a.gp_wgs = gp_wgs;
        p.gp_in_ws = true;
      }
    }
  }
  a.stride = 2 + 2 * D + K;
  const size_t n_part = (size_t)K * a.chunks * a.stride;
  int rc = ensure_dev(ctx, &ctx->d_scratch, &ctx->d_scratch_cap, n_part + n_table);
  if (rc) return rc;
  a.partial = ctx->d_scratch;
  p.table = p.ws ? ctx->d_scratch + n_part : nullptr;
  return 0;
}

// Host-only view of the span arithmetic (the CPU suite checks its invariants; the kernels use the same struct).
extern "C" int vbmc_ws_span_layout(int cus, int pb, int front, int nb, int pad, int K, int64_t* part_lo, int* first_part,
                                   int* rows_per_component) {
  if (cus <= 0 || pb < 0 || pb > cus || front <= 0 || front >= 1000 + (pb == 0) || nb <= 0 || pad < 0 || K <= 0 || !part_lo ||
      !first_part || !rows_per_component)
    return VBMC_E_ARG;
  WsSpan sp;
  sp.cus = cus;
  sp.pb = pb;
  sp.front = front;
  sp.nb = nb;
  sp.pad = pad;
  sp.T = (int64_t)K * sp.nbv();
  for (int u = 0; u <= sp.n_parts(); ++u) part_lo[u] = sp.lo(u);
  int R = 1;
  for (int j = 0; j < K; ++j) {
    first_part[j] = sp.part_of((int64_t)j * sp.nbv());
    R = std::max(R, sp.part_of((int64_t)j * sp.nbv() + nb - 1) - first_part[j] + 1);
  }
  *rows_per_component = R;
  return VBMC_OK;
}

void entmc_fill_prep(const vbmc_ctx* ctx, const EntPlan& p, PrepArgs& a) {
  a.mix = ctx->d_mix;
  a.ml = ctx->ml;
  if (p.ws) {
    a.n_table = ctx->K;
    a.DP = p.DP;
    a.K4 = ws_table_rows(ctx->K);
    a.table = p.table;
  }
}

// Philox mode: have extra blocks of the prep launch generate the draws (they fill the GPU while
// its few table / GP blocks sit in latency chains) and run the entropy kernel in its
// resident-draw form on the identical values.  Call after entmc_plan + entmc_fill_prep.
// VBMC_ELBO_PREGEN=0 keeps the generation inside the entropy kernel.
int entmc_pregen(vbmc_ctx* ctx, EntPlan& p, PrepArgs& pa) {
  const bool on = ctx->opt_elbo_pregen != 0;
  const int D = ctx->D, K = ctx->K;
  const size_t n_eps = (size_t)K * (size_t)p.a.row_count * D;
  if (!on || p.a.eps_mode != VBMC_EPS_PHILOX || n_eps == 0 || n_eps > ((size_t)1 << 28)) return 0;
  vbmc_ctx::AheadDraws& ah = ctx->ahead;
  int cur;
  if (ah.valid && ah.seed == p.a.seed && ah.K == K && ah.D == D && ah.rows == p.a.row_count &&
      ah.n_half == p.a.n_half && ah.row_begin == p.a.row_begin) {
    // the previous evaluation already queued exactly these draws (spare workgroups of its finish launch) -- or the
    // first part of them: the rest is generated by this evaluation's prep launch
    cur = ah.buf;
    p.pregen_hit = true;
    if (ah.frac < 1.0)
      pa.gen = make_gen_slice(ctx->d_epsgen[cur], K, D, p.a.row_count, p.a.n_half, p.a.row_begin, p.a.seed, nullptr, ah.frac, 1.0);
  } else {
    cur = ah.valid ? 1 - ah.buf : 0;  // keep clear of a speculative buffer that is not the one wanted
    int rc = ensure_dev(ctx, &ctx->d_epsgen[cur], &ctx->d_epsgen_cap[cur], n_eps);
    if (rc) return rc;
    pa.gen = make_gen_slice(ctx->d_epsgen[cur], K, D, p.a.row_count, p.a.n_half, p.a.row_begin, p.a.seed, nullptr, 0.0, 1.0);
  }
  ah.valid = false;
  ctx->gen_cur = cur;
  p.a.eps_mode = VBMC_EPS_RESIDENT;
  p.a.eps = ctx->d_epsgen[cur];
  p.a.eps_rows = p.a.row_count;
  return 0;
}

GenSlice entmc_ahead_slice(vbmc_ctx* ctx, const EntPlan& p) {
  const double frac_end = 1.0;
  const EntArgs& a = p.a;
  GenSlice none;
  if (!ctx->opt_elbo_ahead || a.eps == nullptr || a.eps != ctx->d_epsgen[ctx->gen_cur]) return none;
  const int D = ctx->D, K = ctx->K, other = 1 - ctx->gen_cur;
  const size_t n_eps = (size_t)K * (size_t)a.row_count * D;
  if (ctx->d_epsgen_cap[other] < n_eps || !ctx->d_epsgen[other]) {
    // first use only.  ensure_dev would wait for the stream when growing an existing buffer, and
    // that wait must not sit between this evaluation's launches and its result: allocate fresh
    if (ctx->d_epsgen[other]) return none;  // too small: leave it; the next evaluation re-plans
    if (ensure_dev(ctx, &ctx->d_epsgen[other], &ctx->d_epsgen_cap[other], n_eps)) return none;
  }
  vbmc_ctx::AheadDraws& ah = ctx->ahead;
  ah.seed = a.seed + 1;
  ah.K = K;
  ah.D = D;
  ah.rows = a.row_count;
  ah.n_half = a.n_half;
  ah.row_begin = a.row_begin;
  ah.buf = other;
  ah.frac = frac_end;
  ah.valid = true;  // the caller launches the slice right away
  return make_gen_slice(ctx->d_epsgen[other], K, D, a.row_count, a.n_half, a.row_begin, ah.seed, nullptr, 0.0, frac_end);
}

bool entmc_uses_mfma(const vbmc_ctx* ctx, const EntPlan& p) {
  return p.ws && p.a.sp.cus == 0 && !entmc_small_applies(p.a, p.DP) && ctx->opt_entmc_mfma && entmc_mfma_applies(p.a, p.DP);
}

int entmc_launch_main(vbmc_ctx* ctx, const EntPlan& p) {
  const EntArgs& a = p.a;
  // timing: the wave-split launch carries the event pair on its own dispatch packet; the generic
  // kernel is bracketed by two records (each a barrier packet, ~6 us between dependent kernels)
  hipEvent_t e0 = ctx->timing ? ctx->ev[0] : nullptr, e1 = ctx->timing ? ctx->ev[1] : nullptr;
  const bool small = p.ws && a.sp.cus == 0 && entmc_small_applies(a, p.DP);  // (a span-mode plan is the wave-split kernel's)
  const bool mfma = p.ws && a.sp.cus == 0 && !small && ctx->opt_entmc_mfma && entmc_mfma_applies(a, p.DP);
  ctx->last_plan[0] = small ? 2 : mfma ? 3 : (p.ws ? (a.sp.cus > 0 ? 5 : 1) : 0);  // (5: the wave-split kernel in span mode)
  ctx->last_plan[1] = a.rg;
  ctx->last_plan[2] = a.chunks;
  ctx->last_plan[3] = a.eps_mode == VBMC_EPS_RESIDENT ? 1 : 0;
  const bool bracket = ctx->timing && (!p.ws || small);
  if (bracket) HIP_TRY(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
  if (small) {
    launch_entmc_small(ctx->stream, a, p.DP, p.table);
  } else if (mfma) {
    launch_entmc_mfma(ctx->stream, a, p.DP, p.table, e0, e1);
  } else if (p.ws) {
    switch (p.DP) {
#define VBMC_CASE_WS(dp) case dp: launch_entmc_ws_dp##dp(ctx->stream, a, p.table, e0, e1); break;
      VBMC_WS_DPS(VBMC_CASE_WS)
#undef VBMC_CASE_WS
    }
  } else {
    switch (p.DP) {
      case 2: launch_entmc_dp<2>(ctx, a); break;
      case 4: launch_entmc_dp<4>(ctx, a); break;
      case 6: launch_entmc_dp<6>(ctx, a); break;
      case 8: launch_entmc_dp<8>(ctx, a); break;
      case 10: launch_entmc_dp<10>(ctx, a); break;
      case 12: launch_entmc_dp<12>(ctx, a); break;
      case 16: launch_entmc_dp<16>(ctx, a); break;
      case 20: launch_entmc_dp<20>(ctx, a); break;
      case 24: launch_entmc_dp<24>(ctx, a); break;
      default: launch_entmc_dp<32>(ctx, a); break;
    }
  }
  if (ctx->timing) {
    if (bracket) HIP_TRY(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    ctx->ev_valid[0] = true;
  }
  HIP_TRY(ctx, hipGetLastError());
  return 0;
}

namespace {
__global__ __launch_bounds__(256) void eps_gen_kernel(GenSlice g) { gen_slice_block(g, blockIdx.x, threadIdx.x); }
}  // namespace

GenSlice make_gen_slice(double* eps, int K, int D, int64_t rows, int64_t n_half, int64_t row_begin,
                        uint64_t seed, const int* seed_add, double frac_begin, double frac_end) {
  GenSlice g;
  g.eps = eps;
  g.K = K;
  g.D = D;
  g.rows = rows;
  g.n_half = n_half;
  g.row_begin = row_begin;
  g.seed = seed;
  g.seed_add = seed_add;
  g.nb = (D + 3) / 4;
  auto magic = [](uint64_t d) { return (uint32_t)std::min<uint64_t>(d ? 0x100000000ull / d : 0, 0xFFFFFFFFull); };
  g.nb_magic = magic((uint64_t)g.nb);
  g.rows_magic = magic((uint64_t)rows);
  const int64_t total = (int64_t)K * rows * g.nb;
  const int64_t b = (int64_t)(frac_begin * (double)total), e = frac_end >= 1.0 ? total : (int64_t)(frac_end * (double)total);
  g.item_begin = b;
  g.item_count = e > b ? e - b : 0;
  g.n_blocks = (int)((g.item_count + 255) / 256);
  return g;
}

namespace {
// Multi-GPU step: the all-reduced raw vector (device memory, written by the collective in front of
// this launch) goes to the pinned block in one coalesced copy and the sequence number is published;
// the other workgroups generate the next evaluation's draws.
__global__ __launch_bounds__(256) void entmc_publish_kernel(const double* __restrict__ src, DoneSignal done, GenSlice gen) {
  if (blockIdx.x > 0) {
    gen_slice_block(gen, blockIdx.x - 1, threadIdx.x);
    return;
  }
  if (done.ident_dst && threadIdx.x == 255) {
    __hip_atomic_store(done.ident_dst, *done.ident_src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(done.ident_dst + 1, done.ident_seed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  staged_copy_to_host(src, done.host_out, done.host_n);
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(done.flag, done.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace

int entmc_launch_publish(vbmc_ctx* ctx, const double* d_raw, const DoneSignal& done, const GenSlice* gen) {
  const GenSlice g = gen ? *gen : GenSlice();
  hipLaunchKernelGGL(entmc_publish_kernel, dim3(1 + g.n_blocks), dim3(256), 0, ctx->stream, d_raw, done, g);
  HIP_TRY(ctx, hipGetLastError());
  return 0;
}

int launch_eps_gen(vbmc_ctx* ctx, hipStream_t st, const GenSlice& g) {
  if (g.n_blocks <= 0) return 0;
  hipLaunchKernelGGL(eps_gen_kernel, dim3((unsigned)g.n_blocks), dim3(256), 0, st, g);
  HIP_TRY(ctx, hipGetLastError());
  return 0;
}

int entmc_launch_finish(vbmc_ctx* ctx, const EntPlan& p, double* raw_out, const GenSlice* gen,
                        const DoneSignal* done) {
  const int n_out = raw_len(ctx->D, ctx->K);
  if ((uint64_t)ctx->K * (uint64_t)p.a.chunks * (uint64_t)p.a.stride >= ((uint64_t)1 << 31))
    return vbmc_fail(ctx, VBMC_E_UNSUP, "entropy: partial block of %d x %d rows too large", ctx->K, p.a.chunks);
  const GenSlice g = gen ? *gen : GenSlice();
  const DoneSignal ds = done ? *done : DoneSignal();
  hipLaunchKernelGGL(entmc_finish_kernel, dim3((n_out + 3) / 4 + g.n_blocks), dim3(256), 0, ctx->stream,
                     p.a.partial, p.a.chunks, p.a.stride, ctx->d_mix, ctx->ml, p.inv_ns,
                     p.a.want_grad, p.ws ? 1 : 0, raw_out, g, ds);
  HIP_TRY(ctx, hipGetLastError());
  return 0;
}

// Stand-alone entropy: prep (table only) -> main -> finish.
int launch_entmc(vbmc_ctx* ctx, int64_t ns_per_comp, int eps_mode, uint64_t seed,
                 int64_t row_begin, int64_t row_count, int want_grad, double* d_raw) {
  EntPlan p;
  int rc = entmc_plan(ctx, ns_per_comp, eps_mode, seed, row_begin, row_count, want_grad, p);
  if (rc) return rc;
  PrepArgs pa;
  entmc_fill_prep(ctx, p, pa);
  rc = entmc_pregen(ctx, p, pa);
  if (rc) return rc;
  rc = launch_prep(ctx, pa);
  if (rc) return rc;
  rc = entmc_launch_main(ctx, p);
  if (rc) return rc;
  return entmc_launch_finish(ctx, p, d_raw);
}

int launch_entlb(vbmc_ctx* ctx, double* d_res) {
  const int D = ctx->D, K = ctx->K;
  const size_t n_part = (size_t)K + (size_t)K * D, n_G = (size_t)K * K;
  int rc = ensure_dev(ctx, &ctx->d_scratch, &ctx->d_scratch_cap, n_part + n_G + K);
  if (rc) return rc;
  double* part = ctx->d_scratch;
  double* Gm = part + n_part;
  double* gsum = Gm + n_G;
  hipLaunchKernelGGL(entlb_rows_kernel, dim3(K), dim3(256), 0, ctx->stream, ctx->d_mix, ctx->ml, Gm,
                     gsum);
  size_t lds = sizeof(double) * (size_t)(WAVES * (2 * D + 2));
  hipLaunchKernelGGL(entlb_kernel, dim3(K), dim3(256), lds, ctx->stream, ctx->d_mix, ctx->ml, 1,
                     (const double*)Gm, (const double*)gsum, d_res, part);
  hipLaunchKernelGGL(entlb_finish_kernel, dim3((1 + D + 3) / 4), dim3(256), 0, ctx->stream,
                     ctx->d_mix, ctx->ml, 1, (const double*)part, d_res);
  HIP_TRY(ctx, hipGetLastError());
  return 0;
}
