// acquisition-function evaluation, and row a13 (_sq_dist).
//
// AbstractAcqFcn.__call__ (acquisition_functions/abstract_acq_fcn.py:68-147) evaluates, for a
// batch of points, gp.predict(separate_samples=True), the mean / total variance over the GP
// hyper-parameter samples, vp.pdf (or log pdf) and one of the closed-form acquisition
// formulas.  Here the points are uploaded once, the per-sample predictive moments stay on the
// device, and one fused kernel turns them into the acquisition value: the only D2H traffic
// is the M results.
#include <cmath>
#include <cstring>

#include <chrono>

#include "common.h"
#include "fastmath.h"

namespace {

constexpr double kRealMin = 2.2250738585072014e-308;   // sys.float_info.min
constexpr double kRealMax = 1.7976931348623157e+308;   // sys.float_info.max
constexpr double kLogRealMin = -708.3964185322641;     // np.log(sys.float_info.min)

// the acquisition value of one point from its mean prediction, total variance and (log) density
__device__ __forceinline__ double acq_value(int kind, double f_bar, double var_tot, double dens, double sn, double y_max,
                                            double tol_var) {
  double a;
  bool log_flag = false;
  switch (kind) {
    case VBMC_ACQ_LOG: {  // acq_fcn_log.py:43-52
      const double log_p = fmax(dens, kLogRealMin);
      a = -(log(var_tot) + f_bar - y_max + log_p);
      log_flag = true;
      break;
    }
    case VBMC_ACQ_VANILLA: {  // acq_fcn_vanilla.py:38-42
      const double p = fmax(dens, kRealMin);
      a = -var_tot * (p * p);
      break;
    }
    case VBMC_ACQ_NOISY: {  // acq_fcn_noisy.py:33-41
      const double p = fmax(dens, kRealMin);
      a = -var_tot * (1.0 - sn / (var_tot + sn)) * exp(f_bar - y_max) * p;
      break;
    }
    default: {  // VBMC_ACQ_STD, acq_fcn.py:38-45
      const double p = fmax(dens, kRealMin);
      a = -var_tot * exp(f_bar - y_max) * p;
      break;
    }
  }
  // variance regularisation (abstract_acq_fcn.py:112-128)
  if (tol_var > 0.0 && var_tot < tol_var) {
    const double pen = tol_var / var_tot - 1.0;
    if (log_flag)
      a += pen;
    else
      a *= exp(-pen);
  }
  return fmax(a, -kRealMax);  // :130-131
}

// fmu, fs2: [S][M] per-sample predictive moments; dens: pdf (or log pdf for VBMC_ACQ_LOG).
__global__ __launch_bounds__(256) void acq_combine_kernel(
    const double* __restrict__ fmu, const double* __restrict__ fs2, const double* __restrict__ dens,
    const double* __restrict__ sn2, int S, int64_t M, int64_t ld, int kind, double y_max,
    double tol_var, double* __restrict__ acq, double* __restrict__ f_bar_out,
    double* __restrict__ var_tot_out) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  // abstract_acq_fcn.py:82-97
  double fsum = 0.0, vsum = 0.0;
  for (int s = 0; s < S; ++s) {
    fsum += fmu[(size_t)s * ld + m];
    vsum += fs2[(size_t)s * ld + m];
  }
  const double f_bar = fsum / S, var_bar = vsum / S;
  double var_f = 0.0;
  if (S > 1) {
    double q = 0.0;
    for (int s = 0; s < S; ++s) {
      const double t = fmu[(size_t)s * ld + m] - f_bar;
      q += t * t;
    }
    var_f = q / (S - 1);
  }
  const double var_tot = var_f + var_bar;
  acq[m] = acq_value(kind, f_bar, var_tot, dens[m], kind == VBMC_ACQ_NOISY ? sn2[m] : 0.0, y_max, tol_var);
  if (f_bar_out) f_bar_out[m] = f_bar;
  if (var_tot_out) var_tot_out[m] = var_tot;
}

// Small batches (at most 256 points: a CMA-ES population, a single point): what follows the two predict products as ONE
// launch, wave = point -- predict's finish (gp.hip predict_finish_kernel: partial sums -> f_mu, f_s2
// per GP sample, mean function), the mixture density at the point (mixture.hip's wave-per-point form), the
// formula -- with the results stored write-through into pinned host memory, drained, counted, and a completion word published for
// the host to poll (the recipe of the ELBO step's completion word, entropy.hip): three launches, no copy call and no
// stream synchronisation per batch instead of five launches, two copies and a wait.
struct AcqTail {
  const double* part;  // predict partials [S][2][ntiles][M]
  int64_t pstride;
  int ntiles, M, D, P, S, mean_kind, kind, log_dens;
  const double *hyp, *smeta, *xs, *mix, *sn2;
  MixLayout ml;
  double y_max, tol_var;
  double *acq, *f_bar, *var_tot;  // pinned host memory (device addresses); f_bar / var_tot nullable
  uint64_t* flag;
  uint64_t seq;
  int* cnt;  // zero between launches: workgroups that have stored their results
};

__global__ __launch_bounds__(256) void acq_tail_small_kernel(AcqTail a) {
  // wave = point: lanes over the GP samples for predict's finish, lanes over the components for the density
  __shared__ int s_last;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = blockIdx.x * 4 + wave, M = a.M, D = a.D, S = a.S;
  if (m < M) {
    // ---- predict's finish (gp.hip predict_finish_kernel), lane = GP sample ----
    double fmu = 0.0, fs2 = 0.0;
    if (lane < S) {
      const int smp = lane;
      const double* part = a.part + (size_t)smp * a.pstride;
      const double* fpart = part + (size_t)a.ntiles * M;
      const double* hyp = a.hyp + (size_t)smp * a.P;
      const bool chol = a.smeta[3 * smp] != 0.0;
      const double sf2 = exp(2.0 * hyp[D]);
      double sv = 0.0, f = 0.0;
      for (int t = 0; t < a.ntiles; ++t) {
        sv += part[(size_t)t * M + m];
        f += fpart[(size_t)t * M + m];
      }
      fs2 = fmax(chol ? sf2 - sv : sf2 + sv, 0.0);
      double mean = 0.0;
      const double* hm = hyp + D + 2;
      if (a.mean_kind == VBMC_MEAN_CONST) mean = hm[0];
      if (a.mean_kind == VBMC_MEAN_NEGQUAD) {
        mean = hm[0];
        for (int d = 0; d < D; ++d) {
          const double t = (a.xs[m * D + d] - hm[1 + d]) * exp(-hm[1 + D + d]);
          mean -= 0.5 * t * t;
        }
      }
      fmu = mean + f;
    }
    // ---- abstract_acq_fcn.py:82-97 ----
    const double f_bar = fm::wave_sum_dpp(fmu) / S, var_bar = fm::wave_sum_dpp(fs2) / S;
    double var_f = 0.0;
    if (S > 1) {
      const double t = lane < S ? fmu - f_bar : 0.0;
      var_f = fm::wave_sum_dpp(t * t) / (S - 1);
    }
    const double var_tot = var_f + var_bar;
    // ---- the mixture density at the point (variational_posterior.py:450-463), lane = component ----
    const int K = a.ml.K;
    const double* mup = a.mix + a.ml.o_mup;
    const double* is2 = a.mix + a.ml.o_is2;
    const double* wc = a.mix + a.ml.o_wc;
    const double* ilam = a.mix + a.ml.o_ilam;
    double y = 0.0;
    for (int k = lane; k < K; k += 64) {
      const double* mk = mup + k * D;
      double d2 = 0.0;
      for (int d = 0; d < D; ++d) {
        const double u = a.xs[m * D + d] * ilam[d] - mk[d];
        d2 = fma(u, u, d2);
      }
      y += wc[k] * fm::exp2_fast((-0.5 * 0x1.71547652b82fep+0 * is2[k]) * d2);
    }
    y = fm::wave_sum_dpp(y);
    if (lane == 0) {
      const double dens = a.log_dens ? ((y == 0.0) ? -INFINITY : log(y)) : y;
      const double av = acq_value(a.kind, f_bar, var_tot, dens, a.kind == VBMC_ACQ_NOISY ? a.sn2[m] : 0.0, a.y_max, a.tol_var);
      __hip_atomic_store(a.acq + m, av, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (a.f_bar) __hip_atomic_store(a.f_bar + m, f_bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (a.var_tot) __hip_atomic_store(a.var_tot + m, var_tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  // every workgroup: results acknowledged, then ONE count; the last workgroup to count publishes the sequence number
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  __syncthreads();
  if (threadIdx.x == 0)
    s_last = __hip_atomic_fetch_add(a.cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
  __syncthreads();
  if (s_last && threadIdx.x == 0) {
    __hip_atomic_store(a.cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(a.flag, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

}  // namespace

extern "C" int vbmc_acq_eval(vbmc_ctx* ctx, int64_t M, const double* xs_MxD, int kind, double y_max,
                             double tol_gp_var, const double* sn2_M, double* acq_M, double* f_bar_M,
                             double* var_tot_M) {
  if (!ctx || (M > 0 && (!xs_MxD || !acq_M))) return VBMC_E_ARG;
  NEED_DEVICE(ctx);
  if (!ctx->gp.set) return vbmc_fail(ctx, VBMC_E_ARG, "acq_eval: GP not set");
  if (!ctx->mix_set) return vbmc_fail(ctx, VBMC_E_ARG, "acq_eval: mixture not set");
  if (ctx->gp.D != ctx->D) return vbmc_fail(ctx, VBMC_E_ARG, "acq_eval: GP/mixture D mismatch");
  if (kind < VBMC_ACQ_STD || kind > VBMC_ACQ_NOISY)
    return vbmc_fail(ctx, VBMC_E_UNSUP, "acq_eval: unknown acquisition kind %d", kind);
  if (kind == VBMC_ACQ_NOISY && !sn2_M)
    return vbmc_fail(ctx, VBMC_E_ARG, "acq_eval: the noisy acquisition needs sn2 per point");
  if (M == 0) return VBMC_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const GpState& g = ctx->gp;
  const int N = g.N, D = g.D, S = g.S;
  if (D > 32) return vbmc_fail(ctx, VBMC_E_UNSUP, "acq_eval: D=%d > 32 not supported", D);
  const int ntiles = (N + 63) / 64;
  int64_t mb = ((int64_t)1 << 27) / ((int64_t)S * N);  // S kernel matrices of a batch under 1 GiB
  mb = mb > 65536 ? 65536 : (mb < 64 ? 64 : (mb / 64) * 64);
  if (M < mb) mb = M;
  // scratch: xs | Ks [S] | part,fpart [S] | fmu[S] | fs2[S] | dens | sn2 | acq | f_bar | var_tot
  const size_t ks_n = predict_ks_elems(S, mb, N);
  const size_t need = align32((size_t)mb * D) + ks_n + 2 * (size_t)S * ntiles * mb + 2 * (size_t)S * mb + 5 * (size_t)mb;
  int rc = ensure_dev(ctx, &ctx->d_scratch, &ctx->d_scratch_cap, need);
  if (rc) return rc;
  rc = ensure_pinned(ctx, 3 * (size_t)mb);
  if (rc) return rc;
  double* d_xs = ctx->d_scratch;
  double* d_Ks = d_xs + align32((size_t)mb * D);  // 256-byte aligned: read by 16-byte LDS-direct loads
  double* d_part = d_Ks + ks_n;
  double* d_fmu = d_part + 2 * (size_t)S * ntiles * mb;
  double* d_fs2 = d_fmu + (size_t)S * mb;
  double* d_dens = d_fs2 + (size_t)S * mb;
  double* d_sn2 = d_dens + mb;
  double* d_acq = d_sn2 + mb;  // acq | f_bar | var_tot, contiguous
  // ---- small batches (a CMA-ES population, a single point): the CPU writes the points straight into host-writable
  // device memory, the last kernel writes the results into pinned host memory and publishes a completion word that the
  // CPU polls: no copy calls and no stream synchronisation around ~30 us of kernels ----
  if (M <= 256 && M <= mb && S <= 64 && ctx->opt_acq_poll && !ctx->acq_fg_failed) {
    if (!ctx->d_acq_fg) {
      int large_bar = 0;
      if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, ctx->device) != hipSuccess || !large_bar ||
          hipExtMallocWithFlags((void**)&ctx->d_acq_fg, sizeof(double) * 256 * 33, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        ctx->d_acq_fg = nullptr;
        ctx->acq_fg_failed = true;
      }
    }
  }
  // (M <= mb: the scratch and the pinned result rows are laid out for batches of mb points, which S * N can push below M)
  if (M <= 256 && M <= mb && S <= 64 && ctx->opt_acq_poll && ctx->d_acq_fg) {
    if (ctx->spec.armed) spec_disarm(ctx);  // (launches waiting for a theta would sit in front of these)
    const int64_t m = M;
    memcpy(ctx->d_acq_fg, xs_MxD, sizeof(double) * m * D);
    if (kind == VBMC_ACQ_NOISY) memcpy(ctx->d_acq_fg + 256 * 32, sn2_M, sizeof(double) * m);
    __builtin_ia32_sfence();  // write-combined stores drained before the doorbell of the launches
    const double* x_dev = ctx->d_acq_fg;
    rc = launch_gp_predict_products(ctx, m, x_dev, d_Ks, d_part);  // K*, the variance product: their partial sums
    if (rc) return rc;
    const uint64_t seq = ++ctx->acq_seq;
    volatile uint64_t* flag = ctx->h_done + 7;
    AcqTail t;
    t.part = d_part;
    t.ntiles = ntiles;
    t.pstride = 2 * (int64_t)ntiles * m;
    t.M = (int)m;
    t.D = D;
    t.P = g.P;
    t.S = S;
    t.mean_kind = g.mean_kind;
    t.kind = kind;
    t.log_dens = kind == VBMC_ACQ_LOG;
    t.hyp = g.d_hyp;
    t.smeta = g.d_smeta;
    t.xs = x_dev;
    t.mix = ctx->d_mix;
    t.sn2 = ctx->d_acq_fg + 256 * 32;
    t.ml = ctx->ml;
    t.y_max = y_max;
    t.tol_var = tol_gp_var;
    t.acq = ctx->hp_dev;
    t.f_bar = f_bar_M ? ctx->hp_dev + mb : nullptr;
    t.var_tot = var_tot_M ? ctx->hp_dev + 2 * mb : nullptr;
    t.flag = ctx->hd_done + 7;
    t.seq = seq;
    t.cnt = ctx->d_done_cnt + 4;
    hipLaunchKernelGGL(acq_tail_small_kernel, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, ctx->stream, t);
    HIP_TRY(ctx, hipGetLastError());
    const auto t0 = std::chrono::steady_clock::now();
    bool seen = false;
    for (long spin = 0;; ++spin) {
      if (*flag == seq) {
        seen = true;
        break;
      }
      if ((spin & 1023) == 1023 &&
          std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0)
        break;  // (a stuck queue: the stream wait below reports what happened)
      __builtin_ia32_pause();
    }
    if (!seen) HIP_TRY(ctx, stream_wait(ctx));
    if (!seen && *flag != seq) return vbmc_fail(ctx, VBMC_E_HIP, "acq_eval: the completion word did not arrive");
    __atomic_thread_fence(__ATOMIC_ACQUIRE);  // the results were written before the word: read them after it
    ctx->pack_in_flight = false;  // (everything queued before the last kernel has run)
    memcpy(acq_M, ctx->h_pinned, sizeof(double) * m);
    if (f_bar_M) memcpy(f_bar_M, ctx->h_pinned + mb, sizeof(double) * m);
    if (var_tot_M) memcpy(var_tot_M, ctx->h_pinned + 2 * mb, sizeof(double) * m);
    return VBMC_OK;
  }
  for (int64_t o = 0; o < M; o += mb) {
    const int64_t m = (M - o) < mb ? (M - o) : mb;
    HIP_TRY(ctx, hipMemcpyAsync(d_xs, xs_MxD + o * D, sizeof(double) * m * D, hipMemcpyHostToDevice,
                                ctx->stream));
    if (kind == VBMC_ACQ_NOISY)
      HIP_TRY(ctx, hipMemcpyAsync(d_sn2, sn2_M + o, sizeof(double) * m, hipMemcpyHostToDevice, ctx->stream));
    rc = launch_gp_predict_all(ctx, m, d_xs, d_Ks, d_part, 0, d_fmu, d_fs2, mb);
    if (rc) return rc;
    rc = launch_mixture_pdf(ctx, m, d_xs, kind == VBMC_ACQ_LOG, 0, INFINITY, d_dens, nullptr);
    if (rc) return rc;
    hipLaunchKernelGGL(acq_combine_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const double*)d_fmu, (const double*)d_fs2, (const double*)d_dens,
                       (const double*)d_sn2, S, m, mb, kind, y_max, tol_gp_var, d_acq, d_acq + mb,
                       d_acq + 2 * mb);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_pinned, d_acq, sizeof(double) * m, hipMemcpyDeviceToHost, ctx->stream));
    if (f_bar_M)
      HIP_TRY(ctx, hipMemcpyAsync(ctx->h_pinned + mb, d_acq + mb, sizeof(double) * m, hipMemcpyDeviceToHost,
                                  ctx->stream));
    if (var_tot_M)
      HIP_TRY(ctx, hipMemcpyAsync(ctx->h_pinned + 2 * mb, d_acq + 2 * mb, sizeof(double) * m,
                                  hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, stream_wait(ctx));
    memcpy(acq_M + o, ctx->h_pinned, sizeof(double) * m);
    if (f_bar_M) memcpy(f_bar_M + o, ctx->h_pinned + mb, sizeof(double) * m);
    if (var_tot_M) memcpy(var_tot_M + o, ctx->h_pinned + 2 * mb, sizeof(double) * m);
  }
  return VBMC_OK;
}

// a13: c[i][j] = |a_i - b_j|^2 the way AbstractAcqFcn._sq_dist computes it
// (acquisition_functions/abstract_acq_fcn.py:195-222); argmin_n (nullable) = np.argmin(c, axis=1)
// (the nearest-neighbour lookup of _estimate_observation_noise, :244-252).
extern "C" int vbmc_sq_dist(vbmc_ctx* ctx, int64_t n, int64_t m, int D, const double* a_nxD,
                            const double* b_mxD, double* c_nxm, int64_t* argmin_n) {
  if (!ctx || n < 0 || m < 0 || D < 1) return VBMC_E_ARG;
  if (n > 0 && m > 0 && (!a_nxD || !b_mxD)) return VBMC_E_ARG;
  NEED_DEVICE(ctx);
  if (D > 32) return vbmc_fail(ctx, VBMC_E_UNSUP, "sq_dist: D=%d > 32 not supported", D);
  if (m > (int64_t)1 << 24) return vbmc_fail(ctx, VBMC_E_UNSUP, "sq_dist: more than 2^24 columns");
  if (argmin_n && m == 0 && n > 0)
    return vbmc_fail(ctx, VBMC_E_ARG, "sq_dist: argmin of an empty row");
  if (n == 0 || m == 0) return VBMC_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  // mu = m/(n+m) mean(b) + n/(n+m) mean(a)   (:212-215)
  std::vector<double> ma(D, 0.0), mbv(D, 0.0), cen(D);
  for (int64_t i = 0; i < n; ++i)
    for (int d = 0; d < D; ++d) ma[d] += a_nxD[i * D + d];
  for (int64_t j = 0; j < m; ++j)
    for (int d = 0; d < D; ++d) mbv[d] += b_mxD[j * D + d];
  const double tot = (double)(n + m);
  for (int d = 0; d < D; ++d) cen[d] = ((double)m / tot) * (mbv[d] / m) + ((double)n / tot) * (ma[d] / n);
  const int ntiles = (int)((m + 63) / 64);
  // rows per pass: keep the n x m block under ~256 MiB
  int64_t nb = c_nxm ? ((int64_t)1 << 25) / (m > 0 ? m : 1) : ((int64_t)1 << 20);
  nb = nb < 64 ? 64 : nb;
  nb = nb > n ? n : nb;
  const size_t n_c = c_nxm ? (size_t)nb * m : 0;
  const size_t n_p = argmin_n ? 2 * (size_t)ntiles * nb + (size_t)nb : 0;  // tile minima/indices + int64 out
  const size_t need = (size_t)D + (size_t)nb * D + (size_t)m * D + n_c + n_p;
  int rc = ensure_dev(ctx, &ctx->d_scratch, &ctx->d_scratch_cap, need);
  if (rc) return rc;
  double* d_cen = ctx->d_scratch;
  double* d_a = d_cen + D;
  double* d_b = d_a + (size_t)nb * D;
  double* d_c = c_nxm ? d_b + (size_t)m * D : nullptr;
  double* d_p = d_b + (size_t)m * D + n_c;
  int64_t* d_am = argmin_n ? (int64_t*)(d_p + 2 * (size_t)ntiles * nb) : nullptr;
  HIP_TRY(ctx, hipMemcpyAsync(d_cen, cen.data(), sizeof(double) * D, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_b, b_mxD, sizeof(double) * m * D, hipMemcpyHostToDevice, ctx->stream));
  for (int64_t o = 0; o < n; o += nb) {
    const int64_t cnt = (n - o) < nb ? (n - o) : nb;
    HIP_TRY(ctx, hipMemcpyAsync(d_a, a_nxD + o * D, sizeof(double) * cnt * D, hipMemcpyHostToDevice, ctx->stream));
    rc = launch_sq_dist(ctx, d_a, cnt, d_b, (int)m, D, d_cen, d_c, d_p, d_am);
    if (rc) return rc;
    if (c_nxm)
      HIP_TRY(ctx, hipMemcpyAsync(c_nxm + o * m, d_c, sizeof(double) * cnt * m, hipMemcpyDeviceToHost, ctx->stream));
    if (argmin_n)
      HIP_TRY(ctx, hipMemcpyAsync(argmin_n + o, d_am, sizeof(int64_t) * cnt, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, stream_wait(ctx));
  }
  return VBMC_OK;
}
