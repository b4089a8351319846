// The fused objective: one call = one evaluation of the negative ELBO
// (reference vbmc/variational_optimization.py:991-1235 _neg_elcbo) with a single
// host<->device round trip.
#include <chrono>
#include <cstdlib>
#include <cmath>
#include <cstring>

#include "common.h"
#include "entropy_args.h"

// _soft_bound_loss (:645-657)
static double soft_bound_loss(const std::vector<double>& x, const double* lb, const double* ub,
                              double tol_con, std::vector<double>* dy) {
  double y = 0.0;
  if (dy) dy->assign(x.size(), 0.0);
  for (size_t i = 0; i < x.size(); ++i) {
    const double ell = (ub[i] - lb[i]) * tol_con;
    if (x[i] < lb[i]) {
      const double t = (lb[i] - x[i]) / ell;
      y += 0.5 * t * t;
      if (dy) (*dy)[i] = (x[i] - lb[i]) / (ell * ell);
    }
    if (x[i] > ub[i]) {
      const double t = (x[i] - ub[i]) / ell;
      y += 0.5 * t * t;
      if (dy) (*dy)[i] = (x[i] - ub[i]) / (ell * ell);
    }
  }
  return y;
}

extern "C" int vbmc_neg_elcbo(vbmc_ctx* ctx, double* theta, int n_theta,
                              const vbmc_elbo_opts* opts, double* F, double* dF, double* G,
                              double* H, double* mu_KxD, double* sigma_K, double* lambd_D,
                              double* w_K, double* eta_K) {
  if (!ctx || !theta || !opts) return VBMC_E_ARG;
  NEED_DEVICE(ctx);
  if (!ctx->mix_set) return vbmc_fail(ctx, VBMC_E_ARG, "neg_elcbo: mixture (D,K) not set");
  if (!ctx->gp.set) return vbmc_fail(ctx, VBMC_E_ARG, "neg_elcbo: GP not set");
  if (ctx->gp.D != ctx->D) return vbmc_fail(ctx, VBMC_E_ARG, "neg_elcbo: GP/mixture D mismatch");
  using clk = std::chrono::steady_clock;
  const auto t_begin = clk::now();
  auto us_since = [](clk::time_point a) {
    return std::chrono::duration<double, std::micro>(clk::now() - a).count();
  };
  const int D = ctx->D, K = ctx->K;
  const int mask = opts->optimize_mask;
  const bool o_mu = mask & 1, o_sg = mask & 2, o_lm = mask & 4, o_w = mask & 8;
  int rc = vbmc_theta_to_mixture(ctx, theta, n_theta, mask, mu_KxD, sigma_K, lambd_D, w_K, eta_K);
  if (rc) return rc;
  if (o_w) {
    // the reference shifts its caller's theta tail in place (:1082-1085)
    double* e = theta + (n_theta - K);
    double mx = e[0];
    for (int k = 1; k < K; ++k) mx = e[k] > mx ? e[k] : mx;
    for (int k = 0; k < K; ++k) e[k] -= mx;
  }
  const int grad_flags = opts->compute_grad ? mask : 0;
  const GpState& g = ctx->gp;
  const int S = g.S;
  const int st = 1 + 2 * D;
  const size_t n_res = (size_t)S * K * st;
  const int n_raw = raw_len(D, K);
  const bool mc = opts->ns_per_comp > 0;
  const bool lb_dev = !mc && K > 1;

  int64_t row_begin = opts->row_begin, row_count = opts->row_count;
  if (mc) {
    if (opts->ns_per_comp & 1)
      return vbmc_fail(ctx, VBMC_E_ARG, "neg_elcbo: ns_per_comp must be even");
    const int64_t n_half = opts->ns_per_comp / 2;
    if (row_count < 0) {
      row_begin = n_half * ctx->rank / ctx->world;
      row_count = n_half * (ctx->rank + 1) / ctx->world - row_begin;
    }
    if (row_begin < 0 || row_begin + row_count > n_half)
      return vbmc_fail(ctx, VBMC_E_ARG, "neg_elcbo: bad row slice");
    if (opts->eps_mode == VBMC_EPS_RESIDENT &&
        (!ctx->d_eps || ctx->eps_K != K || ctx->eps_D != D || ctx->eps_n_half != n_half ||
         ctx->eps_row_begin != row_begin || ctx->eps_rows != row_count))
      return vbmc_fail(ctx, VBMC_E_ARG, "neg_elcbo: resident eps does not match the request");
  }

  // Results land in pinned host memory: the GP sums are written there directly by the
  // prep kernel and, on one GPU, so is the raw entropy vector by the finish kernel (no
  // D2H copy).  With a communicator the raw vector goes through device memory for the
  // all-reduce and is copied back afterwards.
  rc = ensure_pinned(ctx, n_res + (size_t)n_raw);
  if (rc) return rc;
  double* hp_dev = nullptr;  // device-side address of the pinned block
  HIP_TRY(ctx, hipHostGetDevicePointer((void**)&hp_dev, ctx->h_pinned, 0));
  double* res_out = hp_dev;
  double* raw_host = hp_dev + n_res;
  static const bool force_coll = [] {
    const char* e = getenv("VBMC_FORCE_COLLECTIVE");
    return e && e[0] == '1';
  }();
  const bool multi = ctx->comm != nullptr && (ctx->world > 1 || force_coll);
  double* raw_out = raw_host;
  if (multi && mc) {
    rc = ensure_dev(ctx, &ctx->d_out, &ctx->d_out_cap, (size_t)n_raw);
    if (rc) return rc;
    raw_out = ctx->d_out;
  }

  ctx->host_us[0] = us_since(t_begin);
  const auto t_launch = clk::now();
  PrepArgs pa;
  glj_fill_prep(ctx, grad_flags != 0, res_out, nullptr, pa);
  EntPlan plan;
  if (mc) {
    rc = entmc_plan(ctx, opts->ns_per_comp, opts->eps_mode, opts->seed, row_begin, row_count,
                    grad_flags != 0, plan);
    if (rc) return rc;
    entmc_fill_prep(ctx, plan, pa);
  }
  rc = launch_prep(ctx, pa);  // GP sums + (j,k) table rows, one launch
  if (rc) return rc;
  if (mc) {
    rc = entmc_launch_main(ctx, plan);
    if (rc) return rc;
    rc = entmc_launch_finish(ctx, plan, raw_out);
    if (rc) return rc;
    if (multi) {
      rc = comm_allreduce_sum(ctx, raw_out, n_raw);
      if (rc) return rc;
      HIP_TRY(ctx, hipMemcpyAsync(ctx->h_pinned + n_res, raw_out, sizeof(double) * n_raw,
                                  hipMemcpyDeviceToHost, ctx->stream));
    }
  } else if (lb_dev) {
    rc = launch_entlb(ctx, raw_host);
    if (rc) return rc;
  }
  ctx->host_us[1] = us_since(t_launch);
  const auto t_wait = clk::now();
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ctx->pack_in_flight = false;
  ctx->host_us[2] = us_since(t_wait);
  const auto t_fin = clk::now();

  // ---- host finalisation ---------------------------------------------------
  GljHost o;
  glj_finalize(ctx, ctx->h_pinned, grad_flags != 0, o);
  double Gv = 0.0;
  for (int s = 0; s < S; ++s) Gv += o.G[s];
  Gv /= S;
  std::vector<double> dGv((size_t)n_theta, 0.0), dHv((size_t)n_theta, 0.0);
  if (grad_flags) {
    // average the per-sample blocks (linear), then Jacobians
    std::vector<double> mu((size_t)K * D, 0.0), sg(K, 0.0), lm(D, 0.0), wg(K, 0.0);
    for (int s = 0; s < S; ++s) {
      for (int i = 0; i < K * D; ++i) mu[i] += o.mu[(size_t)s * K * D + i] / S;
      for (int k = 0; k < K; ++k) sg[k] += o.sigma[(size_t)s * K + k] / S;
      for (int d = 0; d < D; ++d) lm[d] += o.lambd[(size_t)s * D + d] / S;
      for (int k = 0; k < K; ++k) wg[k] += o.w[(size_t)s * K + k] / S;
    }
    const int n = glj_pack(ctx, mu.data(), sg.data(), lm.data(), wg.data(), grad_flags, 1, dGv.data());
    if (n != n_theta) return vbmc_fail(ctx, VBMC_E_ARG, "neg_elcbo: gradient length %d != n_theta %d", n, n_theta);
  }
  double Hv = 0.0;
  if (mc || lb_dev) {
    const double* r = ctx->h_pinned + n_res;
    entropy_pack(ctx, r[0], r + 1, r + 1 + D * K, r + 1 + D * K + K, r + 1 + D * K + K + D,
                 grad_flags, 1, &Hv, grad_flags ? dHv.data() : nullptr);
  } else {
    rc = vbmc_entlb(ctx, grad_flags, 1, &Hv, grad_flags ? dHv.data() : nullptr);  // K == 1 closed form
    if (rc) return rc;
  }
  double Fv = -Gv - Hv;
  std::vector<double> dFv((size_t)n_theta, 0.0);
  if (grad_flags)
    for (int i = 0; i < n_theta; ++i) dFv[i] = -dGv[i] - dHv[i];

  // ---- soft bounds and weight penalty (:1195-1229, _vp_bound_loss :537-606) ----
  if (opts->bnd_lb && opts->bnd_ub) {
    std::vector<double> ext;
    int pos = 0;
    std::vector<double> ln_sigma(K), ln_lambd(D);
    if (o_mu) {
      ext.insert(ext.end(), theta, theta + D * K);
      pos = D * K;
    }
    if (o_sg) {
      for (int k = 0; k < K; ++k) ln_sigma[k] = theta[pos + k];
      pos += K;
    } else {
      for (int k = 0; k < K; ++k) ln_sigma[k] = std::log(ctx->sigma[k]);
    }
    if (o_lm) {
      for (int d = 0; d < D; ++d) ln_lambd[d] = theta[pos + d];
    } else {
      for (int d = 0; d < D; ++d) ln_lambd[d] = std::log(ctx->lambd[d]);
    }
    const int sc0 = (int)ext.size();
    if (o_sg || o_lm)
      for (int k = 0; k < K; ++k)
        for (int d = 0; d < D; ++d) ext.push_back(ln_lambd[d] + ln_sigma[k]);  // ravel('F')
    if (o_w) ext.insert(ext.end(), theta + (n_theta - K), theta + n_theta);
    if ((int)ext.size() != opts->n_bnd)
      return vbmc_fail(ctx, VBMC_E_ARG, "neg_elcbo: bounds length %d != %d", opts->n_bnd,
                       (int)ext.size());
    std::vector<double> dL;
    const double L = soft_bound_loss(ext, opts->bnd_lb, opts->bnd_ub, opts->tol_con,
                                     grad_flags ? &dL : nullptr);
    Fv += L;
    if (grad_flags) {
      int q = 0;
      if (o_mu) {
        for (int i = 0; i < D * K; ++i) dFv[q + i] += dL[i];
        q += D * K;
      }
      if (o_sg || o_lm) {
        // the reference reshapes this block C-order (D,K) (:585-587); restated as-is
        if (o_sg) {
          for (int k = 0; k < K; ++k) {
            double a = 0.0;
            for (int d = 0; d < D; ++d) a += dL[sc0 + d * K + k];
            dFv[q + k] += a;
          }
          q += K;
        }
        if (o_lm) {
          for (int d = 0; d < D; ++d) {
            double a = 0.0;
            for (int k = 0; k < K; ++k) a += dL[sc0 + d * K + k];
            dFv[q + d] += a;
          }
          q += D;
        }
      }
      if (o_w)
        for (int k = 0; k < K; ++k) dFv[q + k] += dL[dL.size() - K + k];
    }
    if (o_w) {
      const double th = opts->weight_threshold, pen = opts->weight_penalty;
      double a = 0.0;
      std::vector<double> wg(K);
      for (int k = 0; k < K; ++k) {
        const bool small = ctx->w[k] < th;
        a += small ? ctx->w[k] : th;
        wg[k] = small ? pen : 0.0;
      }
      Fv += a * pen;
      if (grad_flags) {
        std::vector<double> jw(K);
        softmax_jacobian_apply(ctx->eta, wg.data(), jw.data());
        for (int k = 0; k < K; ++k) dFv[n_theta - K + k] += jw[k];
      }
    }
  }
  ctx->host_us[3] = us_since(t_fin);
  ctx->host_us[4] = us_since(t_begin);
  if (F) *F = Fv;
  if (G) *G = Gv;
  if (H) *H = Hv;
  if (dF && grad_flags) memcpy(dF, dFv.data(), sizeof(double) * n_theta);
  return VBMC_OK;
}

extern "C" int vbmc_last_host_us(const vbmc_ctx* ctx, double out[5]) {
  if (!ctx || !out) return VBMC_E_ARG;
  for (int i = 0; i < 5; ++i) out[i] = ctx->host_us[i];
  return VBMC_OK;
}
