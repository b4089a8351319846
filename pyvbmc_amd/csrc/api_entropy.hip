// C-ABI entry points of the entropy estimators + their host-side finalisation
// (Jacobians and packing: entropy/entmc_vbmc.py:114-132, entlb_vbmc.py:161-179).
#include <cmath>
#include <cstring>

#include "common.h"

// w_grad <- J_w @ w_grad, J_w = -ee ee^T / s^2 + diag(ee)/s, ee = exp(eta)
// (entmc_vbmc.py:122-130).  exp(eta) is cached in the context until eta changes: one ELBO
// evaluation applies this Jacobian three times (GP term, entropy, weight penalty).
void softmax_jacobian_apply(vbmc_ctx* ctx, const double* g, double* out) {
  const int K = (int)ctx->eta.size();
  if (!ctx->exp_eta_valid || (int)ctx->exp_eta.size() != K) {
    ctx->exp_eta.resize(K);
    double s = 0.0;
    for (int k = 0; k < K; ++k) {
      ctx->exp_eta[k] = std::exp(ctx->eta[k]);
      s += ctx->exp_eta[k];
    }
    ctx->exp_eta_sum = s;
    ctx->exp_eta_valid = true;
  }
  const double* ee = ctx->exp_eta.data();
  const double s = ctx->exp_eta_sum;
  double dot = 0.0;
  for (int k = 0; k < K; ++k) dot += ee[k] * g[k];
  for (int k = 0; k < K; ++k) out[k] = -ee[k] * dot / (s * s) + ee[k] * g[k] / s;
}

int entropy_pack(vbmc_ctx* ctx, double H, const double* mu, const double* sg,
                 const double* lm, const double* wg, int grad_flags, int jacobian_flag,
                 double* H_out, double* dH_out) {
  const int D = ctx->D, K = ctx->K;
  if (H_out) *H_out = H;
  if (!dH_out) return 0;
  int pos = 0;
  if (grad_flags & 1) {
    memcpy(dH_out + pos, mu, sizeof(double) * D * K);
    pos += D * K;
  }
  if (grad_flags & 2) {
    for (int k = 0; k < K; ++k) dH_out[pos + k] = jacobian_flag ? sg[k] * ctx->sigma[k] : sg[k];
    pos += K;
  }
  if (grad_flags & 4) {
    for (int d = 0; d < D; ++d) dH_out[pos + d] = jacobian_flag ? lm[d] * ctx->lambd[d] : lm[d];
    pos += D;
  }
  if (grad_flags & 8) {
    if (jacobian_flag)
      softmax_jacobian_apply(ctx, wg, dH_out + pos);
    else
      memcpy(dH_out + pos, wg, sizeof(double) * K);
    pos += K;
  }
  return pos;
}

extern "C" {

int vbmc_entmc_finalize(vbmc_ctx* ctx, const double* raw, int grad_flags, int jacobian_flag,
                        double* H, double* dH) {
  if (!ctx || !raw) return VBMC_E_ARG;
  if (!ctx->mix_set) return vbmc_fail(ctx, VBMC_E_ARG, "entmc_finalize: mixture not set");
  const int D = ctx->D, K = ctx->K;
  entropy_pack(ctx, raw[0], raw + 1, raw + 1 + D * K, raw + 1 + D * K + K,
               raw + 1 + D * K + K + D, grad_flags, jacobian_flag, H, dH);
  return VBMC_OK;
}

int vbmc_entmc(vbmc_ctx* ctx, int64_t ns_per_comp, int eps_mode, uint64_t seed, int64_t row_begin,
               int64_t row_count, int grad_flags, int jacobian_flag, double* H, double* dH,
               double* raw_out) {
  if (!ctx) return VBMC_E_ARG;
  if (!ctx->mix_set) return vbmc_fail(ctx, VBMC_E_ARG, "entmc: mixture not set");
  NEED_DEVICE(ctx);
  if (ns_per_comp < 2 || (ns_per_comp & 1))
    return vbmc_fail(ctx, VBMC_E_ARG, "entmc: ns_per_comp=%lld must be even and >= 2",
                     (long long)ns_per_comp);
  const int64_t n_half = ns_per_comp / 2;
  if (row_begin < 0 || row_count < 0 || row_begin + row_count > n_half)
    return vbmc_fail(ctx, VBMC_E_ARG, "entmc: rows [%lld,+%lld) outside [0,%lld)",
                     (long long)row_begin, (long long)row_count, (long long)n_half);
  if (eps_mode == VBMC_EPS_RESIDENT) {
    if (!ctx->d_eps || ctx->eps_K != ctx->K || ctx->eps_D != ctx->D || ctx->eps_n_half != n_half ||
        ctx->eps_row_begin != row_begin || ctx->eps_rows != row_count)
      return vbmc_fail(ctx, VBMC_E_ARG,
                       "entmc: resident eps (K=%d D=%d n_half=%lld rows [%lld,+%lld)) does not match "
                       "the request (K=%d D=%d n_half=%lld rows [%lld,+%lld)); call vbmc_set_eps",
                       ctx->eps_K, ctx->eps_D, (long long)ctx->eps_n_half,
                       (long long)ctx->eps_row_begin, (long long)ctx->eps_rows, ctx->K, ctx->D,
                       (long long)n_half, (long long)row_begin, (long long)row_count);
  } else if (eps_mode != VBMC_EPS_PHILOX) {
    return vbmc_fail(ctx, VBMC_E_ARG, "entmc: unknown eps_mode %d", eps_mode);
  }
  NEED_DEVICE(ctx);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int D = ctx->D, K = ctx->K;
  const int n = raw_len(D, K);
  int rc = ensure_dev(ctx, &ctx->d_out, &ctx->d_out_cap, (size_t)n);
  if (rc) return rc;
  rc = ensure_pinned(ctx, (size_t)n);
  if (rc) return rc;
  rc = launch_entmc(ctx, ns_per_comp, eps_mode, seed, row_begin, row_count, grad_flags != 0,
                    ctx->d_out);
  if (rc) return rc;
  if (ctx->comm) {
    rc = comm_allreduce_sum(ctx, ctx->d_out, n);
    if (rc) return rc;
  }
  HIP_TRY(ctx, hipMemcpyAsync(ctx->h_pinned, ctx->d_out, sizeof(double) * n, hipMemcpyDeviceToHost,
                              ctx->stream));
  HIP_TRY(ctx, stream_wait(ctx));
  if (raw_out) memcpy(raw_out, ctx->h_pinned, sizeof(double) * n);
  return vbmc_entmc_finalize(ctx, ctx->h_pinned, grad_flags, jacobian_flag, H, dH);
}

int vbmc_philox_normals(vbmc_ctx* ctx, int K, int64_t n_half, int D, uint64_t seed, int64_t row_begin,
                        int64_t row_count, double* out) {
  if (!ctx || !out || K < 1 || D < 1 || n_half < 1) return VBMC_E_ARG;
  NEED_DEVICE(ctx);
  if (row_begin < 0 || row_count < 0 || row_begin + row_count > n_half)
    return vbmc_fail(ctx, VBMC_E_ARG, "philox_normals: rows [%lld,+%lld) outside [0,%lld)", (long long)row_begin,
                     (long long)row_count, (long long)n_half);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t n = (size_t)K * (size_t)row_count * (size_t)D;
  if (n == 0) return VBMC_OK;
  HIP_TRY(ctx, stream_wait(ctx));  // (the scratch buffer may be in use by queued launches)
  int rc = ensure_dev(ctx, &ctx->d_scratch, &ctx->d_scratch_cap, n);
  if (rc) return rc;
  const GenSlice g = make_gen_slice(ctx->d_scratch, K, D, row_count, n_half, row_begin, seed, nullptr, 0.0, 1.0);
  rc = launch_eps_gen(ctx, ctx->stream, g);
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(out, ctx->d_scratch, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, stream_wait(ctx));
  return VBMC_OK;
}

int vbmc_entlb(vbmc_ctx* ctx, int grad_flags, int jacobian_flag, double* H, double* dH) {
  if (!ctx) return VBMC_E_ARG;
  if (!ctx->mix_set) return vbmc_fail(ctx, VBMC_E_ARG, "entlb: mixture not set");
  const int D = ctx->D, K = ctx->K;
  if (K == 1) {
    // exact entropy of one Gaussian (entlb_vbmc.py:60-78)
    double Hh = 0.5 * D * (1.0 + std::log(2.0 * M_PI)) + D * std::log(ctx->sigma[0]);
    for (int d = 0; d < D; ++d) Hh += std::log(ctx->lambd[d]);
    std::vector<double> mu((size_t)D, 0.0), lm((size_t)D);
    double sg = D / ctx->sigma[0], wg = 0.0;
    for (int d = 0; d < D; ++d) lm[d] = 1.0 / ctx->lambd[d];
    entropy_pack(ctx, Hh, mu.data(), &sg, lm.data(), &wg, grad_flags, jacobian_flag, H, dH);
    return VBMC_OK;
  }
  NEED_DEVICE(ctx);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int n = raw_len(D, K);
  int rc = ensure_dev(ctx, &ctx->d_out, &ctx->d_out_cap, (size_t)n);
  if (rc) return rc;
  rc = ensure_pinned(ctx, (size_t)n);
  if (rc) return rc;
  rc = launch_entlb(ctx, ctx->d_out);
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(ctx->h_pinned, ctx->d_out, sizeof(double) * n, hipMemcpyDeviceToHost,
                              ctx->stream));
  HIP_TRY(ctx, stream_wait(ctx));
  const double* r = ctx->h_pinned;
  entropy_pack(ctx, r[0], r + 1, r + 1 + D * K, r + 1 + D * K + K, r + 1 + D * K + K + D,
               grad_flags, jacobian_flag, H, dH);
  return VBMC_OK;
}

}  // extern "C"
