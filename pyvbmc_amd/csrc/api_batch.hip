// Batched objective for the sieve: B candidate parameter vectors evaluated in one go.
//
// Reference: vbmc/variational_optimization.py:775-787 (_sieve) calls
//   _neg_elcbo(theta_b, gp, vp0, 0, ns_ent_K_fast = 0, compute_grad = False, theta_bnd)
// once per candidate (up to ceil(50 K) of them per VBMC iteration) -- every call tiny and
// launch-bound.  Here all candidates share ONE upload, ONE GP-sums launch (grid.y = B),
// ONE lower-bound-entropy launch (a workgroup per candidate) and one read-back.
#include <cmath>
#include <cstring>

#include "common.h"
#include "fastmath.h"

namespace {

__device__ __forceinline__ double wave_sum(double v) {
  return fm::wave_sum_dpp(v);
}

// H_b = -sum_i w_i log( sum_j w_j gamma_ij )   (entlb_vbmc.py:84-97), value only.
// One workgroup per candidate; wave w handles rows i = w, w+4, ...; lanes run over j.
__global__ __launch_bounds__(256) void entlb_value_batch_kernel(const double* __restrict__ packs,
                                                                MixLayout ml, size_t stride,
                                                                double* __restrict__ H) {
  __shared__ double sPart[4];
  const int D = ml.D, K = ml.K;
  const double* mix = packs + (size_t)blockIdx.x * stride;
  const double* mup = mix + ml.o_mup;
  const double* sig = mix + ml.o_sig;
  const double* w = mix + ml.o_w;
  const double* lam = mix + ml.o_lam;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double lnc = -0.5 * D * log(2.0 * M_PI);
  for (int d = 0; d < D; ++d) lnc -= log(lam[d]);
  double hacc = 0.0;
  for (int i = wave; i < K; i += 4) {
    double acc = 0.0;
    for (int j = lane; j < K; j += 64) {
      const double s2 = sig[i] * sig[i] + sig[j] * sig[j];
      double d2 = 0.0;
      for (int d = 0; d < D; ++d) {
        const double t = mup[i * D + d] - mup[j * D + d];
        d2 = fma(t, t, d2);
      }
      acc += w[j] * exp(lnc - 0.5 * D * log(s2) - 0.5 * d2 / s2);
    }
    acc = wave_sum(acc);
    hacc -= w[i] * log(acc);
  }
  if (lane == 0) sPart[wave] = hacc;
  __syncthreads();
  if (threadIdx.x == 0) H[blockIdx.x] = (sPart[0] + sPart[1]) + (sPart[2] + sPart[3]);
}

}  // namespace

extern "C" int vbmc_neg_elcbo_batch(vbmc_ctx* ctx, const double* thetas_BxN, int B, int n_theta,
                                    const vbmc_elbo_opts* opts, double* F_B, double* G_B,
                                    double* H_B) {
  if (!ctx || !thetas_BxN || !opts || !F_B || B < 0) return VBMC_E_ARG;
  NEED_DEVICE(ctx);
  if (!ctx->mix_set) return vbmc_fail(ctx, VBMC_E_ARG, "neg_elcbo_batch: mixture (D,K) not set");
  if (!ctx->gp.set) return vbmc_fail(ctx, VBMC_E_ARG, "neg_elcbo_batch: GP not set");
  if (ctx->gp.D != ctx->D) return vbmc_fail(ctx, VBMC_E_ARG, "neg_elcbo_batch: GP/mixture D mismatch");
  if (opts->ns_per_comp != 0 || opts->compute_grad)
    return vbmc_fail(ctx, VBMC_E_UNSUP,
                     "neg_elcbo_batch covers the sieve call only: Ns = 0 (entlb), no gradient");
  if (B == 0) return VBMC_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int D = ctx->D, K = ctx->K, S = ctx->gp.S;
  const int mask = opts->optimize_mask;
  const bool o_mu = mask & 1, o_sg = mask & 2, o_lm = mask & 4, o_w = mask & 8;
  MixLayout ml;
  ml.plan(D, K);
  const size_t stride = (size_t)ml.total;

  // ---- host: theta_b -> mixture_b -> pack_b -------------------------------------------------
  std::vector<double> packs(stride * B);
  std::vector<double> mus((size_t)B * K * D), sgs((size_t)B * K), lms((size_t)B * D), ws((size_t)B * K),
      etas((size_t)B * K);
  for (int b = 0; b < B; ++b) {
    double* mu = mus.data() + (size_t)b * K * D;
    double* sg = sgs.data() + (size_t)b * K;
    double* lm = lms.data() + (size_t)b * D;
    double* w = ws.data() + (size_t)b * K;
    double* eta = etas.data() + (size_t)b * K;
    memcpy(mu, ctx->mu.data(), sizeof(double) * K * D);
    memcpy(sg, ctx->sigma.data(), sizeof(double) * K);
    memcpy(lm, ctx->lambd.data(), sizeof(double) * D);
    memcpy(w, ctx->w.data(), sizeof(double) * K);
    memcpy(eta, ctx->eta.data(), sizeof(double) * K);
    const int st = theta_to_arrays(D, K, thetas_BxN + (size_t)b * n_theta, n_theta, mask, mu, sg, lm, w, eta);
    if (st == -1) return vbmc_fail(ctx, VBMC_E_ARG, "neg_elcbo_batch: theta length %d does not match", n_theta);
    if (st == -2) return vbmc_fail(ctx, VBMC_E_NONFINITE, "neg_elcbo_batch: candidate %d is not finite", b);
    write_mixture_pack(ml, mu, sg, lm, w, packs.data() + stride * b);
  }

  // ---- device ------------------------------------------------------------------------------
  const size_t n_res = (size_t)B * S * K;  // value only: one sum per (b, s, k)
  const size_t n_H = (K > 1) ? (size_t)B : 0;
  int rc = ensure_dev(ctx, &ctx->d_scratch, &ctx->d_scratch_cap, stride * B + n_res * (1 + 2 * D) + n_H);
  if (rc) return rc;
  rc = ensure_pinned(ctx, n_res * (1 + 2 * D) + n_H);
  if (rc) return rc;
  double* d_packs = ctx->d_scratch;
  double* d_res = d_packs + stride * B;
  double* d_H = d_res + n_res * (1 + 2 * D);
  HIP_TRY(ctx, hipMemcpyAsync(d_packs, packs.data(), sizeof(double) * stride * B, hipMemcpyHostToDevice,
                              ctx->stream));
  PrepArgs pa;
  glj_fill_prep(ctx, 0, d_res, nullptr, pa);
  pa.mix = d_packs;
  pa.ml = ml;
  pa.batch = B;
  pa.mix_stride = stride;
  pa.res_stride = (size_t)S * K * (1 + 2 * D);
  rc = launch_prep(ctx, pa);
  if (rc) return rc;
  if (K > 1) {
    hipLaunchKernelGGL(entlb_value_batch_kernel, dim3(B), dim3(256), 0, ctx->stream,
                       (const double*)d_packs, ml, stride, d_H);
    HIP_TRY(ctx, hipGetLastError());
  }
  HIP_TRY(ctx, hipMemcpyAsync(ctx->h_pinned, d_res, sizeof(double) * (n_res * (1 + 2 * D) + n_H),
                              hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, stream_wait(ctx));

  // ---- host finalisation per candidate (glj_finalize's value part + bound losses) ---------------
  const GpState& g = ctx->gp;
  const double* res = ctx->h_pinned;
  const double* Hd = ctx->h_pinned + n_res * (1 + 2 * D);
  const int st1 = 1 + 2 * D;
  std::vector<double> iom2(D), xm(D);
  for (int b = 0; b < B; ++b) {
    const double* mu = mus.data() + (size_t)b * K * D;
    const double* sg = sgs.data() + (size_t)b * K;
    const double* lm = lms.data() + (size_t)b * D;
    const double* w = ws.data() + (size_t)b * K;
    double Gv = 0.0;
    for (int s = 0; s < S; ++s) {
      const double* h = g.hyp.data() + (size_t)s * g.P;
      const bool quad = g.mean_kind == VBMC_MEAN_NEGQUAD;
      const double m0 = g.mean_kind == VBMC_MEAN_ZERO ? 0.0 : h[D + 2];
      if (quad)
        for (int d = 0; d < D; ++d) {
          xm[d] = h[D + 3 + d];
          iom2[d] = std::exp(-2.0 * h[2 * D + 3 + d]);
        }
      for (int k = 0; k < K; ++k) {
        double I_k = res[((size_t)b * S * K + (size_t)s * K + k) * st1] + m0;
        if (quad) {
          double nu = 0.0;
          for (int d = 0; d < D; ++d) {
            const double m = mu[(size_t)k * D + d];
            nu += iom2[d] * (m * m + sg[k] * sg[k] * lm[d] * lm[d] - 2.0 * m * xm[d] + xm[d] * xm[d]);
          }
          I_k += -0.5 * nu;
        }
        Gv += w[k] * I_k;
      }
    }
    Gv /= S;
    double Hv;
    if (K > 1) {
      Hv = Hd[b];
    } else {
      Hv = 0.5 * D * (1.0 + std::log(2.0 * M_PI)) + D * std::log(sg[0]);
      for (int d = 0; d < D; ++d) Hv += std::log(lm[d]);
    }
    double Fv = -Gv - Hv;
    if (opts->bnd_lb && opts->bnd_ub) {
      // soft bounds on (mu, ln sigma + ln lambda, eta) with the max-shifted eta tail, as
      // _neg_elcbo sees them (:1082-1085, :1195-1229)
      const double* th = thetas_BxN + (size_t)b * n_theta;
      int pos = 0, q = 0;
      double L = 0.0;
      auto pen = [&](double x, int i) {
        const double lb = opts->bnd_lb[i], ub = opts->bnd_ub[i];
        const double ell = (ub - lb) * opts->tol_con;
        if (x < lb) L += 0.5 * ((lb - x) / ell) * ((lb - x) / ell);
        if (x > ub) L += 0.5 * ((x - ub) / ell) * ((x - ub) / ell);
      };
      const int n_ext = (o_mu ? D * K : 0) + ((o_sg || o_lm) ? D * K : 0) + (o_w ? K : 0);
      if (n_ext != opts->n_bnd)
        return vbmc_fail(ctx, VBMC_E_ARG, "neg_elcbo_batch: bounds length %d != %d", opts->n_bnd, n_ext);
      if (o_mu) {
        for (int i = 0; i < D * K; ++i) pen(th[i], q++);
        pos = D * K;
      }
      if (o_sg || o_lm) {
        const double* ls = o_sg ? th + pos : nullptr;
        if (o_sg) pos += K;
        const double* ll = o_lm ? th + pos : nullptr;
        for (int k = 0; k < K; ++k)
          for (int d = 0; d < D; ++d)
            pen((ll ? ll[d] : std::log(ctx->lambd[d])) + (ls ? ls[k] : std::log(ctx->sigma[k])), q++);
      }
      if (o_w) {
        const double* e = th + (n_theta - K);
        double mx = e[0];
        for (int k = 1; k < K; ++k) mx = e[k] > mx ? e[k] : mx;
        for (int k = 0; k < K; ++k) pen(e[k] - mx, q++);
        double a = 0.0;
        for (int k = 0; k < K; ++k) a += (w[k] < opts->weight_threshold) ? w[k] : opts->weight_threshold;
        L += a * opts->weight_penalty;
      }
      Fv += L;
    }
    F_B[b] = Fv;
    if (G_B) G_B[b] = Gv;
    if (H_B) H_B[b] = Hv;
  }
  return VBMC_OK;
}
