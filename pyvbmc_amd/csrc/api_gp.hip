// C-ABI entry points for the GP side: state upload, _gp_log_joint, predict.
// Host finalisation follows vbmc/variational_optimization.py:1374-1606.
#include <cmath>
#include <cstring>

#include "common.h"

static const double kTiny = 2.220446049250313e-16;  // np.spacing(1)

extern "C" int vbmc_set_gp(vbmc_ctx* ctx, int N, int D, int S, int P, int mean_kind,
                           const double* X_NxD, const double* hyp_SxP, const double* alpha_SxN,
                           const double* L_SxNxN, const int32_t* L_chol_S, const double* sW_SxN,
                           const double* sn2_mult_S) {
  if (!ctx || !X_NxD || !hyp_SxP || !alpha_SxN || !L_SxNxN || !L_chol_S || !sW_SxN)
    return VBMC_E_ARG;
  if (N < 1 || D < 1 || S < 1) return vbmc_fail(ctx, VBMC_E_ARG, "set_gp: bad N=%d D=%d S=%d", N, D, S);
  const int mean_n = mean_kind == VBMC_MEAN_ZERO ? 0 : mean_kind == VBMC_MEAN_CONST ? 1 : 1 + 2 * D;
  if (mean_kind < 0 || mean_kind > 2 || P != D + 2 + mean_n)
    return vbmc_fail(ctx, VBMC_E_ARG, "set_gp: P=%d does not match D+2+mean(%d)=%d", P, mean_kind,
                     D + 2 + mean_n);
  NEED_DEVICE(ctx);
  ctx->gp_watch_ptrs.clear();  // (vbmc_set_gp_watch: the caller sets it again for the arrays of THIS upload)
  ctx->gp_watch_lens.clear();
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, stream_wait(ctx));
  GpState& g = ctx->gp;
  g.set = false;
  // device buffers are kept across GP updates and only grown (active sampling updates the GP after
  // every new point: N creeps up by one)
  const size_t nn = (size_t)N * N;
  auto grow = [&](double** p, size_t* cap, size_t n) -> int {
    if (*p && *cap >= n) return 0;
    if (*p) HIP_TRY(ctx, hipFree(*p));
    *p = nullptr;
    const size_t want = n + n / 8 + 64;
    HIP_TRY(ctx, hipMalloc((void**)p, sizeof(double) * want));
    *cap = want;
    return 0;
  };
  int rc;
  if ((rc = grow(&g.d_X, &g.cap_X, (size_t)N * D)) || (rc = grow(&g.d_XT, &g.cap_XT, (size_t)N * D)) ||
      (rc = grow(&g.d_alpha, &g.cap_alpha, (size_t)S * N)) ||
      (rc = grow(&g.d_L, &g.cap_L, (size_t)S * nn)) || (rc = grow(&g.d_Linv, &g.cap_Linv, (size_t)S * nn)) ||
      (rc = grow(&g.d_LinvP, &g.cap_LinvP, (size_t)S * predict_ld(N) * predict_ld(N))) ||
      (rc = grow(&g.d_sW, &g.cap_sW, (size_t)S * N)) || (rc = grow(&g.d_hyp, &g.cap_hyp, (size_t)S * P)) ||
      (rc = grow(&g.d_xc, &g.cap_xc, (size_t)D)) || (rc = grow(&g.d_smeta, &g.cap_smeta, (size_t)3 * S)))
    return rc;
  g.N = N; g.D = D; g.S = S; g.P = P; g.mean_kind = mean_kind;
  g.hyp.assign(hyp_SxP, hyp_SxP + (size_t)S * P);
  g.L_chol.assign(L_chol_S, L_chol_S + S);
  g.sn2_eff.resize(S);
  g.sn2_mult.resize(S);
  for (int s = 0; s < S; ++s) {
    const double sw0 = sW_SxN[(size_t)s * N];
    g.sn2_eff[s] = 1.0 / (sw0 * sw0);  // variational_optimization.py:1398
    g.sn2_mult[s] = sn2_mult_S ? sn2_mult_S[s] : 1.0;
  }
  // centre of the |a|^2 + |b|^2 - 2 a.b expansion in predict (cf. _sq_dist's mean shift,
  // acquisition_functions/abstract_acq_fcn.py:212-217)
  g.h_small.assign((size_t)D + 3 * (size_t)S, 0.0);
  double* xc = g.h_small.data();
  for (int n = 0; n < N; ++n)
    for (int d = 0; d < D; ++d) xc[d] += X_NxD[(size_t)n * D + d];
  for (int d = 0; d < D; ++d) xc[d] /= N;
  double* smeta = xc + D;
  for (int s = 0; s < S; ++s) {
    smeta[3 * s] = g.L_chol[s] ? 1.0 : 0.0;
    smeta[3 * s + 1] = g.sn2_mult[s];
    smeta[3 * s + 2] = 1.0 / g.sn2_eff[s];
  }
  // (the caller's arrays are pageable: these copies return once the source has been staged, and
  // everything queued here is waited for below before the caller's memory can change)
  HIP_TRY(ctx, hipMemcpyAsync(g.d_smeta, smeta, sizeof(double) * 3 * S, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(g.d_L, L_SxNxN, sizeof(double) * S * nn, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(g.d_X, X_NxD, sizeof(double) * N * D, hipMemcpyHostToDevice, ctx->stream));
  g.h_XT.resize((size_t)N * D);
  for (int n = 0; n < N; ++n)
    for (int d = 0; d < D; ++d) g.h_XT[(size_t)d * N + n] = X_NxD[(size_t)n * D + d];
  HIP_TRY(ctx, hipMemcpyAsync(g.d_XT, g.h_XT.data(), sizeof(double) * N * D, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(g.d_alpha, alpha_SxN, sizeof(double) * S * N, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(g.d_sW, sW_SxN, sizeof(double) * S * N, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(g.d_hyp, hyp_SxP, sizeof(double) * S * P, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(g.d_xc, xc, sizeof(double) * D, hipMemcpyHostToDevice, ctx->stream));
  // L^-1 of the Cholesky samples, once per GP update
  rc = launch_trinv(ctx);
  if (rc) return rc;
  HIP_TRY(ctx, stream_wait(ctx));
  g.set = true;
  return VBMC_OK;
}

void glj_finalize(const vbmc_ctx* ctx, const double* res, int want_grad, GljHost& o) {
  const GpState& g = ctx->gp;
  const int D = ctx->D, K = ctx->K, S = g.S;
  const int st = 1 + 2 * D;
  o.G.assign(S, 0.0);
  o.I_sk.assign((size_t)S * K, 0.0);
  o.mu.assign((size_t)S * K * D, 0.0);
  o.sigma.assign((size_t)S * K, 0.0);
  o.lambd.assign((size_t)S * D, 0.0);
  o.w.assign((size_t)S * K, 0.0);
  std::vector<double> ell2(D), xm(D, 0.0), iom2(D, 0.0);
  for (int s = 0; s < S; ++s) {
    const double* h = g.hyp.data() + (size_t)s * g.P;
    for (int d = 0; d < D; ++d) ell2[d] = std::exp(2.0 * h[d]);
    const bool quad = g.mean_kind == VBMC_MEAN_NEGQUAD;
    const double m0 = g.mean_kind == VBMC_MEAN_ZERO ? 0.0 : h[D + 2];
    if (quad)
      for (int d = 0; d < D; ++d) {
        xm[d] = h[D + 3 + d];
        iom2[d] = std::exp(-2.0 * h[2 * D + 3 + d]);
      }
    for (int k = 0; k < K; ++k) {
      const double* r = res + ((size_t)s * K + k) * st;
      const double sg = ctx->sigma[k], wk = ctx->w[k];
      double I_k = r[0] + m0;
      if (quad) {
        double nu = 0.0;
        for (int d = 0; d < D; ++d) {
          const double m = ctx->mu[(size_t)k * D + d], lam = ctx->lambd[d];
          nu += iom2[d] * (m * m + sg * sg * lam * lam - 2.0 * m * xm[d] + xm[d] * xm[d]);
        }
        I_k += -0.5 * nu;
      }
      o.G[s] += wk * I_k;
      o.I_sk[(size_t)s * K + k] = I_k;
      o.w[(size_t)s * K + k] = I_k;
      if (!want_grad) continue;
      double gs = 0.0;
      for (int d = 0; d < D; ++d) {
        const double lam = ctx->lambd[d];
        const double tau2 = sg * sg * lam * lam + ell2[d];
        const double itau2 = 1.0 / tau2;            // the only division of this (k, d)
        const double itau = std::sqrt(tau2) * itau2;  // 1 / tau
        const double U = r[1 + d], T = r[1 + D + d] - r[0];
        double gm = wk * (-U * itau);
        double gl = wk * (sg * sg * itau2) * lam * T;
        gs += (lam * lam * itau2) * T;
        if (quad) {
          gm -= wk * iom2[d] * (ctx->mu[(size_t)k * D + d] - xm[d]);
          gl -= wk * sg * sg * iom2[d] * lam;
        }
        o.mu[((size_t)s * K + k) * D + d] = gm;
        o.lambd[(size_t)s * D + d] += gl;
      }
      double gsig = wk * sg * gs;
      if (quad) {
        double q = 0.0;
        for (int d = 0; d < D; ++d) q += iom2[d] * ctx->lambd[d] * ctx->lambd[d];
        gsig -= wk * sg * q;
      }
      o.sigma[(size_t)s * K + k] = gsig;
    }
  }
}

// dG blocks: mu always; sigma/lambda/w only under jacobian_flag (:1528-1546).
int glj_pack(vbmc_ctx* ctx, const double* mu, const double* sg, const double* lm,
             const double* wg, int grad_flags, int jacobian_flag, double* out) {
  const int D = ctx->D, K = ctx->K;
  int pos = 0;
  if (grad_flags & 1) {
    if (out) memcpy(out, mu, sizeof(double) * D * K);
    pos += D * K;
  }
  if (jacobian_flag && (grad_flags & 2)) {
    if (out)
      for (int k = 0; k < K; ++k) out[pos + k] = sg[k] * ctx->sigma[k];
    pos += K;
  }
  if (jacobian_flag && (grad_flags & 4)) {
    if (out)
      for (int d = 0; d < D; ++d) out[pos + d] = lm[d] * ctx->lambd[d];
    pos += D;
  }
  if (jacobian_flag && (grad_flags & 8)) {
    if (out) softmax_jacobian_apply(ctx, wg, out + pos);
    pos += K;
  }
  return pos;
}

extern "C" int vbmc_gp_log_joint(vbmc_ctx* ctx, int grad_flags, int avg_flag, int jacobian_flag,
                                 int compute_var, double* G, double* dG, double* varG,
                                 double* var_ss, double* I_SxK, double* J_SxKxK) {
  if (!ctx) return VBMC_E_ARG;
  NEED_DEVICE(ctx);
  if (!ctx->mix_set) return vbmc_fail(ctx, VBMC_E_ARG, "gp_log_joint: mixture not set");
  if (!ctx->gp.set) return vbmc_fail(ctx, VBMC_E_ARG, "gp_log_joint: GP not set");
  if (ctx->gp.D != ctx->D)
    return vbmc_fail(ctx, VBMC_E_ARG, "gp_log_joint: GP D=%d != mixture D=%d", ctx->gp.D, ctx->D);
  if (compute_var == 2)
    return vbmc_fail(ctx, VBMC_E_UNSUP,
                     "Diagonal approximation of GP log-joint variance not implemented.");
  if (compute_var && grad_flags)
    return vbmc_fail(ctx, VBMC_E_UNSUP,
                     "Computation of gradient of log joint variance is currently available only "
                     "for diagonal approximation of the variance.");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const GpState& g = ctx->gp;
  const int D = ctx->D, K = ctx->K, S = g.S, N = g.N;
  const int st = 1 + 2 * D;
  const size_t n_res = (size_t)S * K * st;
  const size_t n_Z = compute_var ? (size_t)S * K * N : 0;
  const size_t n_Q = compute_var ? (size_t)S * K * K : 0;
  int rc = ensure_dev(ctx, &ctx->d_scratch, &ctx->d_scratch_cap, n_res + 2 * n_Z + n_Q);
  if (rc) return rc;
  rc = ensure_pinned(ctx, n_res + n_Q);
  if (rc) return rc;
  double* d_res = ctx->d_scratch;
  double* d_Z = compute_var ? d_res + n_res : nullptr;
  double* d_V = compute_var ? d_Z + n_Z : nullptr;
  double* d_Q = compute_var ? d_V + n_Z : nullptr;
  rc = launch_gp_log_joint(ctx, grad_flags != 0, d_res, d_Z);
  if (rc) return rc;
  if (compute_var) {
    rc = launch_gp_var(ctx, d_Z, d_V, d_Q);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_pinned + n_res, d_Q, sizeof(double) * n_Q,
                                hipMemcpyDeviceToHost, ctx->stream));
  }
  HIP_TRY(ctx, hipMemcpyAsync(ctx->h_pinned, d_res, sizeof(double) * n_res, hipMemcpyDeviceToHost,
                              ctx->stream));
  HIP_TRY(ctx, stream_wait(ctx));

  GljHost o;
  glj_finalize(ctx, ctx->h_pinned, grad_flags != 0, o);
  if (I_SxK) memcpy(I_SxK, o.I_sk.data(), sizeof(double) * S * K);

  std::vector<double> vG(S, 0.0);
  if (compute_var) {
    const double* Jd = ctx->h_pinned + n_res;  // J_sjk as computed on the device (gram_kernel)
    for (int s = 0; s < S; ++s) {
      for (int k = 0; k < K; ++k)
        for (int j = 0; j <= k; ++j) {
          const double J = Jd[((size_t)s * K + j) * K + k];
          if (j == k)
            vG[s] += ctx->w[k] * ctx->w[k] * (J > kTiny ? J : kTiny);
          else
            vG[s] += 2.0 * ctx->w[j] * ctx->w[k] * J;
        }
      if (vG[s] < kTiny) vG[s] = kTiny;
    }
    if (J_SxKxK) memcpy(J_SxKxK, Jd, sizeof(double) * (size_t)S * K * K);
  }

  // gradients per sample
  const int n_dG = glj_pack(ctx, nullptr, nullptr, nullptr, nullptr, grad_flags, jacobian_flag, nullptr);
  std::vector<double> dGs((size_t)n_dG * S, 0.0);  // [S][n_dG]
  if (grad_flags)
    for (int s = 0; s < S; ++s)
      glj_pack(ctx, o.mu.data() + (size_t)s * K * D, o.sigma.data() + (size_t)s * K,
               o.lambd.data() + (size_t)s * D, o.w.data() + (size_t)s * K, grad_flags,
               jacobian_flag, dGs.data() + (size_t)s * n_dG);

  double vss_out = 0.0;
  if (S > 1 && avg_flag) {
    double Gbar = 0.0;
    for (int s = 0; s < S; ++s) Gbar += o.G[s];
    Gbar /= S;
    if (compute_var) {
      double vss = 0.0, vmean = 0.0, vstd = 0.0;
      for (int s = 0; s < S; ++s) {
        vss += (o.G[s] - Gbar) * (o.G[s] - Gbar);
        vmean += vG[s];
      }
      vss /= (S - 1);
      vmean /= S;
      for (int s = 0; s < S; ++s) vstd += (vG[s] - vmean) * (vG[s] - vmean);
      vstd = std::sqrt(vstd / (S - 1));
      vss_out = vss + vstd;
      double sum = 0.0;
      for (int s = 0; s < S; ++s) sum += vG[s];
      if (varG) varG[0] = sum / S + vss;
    }
    if (G) G[0] = Gbar;
    if (dG && grad_flags)
      for (int i = 0; i < n_dG; ++i) {
        double a = 0.0;
        for (int s = 0; s < S; ++s) a += dGs[(size_t)s * n_dG + i];
        dG[i] = a / S;
      }
  } else {
    if (G) memcpy(G, o.G.data(), sizeof(double) * S);
    if (varG && compute_var) memcpy(varG, vG.data(), sizeof(double) * S);
    if (dG && grad_flags)
      for (int i = 0; i < n_dG; ++i)
        for (int s = 0; s < S; ++s) dG[(size_t)i * S + s] = dGs[(size_t)s * n_dG + i];
  }
  if (var_ss) *var_ss = vss_out;
  return VBMC_OK;
}

extern "C" int vbmc_gp_predict(vbmc_ctx* ctx, int64_t M, const double* xs_MxD, int add_noise,
                               int separate_samples, double* fmu, double* fs2) {
  if (!ctx || (M > 0 && (!xs_MxD || !fmu || !fs2))) return VBMC_E_ARG;
  NEED_DEVICE(ctx);
  if (!ctx->gp.set) return vbmc_fail(ctx, VBMC_E_ARG, "gp_predict: GP not set");
  if (M == 0) return VBMC_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const GpState& g = ctx->gp;
  const int N = g.N, D = g.D, S = g.S;
  if (D > 32) return vbmc_fail(ctx, VBMC_E_UNSUP, "gp_predict: D=%d > 32 not supported", D);
  const int ntiles = (N + 63) / 64;
  // batch so that the S kernel matrices of a batch stay under 1 GiB
  int64_t mb = ((int64_t)1 << 27) / ((int64_t)S * N);
  mb = mb > 65536 ? 65536 : (mb < 64 ? 64 : (mb / 64) * 64);
  if (M < mb) mb = M;
  // scratch: xs (mb*D) | Ks [S](mb*N) | part, fpart [S](2*ntiles*mb) | fmu [S][mb] | fs2 [S][mb]
  const size_t ks_n = predict_ks_elems(S, mb, N);
  const size_t need = align32((size_t)mb * D) + ks_n + 2 * (size_t)S * ntiles * mb + 2 * (size_t)S * mb;
  int rc = ensure_dev(ctx, &ctx->d_scratch, &ctx->d_scratch_cap, need);
  if (rc) return rc;
  rc = ensure_pinned(ctx, 2 * (size_t)S * mb);
  if (rc) return rc;
  double* d_xs = ctx->d_scratch;
  double* d_Ks = d_xs + align32((size_t)mb * D);  // 256-byte aligned: read by 16-byte LDS-direct loads
  double* d_part = d_Ks + ks_n;
  double* d_fmu = d_part + 2 * (size_t)S * ntiles * mb;
  double* d_fs2 = d_fmu + (size_t)S * mb;
  std::vector<double> mu_s, s2_s;
  if (!separate_samples) {
    mu_s.resize((size_t)mb * S);
    s2_s.resize((size_t)mb * S);
  }
  for (int64_t o = 0; o < M; o += mb) {
    const int64_t m = (M - o) < mb ? (M - o) : mb;
    HIP_TRY(ctx, hipMemcpyAsync(d_xs, xs_MxD + o * D, sizeof(double) * m * D, hipMemcpyHostToDevice,
                                ctx->stream));
    // every hyper-parameter sample in the same three launches (grid.z = sample)
    if (ctx->timing) HIP_TRY(ctx, hipEventRecord(ctx->ev[6], ctx->stream));
    rc = launch_gp_predict_all(ctx, m, d_xs, d_Ks, d_part, add_noise, d_fmu, d_fs2, mb);
    if (rc) return rc;
    if (ctx->timing) {
      HIP_TRY(ctx, hipEventRecord(ctx->ev[7], ctx->stream));
      ctx->ev_valid[3] = true;
    }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_pinned, d_fmu, sizeof(double) * 2 * S * mb, hipMemcpyDeviceToHost,
                                ctx->stream));
    HIP_TRY(ctx, stream_wait(ctx));
    for (int s = 0; s < S; ++s) {
      const double* hm = ctx->h_pinned + (size_t)s * mb;
      const double* hv = ctx->h_pinned + (size_t)(S + s) * mb;
      if (separate_samples) {
        for (int64_t i = 0; i < m; ++i) {
          fmu[(o + i) * S + s] = hm[i];
          fs2[(o + i) * S + s] = hv[i];
        }
      } else {
        for (int64_t i = 0; i < m; ++i) {
          mu_s[(size_t)i * S + s] = hm[i];
          s2_s[(size_t)i * S + s] = hv[i];
        }
      }
    }
    if (!separate_samples) {
      // mean over s; fs2 = mean_s fs2 + var_s(fmu, ddof=1)
      for (int64_t i = 0; i < m; ++i) {
        double a = 0.0, v = 0.0;
        for (int s = 0; s < S; ++s) {
          a += mu_s[(size_t)i * S + s];
          v += s2_s[(size_t)i * S + s];
        }
        a /= S;
        v /= S;
        if (S > 1) {
          double q = 0.0;
          for (int s = 0; s < S; ++s) {
            const double t = mu_s[(size_t)i * S + s] - a;
            q += t * t;
          }
          v += q / (S - 1);
        }
        fmu[o + i] = a;
        fs2[o + i] = v;
      }
    }
  }
  return VBMC_OK;
}
