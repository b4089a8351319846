// Batched objective for the sieve: B candidate parameter vectors evaluated in one go.
//
// Reference: vbmc/variational_optimization.py:775-787 (_sieve) calls
//   _neg_elcbo(theta_b, gp, vp0, 0, ns_ent_K_fast = 0, compute_grad = False, theta_bnd)
// once per candidate (up to ceil(50 K) of them per VBMC iteration) -- every call tiny and
// launch-bound.  Here the candidate vectors go up in ONE copy and four launches do everything:
//   batch_pack_kernel          a workgroup per candidate: set_parameters (variational_posterior.py:680-759) with the
//                              eta max-shift and the mixture pack (adam_dev::pack_from_theta, the optimiser loop's own)
//   glj_value_batch_kernel     the GP expected-log-joint values: lane = (candidate, component), points through scalar loads
//   entlb_value_batch_kernel   the lower-bound entropy, a workgroup per candidate
//   batch_finalize_kernel      a workgroup per candidate: G from the sums (api_gp.hip glj_finalize's value part), the
//                              soft bounds and the weight penalty (:1195-1229), F = -G - H + loss
// and 3 B doubles come back.  (Rounds 1-2 made the packs and the finalisation on the host, one candidate after the
// other, and moved 26 MB of packs up and 21 MB of sums down per 2 500 candidates: 11.0 ms, of which the GPU worked 1.8.)
#include <cmath>
#include <cstring>

#include "adam_dev.h"
#include "common.h"
#include "fastmath.h"

namespace {

__device__ __forceinline__ double wave_sum(double v) {
  return fm::wave_sum_dpp(v);
}

// H_b = -sum_i w_i log( sum_j w_j gamma_ij )   (entlb_vbmc.py:84-97), value only.
// One workgroup per candidate; wave w handles rows i = w, w+4, ...; lanes run over j.  The candidate's means, sigmas
// and weights are staged in LDS once (they were re-read from memory for every (i, j) pair: 117 us per 2 500 candidates).
__global__ __launch_bounds__(256) void entlb_value_batch_kernel(const double* __restrict__ packs,
                                                                MixLayout ml, size_t stride,
                                                                double* __restrict__ H) {
  extern __shared__ double sm[];  // mup [K][D] | sigma^2 [K] | w [K]
  __shared__ double sPart[4];
  const int D = ml.D, K = ml.K;
  const double* mix = packs + (size_t)blockIdx.x * stride;
  const double* lam = mix + ml.o_lam;
  double* sMup = sm;
  double* sS2 = sMup + K * D;
  double* sW = sS2 + K;
  for (int i = threadIdx.x; i < K * D; i += 256) sMup[i] = mix[ml.o_mup + i];
  for (int i = threadIdx.x; i < K; i += 256) {
    const double sg = mix[ml.o_sig + i];
    sS2[i] = sg * sg;
    sW[i] = mix[ml.o_w + i];
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double lnc = -0.5 * D * log(2.0 * M_PI);
  for (int d = 0; d < D; ++d) lnc -= log(lam[d]);
  __syncthreads();
  constexpr double LOG2E = 0x1.71547652b82fep+0;
  double hacc = 0.0;
  for (int i = wave; i < K; i += 4) {
    double acc = 0.0;
    for (int j = lane; j < K; j += 64) {
      const double s2 = sS2[i] + sS2[j];
      double d2 = 0.0;
      for (int d = 0; d < D; ++d) {
        const double t = sMup[i * D + d] - sMup[j * D + d];
        d2 = fma(t, t, d2);
      }
      acc += sW[j] * fm::exp2_fast(LOG2E * (lnc - 0.5 * D * fm::log_fast(s2) - 0.5 * d2 * fm::rcp_fast(s2)));
    }
    acc = wave_sum(acc);
    hacc -= sW[i] * fm::log_fast(acc);
  }
  if (lane == 0) sPart[wave] = hacc;
  __syncthreads();
  if (threadIdx.x == 0) H[blockIdx.x] = (sPart[0] + sPart[1]) + (sPart[2] + sPart[3]);
}

// theta_b -> mixture_b -> pack_b on the device; the attributes (normalised mu | sigma | lambda | w | eta) stay in
// auxs for the finalisation; theta's eta tail is max-shifted in place (on the device copy)
__global__ __launch_bounds__(256) void batch_pack_kernel(adam_dev::AdamDev a, double* __restrict__ thetas,
                                                         double* __restrict__ auxs, const double* __restrict__ base_aux,
                                                         double* __restrict__ packs, size_t stride, int n_aux,
                                                         int* __restrict__ bad) {
  __shared__ double red[16];
  const int b = blockIdx.x, tid = threadIdx.x;
  double* th = thetas + (size_t)b * a.n_theta;
  double* aux = auxs + (size_t)b * n_aux;
  for (int i = tid; i < n_aux; i += 256) aux[i] = base_aux[i];  // the blocks theta does not carry
  int nf = 0;
  for (int i = tid; i < a.n_theta; i += 256) nf |= !isfinite(th[i]);
  if (nf) atomicMin(bad, b);
  __syncthreads();
  adam_dev::pack_from_theta<256>(a, th, aux, red, packs + (size_t)b * stride);
}

// The GP expected-log-joint VALUES of a batch (variational_optimization.py:1400-1406,1466: I_sk = sum_n z_n alpha_n with
// z_n = exp(lnnf - 1/2 sum_d ((mu_dk - X_nd) / tau_dk)^2)): lane = one (candidate b, component k) pair, grid.y = GP sample s.
// Everything indexed by the point n -- X_n, alpha_n -- is then wave-uniform and arrives through SCALAR loads as SGPR
// operands, as the table rows of the entropy kernel do; the lane keeps mu_dk and 1/tau_dk in registers and runs over the
// N points with no memory instruction of its own: 2 D + ~20 float64 instructions per (pair, point).  (The block kernel,
// glj_block.h, is built for ONE mixture and its gradient: a workgroup per (s,k) re-reads X for every candidate --
// 125 000 workgroups and 4 GB of L2 reads per 2 500 candidates, 0.83 ms.)
template <int DP>
__global__ __launch_bounds__(256) void glj_value_batch_kernel(const double* __restrict__ packs, MixLayout ml, size_t stride,
                                                              int n_pairs, const double* __restrict__ X,
                                                              const double* __restrict__ alpha, const double* __restrict__ hyp,
                                                              int N, int P, int S, double* __restrict__ res) {
  const int D = ml.D, K = ml.K;
  const int s = blockIdx.y;
  const int pair = blockIdx.x * 256 + threadIdx.x;
  const bool live = pair < n_pairs;
  const int pc = live ? pair : n_pairs - 1;
  const int b = pc / K, k = pc - b * K;
  const double* mix = packs + (size_t)b * stride;
  const double* h = hyp + (size_t)s * P;
  const double sigk = mix[ml.o_sig + k];
  double mu[DP], itau[DP], term = 0.0;
#pragma unroll
  for (int d = 0; d < DP; ++d) {
    if (d < D) {
      const double ell = fm::exp2_fast(0x1.71547652b82fep+0 * h[d]);  // exp(h_d)
      const double lam = mix[ml.o_lam + d];
      const double tau2 = sigk * sigk * lam * lam + ell * ell;
      itau[d] = fm::rsqrt_fast(tau2);
      mu[d] = mix[ml.o_mu + k * D + d] * itau[d];  // mu_dk / tau_dk: (mu - x) / tau is then one FMA per dimension
      term += h[d] - 0.5 * fm::log_fast(tau2);
    } else {
      itau[d] = 0.0;
      mu[d] = 0.0;
    }
  }
  const double lnnf = 2.0 * h[D] + term;
  const double* al = alpha + (size_t)s * N;
  // four points per step: their scalar loads are issued together, ahead of the arithmetic (one point per step left every
  // iteration waiting for its own operands: 0.30 ms per 2 500 candidates instead of 0.1x)
  double acc = 0.0;
  constexpr int PU = 4;
  for (int n0 = 0; n0 < N; n0 += PU) {
    double d2[PU], a_n[PU];
#pragma unroll
    for (int u = 0; u < PU; ++u) {
      const int n = n0 + u < N ? n0 + u : N - 1;
      const double* x = X + (size_t)n * D;  // wave-uniform: scalar loads
      a_n[u] = n0 + u < N ? al[n] : 0.0;
      double q = 0.0;
#pragma unroll
      for (int d = 0; d < DP; ++d) {
        const double dl = fma(-x[d < D ? d : 0], itau[d], mu[d]);
        q = fma(dl, dl, q);
      }
      d2[u] = q;
    }
#pragma unroll
    for (int u = 0; u < PU; ++u) {
      const double z = fm::exp2_fast(0x1.71547652b82fep+0 * (lnnf - 0.5 * d2[u]));
      acc = fma(z, a_n[u], acc);
    }
  }
  if (live) res[((size_t)b * S * K + (size_t)s * K + k) * (1 + 2 * D)] = acc;
}

template <int DP>
static void launch_glj_value_batch(hipStream_t st, const double* packs, const MixLayout& ml, size_t stride, int B, const GpState& g,
                                   double* res) {
  const int n_pairs = B * ml.K;
  hipLaunchKernelGGL((glj_value_batch_kernel<DP>), dim3((n_pairs + 255) / 256, g.S), dim3(256), 0, st, packs, ml, stride, n_pairs,
                     (const double*)g.d_X, (const double*)g.d_alpha, (const double*)g.d_hyp, g.N, g.P, g.S, res);
}

struct BatchFin {
  int D, K, S, P, mean_kind, mask, n_theta, n_aux, n_bnd;
  const double *thetas, *auxs, *base_aux, *res, *hyp, *Hd, *blb, *bub;
  double tol_con, w_thresh, w_pen;
  double *F, *G, *H;
};

__global__ __launch_bounds__(256) void batch_finalize_kernel(BatchFin f) {
  extern __shared__ double sh[];  // iom2 [S][D]
  __shared__ double red[8];
  const int D = f.D, K = f.K, S = f.S, b = blockIdx.x, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const double* th = f.thetas + (size_t)b * f.n_theta;
  const double* aux = f.auxs + (size_t)b * f.n_aux;
  const double* mu = aux;
  const double* sg = mu + K * D;
  const double* lm = sg + K;
  const double* w = lm + D;
  const bool quad = f.mean_kind == VBMC_MEAN_NEGQUAD;
  const bool o_mu = f.mask & 1, o_sg = f.mask & 2, o_lm = f.mask & 4, o_w = f.mask & 8;
  const int st1 = 1 + 2 * D;
  if (quad)
    for (int i = tid; i < S * D; i += 256) {
      const int s = i / D, d = i - s * D;
      sh[i] = exp(-2.0 * f.hyp[(size_t)s * f.P + 2 * D + 3 + d]);
    }
  __syncthreads();
  // ---- G = mean_s sum_k w_k I_sk  (variational_optimization.py:1466-1476) ----
  double gacc = 0.0;
  for (int idx = tid; idx < S * K; idx += 256) {
    const int s = idx / K, k = idx - s * K;
    const double* h = f.hyp + (size_t)s * f.P;
    double I = f.res[((size_t)b * S * K + idx) * st1] + (f.mean_kind == VBMC_MEAN_ZERO ? 0.0 : h[D + 2]);
    if (quad) {
      double nu = 0.0;
      for (int d = 0; d < D; ++d) {
        const double m = mu[(size_t)k * D + d], xm = h[D + 3 + d];
        nu += sh[s * D + d] * (m * m + sg[k] * sg[k] * lm[d] * lm[d] - 2.0 * m * xm + xm * xm);
      }
      I += -0.5 * nu;
    }
    gacc += w[k] * I;
  }
  // ---- soft bounds on (mu, ln sigma + ln lambda, eta) and the weight penalty (:1195-1229) ----
  double L = 0.0;
  if (f.n_bnd > 0) {
    const int n_mu = o_mu ? D * K : 0, n_sc = (o_sg || o_lm) ? D * K : 0;
    const int p_sg = n_mu, p_lm = p_sg + (o_sg ? K : 0), p_w = f.n_theta - K;
    const double* bsg = f.base_aux + K * D;
    const double* blm = bsg + K;
    for (int q = tid; q < f.n_bnd; q += 256) {
      double x;
      if (q < n_mu) {
        x = th[q];
      } else if (q < n_mu + n_sc) {
        const int r = q - n_mu, k = r / D, d = r - k * D;
        x = (o_lm ? th[p_lm + d] : log(blm[d])) + (o_sg ? th[p_sg + k] : log(bsg[k]));
      } else {
        x = th[p_w + (q - n_mu - n_sc)];  // (the tail was max-shifted by the pack kernel)
      }
      const double lb = f.blb[q], ub = f.bub[q];
      const double ell = (ub - lb) * f.tol_con;
      if (x < lb) L += 0.5 * ((lb - x) / ell) * ((lb - x) / ell);
      if (x > ub) L += 0.5 * ((x - ub) / ell) * ((x - ub) / ell);
    }
    if (o_w)
      for (int k = tid; k < K; k += 256) L += ((w[k] < f.w_thresh) ? w[k] : f.w_thresh) * f.w_pen;
  }
  gacc = wave_sum(gacc);
  L = wave_sum(L);
  if (lane == 0) {
    red[wave] = gacc;
    red[4 + wave] = L;
  }
  __syncthreads();
  if (tid == 0) {
    const double Gv = ((red[0] + red[1]) + (red[2] + red[3])) / S;
    const double Lv = (red[4] + red[5]) + (red[6] + red[7]);
    double Hv;
    if (K > 1) {
      Hv = f.Hd[b];
    } else {
      Hv = 0.5 * D * (1.0 + log(2.0 * M_PI)) + D * log(sg[0]);
      for (int d = 0; d < D; ++d) Hv += log(lm[d]);
    }
    f.F[b] = -Gv - Hv + Lv;
    f.G[b] = Gv;
    f.H[b] = Hv;
  }
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
  const int need = (o_mu ? D * K : 0) + (o_sg ? K : 0) + (o_lm ? D : 0) + (o_w ? K : 0);
  if (n_theta != need) return vbmc_fail(ctx, VBMC_E_ARG, "neg_elcbo_batch: theta length %d does not match", n_theta);
  const bool has_bnd = opts->bnd_lb && opts->bnd_ub;
  const int n_ext = (o_mu ? D * K : 0) + ((o_sg || o_lm) ? D * K : 0) + (o_w ? K : 0);
  if (has_bnd && n_ext != opts->n_bnd)
    return vbmc_fail(ctx, VBMC_E_ARG, "neg_elcbo_batch: bounds length %d != %d", opts->n_bnd, n_ext);
  const int n_bnd = has_bnd ? opts->n_bnd : 0;
  MixLayout ml;
  ml.plan(D, K);
  const size_t stride = (size_t)ml.total;
  const int n_aux = (int)adam_dev::aux_len(D, K);

  // ---- device memory: thetas | attributes | packs | sums | H | F G H | base attributes | bounds | flags ----
  const size_t n_res = (size_t)B * S * K * (1 + 2 * D);
  const size_t total = (size_t)B * n_theta + (size_t)B * n_aux + stride * B + n_res + 4 * (size_t)B + n_aux + 2 * (size_t)n_bnd + 4;
  int rc = ensure_dev(ctx, &ctx->d_scratch, &ctx->d_scratch_cap, total);
  if (rc) return rc;
  rc = ensure_pinned(ctx, 3 * (size_t)B + n_aux + 2 * (size_t)n_bnd + 2);
  if (rc) return rc;
  double* d_th = ctx->d_scratch;
  double* d_aux = d_th + (size_t)B * n_theta;
  double* d_packs = d_aux + (size_t)B * n_aux;
  double* d_res = d_packs + stride * B;
  double* d_H = d_res + n_res;
  double* d_out = d_H + B;  // F | G | H
  double* d_base = d_out + 3 * (size_t)B;
  double* d_bnd = d_base + n_aux;
  int* d_flags = (int*)(d_bnd + 2 * (size_t)n_bnd);  // [0] non-finite status of the pack, [1] first bad candidate
  hipStream_t sm = ctx->stream;
  // the small host-side inputs through the pinned buffer: the ctx mixture's attributes, the bounds
  double* hp = ctx->h_pinned + 3 * (size_t)B;
  memcpy(hp, ctx->mu.data(), sizeof(double) * K * D);
  memcpy(hp + K * D, ctx->sigma.data(), sizeof(double) * K);
  memcpy(hp + K * D + K, ctx->lambd.data(), sizeof(double) * D);
  memcpy(hp + K * D + K + D, ctx->w.data(), sizeof(double) * K);
  memcpy(hp + K * D + 2 * K + D, ctx->eta.data(), sizeof(double) * K);
  if (has_bnd) {
    memcpy(hp + n_aux, opts->bnd_lb, sizeof(double) * n_bnd);
    memcpy(hp + n_aux + n_bnd, opts->bnd_ub, sizeof(double) * n_bnd);
  }
  HIP_TRY(ctx, hipMemcpyAsync(d_base, hp, sizeof(double) * (n_aux + 2 * (size_t)n_bnd), hipMemcpyHostToDevice, sm));
  HIP_TRY(ctx, hipMemcpyAsync(d_th, thetas_BxN, sizeof(double) * (size_t)B * n_theta, hipMemcpyHostToDevice, sm));
  const int flags0[2] = {0, 0x7fffffff};
  HIP_TRY(ctx, hipMemcpyAsync(d_flags, flags0, sizeof(flags0), hipMemcpyHostToDevice, sm));

  adam_dev::AdamDev a;
  memset(&a, 0, sizeof(a));
  a.ml = ml;
  a.D = D;
  a.K = K;
  a.S = S;
  a.mask = mask;
  a.n_theta = n_theta;
  a.c_norm = 1.0 / std::pow(2.0 * M_PI, 0.5 * D);
  a.status = d_flags;
  hipLaunchKernelGGL(batch_pack_kernel, dim3(B), dim3(256), 0, sm, a, d_th, d_aux, (const double*)d_base, d_packs, stride, n_aux,
                     d_flags + 1);
  HIP_TRY(ctx, hipGetLastError());

  if (D <= 32 && (int64_t)B * K < ((int64_t)1 << 31)) {
    const GpState& g = ctx->gp;
    if (D <= 4) launch_glj_value_batch<4>(sm, d_packs, ml, stride, B, g, d_res);
    else if (D <= 8) launch_glj_value_batch<8>(sm, d_packs, ml, stride, B, g, d_res);
    else if (D <= 12) launch_glj_value_batch<12>(sm, d_packs, ml, stride, B, g, d_res);
    else if (D <= 16) launch_glj_value_batch<16>(sm, d_packs, ml, stride, B, g, d_res);
    else if (D <= 24) launch_glj_value_batch<24>(sm, d_packs, ml, stride, B, g, d_res);
    else launch_glj_value_batch<32>(sm, d_packs, ml, stride, B, g, d_res);
    HIP_TRY(ctx, hipGetLastError());
  } else {  // more dimensions than the lane-per-pair kernel keeps in registers: the block kernel, grid.y = candidate
    PrepArgs pa;
    glj_fill_prep(ctx, 0, d_res, nullptr, pa);
    pa.mix = d_packs;
    pa.ml = ml;
    pa.batch = B;
    pa.mix_stride = stride;
    pa.res_stride = (size_t)S * K * (1 + 2 * D);
    rc = launch_prep(ctx, pa);
    if (rc) return rc;
  }
  if (K > 1) {
    const size_t lds_e = sizeof(double) * ((size_t)K * D + 2 * (size_t)K);
    if (lds_e > 150 * 1024) return vbmc_fail(ctx, VBMC_E_UNSUP, "neg_elcbo_batch: K=%d D=%d too large", K, D);
    if (lds_e > 64 * 1024)
      HIP_TRY(ctx, hipFuncSetAttribute((const void*)entlb_value_batch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_e));
    hipLaunchKernelGGL(entlb_value_batch_kernel, dim3(B), dim3(256), lds_e, sm,
                       (const double*)d_packs, ml, stride, d_H);
    HIP_TRY(ctx, hipGetLastError());
  }
  BatchFin f;
  f.D = D; f.K = K; f.S = S; f.P = ctx->gp.P; f.mean_kind = ctx->gp.mean_kind; f.mask = mask;
  f.n_theta = n_theta; f.n_aux = n_aux; f.n_bnd = n_bnd;
  f.thetas = d_th; f.auxs = d_aux; f.base_aux = d_base; f.res = d_res; f.hyp = ctx->gp.d_hyp; f.Hd = d_H;
  f.blb = d_bnd; f.bub = d_bnd + n_bnd;
  f.tol_con = opts->tol_con; f.w_thresh = opts->weight_threshold; f.w_pen = opts->weight_penalty;
  f.F = d_out; f.G = d_out + B; f.H = d_out + 2 * (size_t)B;
  hipLaunchKernelGGL(batch_finalize_kernel, dim3(B), dim3(256), sizeof(double) * ((size_t)S * D + 1), sm, f);
  HIP_TRY(ctx, hipGetLastError());
  int flags[2] = {0, 0};
  HIP_TRY(ctx, hipMemcpyAsync(ctx->h_pinned, d_out, sizeof(double) * 3 * (size_t)B, hipMemcpyDeviceToHost, sm));
  HIP_TRY(ctx, hipMemcpyAsync(flags, d_flags, sizeof(flags), hipMemcpyDeviceToHost, sm));
  HIP_TRY(ctx, stream_wait(ctx));
  if (flags[1] != 0x7fffffff || flags[0])
    return vbmc_fail(ctx, VBMC_E_NONFINITE, "neg_elcbo_batch: candidate %d is not finite", flags[1] != 0x7fffffff ? flags[1] : -1);
  memcpy(F_B, ctx->h_pinned, sizeof(double) * B);
  if (G_B) memcpy(G_B, ctx->h_pinned + B, sizeof(double) * B);
  if (H_B) memcpy(H_B, ctx->h_pinned + 2 * (size_t)B, sizeof(double) * B);
  return VBMC_OK;
}
