// the stochastic optimiser's inner loop kept on the device.
//
// The reference's minimize_adam (vbmc/minimize_adam.py:84-105) calls the objective
// _neg_elcbo(theta, gp, vp, beta, NsK, compute_grad=True, theta_bnd=...)
// (vbmc/variational_optimization.py:238-249) once per iteration and updates theta on the
// host: one host<->device round trip, one synchronisation and ~50 us of host arithmetic
// per iteration.  Here one iteration is four launches on one stream and no
// synchronisation:
//     elbo_prep_kernel -> entmc_ws_kernel -> entmc_finish_kernel [-> all-reduce] ->
//     adam_step_kernel
// adam_step_kernel (one workgroup) does everything the host did between two entropy
// launches: GP-sum finalisation (api_gp.hip glj_finalize), entropy Jacobians
// (api_entropy.hip entropy_pack), soft bounds + weight penalty (api_elbo.hip), the Adam
// update with the box clamp, the in-place max-shift of the eta tail
// (variational_optimization.py:1082-1085 -- minimize_adam's x IS the array _neg_elcbo
// shifts), set_parameters with the lambda renormalisation
// (variational_posterior.py:680-759) and the mixture pack of the next iterate.
// The host only decides when to stop (every batch_size = 20 iterations, as the reference
// does) from the y_tab / x_tab rows it copies back.
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "common.h"
#include "entropy_args.h"

namespace {

// One contiguous block of doubles holds everything the step kernel reads; the kernel
// mirrors a prefix of it (a tier boundary) into LDS with one deep batch of loads.
struct AdamLayout {
  int n = 0, n_aux = 0, n_hyp = 0, n_raw = 0, n_res = 0, n_bnd = 0;
  __host__ __device__ int o_theta() const { return 0; }               // [n] current x
  __host__ __device__ int o_aux() const { return n; }                 // mu K*D | sigma K | lambd D | w K | eta K
  __host__ __device__ int o_hyp() const { return o_aux() + n_aux; }   // [S][P]
  __host__ __device__ int o_raw() const { return o_hyp() + n_hyp; }   // normalised entropy accumulator
  __host__ __device__ int o_res() const { return o_raw() + n_raw; }   // [S][K][1+2D] GP sums     (tier 1 ends)
  __host__ __device__ int o_m() const { return o_res() + n_res; }     // Adam first moment        (tier 2 ends)
  __host__ __device__ int o_v() const { return o_m() + n; }
  __host__ __device__ int o_blb() const { return o_v() + n; }         // soft bounds
  __host__ __device__ int o_bub() const { return o_blb() + n_bnd; }
  __host__ __device__ int o_xlb() const { return o_bub() + n_bnd; }   // box
  __host__ __device__ int o_xub() const { return o_xlb() + n; }
  __host__ __device__ int end() const { return o_xub() + n; }         //                          (tier 3 ends)
};

struct AdamDev {
  MixLayout ml;
  AdamLayout lay;
  int D, K, S, P, mean_kind, mask, n_theta, n_bnd, has_box, has_bnd;
  int n_stage;    // doubles of `state` mirrored in LDS (0, or a tier boundary of AdamLayout)
  int work_lds;   // scratch arrays in LDS (after the mirror) instead of `work`
  double* mix;    // mixture pack of the current iterate (rewritten for the next one)
  double* state;  // AdamLayout block
  double* work;   // scratch when it does not fit the LDS: see work_len()
  double tol_con, w_thresh, w_pen;
  double step, c1, c2, fudge, beta1, beta2;  // step size and 1/(1-beta^(i+1)) of this iteration
  double* x_row;  // [n_theta] row i of x_tab
  double* y_out;  // y_tab[i], then G, H (3 doubles per iteration)
  int* status;    // != 0: a non-finite iterate was produced
};

__host__ __device__ inline size_t aux_len(int D, int K) { return (size_t)K * D + 3 * (size_t)K + D; }
// scratch: ell2, iom2 [S][D] | gmu, tgs, tnu [K][D] | gsg, gw, ee [K] | glm, bl [D] | dL [n_bnd] | dF [n]
__host__ __device__ inline size_t work_len(int D, int K, int S, int n_bnd, int n) {
  return 2 * (size_t)S * D + 3 * (size_t)K * D + 3 * (size_t)K + 2 * (size_t)D + (size_t)n_bnd + (size_t)n;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// all four waves get the sum; red[0..3] is scratch (barriers on both sides)
__device__ double block_sum(double v, double* red) {
  const int tid = threadIdx.x;
  v = wave_sum(v);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

__device__ double block_max(double v, double* red) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  return fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}

// set_parameters(theta) + eta max-shift + mixture pack; theta's eta tail is shifted in
// place.  theta / aux may live in LDS; the pack goes to a.mix.
__device__ void pack_from_theta(const AdamDev& a, double* theta, double* aux, double* red) {
  const int D = a.D, K = a.K, tid = threadIdx.x, n = a.n_theta;
  const bool o_mu = a.mask & 1, o_sg = a.mask & 2, o_lm = a.mask & 4, o_w = a.mask & 8;
  const int p_sg = o_mu ? D * K : 0, p_lm = p_sg + (o_sg ? K : 0), p_w = n - K;
  double* mu = aux;
  double* sg = mu + K * D;
  double* lm = sg + K;
  double* w = lm + D;
  double* eta = w + K;
  int bad = 0;
  for (int i = tid; i < n; i += 256) bad |= !isfinite(theta[i]);
  if (bad) atomicOr(a.status, 1);
  if (o_mu)
    for (int i = tid; i < D * K; i += 256) mu[i] = theta[i];
  if (o_sg)
    for (int k = tid; k < K; k += 256) sg[k] = exp(theta[p_sg + k]);
  if (o_lm)
    for (int d = tid; d < D; d += 256) lm[d] = exp(theta[p_lm + d]);
  if (o_w) {
    double mx = -INFINITY;
    for (int k = tid; k < K; k += 256) mx = fmax(mx, theta[p_w + k]);
    mx = block_max(mx, red);
    for (int k = tid; k < K; k += 256) {
      const double e = theta[p_w + k] - mx;
      theta[p_w + k] = e;
      eta[k] = e;
      w[k] = exp(e);
    }
  }
  __syncthreads();
  // lambda -> unit RMS, sigma absorbs it; weights normalised
  double s2 = 0.0, wsum = 0.0;
  for (int d = tid; d < D; d += 256) s2 += lm[d] * lm[d];
  if (o_w)
    for (int k = tid; k < K; k += 256) wsum += w[k];
  s2 = block_sum(s2, red);
  wsum = block_sum(wsum, red);
  const double nl = sqrt(s2 / D);
  __syncthreads();
  for (int d = tid; d < D; d += 256) lm[d] /= nl;
  for (int k = tid; k < K; k += 256) {
    sg[k] *= nl;
    if (o_w) w[k] /= wsum;
  }
  __syncthreads();
  if (tid < 64) {
    double pr = 1.0;
    for (int d = tid; d < D; d += 64) pr *= lm[d];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) pr *= __shfl_xor(pr, off, 64);
    if (tid == 0) red[8] = 1.0 / pow(2.0 * M_PI, 0.5 * D) / pr;
  }
  __syncthreads();
  const double nconst = red[8];
  const double l2n = log2(nconst);
  const MixLayout& ml = a.ml;
  double* p = a.mix;
  for (int i = tid; i < K * D; i += 256) {
    const int d = i % D;
    p[ml.o_mu + i] = mu[i];
    p[ml.o_mup + i] = mu[i] / lm[d];
  }
  for (int k = tid; k < K; k += 256) {
    const double s = sg[k];
    const double sD = pow(s, (double)D);
    p[ml.o_is2 + k] = 1.0 / (s * s);
    p[ml.o_rc + k] = nconst / sD;
    p[ml.o_lrc + k] = l2n - D * log2(s);
    p[ml.o_wc + k] = w[k] * nconst / sD;
    p[ml.o_sig + k] = s;
    p[ml.o_w + k] = w[k];
  }
  for (int d = tid; d < D; d += 256) {
    p[ml.o_lam + d] = lm[d];
    p[ml.o_ilam + d] = 1.0 / lm[d];
  }
}

// MODE 2: the whole state block and the scratch are in LDS; MODE 0: everything in global memory;
// MODE 1: a prefix tier in LDS.  Compile-time for modes 0 and 2 so that every access is a true
// LDS (ds_*) or global access: a pointer that may be either one at run time makes the compiler
// emit flat_* instructions, whose LDS latency is several times that of ds_*.
template <int MODE>
__global__ __launch_bounds__(256) void adam_step_kernel(AdamDev a, int do_step) {
  extern __shared__ double sh[];
  __shared__ double red[16];
  const int D = a.D, K = a.K, S = a.S, tid = threadIdx.x, n = a.n_theta;
  const int lane = tid & 63, wave = tid >> 6;
  const int st = 1 + 2 * D;
  
  // ---- mirror the state block (or a prefix) in LDS: one deep batch of loads, then every
  // later access is an LDS access instead of a ~1 us global round trip ----
  const AdamLayout& L = a.lay;
  {
    const int cnt = do_step ? a.n_stage : min(a.n_stage, L.o_hyp());
    constexpr int U = 16;
    for (int base = 0; base < cnt; base += 256 * U) {
      double r[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = base + u * 256 + tid;
        r[u] = i < cnt ? a.state[i] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = base + u * 256 + tid;
        if (i < cnt) sh[i] = r[u];
      }
    }
    if (cnt > 0) __syncthreads();
  }
  auto at = [&](int off) -> double* {
    if (MODE == 2) return sh + off;
    if (MODE == 0) return a.state + off;
    return off < a.n_stage ? sh + off : a.state + off;
  };
  double* theta = at(L.o_theta());
  double* aux = at(L.o_aux());
  double* work = MODE == 2 ? sh + a.n_stage : MODE == 0 ? a.work : (a.work_lds ? sh + a.n_stage : a.work);
  const double* raw = at(L.o_raw());
  const double* res = at(L.o_res());
  const double* hyp = at(L.o_hyp());
  double* am = at(L.o_m());
  double* av = at(L.o_v());
  const double* bnd_lb = at(L.o_blb());
  const double* bnd_ub = at(L.o_bub());
  const double* box_lb = at(L.o_xlb());
  const double* box_ub = at(L.o_xub());

  if (do_step) {
    const bool o_mu = a.mask & 1, o_sg = a.mask & 2, o_lm = a.mask & 4, o_w = a.mask & 8;
    const int p_sg = o_mu ? D * K : 0, p_lm = p_sg + (o_sg ? K : 0), p_w = n - K;
    const double* mu = aux;
    const double* sg = mu + K * D;
    const double* lm = sg + K;
    const double* w = lm + D;
    const double* eta = w + K;
    double* ell2 = work;              // [S][D]
    double* iom2 = ell2 + S * D;      // [S][D]
    double* gmu = iom2 + S * D;       // [K][D]  d G / d mu   (averaged over s)
    double* gsg = gmu + K * D;        // [K]     d G / d sigma (pre-Jacobian)
    double* glm = gsg + K;            // [D]
    double* gw = glm + D;             // [K]     combined pre-Jacobian weight gradient of F
    double* ee = gw + K;              // [K]
    double* tgs = ee + K;             // [K][D]  per-(k,d) terms of d G / d sigma
    double* tnu = tgs + K * D;        // [K][D]  per-(k,d) terms of the quadratic-mean part
    double* bl = tnu + K * D;         // [D]     soft-bound gradient folded onto lambda
    double* dL = bl + D;              // [n_bnd]
    double* dF = dL + a.n_bnd;        // [n_theta]
    const bool quad = a.mean_kind == VBMC_MEAN_NEGQUAD;

    for (int i = tid; i < S * D; i += 256) {
      const int s = i / D, d = i - s * D;
      const double* h = hyp + (size_t)s * a.P;
      ell2[i] = exp(2.0 * h[d]);
      iom2[i] = quad ? exp(-2.0 * h[2 * D + 3 + d]) : 0.0;
    }
    __syncthreads();

    // ---- GP expected log joint (host twin: api_gp.hip glj_finalize), in two steps so that
    // all 256 lanes work: per (k, d) term first, per component after the barrier ----
    for (int idx = tid; idx < K * D; idx += 256) {
      const int k = idx / D, d = idx - k * D;
      const double sgk = sg[k], wk = w[k], lam = lm[d], m = mu[idx];
      double gm_acc = 0.0, gs_acc = 0.0, nu_acc = 0.0;
      for (int s = 0; s < S; ++s) {
        const double* h = hyp + (size_t)s * a.P;
        const double* r = res + ((size_t)s * K + k) * st;
        const double tau2 = sgk * sgk * lam * lam + ell2[s * D + d];
        const double tau = sqrt(tau2);
        const double U = r[1 + d], T = r[1 + D + d] - r[0];
        double gm = wk * (-U / tau);
        gs_acc += (lam * lam / tau2) * T / S;
        if (quad) {
          const double xm = h[D + 3 + d], io = iom2[s * D + d];
          gm -= wk * io * (m - xm);
          nu_acc += io * (m * m + sgk * sgk * lam * lam - 2.0 * m * xm + xm * xm) / S;
        }
        gm_acc += gm / S;
      }
      gmu[idx] = gm_acc;
      tgs[idx] = gs_acc;
      tnu[idx] = nu_acc;
    }
    __syncthreads();
    double gpart = 0.0;
    for (int k = tid; k < K; k += 256) {
      const double sgk = sg[k], wk = w[k];
      double gs = 0.0, nu = 0.0, base = 0.0, qbar = 0.0;
      for (int d = 0; d < D; ++d) {
        gs += tgs[k * D + d];
        nu += tnu[k * D + d];
      }
      for (int s = 0; s < S; ++s) {
        const double* h = hyp + (size_t)s * a.P;
        base += (res[((size_t)s * K + k) * st] + (a.mean_kind == VBMC_MEAN_ZERO ? 0.0 : h[D + 2])) / S;
        if (quad)
          for (int d = 0; d < D; ++d) qbar += iom2[s * D + d] * lm[d] * lm[d] / S;
      }
      const double wI = base - 0.5 * nu;  // mean over s of I_sk
      gpart += wk * wI;
      gsg[k] = wk * sgk * (gs - qbar);
      // pre-Jacobian weight gradient of F = -G - H (+ penalty below)
      gw[k] = -wI - raw[1 + D * K + K + D + k];
    }
    // ---- lambda gradient: one wave per dimension, lanes over the (s, k) terms ----
    for (int d = wave; d < D; d += 4) {
      const double lam = lm[d];
      double acc = 0.0;
      for (int idx = lane; idx < S * K; idx += 64) {
        const int s = idx / K, k = idx - s * K;
        const double* r = res + (size_t)idx * st;
        const double sgk = sg[k], wk = w[k];
        const double tau2 = sgk * sgk * lam * lam + ell2[s * D + d];
        const double T = r[1 + D + d] - r[0];
        double gl = wk * (sgk * sgk / tau2) * lam * T;
        if (quad) gl -= wk * sgk * sgk * iom2[s * D + d] * lam;
        acc += gl / S;
      }
      acc = wave_sum(acc);
      if (lane == 0) glm[d] = acc;
    }
    const double G = block_sum(gpart, red);

    // ---- soft bounds (_vp_bound_loss :537-606) and weight penalty (:1211-1229) ----
    double loss = 0.0;
    if (a.has_bnd) {
      const int n_mu = o_mu ? D * K : 0, n_sc = (o_sg || o_lm) ? D * K : 0;
      for (int i = tid; i < a.n_bnd; i += 256) {
        double x;
        if (i < n_mu) {
          x = theta[i];
        } else if (i < n_mu + n_sc) {
          const int q = i - n_mu, k = q / D, d = q - k * D;  // ravel('F') of the (D,K) array
          const double ls = o_sg ? theta[p_sg + k] : log(sg[k]);
          const double ll = o_lm ? theta[p_lm + d] : log(lm[d]);
          x = ll + ls;
        } else {
          x = theta[p_w + (i - n_mu - n_sc)];
        }
        const double lb = bnd_lb[i], ub = bnd_ub[i];
        const double ell = (ub - lb) * a.tol_con;
        double g = 0.0;
        if (x < lb) {
          const double t = (lb - x) / ell;
          loss += 0.5 * t * t;
          g = (x - lb) / (ell * ell);
        }
        if (x > ub) {
          const double t = (x - ub) / ell;
          loss += 0.5 * t * t;
          g = (x - ub) / (ell * ell);
        }
        dL[i] = g;
      }
      if (o_w) {
        for (int k = tid; k < K; k += 256) {  // same thread wrote gw[k] above
          const bool small = w[k] < a.w_thresh;
          loss += (small ? w[k] : a.w_thresh) * a.w_pen;
          if (small) gw[k] += a.w_pen;
        }
      }
    }
    loss = block_sum(loss, red);  // its barriers also order the writes above before the reads below
    if (a.has_bnd && o_lm) {
      // the reference reshapes the scale block C-order (D,K) (:585-587); restated as-is
      const int sc0 = o_mu ? D * K : 0;
      for (int d = wave; d < D; d += 4) {
        double acc = 0.0;
        for (int k = lane; k < K; k += 64) acc += dL[sc0 + d * K + k];
        acc = wave_sum(acc);
        if (lane == 0) bl[d] = acc;
      }
    }
    __syncthreads();

    // ---- softmax Jacobian of the combined weight gradient (entmc_vbmc.py:122-130) ----
    double sm_s = 1.0, sm_dot = 0.0;
    if (o_w) {
      double ps = 0.0, pd = 0.0;
      for (int k = tid; k < K; k += 256) {
        const double e = exp(eta[k]);
        ee[k] = e;
        ps += e;
        pd += e * gw[k];
      }
      sm_s = block_sum(ps, red);
      sm_dot = block_sum(pd, red);
    }

    // ---- assemble dF ----
    const double* rmu = raw + 1;
    const double* rsg = rmu + D * K;
    const double* rlm = rsg + K;
    const int sc0 = o_mu ? D * K : 0;
    for (int i = tid; i < n; i += 256) {
      double g;
      if (o_mu && i < D * K) {
        g = -gmu[i] - rmu[i];
        if (a.has_bnd) g += dL[i];
      } else if (o_sg && i >= p_sg && i < p_sg + K) {
        const int k = i - p_sg;
        g = -gsg[k] * sg[k] - rsg[k] * sg[k];
        if (a.has_bnd) {
          // the reference reshapes this block C-order (D,K) (:585-587); restated as-is
          double acc = 0.0;
          for (int d = 0; d < D; ++d) acc += dL[sc0 + d * K + k];
          g += acc;
        }
      } else if (o_lm && i >= p_lm && i < p_lm + D) {
        const int d = i - p_lm;
        g = -glm[d] * lm[d] - rlm[d] * lm[d];
        if (a.has_bnd) g += bl[d];
      } else {
        const int k = i - p_w;
        g = -ee[k] * sm_dot / (sm_s * sm_s) + ee[k] * gw[k] / sm_s;
        if (a.has_bnd) g += dL[a.n_bnd - K + k];
      }
      dF[i] = g;
    }
    if (tid == 0) {
      const double H = raw[0];
      a.y_out[0] = -G - H + loss;
      a.y_out[1] = G;
      a.y_out[2] = H;
    }
    __syncthreads();

    // ---- Adam update (minimize_adam.py:89-105) ----
    for (int i = tid; i < n; i += 256) {
      const double g = dF[i];
      const double m = a.beta1 * am[i] + (1.0 - a.beta1) * g;
      const double v = a.beta2 * av[i] + (1.0 - a.beta2) * (g * g);
      am[i] = m;
      av[i] = v;
      const double m_hat = m * a.c1, v_hat = v * a.c2;
      double x = theta[i] - a.step * m_hat / (sqrt(v_hat) + a.fudge);
      if (a.has_box) x = fmin(box_ub[i], fmax(box_lb[i], x));
      theta[i] = x;
      a.x_row[i] = x;
    }
    __syncthreads();
  }

  pack_from_theta(a, theta, aux, red);
  if (a.n_stage > 0) {  // write the mirrored, modified arrays back
    __syncthreads();
    for (int i = tid; i < L.o_hyp(); i += 256) a.state[i] = sh[i];  // theta | aux
    if (do_step && a.n_stage > L.o_m())
      for (int i = L.o_m() + tid; i < L.o_blb(); i += 256) a.state[i] = sh[i];  // m | v
  }
}

}  // namespace

// ---------------------------------------------------------------------------
struct AdamState {
  bool active = false;
  int n_theta = 0, n_bnd = 0, max_iter = 0, iter = 0, mask = 0;
  int64_t ns = 0, row_begin = 0, row_count = 0;
  int eps_mode = 0;
  uint64_t seed = 0;
  double master_min = 0, master_max = 0, master_decay = 0;
  double tol_con = 0, w_thresh = 0, w_pen = 0;
  bool has_box = false, has_bnd = false;
  AdamLayout lay;
  int n_stage = 0, work_lds = 0;  // what the step kernel keeps in LDS
  size_t lds_bytes = 0;
  double* d_buf = nullptr;
  size_t d_cap = 0;
  int* d_status = nullptr;
  // carve of d_buf
  double *state = nullptr, *work = nullptr, *x_tab = nullptr, *y_tab = nullptr;
  // draws generated one iteration ahead on a second stream (Philox mode)
  bool pregen = false;
  double* d_eps2[2] = {nullptr, nullptr};
  size_t eps2_cap = 0;               // doubles per buffer
  hipStream_t gen_stream = nullptr;
  hipEvent_t ev_gen[2] = {nullptr, nullptr};  // buffer b holds the draws of its iteration
  hipEvent_t ev_ent[2] = {nullptr, nullptr};  // the entropy kernel reading buffer b has finished
  bool ent_recorded[2] = {false, false};
};

static AdamState* adam_of(vbmc_ctx* ctx) {
  if (!ctx->adam) ctx->adam = new AdamState();
  return (AdamState*)ctx->adam;
}

void adam_free(vbmc_ctx* ctx) {
  AdamState* st = (AdamState*)ctx->adam;
  if (!st) return;
  if (st->d_buf) (void)hipFree(st->d_buf);
  if (st->d_status) (void)hipFree(st->d_status);
  for (int b = 0; b < 2; ++b) {
    if (st->d_eps2[b]) (void)hipFree(st->d_eps2[b]);
    if (st->ev_gen[b]) (void)hipEventDestroy(st->ev_gen[b]);
    if (st->ev_ent[b]) (void)hipEventDestroy(st->ev_ent[b]);
  }
  if (st->gen_stream) (void)hipStreamDestroy(st->gen_stream);
  delete st;
  ctx->adam = nullptr;
}

static void launch_step(const AdamState& st, hipStream_t sm, const AdamDev& a, int do_step) {
  const int mode = st.n_stage == 0 ? 0 : (st.n_stage == st.lay.end() ? 2 : 1);
  if (mode == 2) hipLaunchKernelGGL(adam_step_kernel<2>, dim3(1), dim3(256), st.lds_bytes, sm, a, do_step);
  else if (mode == 1) hipLaunchKernelGGL(adam_step_kernel<1>, dim3(1), dim3(256), st.lds_bytes, sm, a, do_step);
  else hipLaunchKernelGGL(adam_step_kernel<0>, dim3(1), dim3(256), st.lds_bytes, sm, a, do_step);
}

static void fill_dev(const vbmc_ctx* ctx, const AdamState& st, AdamDev& a) {
  const GpState& g = ctx->gp;
  a.ml = ctx->ml;
  a.D = ctx->D;
  a.K = ctx->K;
  a.S = g.S;
  a.P = g.P;
  a.mean_kind = g.mean_kind;
  a.mask = st.mask;
  a.n_theta = st.n_theta;
  a.n_bnd = st.n_bnd;
  a.has_box = st.has_box;
  a.has_bnd = st.has_bnd;
  a.lay = st.lay;
  a.n_stage = st.n_stage;
  a.work_lds = st.work_lds;
  a.mix = ctx->d_mix;
  a.state = st.state;
  a.work = st.work;
  a.tol_con = st.tol_con;
  a.w_thresh = st.w_thresh;
  a.w_pen = st.w_pen;
  a.step = a.c1 = a.c2 = 0.0;
  a.fudge = std::sqrt(2.220446049250313e-16);  // sqrt(np.spacing(1))
  a.beta1 = 0.9;
  a.beta2 = 0.999;
  a.x_row = nullptr;
  a.y_out = nullptr;
  a.status = st.d_status;
}

// Enqueue the generation of iteration `iter`'s draws into buffer iter & 1 on the second stream.
static int enqueue_gen(vbmc_ctx* ctx, AdamState* st, int iter) {
  const int b = iter & 1;
  if (st->ent_recorded[b]) HIP_TRY(ctx, hipStreamWaitEvent(st->gen_stream, st->ev_ent[b], 0));
  // The draws of iteration `iter`, produced while the kernels of iteration iter - 1 run: the
  // entropy kernel issues ~63 % of its FP64 slots and the finish / step / prep kernels leave the
  // GPU almost idle; this small-footprint kernel fills that time instead of adding its ~20 us to
  // the critical path.
  int rc = launch_eps_gen(ctx, st->gen_stream, st->d_eps2[b], st->ns / 2, st->row_begin, st->row_count,
                          st->seed + (uint64_t)iter);
  if (rc) return rc;
  HIP_TRY(ctx, hipEventRecord(st->ev_gen[b], st->gen_stream));
  return 0;
}

extern "C" int vbmc_adam_begin(vbmc_ctx* ctx, const double* theta0, int n_theta,
                               const vbmc_elbo_opts* opts, const double* lb, const double* ub,
                               int max_iter, double master_min, double master_max,
                               double master_decay) {
  if (!ctx || !theta0 || !opts || max_iter < 1) return VBMC_E_ARG;
  NEED_DEVICE(ctx);
  if (!ctx->mix_set) return vbmc_fail(ctx, VBMC_E_ARG, "adam_begin: mixture (D,K) not set");
  if (!ctx->gp.set) return vbmc_fail(ctx, VBMC_E_ARG, "adam_begin: GP not set");
  if (ctx->gp.D != ctx->D) return vbmc_fail(ctx, VBMC_E_ARG, "adam_begin: GP/mixture D mismatch");
  if (opts->ns_per_comp <= 0 || (opts->ns_per_comp & 1))
    return vbmc_fail(ctx, VBMC_E_UNSUP,
                     "adam_begin: the stochastic optimiser needs an even ns_per_comp > 0");
  if (!opts->compute_grad) return vbmc_fail(ctx, VBMC_E_ARG, "adam_begin: compute_grad must be set");
  if ((lb == nullptr) != (ub == nullptr)) return vbmc_fail(ctx, VBMC_E_ARG, "adam_begin: lb/ub must both be given");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int D = ctx->D, K = ctx->K, S = ctx->gp.S;
  const int mask = opts->optimize_mask;
  const int need = ((mask & 1) ? D * K : 0) + ((mask & 2) ? K : 0) + ((mask & 4) ? D : 0) +
                   ((mask & 8) ? K : 0);
  if (n_theta != need || n_theta < 1)
    return vbmc_fail(ctx, VBMC_E_ARG, "adam_begin: theta length %d does not match D=%d K=%d mask=%d",
                     n_theta, D, K, mask);
  for (int i = 0; i < n_theta; ++i)
    if (!std::isfinite(theta0[i])) return vbmc_fail(ctx, VBMC_E_NONFINITE, "theta has a non-finite entry");
  AdamState* st = adam_of(ctx);
  st->active = false;
  ctx->pack_valid = false;  // the loop rewrites d_mix on the device; vbmc_adam_end re-syncs the host copies
  st->has_bnd = opts->bnd_lb && opts->bnd_ub;
  st->n_bnd = st->has_bnd ? opts->n_bnd : 0;
  if (st->has_bnd) {
    const int want = ((mask & 1) ? D * K : 0) + ((mask & 6) ? D * K : 0) + ((mask & 8) ? K : 0);
    if (st->n_bnd != want)
      return vbmc_fail(ctx, VBMC_E_ARG, "adam_begin: bounds length %d != %d", st->n_bnd, want);
  }
  st->has_box = lb != nullptr;
  st->n_theta = n_theta;
  st->max_iter = max_iter;
  st->iter = 0;
  st->mask = mask;
  st->ns = opts->ns_per_comp;
  st->eps_mode = opts->eps_mode;
  st->seed = opts->seed;
  st->master_min = master_min;
  st->master_max = master_max;
  st->master_decay = master_decay;
  st->tol_con = opts->tol_con;
  st->w_thresh = opts->weight_threshold;
  st->w_pen = opts->weight_penalty;
  const int64_t n_half = st->ns / 2;
  st->row_begin = opts->row_begin;
  st->row_count = opts->row_count;
  if (st->row_count < 0) {
    st->row_begin = n_half * ctx->rank / ctx->world;
    st->row_count = n_half * (ctx->rank + 1) / ctx->world - st->row_begin;
  }
  if (st->row_begin < 0 || st->row_begin + st->row_count > n_half)
    return vbmc_fail(ctx, VBMC_E_ARG, "adam_begin: bad row slice");
  if (st->eps_mode == VBMC_EPS_RESIDENT &&
      (!ctx->d_eps || ctx->eps_K != K || ctx->eps_D != D || ctx->eps_n_half != n_half ||
       ctx->eps_row_begin != st->row_begin || ctx->eps_rows != st->row_count))
    return vbmc_fail(ctx, VBMC_E_ARG, "adam_begin: resident eps does not match the request");

  // one allocation, carved: the state block, scratch, x_tab, y_tab
  AdamLayout& L = st->lay;
  L.n = n_theta;
  L.n_aux = (int)aux_len(D, K);
  L.n_hyp = S * ctx->gp.P;
  L.n_raw = raw_len(D, K);
  L.n_res = S * K * (1 + 2 * D);
  L.n_bnd = st->n_bnd;
  const size_t n_work = work_len(D, K, S, st->n_bnd, n_theta);
  {
    // LDS plan of the step kernel: the largest tier of the state block that fits next to
    // the scratch arrays
    const char* no_lds = getenv("VBMC_ADAM_NO_LDS");  // test hook: force the global-memory path
    const size_t cap = (no_lds && no_lds[0] == '1') ? 0 : 150 * 1024 / sizeof(double);
    st->n_stage = 0;
    for (int tier : {L.end(), L.o_m(), L.o_res()})
      if ((size_t)tier + n_work <= cap) {
        st->n_stage = tier;
        break;
      }
    st->work_lds = st->n_stage > 0;
    st->lds_bytes = st->work_lds ? sizeof(double) * ((size_t)st->n_stage + n_work) : 0;
    if (st->lds_bytes > 64 * 1024)
    {
      HIP_TRY(ctx, hipFuncSetAttribute((const void*)adam_step_kernel<1>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)st->lds_bytes));
      HIP_TRY(ctx, hipFuncSetAttribute((const void*)adam_step_kernel<2>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)st->lds_bytes));
    }
  }
  const size_t total = (size_t)L.end() + n_work + (size_t)max_iter * n_theta + 3 * (size_t)max_iter + 64;
  int rc = ensure_dev(ctx, &st->d_buf, &st->d_cap, total);
  if (rc) return rc;
  if (!st->d_status) HIP_TRY(ctx, hipMalloc((void**)&st->d_status, sizeof(int)));
  st->state = st->d_buf;
  st->work = st->state + L.end();
  st->x_tab = st->work + n_work;
  st->y_tab = st->x_tab + (size_t)max_iter * n_theta;

  hipStream_t sm = ctx->stream;
  double* sb = st->state;
  HIP_TRY(ctx, hipMemsetAsync(sb, 0, sizeof(double) * L.end(), sm));  // m = v = 0
  HIP_TRY(ctx, hipMemcpyAsync(sb + L.o_theta(), theta0, sizeof(double) * n_theta, hipMemcpyHostToDevice, sm));
  HIP_TRY(ctx, hipMemsetAsync(st->d_status, 0, sizeof(int), sm));
  // attributes of the blocks theta does not carry start from the ctx mixture
  std::vector<double> aux(L.n_aux);
  memcpy(aux.data(), ctx->mu.data(), sizeof(double) * K * D);
  memcpy(aux.data() + K * D, ctx->sigma.data(), sizeof(double) * K);
  memcpy(aux.data() + K * D + K, ctx->lambd.data(), sizeof(double) * D);
  memcpy(aux.data() + K * D + K + D, ctx->w.data(), sizeof(double) * K);
  memcpy(aux.data() + K * D + 2 * K + D, ctx->eta.data(), sizeof(double) * K);
  HIP_TRY(ctx, hipMemcpyAsync(sb + L.o_aux(), aux.data(), sizeof(double) * L.n_aux, hipMemcpyHostToDevice, sm));
  HIP_TRY(ctx, hipMemcpyAsync(sb + L.o_hyp(), ctx->gp.d_hyp, sizeof(double) * L.n_hyp, hipMemcpyDeviceToDevice, sm));
  if (st->has_bnd) {
    HIP_TRY(ctx, hipMemcpyAsync(sb + L.o_blb(), opts->bnd_lb, sizeof(double) * st->n_bnd, hipMemcpyHostToDevice, sm));
    HIP_TRY(ctx, hipMemcpyAsync(sb + L.o_bub(), opts->bnd_ub, sizeof(double) * st->n_bnd, hipMemcpyHostToDevice, sm));
  }
  if (st->has_box) {
    HIP_TRY(ctx, hipMemcpyAsync(sb + L.o_xlb(), lb, sizeof(double) * n_theta, hipMemcpyHostToDevice, sm));
    HIP_TRY(ctx, hipMemcpyAsync(sb + L.o_xub(), ub, sizeof(double) * n_theta, hipMemcpyHostToDevice, sm));
  }
  HIP_TRY(ctx, hipStreamSynchronize(sm));  // `aux` is pageable stack-adjacent memory
  ctx->pack_in_flight = false;
  AdamDev a;
  fill_dev(ctx, *st, a);
  launch_step(*st, sm, a, 0);
  HIP_TRY(ctx, hipGetLastError());
  // draws one iteration ahead (Philox mode, unless switched off or too large)
  {
    const char* off = getenv("VBMC_ADAM_PREGEN");
    const size_t n_eps = (size_t)K * (size_t)st->row_count * D;
    st->pregen = st->eps_mode == VBMC_EPS_PHILOX && !(off && off[0] == '0') && n_eps > 0 &&
                 n_eps <= ((size_t)1 << 28);  // <= 2 GiB per buffer
    if (st->pregen) {
      if (!st->gen_stream) {
        int lo = 0, hi = 0;
        HIP_TRY(ctx, hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIP_TRY(ctx, hipStreamCreateWithPriority(&st->gen_stream, hipStreamNonBlocking, lo));
        for (int b = 0; b < 2; ++b) {
          HIP_TRY(ctx, hipEventCreateWithFlags(&st->ev_gen[b], hipEventDisableTiming));
          HIP_TRY(ctx, hipEventCreateWithFlags(&st->ev_ent[b], hipEventDisableTiming));
        }
      }
      if (st->eps2_cap < n_eps) {
        HIP_TRY(ctx, hipStreamSynchronize(st->gen_stream));
        for (int b = 0; b < 2; ++b) {
          if (st->d_eps2[b]) HIP_TRY(ctx, hipFree(st->d_eps2[b]));
          st->d_eps2[b] = nullptr;
          HIP_TRY(ctx, hipMalloc((void**)&st->d_eps2[b], sizeof(double) * n_eps));
        }
        st->eps2_cap = n_eps;
      }
      st->ent_recorded[0] = st->ent_recorded[1] = false;
      rc = enqueue_gen(ctx, st, 0);
      if (rc) return rc;
    }
  }
  st->active = true;
  return VBMC_OK;
}

extern "C" int vbmc_adam_run(vbmc_ctx* ctx, int n_iters, double* y_tab_out, double* x_tab_out,
                             double* G_out, double* H_out) {
  if (!ctx || n_iters < 0) return VBMC_E_ARG;
  NEED_DEVICE(ctx);
  AdamState* st = (AdamState*)ctx->adam;
  if (!st || !st->active) return vbmc_fail(ctx, VBMC_E_ARG, "adam_run: vbmc_adam_begin not called");
  if (st->iter + n_iters > st->max_iter)
    return vbmc_fail(ctx, VBMC_E_ARG, "adam_run: %d + %d iterations exceed max_iter %d", st->iter,
                     n_iters, st->max_iter);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int n = st->n_theta;
  static const bool force_coll = [] {
    const char* e = getenv("VBMC_FORCE_COLLECTIVE");
    return e && e[0] == '1';
  }();
  const bool multi = ctx->comm != nullptr && (ctx->world > 1 || force_coll);
  struct TimingOff {  // no per-kernel event pairs inside the loop; restored on every exit path
    vbmc_ctx* c;
    bool was;
    explicit TimingOff(vbmc_ctx* c_) : c(c_), was(c_->timing) { c->timing = false; }
    ~TimingOff() { c->timing = was; }
  } timing_off(ctx);
  AdamDev a;
  fill_dev(ctx, *st, a);
  const int i0 = st->iter;
  int rc = 0;
  for (int it = 0; it < n_iters && rc == 0; ++it) {
    const int i = i0 + it;
    PrepArgs pa;
    glj_fill_prep(ctx, 1, st->state + st->lay.o_res(), nullptr, pa);
    EntPlan plan;
    rc = entmc_plan(ctx, st->ns, st->pregen ? VBMC_EPS_RESIDENT : st->eps_mode, st->seed + (uint64_t)i,
                    st->row_begin, st->row_count, 1, plan);
    if (rc) break;
    const int b = i & 1;
    if (st->pregen) {
      plan.a.eps = st->d_eps2[b];
      plan.a.eps_rows = st->row_count;
    }
    entmc_fill_prep(ctx, plan, pa);
    rc = launch_prep(ctx, pa);
    if (rc) break;
    if (st->pregen) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, st->ev_gen[b], 0));
    rc = entmc_launch_main(ctx, plan);
    if (rc) break;
    if (st->pregen) {
      HIP_TRY(ctx, hipEventRecord(st->ev_ent[b], ctx->stream));
      st->ent_recorded[b] = true;
      if (i + 1 < st->max_iter) {
        rc = enqueue_gen(ctx, st, i + 1);  // overlaps the entropy kernel just launched
        if (rc) break;
      }
    }
    rc = entmc_launch_finish(ctx, plan, st->state + st->lay.o_raw());
    if (rc) break;
    if (multi) {
      rc = comm_allreduce_sum(ctx, st->state + st->lay.o_raw(), raw_len(ctx->D, ctx->K));
      if (rc) break;
    }
    // minimize_adam.py:92-98
    a.c1 = 1.0 / (1.0 - std::pow(a.beta1, (double)(i + 1)));
    a.c2 = 1.0 / (1.0 - std::pow(a.beta2, (double)(i + 1)));
    a.step = st->master_min + (st->master_max - st->master_min) * std::exp(-(double)(i + 1) / st->master_decay);
    a.x_row = st->x_tab + (size_t)i * n;
    a.y_out = st->y_tab + 3 * (size_t)i;
    launch_step(*st, ctx->stream, a, 1);
  }
  if (rc) return rc;
  HIP_TRY(ctx, hipGetLastError());
  st->iter = i0 + n_iters;
  int status = 0;
  std::vector<double> y3(3 * (size_t)n_iters);
  if (n_iters > 0) {
    HIP_TRY(ctx, hipMemcpyAsync(y3.data(), st->y_tab + 3 * (size_t)i0, sizeof(double) * 3 * n_iters,
                                hipMemcpyDeviceToHost, ctx->stream));
    if (x_tab_out)
      HIP_TRY(ctx, hipMemcpyAsync(x_tab_out, st->x_tab + (size_t)i0 * n, sizeof(double) * (size_t)n_iters * n,
                                  hipMemcpyDeviceToHost, ctx->stream));
  }
  HIP_TRY(ctx, hipMemcpyAsync(&status, st->d_status, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  for (int it = 0; it < n_iters; ++it) {
    if (y_tab_out) y_tab_out[it] = y3[3 * (size_t)it];
    if (G_out) G_out[it] = y3[3 * (size_t)it + 1];
    if (H_out) H_out[it] = y3[3 * (size_t)it + 2];
  }
  if (status) {
    st->active = false;
    return vbmc_fail(ctx, VBMC_E_NONFINITE, "adam_run: an iterate became non-finite");
  }
  return VBMC_OK;
}

extern "C" int vbmc_adam_end(vbmc_ctx* ctx, double* theta_out, double* mu_KxD, double* sigma_K,
                             double* lambd_D, double* w_K, double* eta_K, int* iterations) {
  if (!ctx) return VBMC_E_ARG;
  NEED_DEVICE(ctx);
  AdamState* st = (AdamState*)ctx->adam;
  if (!st || !st->active) return vbmc_fail(ctx, VBMC_E_ARG, "adam_end: no optimisation in progress");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int D = ctx->D, K = ctx->K;
  const size_t n_aux = aux_len(D, K);
  std::vector<double> aux(n_aux), th(st->n_theta);
  HIP_TRY(ctx, hipMemcpyAsync(aux.data(), st->state + st->lay.o_aux(), sizeof(double) * n_aux, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(th.data(), st->state + st->lay.o_theta(), sizeof(double) * st->n_theta, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (st->gen_stream) HIP_TRY(ctx, hipStreamSynchronize(st->gen_stream));
  // the device pack in d_mix is the mixture of the last iterate: make the host copies agree
  ctx->mu.assign(aux.begin(), aux.begin() + K * D);
  ctx->sigma.assign(aux.begin() + K * D, aux.begin() + K * D + K);
  ctx->lambd.assign(aux.begin() + K * D + K, aux.begin() + K * D + K + D);
  ctx->w.assign(aux.begin() + K * D + K + D, aux.begin() + K * D + 2 * K + D);
  ctx->eta.assign(aux.begin() + K * D + 2 * K + D, aux.end());
  ctx->pack_valid = true;
  if (theta_out) memcpy(theta_out, th.data(), sizeof(double) * st->n_theta);
  if (mu_KxD) memcpy(mu_KxD, ctx->mu.data(), sizeof(double) * K * D);
  if (sigma_K) memcpy(sigma_K, ctx->sigma.data(), sizeof(double) * K);
  if (lambd_D) memcpy(lambd_D, ctx->lambd.data(), sizeof(double) * D);
  if (w_K) memcpy(w_K, ctx->w.data(), sizeof(double) * K);
  if (eta_K) memcpy(eta_K, ctx->eta.data(), sizeof(double) * K);
  if (iterations) *iterations = st->iter;
  st->active = false;
  return VBMC_OK;
}
