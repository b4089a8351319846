// Device-side pieces of the Adam loop (adam.hip) that other translation units need: the state
// layout, the argument block, and the body of the "pre" workgroup, which the wave-split entropy
// kernel (entropy_ws.hip) runs as an extra row of its own launch.
#pragma once
#include "common.h"
#include "fastmath.h"

namespace adam_dev {

// One contiguous block of doubles holds the optimiser state.  The prefix up to o_raw() is what the
// side-stream kernel (adam_pre_kernel) reads and mirrors into LDS with one deep batch of loads.
struct AdamLayout {
  int n = 0, n_aux = 0, n_hyp = 0, n_raw = 0, n_res = 0, n_bnd = 0;
  __host__ __device__ int o_theta() const { return 0; }               // [n] current x
  __host__ __device__ int o_aux() const { return n; }                 // mu K*D | sigma K | lambd D | w K | eta K
  __host__ __device__ int o_hyp() const { return o_aux() + n_aux; }   // [S][P]
  __host__ __device__ int o_res() const { return o_hyp() + n_hyp; }   // [S][K][1+2D] GP sums
  __host__ __device__ int o_blb() const { return o_res() + n_res; }   // soft bounds
  __host__ __device__ int o_bub() const { return o_blb() + n_bnd; }
  __host__ __device__ int o_raw() const { return o_bub() + n_bnd; }   // normalised entropy accumulator
  __host__ __device__ int o_m() const { return o_raw() + n_raw; }     // Adam first moment
  __host__ __device__ int o_v() const { return o_m() + n; }
  __host__ __device__ int o_xlb() const { return o_v() + n; }         // box
  __host__ __device__ int o_xub() const { return o_xlb() + n; }
  __host__ __device__ int end() const { return o_xub() + n; }
};

struct AdamDev {
  MixLayout ml;
  AdamLayout lay;
  int D, K, S, P, mean_kind, mask, n_theta, n_bnd, has_box, has_bnd;
  double* mix;    // mixture pack of the current iterate (rewritten for the next one)
  double* state;  // AdamLayout block
  double* work;   // adam_pre_kernel scratch when it does not fit the LDS: see work_len()
  double* ee;     // [K] adam_step_kernel scratch when it does not use the LDS
  double* pre;    // [n_theta + 2]: the part of dF that does not depend on the entropy, then G, loss
  double tol_con, w_thresh, w_pen;
  double fudge, beta1, beta2;
  double c_norm;  // 1 / (2 pi)^(D/2)
  double l2_beta1, l2_beta2, l2e_over_decay;  // log2 beta1, log2 beta2, log2(e) / master_decay
  double master_min, master_max, master_decay;  // step-size schedule (minimize_adam.py:92-98)
  // The iteration index is iter_base[0] + it_off: the base lives in device memory (set once per
  // vbmc_adam_run call) and the offset is a launch constant, so the launch arguments of a batch
  // do not depend on where in the optimisation it starts.  (That makes a batch replayable as a
  // hipGraph; measured, the replay gains nothing -- 4 launches cost the host ~20 us against
  // >= 50 us of device time per iteration -- so batches are launched directly.)
  const int* iter_base;
  int it_off;
  double* x_tab;  // [max_iter][n_theta]
  double* y_tab;  // [max_iter][3]: y, G, H
  int* status;    // != 0: a non-finite iterate was produced
};

// Argument block of the fused loop (adam_fused.hip): a batch of iterations in one launch.
struct FusedArgs {
  AdamDev a;
  const double* XT = nullptr;     // [D][N]
  const double* alpha = nullptr;  // [S][N]
  int N = 0;
  int eps_mode = 0;               // VBMC_EPS_PHILOX: Philox(seed + iteration), else the resident block eps
  const double* eps = nullptr;    // [K][eps_rows][D]
  long long eps_rows = 0, n_half = 0, row_begin = 0;
  int rows = 0;                   // antithetic rows per entropy WORKGROUP (adam_fused_plan: in = per component)
  int rows_total = 0;             // antithetic rows per component
  int cus = 0;                    // compute units of the device (every workgroup of the launch must be resident)
  unsigned long long seed = 0;
  double inv_ns = 0.0;
  double* xch = nullptr;                 // [2][K (2 + 2D + K) + S K (2D + 4)] exchange records
  double* backup = nullptr;              // [o_hyp + 2 n_theta + pack]: theta | aux, m, v and the pack as this launch found them
  unsigned long long* flags = nullptr;   // [n_ent + n_gp], zeroed before the launch: the iteration a workgroup has published
  unsigned long long timeout = 2000000;  // wall-clock ticks (100 MHz) a workgroup waits for the others: 20 ms
  unsigned long long* times = nullptr;   // optional [2][64][16] phase stamps (VBMC_FUSED_TIMES=1)
  int n_ent = 0, n_gp = 0;        // workgroups: K R entropy (R row slices per component) + n_gp GP-sum workers
  int i0 = 0, n_iters = 0;
  int stop_rule = 0;              // 1: every workgroup applies minimize_adam's stopping rule itself (vbmc_adam_run_auto)
  double tol_fun = 0.0;
  int* n_done = nullptr;          // iterations this launch ran (written by workgroup 0)
  int rel_acq = 0;                // option "adam_fused" = 3: the flags as release stores / acquire fences (adam_fused.hip, exchange)
  int test_absent = 0;            // test hook (option "adam_fused" = 2): also wait for a workgroup that does not exist
  int o_pack = 0, o_ee = 0, o_recs = 0, o_eps = 0, o_part = 0, o_out = 0, o_gp = 0, o_xt = 0,
      o_alpha = 0;  // LDS carve, in doubles (adam_fused_plan)
  int part_waves = 8;  // waves whose per-lane sums are laid down side by side in the entropy workgroups' reduction scratch (adam_fused_plan)
};
size_t adam_fused_plan(FusedArgs& f);
int adam_fused_launch(vbmc_ctx* ctx, hipStream_t st, const FusedArgs& f, size_t lds_bytes);

__host__ __device__ inline size_t aux_len(int D, int K) { return (size_t)K * D + 3 * (size_t)K + D; }
// scratch: ell2, iom2 [S][D] | gmu, tgs, tnu [K][D] | gsg, gw, ee [K] | glm, bl [D] | dL [n_bnd]
__host__ __device__ inline size_t work_len(int D, int K, int S, int n_bnd) {
  return 2 * (size_t)S * D + 3 * (size_t)K * D + 3 * (size_t)K + 2 * (size_t)D + (size_t)n_bnd;
}

static __device__ __forceinline__ double wave_sum(double v) {
  return fm::wave_sum_dpp(v);
}

// ---------------------------------------------------------------------------
// The "pre" workgroup: everything of one iteration's dF that does NOT depend on the Monte-Carlo
// entropy.  It runs as an extra row of the entropy launch (entropy_ws.hip), i.e. beside the entropy
// workgroups and ordered after the GP sums by the stream alone -- no second stream and no events,
// whose cross-queue latency (6-13 us each on MI355X) would cost more than the work itself.  Only
// the short adam_step_kernel (adam.hip) then sits between two entropy launches.
//   GP-sum finalisation (host twin: api_gp.hip glj_finalize), soft bounds + weight penalty
//   (api_elbo.hip), their Jacobians, and the softmax Jacobian of the entropy-free part of the
//   weight gradient (the Jacobian is linear, so the entropy part is added by the step kernel).
// LDS = true: the state prefix [theta | aux | hyp | res] and the scratch live in LDS (the 2 n_bnd doubles of the soft
// bounds, read once each, do not: with them the working set of BASELINE config 3 was 62 KB, past what two entropy
// workgroups per CU leave, and the workgroup ran from global memory at twice the time);
// compile-time so that every access is a true ds_* or global access (a pointer that may be
// either at run time makes the compiler emit flat_* instructions, several times slower on LDS).
// PRELOADED (with LDS): sh already holds the state prefix (the fused loop, adam_fused.hip, keeps it there
// across iterations); pre_out: where the result goes (a.pre, or that loop's LDS copy).
// UB: soft-bound pairs requested together per thread (1 inside the wave-split entropy kernel, whose two-waves-per-SIMD builds
// have no registers to spare: eight there cost the HEADLINE kernel its second wave -- 272 registers, "failed to meet
// occupancy target"; 8 in the matrix-pipe kernel and the stand-alone launch)
template <bool LDS, bool PRELOADED = false, int UB = 1>
static __device__ void adam_pre_body(const AdamDev& a, double* sh, double* red, double* pre_out) {
  const int D = a.D, K = a.K, S = a.S, tid = threadIdx.x, n = a.n_theta;
  const int lane = tid & 63, wave = tid >> 6;
  const int st = 1 + 2 * D;
  const AdamLayout& L = a.lay;
  if (LDS && !PRELOADED) {
    const int cnt = L.o_blb();  // theta | aux | hyp | res -- the soft bounds are read once each, straight from memory
    constexpr int U = 24;  // BASELINE config 3 (4 493 doubles) in one batch of loads
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
    __syncthreads();
  }
  const double* base = LDS ? sh : a.state;
  const double* theta = base + L.o_theta();
  const double* aux = base + L.o_aux();
  const double* hyp = base + L.o_hyp();
  const double* res = base + L.o_res();
  const double* bnd_lb = a.state + L.o_blb();
  const double* bnd_ub = a.state + L.o_bub();
  double* work = LDS ? sh + L.o_blb() : a.work;

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
  double* gw = glm + D;             // [K]     entropy-free pre-Jacobian weight gradient of F
  double* ee = gw + K;              // [K]
  double* tgs = ee + K;             // [K][D]  per-(k,d) terms of d G / d sigma
  double* tnu = tgs + K * D;        // [K][D]  per-(k,d) terms of the quadratic-mean part
  double* bl = tnu + K * D;         // [D]     soft-bound gradient folded onto lambda
  double* dL = bl + D;              // [n_bnd]
  const bool quad = a.mean_kind == VBMC_MEAN_NEGQUAD;

  for (int i = tid; i < S * D; i += 256) {
    const int s = i / D, d = i - s * D;
    const double* h = hyp + (size_t)s * a.P;
    ell2[i] = fm::exp2_fast(2.0 * 0x1.71547652b82fep+0 * h[d]);  // exp(2 h_d)
    iom2[i] = quad ? fm::exp2_fast(-2.0 * 0x1.71547652b82fep+0 * h[2 * D + 3 + d]) : 0.0;
  }
  __syncthreads();

  // ---- phase 1 (all 256 lanes, two independent jobs):
  //  * GP expected log joint, per (k, d) term (host twin: api_gp.hip glj_finalize)
  //  * soft bounds (_vp_bound_loss :537-606): gradient dL and this thread's share of the loss
  // (1/tau by v_rsq + Newton and 1/S as a factor: a float64 division is a ~20-instruction dependent
  // chain, and this workgroup is nothing but dependent chains)
  const double inv_S = 1.0 / S;
  for (int idx = tid; idx < K * D; idx += 256) {
    const int k = idx / D, d = idx - k * D;
    const double sgk = sg[k], wk = w[k], lam = lm[d], m = mu[idx];
    double gm_acc = 0.0, gs_acc = 0.0, nu_acc = 0.0;
    for (int s = 0; s < S; ++s) {
      const double* h = hyp + (size_t)s * a.P;
      const double* r = res + ((size_t)s * K + k) * st;
      const double tau2 = sgk * sgk * lam * lam + ell2[s * D + d];
      const double rtau = fm::rsqrt_fast(tau2);
      const double U = r[1 + d], T = r[1 + D + d] - r[0];
      double gm = wk * (-U * rtau);
      gs_acc += (lam * lam * (rtau * rtau)) * T * inv_S;
      if (quad) {
        const double xm = h[D + 3 + d], io = iom2[s * D + d];
        gm -= wk * io * (m - xm);
        nu_acc += io * (m * m + sgk * sgk * lam * lam - 2.0 * m * xm + xm * xm) * inv_S;
      }
      gm_acc += gm * inv_S;
    }
    gmu[idx] = gm_acc;
    tgs[idx] = gs_acc;
    tnu[idx] = nu_acc;
  }
  double loss = 0.0;
  if (a.has_bnd) {
    const int n_mu = o_mu ? D * K : 0, n_sc = (o_sg || o_lm) ? D * K : 0;
    // (UB > 1: the bounds of UB entries per thread are requested together: one pair per round makes this loop a chain of
    // n_bnd / 256 memory latencies -- sixteen at K = 100, D = 20, where this workgroup IS its launch)
    for (int b0 = 0; b0 < a.n_bnd; b0 += 256 * UB) {
      double lbv[UB], ubv[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int ic = min(b0 + u * 256 + tid, a.n_bnd - 1);
        lbv[u] = bnd_lb[ic];
        ubv[u] = bnd_ub[ic];
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int i = b0 + u * 256 + tid;
        if (i >= a.n_bnd) continue;
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
        const double lb = lbv[u], ub = ubv[u];
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
    }
  }
  __syncthreads();

  // ---- phase 2: per component (lanes k), then per dimension in 16-lane groups ----
  double gpart = 0.0, ps = 0.0, pd = 0.0;
  for (int k = tid; k < K; k += 256) {
    const double sgk = sg[k], wk = w[k];
    double gs = 0.0, nu = 0.0, b0 = 0.0, qbar = 0.0;
    for (int d = 0; d < D; ++d) {
      gs += tgs[k * D + d];
      nu += tnu[k * D + d];
    }
    for (int s = 0; s < S; ++s) {
      const double* h = hyp + (size_t)s * a.P;
      b0 += (res[((size_t)s * K + k) * st] + (a.mean_kind == VBMC_MEAN_ZERO ? 0.0 : h[D + 2])) * inv_S;
      if (quad)
        for (int d = 0; d < D; ++d) qbar += iom2[s * D + d] * lm[d] * lm[d] * inv_S;
    }
    const double wI = b0 - 0.5 * nu;  // mean over s of I_sk
    gpart += wk * wI;
    gsg[k] = wk * sgk * (gs - qbar);
    double g = -wI;  // d(-G)/dw_k; the entropy part is the step kernel's
    if (a.has_bnd && o_w) {  // weight penalty (:1211-1229)
      const bool small = wk < a.w_thresh;
      loss += (small ? wk : a.w_thresh) * a.w_pen;
      if (small) g += a.w_pen;
    }
    gw[k] = g;
    if (o_w) {  // softmax Jacobian of the entropy-free weight gradient (entmc_vbmc.py:122-130)
      const double e = fm::exp2_fast(0x1.71547652b82fep+0 * eta[k]);
      ee[k] = e;
      ps += e;
      pd += e * g;
    }
  }
  {
    // 16 groups of 16 lanes, one dimension per group and round: the lambda gradient over the
    // (s, k) terms and the soft-bound gradient folded onto lambda (the reference reshapes the
    // scale block C-order (D,K), :585-587; restated as-is)
    const int ns = tid & 15, g16 = tid >> 4;
    const int sc0 = o_mu ? D * K : 0;
    for (int d = g16; d < D; d += 16) {
      const double lam = lm[d];
      double acc = 0.0, accb = 0.0;
      for (int idx = ns; idx < S * K; idx += 16) {
        const int s = idx / K, k = idx - s * K;
        const double* r = res + (size_t)idx * st;
        const double sgk = sg[k], wk = w[k];
        const double tau2 = sgk * sgk * lam * lam + ell2[s * D + d];
        const double T = r[1 + D + d] - r[0];
        double gl = wk * (sgk * sgk * fm::rcp_fast(tau2)) * lam * T;
        if (quad) gl -= wk * sgk * sgk * iom2[s * D + d] * lam;
        acc += gl * inv_S;
      }
      if (a.has_bnd && o_lm)
        for (int k = ns; k < K; k += 16) accb += dL[sc0 + d * K + k];
      acc = fm::row16_sum_dpp(acc);
      accb = fm::row16_sum_dpp(accb);
      if (ns == 0) {
        glm[d] = acc;
        bl[d] = accb;
      }
    }
  }
  // one reduction round for the four block sums
  gpart = wave_sum(gpart);
  loss = wave_sum(loss);
  ps = wave_sum(ps);
  pd = wave_sum(pd);
  if (lane == 0) {
    red[wave] = gpart;
    red[4 + wave] = loss;
    red[8 + wave] = ps;
    red[12 + wave] = pd;
  }
  __syncthreads();  // also orders gsg / gw / ee / glm / bl before the reads below
  const double G = (red[0] + red[1]) + (red[2] + red[3]);
  loss = (red[4] + red[5]) + (red[6] + red[7]);
  const double sm_s = o_w ? (red[8] + red[9]) + (red[10] + red[11]) : 1.0;
  const double sm_dot = o_w ? (red[12] + red[13]) + (red[14] + red[15]) : 0.0;

  // ---- the entropy-free part of dF ----
  const int sc0 = o_mu ? D * K : 0;
  for (int i = tid; i < n; i += 256) {
    double g;
    if (o_mu && i < D * K) {
      g = -gmu[i];
      if (a.has_bnd) g += dL[i];
    } else if (o_sg && i >= p_sg && i < p_sg + K) {
      const int k = i - p_sg;
      g = -gsg[k] * sg[k];
      if (a.has_bnd) {
        // the reference reshapes this block C-order (D,K) (:585-587); restated as-is
        double acc = 0.0;
        for (int d = 0; d < D; ++d) acc += dL[sc0 + d * K + k];
        g += acc;
      }
    } else if (o_lm && i >= p_lm && i < p_lm + D) {
      const int d = i - p_lm;
      g = -glm[d] * lm[d];
      if (a.has_bnd) g += bl[d];
    } else {
      const int k = i - p_w;
      g = -ee[k] * sm_dot / (sm_s * sm_s) + ee[k] * gw[k] / sm_s;
      if (a.has_bnd) g += dL[a.n_bnd - K + k];
    }
    pre_out[i] = g;
  }
  if (tid == 0) {
    pre_out[n] = G;
    pre_out[n + 1] = loss;
  }
}

// set_parameters(theta) + eta max-shift + mixture pack; theta's eta tail is shifted in
// place.  theta / aux may live in LDS; the pack goes to p (a.mix, or the fused loop's LDS copy).  Two reduction rounds:
// (sum lambda^2, max eta), then (sum exp(eta - max), prod lambda); red needs 4 NT / 64 doubles.
// NT: threads of the workgroup (256: the launches of adam.hip; 512: the fused loop).
template <int NT = 256>
static __device__ void pack_from_theta(const AdamDev& a, double* theta, double* aux, double* red, double* p) {
  constexpr int NW = NT / 64;
  const int D = a.D, K = a.K, tid = threadIdx.x, n = a.n_theta;
  const int lane = tid & 63, wave = tid >> 6;
  const bool o_mu = a.mask & 1, o_sg = a.mask & 2, o_lm = a.mask & 4, o_w = a.mask & 8;
  const int p_sg = o_mu ? D * K : 0, p_lm = p_sg + (o_sg ? K : 0), p_w = n - K;
  double* mu = aux;
  double* sg = mu + K * D;
  double* lm = sg + K;
  double* w = lm + D;
  double* eta = w + K;
  int bad = 0;
  for (int i = tid; i < n; i += NT) bad |= !isfinite(theta[i]);
  if (bad) atomicOr(a.status, 1);
  // ---- round 1: raw lambda and its sum of squares; max of the eta tail ----
  double s2 = 0.0, mx = -INFINITY;
  for (int d = tid; d < D; d += NT) {
    const double l = o_lm ? fm::exp2_fast(0x1.71547652b82fep+0 * theta[p_lm + d]) : lm[d];  // exp(.)
    lm[d] = l;
    s2 = fma(l, l, s2);
  }
  if (o_w)
    for (int k = tid; k < K; k += NT) mx = fmax(mx, theta[p_w + k]);
  s2 = wave_sum(s2);
  mx = fm::wave_max_dpp(mx);
  __syncthreads();
  if (lane == 0) {
    red[wave] = s2;
    red[NW + wave] = mx;
  }
  __syncthreads();
  s2 = (red[0] + red[1]) + (red[2] + red[3]);
  mx = fmax(fmax(red[NW], red[NW + 1]), fmax(red[NW + 2], red[NW + 3]));
  if (NW == 8) {
    s2 += (red[4] + red[5]) + (red[6] + red[7]);
    mx = fmax(mx, fmax(fmax(red[NW + 4], red[NW + 5]), fmax(red[NW + 6], red[NW + 7])));
  }
  const double nl = sqrt(s2 / D);  // lambda -> unit RMS, sigma absorbs it
  const double inl = 1.0 / nl;
  // ---- round 2: unnormalised weights and their sum; product of the normalised lambdas ----
  double wsum = 0.0, pr = 1.0;
  if (o_w)
    for (int k = tid; k < K; k += NT) {
      const double e = theta[p_w + k] - mx;
      theta[p_w + k] = e;
      eta[k] = e;
      const double we = fm::exp2_fast(0x1.71547652b82fep+0 * e);
      w[k] = we;
      wsum += we;
    }
  for (int d = tid; d < D; d += NT) pr *= lm[d] * inl;  // this thread's own entries of round 1
  wsum = wave_sum(wsum);
  pr = fm::wave_prod_dpp(pr);
  if (lane == 0) {
    red[2 * NW + wave] = wsum;
    red[3 * NW + wave] = pr;
  }
  __syncthreads();
  wsum = (red[2 * NW] + red[2 * NW + 1]) + (red[2 * NW + 2] + red[2 * NW + 3]);
  pr = (red[3 * NW] * red[3 * NW + 1]) * (red[3 * NW + 2] * red[3 * NW + 3]);
  if (NW == 8) {
    wsum += (red[2 * NW + 4] + red[2 * NW + 5]) + (red[2 * NW + 6] + red[2 * NW + 7]);
    pr *= (red[3 * NW + 4] * red[3 * NW + 5]) * (red[3 * NW + 6] * red[3 * NW + 7]);
  }
  const double nconst = a.c_norm / pr;  // 1 / (2 pi)^(D/2) / prod(lambda)   (entmc_vbmc.py:54-56)
  const double l2n = 0x1.71547652b82fep+0 * fm::log_fast(nconst);
  // ---- the pack and the final attributes (lm stays raw until every reader is through) ----
  const MixLayout& ml = a.ml;
  for (int i = tid; i < K * D; i += NT) {
    const int d = i % D;
    const double m = o_mu ? theta[i] : mu[i];
    mu[i] = m;
    p[ml.o_mu + i] = m;
    p[ml.o_mup + i] = m * fm::rcp_fast(lm[d] * inl);
  }
  for (int k = tid; k < K; k += NT) {
    const double s = (o_sg ? fm::exp2_fast(0x1.71547652b82fep+0 * theta[p_sg + k]) : sg[k]) * nl;
    const double wk = o_w ? w[k] / wsum : w[k];
    double sD = 1.0, b = s;  // sigma^D by repeated squaring, as the host pack (ctx.hip)
    for (int e = D; e > 0; e >>= 1) {
      if (e & 1) sD *= b;
      b *= b;
    }
    sg[k] = s;
    w[k] = wk;
    const double rsD = nconst * fm::rcp_fast(sD);
    p[ml.o_is2 + k] = fm::rcp_fast(s * s);
    p[ml.o_rc + k] = rsD;
    p[ml.o_lrc + k] = l2n - D * (0x1.71547652b82fep+0 * fm::log_fast(s));
    p[ml.o_wc + k] = wk * rsD;
    p[ml.o_sig + k] = s;
    p[ml.o_w + k] = wk;
  }
  __syncthreads();
  for (int d = tid; d < D; d += NT) {
    const double l = lm[d] * inl;
    lm[d] = l;
    p[ml.o_lam + d] = l;
    p[ml.o_ilam + d] = fm::rcp_fast(l);
  }
}


}  // namespace adam_dev
