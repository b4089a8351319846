// The optimiser loop at the reference's OWN sample counts: one launch per batch of iterations -- or per optimisation.
//
// `optimize_vp` runs minimize_adam (vbmc/minimize_adam.py:84-137) around _neg_elcbo
// (variational_optimization.py:238-249) with ns_ent = 100 K^(2/3) samples in total
// (option_configs/advanced_vbmc_options.ini:43): 28 per component at K = 50, 14 antithetic rows.
// At that size every kernel of the four-launch iteration (adam.hip) sits at its latency floor:
// 7.5 + 11 + 5 + 7 = 30 us, of which hardly 5 are arithmetic.  Here K + min(S K, 128) workgroups of 512
// threads stay resident and meet once per iteration:
//
//   phase A (workgroups side by side, from the mixture pack every workgroup holds in its LDS)
//     workgroup j < K       Monte-Carlo entropy sums of component j (entropy_small.hip's form: lane =
//                           component k, the eight waves split the rows); its (j,k) table rows are made in
//                           place, its rows' Philox normals were made during the previous exchange; the
//                           2D + 1 per-lane sums go through LDS once and wave c % 8 finishes item c and
//                           stores the record entry; the entries of the raw gradient that only need
//                           component j (mu_j, sigma_j) are FINISHED here
//     workgroup K + q       GP expected-log-joint sums of blocks (s,k) = q, q + n_gp, ... (the arithmetic of
//                           glj_block.h, X^T and alpha resident in its LDS), then the block's CONTRIBUTIONS
//                           to the entropy-free part of dF (adam_dev::adam_pre_body is linear in them)
//   exchange                every workgroup stores its record write-through (sc1), drains the stores and
//                           sets its own flag to the iteration number; while the flags travel: the next
//                           normals, the soft bounds, the step size; wave 0 polls the flags, then everybody
//                           reads ALL records (sc1 loads, ten in flight) into LDS: one all-gather of
//                           K (2 + 2D + K) + S K (2D + 4) doubles (37 KB at K = 50, D = 10), no fence,
//                           two buffers alternating between iterations
//   phase B (every workgroup, redundantly and identically, all operands in LDS / registers)
//                           the sums over j and (s,k), dF per entry, the Adam update with the moments in
//                           registers, set_parameters + the pack of the next iterate (pack_waves), and --
//                           with f.stop_rule -- minimize_adam's stopping rule every 20 iterations
//
// so an iteration has ONE inter-workgroup exchange and no launch boundary, no table in memory, no draw
// buffer, no partial rows, and the host is not in the loop.  Workgroup 0 writes the iterate and (y, G, H)
// rows, and the state back at the end.  Every spin is bounded (a workgroup that never publishes -- it
// cannot happen with <= 192 workgroups on 256 CUs, but a hung GPU is not an acceptable failure mode --
// raises status bit 4 and every workgroup leaves without writing the state back; the host then runs
// four launches per iteration from the same state).
//
// Used when the loop runs on one rank, K <= 64, D <= 24 (round 5: was 16), at most 64 antithetic rows per component and
// the LDS plan fits (adam_fused_plan); everything else keeps the four-launch iteration.  Same draws
// (Philox(seed + i), philox.h), same formulas: tests/test_adam.py runs both against the oracle loop.
// Measured steps and the phase times: DESIGN.md section 4.6b.
#include <cstdio>
#include <cstdlib>

#include "adam_dev.h"
#include "common.h"
#include "entropy_args.h"
#include "fastmath.h"
#include "philox.h"

using namespace adam_dev;

namespace {

constexpr int SW = 8;        // waves per workgroup: two per SIMD (a float64 instruction every ~4 cycles instead of ~7)
constexpr int NT = 64 * SW;  // threads
constexpr double LOG2E = 0x1.71547652b82fep+0;

__device__ __forceinline__ void st_wt(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_wt(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void drain() { __builtin_amdgcn_s_waitcnt(0x0F70); }  // vmcnt(0)

// a wave-uniform double moved to scalar registers (an LDS read lands in vector registers)
__device__ __forceinline__ double to_sgpr(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// the SW per-wave partials r[0 .. SW) in a fixed order
__device__ __forceinline__ double sumw(const double* r) {
  double v = (r[0] + r[1]) + (r[2] + r[3]);
  if (SW == 8) v += (r[4] + r[5]) + (r[6] + r[7]);
  return v;
}

// adam_dev::pack_from_theta for K <= 64, D <= 64 (lane = d): set_parameters (variational_posterior.py:680-759) with the
// eta max-shift (variational_optimization.py:1082-1085) and the mixture pack.  EVERY wave computes the
// per-component and per-dimension quantities itself (lane = k, lane = d) and so owns the four wave-wide
// reductions: one barrier (before theta's eta tail is shifted in place) instead of five.  The waves share the
// stores.  theta, aux and p are LDS arrays.
__device__ void pack_waves(const AdamDev& a, double* theta, double* aux, double* p, int tid) {
  const int D = a.D, K = a.K, n = a.n_theta;
  const int lane = tid & 63, wave = tid >> 6;
  const bool o_mu = a.mask & 1, o_sg = a.mask & 2, o_lm = a.mask & 4, o_w = a.mask & 8;
  const int p_sg = o_mu ? D * K : 0, p_lm = p_sg + (o_sg ? K : 0), p_w = n - K;
  double* mu = aux;
  double* sg = mu + K * D;
  double* lm = sg + K;
  double* w = lm + D;
  double* eta = w + K;
  const MixLayout& ml = a.ml;
  int bad = 0;
  for (int i = tid; i < n; i += NT) bad |= !isfinite(theta[i]);
  if (bad) atomicOr(a.status, 1);
  const bool isd = lane < D, isk = lane < K;
  // raw lambda (lane = d), the eta tail (lane = k)
  const double l_raw = isd ? (o_lm ? fm::exp2_fast(LOG2E * theta[p_lm + lane]) : lm[lane]) : 0.0;
  const double th_w = (o_w && isk) ? theta[p_w + lane] : -INFINITY;
  const double th_s = (o_sg && isk) ? theta[p_sg + lane] : 0.0;
  const double sg_old = isk ? sg[lane] : 1.0, w_old = isk ? w[lane] : 0.0;
  const double s2 = fm::wave_sum_dpp(l_raw * l_raw);
  const double mx = fm::wave_max_dpp(th_w);
  const double nl = sqrt(s2 / D);  // lambda -> unit RMS, sigma absorbs it
  const double inl = 1.0 / nl;
  const double e = th_w - mx;
  const double we = (o_w && isk) ? fm::exp2_fast(LOG2E * e) : 0.0;
  const double wsum = fm::wave_sum_dpp(we);
  const double l_n = l_raw * inl;
  const double pr = fm::wave_prod_dpp(isd ? l_n : 1.0);
  const double nconst = a.c_norm / pr;  // 1 / (2 pi)^(D/2) / prod(lambda)   (entmc_vbmc.py:54-56)
  const double l2n = LOG2E * fm::log_fast(nconst);
  __syncthreads();  // every wave has read theta's eta tail and the old attributes
  // per component (lane = k): wave 0 stores
  if (wave == 0 && isk) {
    const int k = lane;
    const double s = (o_sg ? fm::exp2_fast(LOG2E * th_s) : sg_old) * nl;
    const double wk = o_w ? we / wsum : w_old;
    double sD = 1.0, b = s;  // sigma^D by repeated squaring, as the host pack (ctx.hip)
    for (int ex = D; ex > 0; ex >>= 1) {
      if (ex & 1) sD *= b;
      b *= b;
    }
    if (o_w) {
      theta[p_w + k] = e;
      eta[k] = e;
    }
    sg[k] = s;
    w[k] = wk;
    const double rsD = nconst * fm::rcp_fast(sD);
    p[ml.o_is2 + k] = fm::rcp_fast(s * s);
    p[ml.o_rc + k] = rsD;
    p[ml.o_lrc + k] = l2n - D * (LOG2E * fm::log_fast(s));
    p[ml.o_wc + k] = wk * rsD;
    p[ml.o_sig + k] = s;
    p[ml.o_w + k] = wk;
  }
  // per dimension (lane = d): wave 1 stores
  if (wave == 1 && isd) {
    lm[lane] = l_n;
    p[ml.o_lam + lane] = l_n;
    p[ml.o_ilam + lane] = fm::rcp_fast(l_n);
  }
  // the means (K D entries): waves 2 .. SW - 1; 1 / lambda_d from the lane that holds dimension d
  {
    const double il = fm::rcp_fast(l_n);  // lane d: 1 / (lambda_d / nl)
    for (int base = (wave - 2) * 64; wave >= 2 && base < K * D; base += (SW - 2) * 64) {
      const int i = base + lane;
      const int d = i % D;
      const int ilo = __builtin_amdgcn_ds_bpermute(d << 2, __double2loint(il));
      const int ihi = __builtin_amdgcn_ds_bpermute(d << 2, __double2hiint(il));
      if (i < K * D) {
        const double m = o_mu ? theta[i] : mu[i];
        mu[i] = m;
        p[ml.o_mu + i] = m;
        p[ml.o_mup + i] = m * __hiloint2double(ihi, ilo);
      }
    }
  }
}

template <int DP>
__global__ __launch_bounds__(NT) void adam_fused_kernel(FusedArgs f) {
  extern __shared__ double sh[];
  __shared__ double red[9 * SW];
  __shared__ double red2[SW][128];
  __shared__ int s_ok;
  const AdamDev& a = f.a;
  const AdamLayout& L = a.lay;
  const MixLayout& ml = a.ml;
  const int D = a.D, K = a.K, S = a.S, n = a.n_theta, N = f.N;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = blockIdx.x, G = gridDim.x;
  const bool o_mu = a.mask & 1, o_sg = a.mask & 2, o_lm = a.mask & 4, o_w = a.mask & 8;
  const int p_sg = o_mu ? D * K : 0, p_lm = p_sg + (o_sg ? K : 0), p_w = n - K;
  const int RE = 2 + 2 * D + K;       // entropy record: slog | raw mu_j (D) | raw sigma_j | lam_j (D) | W_j (K)
  const int RC = 2 * D + 4;           // GP contribution record: gmu (D) | glm (D) | gs | nu | b0 | qbar
  const int n_blocks = S * K;
  const int RT = K * RE + n_blocks * RC;  // the gathered records (LDS)
  // Round 5: R = n_ent / K workgroups per component, each with a slice of the component's rows (plane r = g / K holds the
  // partial records of slice r: every entry of an entropy record is a sum over rows, so the gather adds the planes up).
  const int R = f.n_ent / K;
  const int KRE = K * RE;
  const int RTX = R * KRE + n_blocks * RC;  // the exchange buffer: R planes of entropy records, then the GP records

  // ---- LDS carve: [theta | aux | hyp] as in the state block, then this kernel's own arrays ----
  double* theta = sh + L.o_theta();
  double* aux = sh + L.o_aux();
  const double* hyp = sh + L.o_hyp();
  double* work = sh + L.o_res();  // phase B scratch (the LDS image of the state ends with the hyper-parameters)
  double* pack = sh + f.o_pack;
  double* ee = sh + f.o_ee;
  double* recs = sh + f.o_recs;
  double* sE = sh + f.o_eps;
  double* part = sh + f.o_part;   // entropy workgroups' reduction scratch, laid over recs | gsc | sXT | sAl (dead / unused there)
  double* out = sh + f.o_out;      // [1 + 2D] a GP block's sums
  double* gsc = sh + f.o_gp;
  // X^T: resident in the GP workgroups' LDS where the plan has room for it, else read from memory (L2) by both passes
  const double* sXT = f.o_xt >= 0 ? sh + f.o_xt : f.XT;
  double* sAl = sh + f.o_alpha;

  // ---- what persists across the iterations of this launch ----
  for (int i = tid; i < L.o_res(); i += NT) sh[i] = a.state[i];  // theta | aux | hyp
  for (int i = tid; i < ml.total; i += NT) pack[i] = a.mix[i];
  // Adam's moments of this thread's entries of theta (n_theta <= NU NT, adam_fused_plan; the third entry only in the
  // builds for D > 16, whose theta reaches 1 536 entries: the narrower builds keep their register count)
  constexpr int NU = DP > 16 ? 3 : 2;
  double r_m[NU], r_v[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int i = u * NT + tid;
    r_m[u] = i < n ? a.state[L.o_m() + i] : 0.0;
    r_v[u] = i < n ? a.state[L.o_v() + i] : 0.0;
  }
  // minimize_adam's stopping rule (minimize_adam.py:107-140), applied by every workgroup to the same numbers: the sums of
  // its theta entries over the current and the previous batch of 20 iterations, the batch's objective values
  constexpr int BATCH = 20;
  __shared__ double ywin[BATCH];
  // loop-invariant doubles of phase B: read from LDS where they are used (as values hoisted out of the iteration loop
  // they sat in vector registers across phase A, or in scratch)
  __shared__ double cst[6];
  if (tid == 0) {
    cst[0] = 1.0 - a.beta1;
    cst[1] = 1.0 - a.beta2;
    cst[2] = 1.0 / S;
    const double tol_max = f.tol_fun * 100.0;
    cst[3] = f.tol_fun * f.tol_fun;
    cst[4] = tol_max * tol_max;
    cst[5] = a.master_max - a.master_min;
  }
  int n_ran = f.n_iters;
  if (g == 0 && f.backup) {  // what a launch that gives up is rolled back to (adam.hip fused_restore)
    double* b = f.backup;
    for (int i = tid; i < L.o_hyp(); i += NT) b[i] = a.state[i];
    b += L.o_hyp();
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int i = u * NT + tid;
      if (i < n) {
        b[i] = r_m[u];
        b[n + i] = r_v[u];
      }
    }
    b += 2 * n;
    for (int i = tid; i < ml.total; i += NT) b[i] = a.mix[i];
  }
  if (g >= f.n_ent) {
    if (f.o_xt >= 0)
      for (int i = tid; i < D * N; i += NT) sh[f.o_xt + i] = f.XT[i];
    for (int i = tid; i < S * N; i += NT) sAl[i] = f.alpha[i];
  }
  __syncthreads();


  // phase stamps of workgroups 0 (entropy) and n_ent (GP sums), VBMC_FUSED_TIMES=1: a measurement aid
  const int tslot = f.times == nullptr ? -1 : (g == 0 ? 0 : (g == f.n_ent ? 1 : -1));
  auto stamp = [&](int t, int p) {
    if (tslot >= 0 && tid == 0 && t < 64) f.times[((size_t)tslot * 64 + t) * 16 + p] = wall_clock64();
  };

  // the normals of component g's rows at iteration `it` (philox.h: row j n_half + row_begin + i, block d / 4)
  const int pl = g < f.n_ent ? g / K : 0, jc = g - pl * K;  // entropy workgroup: its plane (row slice) and component
  const int row0 = pl * f.rows;                              // first row of the slice; rows_here of them
  const int rows_here = g < f.n_ent ? (f.rows_total - row0 < f.rows ? (f.rows_total - row0 > 0 ? f.rows_total - row0 : 0) : f.rows) : 0;
  auto make_draws = [&](int it, int tid) {
    const int rows = rows_here;
    if (f.eps_mode == VBMC_EPS_PHILOX) {
      const int nb = (D + 3) >> 2;
      const uint64_t seed = f.seed + (uint64_t)it;
      // two threads per Philox block, one Box-Muller pair each (the block itself is computed twice: it is the short part)
      for (int q = tid; q < 2 * rows * nb; q += NT) {
        const int h = q & 1, ib = q >> 1, i = ib / nb, b = ib - i * nb;
        const uint64_t row = (uint64_t)jc * (uint64_t)f.n_half + (uint64_t)(f.row_begin + row0 + i);
        const Philox4 r = philox4x32_10((uint32_t)row, (uint32_t)(row >> 32), (uint32_t)b, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
        double z0, z1;
        philox_bm32(h ? r.x[2] : r.x[0], h ? r.x[3] : r.x[1], z0, z1);
        const int d0 = 4 * b + 2 * h;
        if (d0 < D) sE[i * D + d0] = z0;
        if (d0 + 1 < D) sE[i * D + d0 + 1] = z1;
      }
    } else if (it == f.i0) {  // resident draws: the same block every iteration
      const double* src = f.eps + ((int64_t)jc * f.eps_rows + row0) * D;
      for (int i = tid; i < rows * D; i += NT) sE[i] = src[i];
    }
  };
  if (g < f.n_ent) make_draws(f.i0, tid);

  const int tid_launch = tid;
  for (int t = 0; t < f.n_iters; ++t) {
    // (an opaque copy of the thread index ties everything derived from it to the iteration: hoisted out of this long loop,
    // the address arithmetic of its ~40 inner loops was worth 75 spilled registers)
    int tid;
    asm volatile("v_mov_b32 %0, %1" : "=v"(tid) : "v"(tid_launch));
    const int lane = tid & 63;
    const int iter = f.i0 + t;
    double* xb = f.xch + (size_t)(t & 1) * RTX;
    stamp(t, 0);

    if (g < f.n_ent) {
      // ================= phase A, entropy of component j (entropy_small.hip; entmc_vbmc.py:64-112) =================
      const int j = jc, k = lane;
      const bool live = k < K;
      const int rows = rows_here;
      // the lane's table row (prep.hip's table block, from the pack in LDS)
      const double* mup = pack + ml.o_mup;
      double sj2, two_sj;
      {
        const double sig_j = pack[ml.o_sig + j];
        sj2 = to_sgpr(sig_j * sig_j);
        two_sj = to_sgpr(2.0 * sig_j);
      }
      double dl[DP], d2s = 0.0;
#pragma unroll
      for (int d = 0; d < DP; ++d) {
        const double v = (d < D && live) ? (mup[j * D + d] - mup[k * D + d]) : 0.0;
        dl[d] = v;
        d2s += v * v;
      }
      double c0 = -2000.0, ak = 0.0, wk = 0.0, wis2 = 0.0;
      if (live) {
        const double is2 = pack[ml.o_is2 + k];
        wk = pack[ml.o_w + k];
        ak = -0.5 * LOG2E * is2;
        c0 = fma(ak, d2s, pack[ml.o_lrc + k]);
        wis2 = wk * is2;
      }
      double slog = 0.0, W = 0.0, A[DP], B[DP];
#pragma unroll
      for (int d = 0; d < DP; ++d) A[d] = B[d] = 0.0;
      __syncthreads();
      stamp(t, 9);
      for (int i = wave; i < rows; i += SW) {
        const double* rp = sE + i * D;
        double e[DP], e2 = 0.0;
#pragma unroll
        for (int d = 0; d < DP; ++d) {
          // (a row's normals are the same in every lane: the builds for D > 16 keep them in scalar registers -- 40 / 48 vector
          // registers they do not have: with them and Delta_jk in vector registers those builds carried 152 / 276 B of scratch per
          // thread, ran 1.5 / 3.5 us per iteration faster -- and about one launch in 10^5 gave the one-launch form up after its
          // 20 ms wait, which no scratch-free build has ever done: profiles/r05_notes.md)
          e[d] = (d < D) ? (DP > 16 ? to_sgpr(rp[d]) : rp[d]) : 0.0;
          e2 = fma(e[d], e[d], e2);
        }
        const double bq = sj2 * e2;
        double c = 0.0;
        if constexpr (DP > 16) {
          // (the builds for D > 16 do not hold Delta_jk across the row loop either: re-formed from the pack in LDS, behind an
          // opaque zero offset so that the reads stay inside the loop -- 40 / 48 more vector registers)
          int zoff;
          asm volatile("v_mov_b32 %0, 0" : "=v"(zoff));
          const double* mq = mup + zoff;
#pragma unroll
          for (int d = 0; d < DP; ++d)
            if (d < D) c = fma(live ? mq[j * D + d] - mq[k * D + d] : 0.0, e[d], c);
        } else {
#pragma unroll
          for (int d = 0; d < DP; ++d) c = fma(dl[d], e[d], c);
        }
        const double sp = fma(two_sj, c, bq), sm = fma(-two_sj, c, bq);
        const double r1 = fm::exp2_fast(fma(ak, sp, c0)), r2 = fm::exp2_fast(fma(ak, sm, c0));
        const double qp = fm::wave_sum_dpp(wk * r1), qm = fm::wave_sum_dpp(wk * r2);
        slog = to_sgpr(slog + (fm::log_fast(qp) + fm::log_fast(qm)));  // (the same value in every lane)
        const double t1 = r1 * fm::rcp_fast(qp), t2 = r2 * fm::rcp_fast(qm);
        const double ts = t1 + t2, td = t1 - t2;
        W += ts;
        const double gs = ts * wis2, gd = td * wis2;
#pragma unroll
        for (int d = 0; d < DP; ++d) {
          A[d] = fma(e[d], gd, A[d]);
          B[d] = fma(e[d] * e[d], gs, B[d]);
        }
      }
      // component j's entries need, summed over the lanes (= k) and the waves (= rows):
      //   mu_d  : sigma_j A_d(k) + w_k/sigma_k^2 Delta_jk,d W(k)       (entropy.hip finish, mu_from_w)
      //   lam_d : sigma_j B_d(k) + Delta_jk,d A_d(k)
      // every wave lays its 2D per-lane values down in LDS, then wave c % SW sums item c over the waves and its lanes
      stamp(t, 10);
      const int NI = 2 * D + 1;
      asm volatile("" ::: "memory");  // (sigma_j is read again rather than held across the row loop: the <16> build's last spill)
      const double sig_j = pack[ml.o_sig + j];
      // (f.part_waves = SW / 2 where [SW][2D + 1][K] does not fit the LDS plan -- D = 20, K = 50: 131 KB --: the upper half of
      // the waves lays its values down first, the lower half adds its own to them behind a barrier)
      const int PWV = DP > 16 ? f.part_waves : SW;  // (a constant in the builds for D <= 16: they keep their code)
      double* pc = part;                       // [PWV][2D + 1][K]
      double* pW = pc + (size_t)PWV * NI * K;  // [SW][64]
      double* pS = pW + SW * 64;               // [SW] slog
      for (int pass = (PWV == SW ? 1 : 0); pass < 2; ++pass) {
        const bool mine = PWV == SW || (pass == 0 ? wave >= PWV : wave < PWV);
        const bool add = PWV != SW && pass == 1;
        const int slot = wave >= PWV ? wave - PWV : wave;
        if (live && mine) {
          double sl = 0.0;
          double* pq = pc + (size_t)slot * NI * K + lane;
#pragma unroll
          for (int d = 0; d < DP; ++d)
            if (d < D) {
              const double dld = DP > 16 ? mup[j * D + d] - mup[k * D + d] : dl[d];  // (live lanes only here)
              const double cl = fma(sig_j, B[d], dld * A[d]);
              const double cm = fma(sig_j, A[d], (wis2 * dld) * W);
              pq[(size_t)d * K] = add ? pq[(size_t)d * K] + cm : cm;
              pq[(size_t)(D + d) * K] = add ? pq[(size_t)(D + d) * K] + cl : cl;
              sl += cl;
            }
          pq[(size_t)(2 * D) * K] = add ? pq[(size_t)(2 * D) * K] + sl : sl;  // item 2D: sigma_j's entry is the sum of the lambda items
        }
        if (pass == 0) __syncthreads();
      }
      pW[wave * 64 + lane] = W;
      if (lane == 0) pS[wave] = slog;
      __syncthreads();
      stamp(t, 11);
      // item c is summed over the waves and the lanes by wave c % SW, whose lane 0 stores the record entry itself
      {
        double* rec = xb + (size_t)g * RE;  // plane pl, component j
        const double sc = pack[ml.o_w + j] * f.inv_ns;
        for (int c = wave; c < NI; c += SW) {
          double v = 0.0;
          if (live) {
#pragma unroll
            for (int wv = 0; wv < SW; ++wv)
              if (wv < PWV) v += pc[((size_t)wv * NI + c) * K + lane];
          }
          v = fm::wave_sum_dpp(v);
          if (lane == 0) {
            if (c < D) st_wt(rec + 1 + c, v * sc * pack[ml.o_ilam + c]);
            else if (c < 2 * D) st_wt(rec + 2 + c, v);  // 2 + D + (c - D)
            else st_wt(rec + 1 + D, v * sc);
          }
        }
        if (wave == SW - 1 && live) {
          double Wk = 0.0;
#pragma unroll
          for (int wv = 0; wv < SW; ++wv) Wk += pW[wv * 64 + lane];
          st_wt(rec + 2 + 2 * D + lane, Wk);
        }
        if (wave == SW - 2 && lane == 0) st_wt(rec, sumw(pS));
      }
    } else {
      // ================= phase A, GP sums of blocks (s,k) (glj_block.h; variational_optimization.py:1400-1465) =================
      double* sItau = gsc;
      double* sMu = sItau + D;
      double* sZa = sMu + D;
      double* sPart = sZa + N;
      double* sMisc = sPart + SW;
      for (int b = g - f.n_ent; b < n_blocks; b += f.n_gp) {
        const int s = b / K, k = b - s * K;
        const double* h = hyp + (size_t)s * a.P;
        const double sigk = pack[ml.o_sig + k];
        if (tid < 64) {
          double term = 0.0;
          for (int d = tid; d < D; d += 64) {
            const double ell = fm::exp2_fast(LOG2E * h[d]);
            const double lam = pack[ml.o_lam + d];
            const double tau2 = sigk * sigk * lam * lam + ell * ell;
            sItau[d] = fm::rsqrt_fast(tau2);
            sMu[d] = pack[ml.o_mu + k * D + d];
            term += h[d] - 0.5 * fm::log_fast(tau2);
          }
          term = fm::wave_sum_dpp(term);
          if (tid == 0) sMisc[0] = 2.0 * h[D] + term;
        }
        __syncthreads();
        const double lnnf = sMisc[0];
        for (int nn = tid; nn < N; nn += NT) {
          double d2 = 0.0;
          for (int d = 0; d < D; ++d) {
            const double dlt = (sMu[d] - sXT[d * N + nn]) * sItau[d];
            d2 = fma(dlt, dlt, d2);
          }
          const double z = fm::exp2_fast(LOG2E * (lnnf - 0.5 * d2));
          sZa[nn] = z * sAl[s * N + nn];
        }
        __syncthreads();
        {
          double acc = 0.0;
          for (int nn = tid; nn < N; nn += NT) acc += sZa[nn];
          acc = fm::wave_sum_dpp(acc);
          if (lane == 0) sPart[wave] = acc;
        }
        {
          const int ns = tid & 15, ds = tid >> 4;
          for (int d = ds; d < D; d += NT / 16) {
            const double m = sMu[d], itau = sItau[d];
            double au = 0.0, at = 0.0;
            for (int nn = ns; nn < N; nn += 16) {
              const double dlt = (m - sXT[d * N + nn]) * itau;
              const double tt = dlt * sZa[nn];
              au += tt;
              at = fma(dlt, tt, at);
            }
            au = fm::row16_sum_dpp(au);
            at = fm::row16_sum_dpp(at);
            if (ns == 0) {
              out[1 + d] = au;
              out[1 + D + d] = at;
            }
          }
        }
        __syncthreads();
        // this block's CONTRIBUTIONS to the entropy-free part of dF (adam_dev::adam_pre_body, phases 1-2: every term of
        // it is linear in the per-(s,k) quantities, so the sums over s and k are left to phase B): lane = d
        if (wave == 0) {
          double r0 = (sPart[0] + sPart[1]) + (sPart[2] + sPart[3]);
          if (SW == 8) r0 += (sPart[4] + sPart[5]) + (sPart[6] + sPart[7]);
          const bool quad = a.mean_kind == VBMC_MEAN_NEGQUAD;
          const double inv_S = cst[2];
          const double wk = pack[ml.o_w + k];
          double c_gs = 0.0, c_nu = 0.0, c_qb = 0.0;
          double* rec = xb + (size_t)R * KRE + (size_t)b * RC;
          if (lane < D) {
            const int d = lane;
            const double lam = pack[ml.o_lam + d], m = sMu[d];
            const double ell2 = fm::exp2_fast(2.0 * LOG2E * h[d]);  // exp(2 h_d)
            const double io = quad ? fm::exp2_fast(-2.0 * LOG2E * h[2 * D + 3 + d]) : 0.0;
            const double tau2 = sigk * sigk * lam * lam + ell2;
            const double rtau = fm::rsqrt_fast(tau2);
            const double Ud = out[1 + d], T = out[1 + D + d] - r0;
            double gm = wk * (-Ud * rtau);
            c_gs = (lam * lam * (rtau * rtau)) * T * inv_S;
            double gl = wk * (sigk * sigk * fm::rcp_fast(tau2)) * lam * T;
            if (quad) {
              const double xm = h[D + 3 + d];
              gm -= wk * io * (m - xm);
              c_nu = io * (m * m + sigk * sigk * lam * lam - 2.0 * m * xm + xm * xm) * inv_S;
              gl -= wk * sigk * sigk * io * lam;
              c_qb = io * lam * lam * inv_S;
            }
            st_wt(rec + d, gm * inv_S);
            st_wt(rec + D + d, gl * inv_S);
          }
          c_gs = fm::wave_sum_dpp(c_gs);
          c_nu = fm::wave_sum_dpp(c_nu);
          c_qb = fm::wave_sum_dpp(c_qb);
          if (lane == 0) {
            st_wt(rec + 2 * D, c_gs);
            st_wt(rec + 2 * D + 1, c_nu);
            st_wt(rec + 2 * D + 2, (r0 + (a.mean_kind == VBMC_MEAN_ZERO ? 0.0 : h[D + 2])) * inv_S);
            st_wt(rec + 2 * D + 3, c_qb);
          }
        }
        __syncthreads();
      }
    }

    // ================= exchange: drained write-through records, one flag per workgroup, all-gather =================
    stamp(t, 1);
    // Default: every record entry is a write-through store (st_wt: past the XCD's L2, which the other XCDs do not
    // snoop) and every gathered one a load that bypasses the caches (ld_wt), so the flag itself needs no fence -- the
    // wave's stores have left (vmcnt(0)) before the workgroup's barrier, the flag is stored after it.  rel_acq = 1 is
    // the same exchange written with the memory model's own words: the flag is an agent-scope RELEASE store (the
    // compiler adds the write-back of the L2's dirty lines, buffer_wbl2 sc1), the reader runs an agent-scope ACQUIRE
    // fence behind the spin (buffer_inv sc1: its L2 and L1 drop what they hold).  Measured at K = 50, NsK = 28:
    // DESIGN.md section 4.6b; the default stays the relaxed form, the fenced one is the documented fallback
    // (option "adam_fused" = 3) and tests/test_adam.py runs both.
    drain();
    __syncthreads();
    if (tid == 0) {
      if (f.rel_acq) __hip_atomic_store(f.flags + g, (unsigned long long)(t + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      else __hip_atomic_store(f.flags + g, (unsigned long long)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    stamp(t, 2);

    // (a second opaque copy: what phase B derives from the thread index is not computed ahead of phase A and carried
    // across its register-hungry loop -- with one zero per iteration that was 100 B of scratch per thread at D = 16)
    int tid_b;
    asm volatile("v_mov_b32 %0, %1" : "=v"(tid_b) : "v"(tid_launch));
    {
    const int tid = tid_b, lane = tid & 63;
    // ---- while the flags travel: the next iteration's normals, and the soft bounds (_vp_bound_loss :537-606),
    // which need theta only ----
    if (g < f.n_ent && t + 1 < f.n_iters) make_draws(iter + 1, tid);
    double* dL = work;              // [n_bnd]
    double* gsg = dL + a.n_bnd;     // [K]
    double* gw = gsg + K;           // [K]
    double* glm = gw + K;           // [D]
    double* bl = glm + D;           // [D]
    const double* mu = aux;
    const double* sg = mu + K * D;
    const double* lm = sg + K;
    const double* wv = lm + D;
    const double* eta = wv + K;
    // (and the iteration's bias corrections and step size, minimize_adam.py:92-98: off the path behind the gather)
    const double it1 = (double)(iter + 1);
    const double c1 = 1.0 / (1.0 - fm::exp2_fast(it1 * a.l2_beta1));
    const double c2 = 1.0 / (1.0 - fm::exp2_fast(it1 * a.l2_beta2));
    const double step = a.master_min + cst[5] * fm::exp2_fast(-it1 * a.l2e_over_decay);
    if (o_w)
      for (int k = tid; k < K; k += NT) ee[k] = fm::exp2_fast(LOG2E * eta[k]);  // softmax terms of the current iterate
    double loss = 0.0;
    if (a.has_bnd) {
      // (the bounds stay in memory: 2 n_bnd doubles of LDS are what would limit N in adam_fused_plan; requesting them
      // ahead of the normals, into registers, was measured and cost more in spills than the ~0.3 us it hides)
      const double* bnd_lb = a.state + L.o_blb();
      const double* bnd_ub = a.state + L.o_bub();
      const int n_mu = o_mu ? D * K : 0, n_sc = (o_sg || o_lm) ? D * K : 0;
      for (int i = tid; i < a.n_bnd; i += NT) {
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
        double gg = 0.0;
        if (x < lb) {
          const double tt = (lb - x) / ell;
          loss += 0.5 * tt * tt;
          gg = (x - lb) / (ell * ell);
        }
        if (x > ub) {
          const double tt = (x - ub) / ell;
          loss += 0.5 * tt * tt;
          gg = (x - ub) / (ell * ell);
        }
        dL[i] = gg;
      }
    }

    if (wave == 0) {
      const unsigned long long want = (unsigned long long)(t + 1);
      const unsigned long long t0 = wall_clock64();
      int ok = 1;
      for (;;) {
        bool all = true;
        for (int q = lane; q < G + f.test_absent; q += 64)
          all = all && __hip_atomic_load(f.flags + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want;
        if (__all(all)) break;
        // (the first exchange also waits for every workgroup to be DISPATCHED: beside another queue's launches that
        // keep the CUs' LDS taken -- tests/test_adam.py's soak -- that has been seen to take longer than 20 ms; 10x)
        if (__builtin_amdgcn_readfirstlane((int)((wall_clock64() - t0) > (t == 0 ? 10 * f.timeout : f.timeout)))) {
          ok = 0;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      if (lane == 0) {
        if (!ok) atomicOr(a.status, 4);
        s_ok = ok;
      }
    }
    __syncthreads();  // (also: dL complete)
    if (!s_ok) return;
    if (f.rel_acq) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    stamp(t, 3);
    if (R == 1) {
      for (int base = 0; base < RT; base += NT * 10) {
        double v[10];
#pragma unroll
        for (int u = 0; u < 10; ++u) v[u] = ld_wt(xb + min(base + u * NT + tid, RT - 1));
#pragma unroll
        for (int u = 0; u < 10; ++u) {
          const int i = base + u * NT + tid;
          if (i < RT) recs[i] = v[u];  // entropy records, then the GP contribution records
        }
      }
    } else {
      // the planes of an entropy entry are added up in plane order (R <= 4: twenty loads in flight)
      for (int base = 0; base < RT; base += NT * 5) {
        double v[5][4];
#pragma unroll
        for (int u = 0; u < 5; ++u) {
          const int i = min(base + u * NT + tid, RT - 1);
#pragma unroll
          for (int r = 0; r < 4; ++r)
            v[u][r] = (r == 0 || (r < R && i < KRE)) ? ld_wt(xb + (i < KRE ? (size_t)r * KRE + i : (size_t)(R - 1) * KRE + i)) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 5; ++u) {
          const int i = base + u * NT + tid;
          if (i < RT) recs[i] = ((v[u][0] + v[u][1]) + v[u][2]) + v[u][3];
        }
      }
    }
    __syncthreads();
    stamp(t, 4);
    const double* crec = recs + K * RE;  // [S K][2D + 4]: gmu (D) | glm (D) | gs | nu | b0 | qbar

    // ================= phase B (every workgroup): dF and the update, from the records in LDS =================
    // stage 1: every wave takes its share of the sums over j (H, lambda_d, w_u); beside that, three independent
    // jobs, each a chain of LDS round trips, on three groups of waves
    const int n_out = 1 + D + K;
    for (int o = lane; o < n_out; o += 64) {
      // lane = output, the waves split the components j
      const int col = o == 0 ? 0 : (o <= D ? 2 + D + (o - 1) : 2 + 2 * D + (o - 1 - D));
      double acc = 0.0;
      for (int j = wave; j < K; j += SW) {
        const double wj = pack[ml.o_w + j];
        const double cf = (o >= 1 && o <= D) ? wj * pack[ml.o_sig + j] : wj;
        acc = fma(cf, recs[j * RE + col], acc);
      }
      red2[wave][o] = acc;
    }
    if (wave == 0) {
      // per component (lane = k; K <= 64): the sums over s, the entropy-free weight gradient, the softmax terms
      double gpart = 0.0, ps = 0.0, pd = 0.0;
      if (lane < K) {
        const int k = lane;
        const double sgk = sg[k], wk = wv[k];
        double gs = 0.0, nu = 0.0, b0 = 0.0, qbar = 0.0;
        for (int sidx = 0; sidx < S; ++sidx) {
          const double* c = crec + (size_t)(sidx * K + k) * RC + 2 * D;
          gs += c[0];
          nu += c[1];
          b0 += c[2];
          qbar += c[3];
        }
        const double wI = b0 - 0.5 * nu;  // mean over s of I_sk
        gpart = wk * wI;
        gsg[k] = wk * sgk * (gs - qbar);
        double gg = -wI;  // d(-G)/dw_k; the entropy part is added by the update below
        if (a.has_bnd && o_w) {  // weight penalty (:1211-1229)
          const bool small = wk < a.w_thresh;
          loss += (small ? wk : a.w_thresh) * a.w_pen;
          if (small) gg += a.w_pen;
        }
        gw[k] = gg;
        if (o_w) {
          const double e = ee[k];
          ps = e;
          pd = e * gg;
        }
      }
      gpart = fm::wave_sum_dpp(gpart);
      ps = fm::wave_sum_dpp(ps);
      pd = fm::wave_sum_dpp(pd);
      if (lane == 0) {
        red[0] = gpart;
        red[2 * SW] = ps;
        red[3 * SW] = pd;
      }
    } else if (wave <= 4) {
      // per dimension (a 16-lane group each, sixteen groups; D > 16: a second round): lambda's sums over (s, k), the soft
      // bounds folded onto lambda
      const int ns = tid & 15;
      const int sc0 = o_mu ? D * K : 0;
      for (int d = (tid - 64) >> 4; d < D; d += 16) {
        double acc = 0.0, accb = 0.0;
        for (int idx = ns; idx < n_blocks; idx += 16) acc += crec[(size_t)idx * RC + D + d];
        if (a.has_bnd && o_lm)
          for (int k = ns; k < K; k += 16) accb += dL[sc0 + d * K + k];
        acc = fm::row16_sum_dpp(acc);
        accb = fm::row16_sum_dpp(accb);
        if (ns == 0) {
          glm[d] = acc;
          bl[d] = accb;
        }
      }
    } else if (o_w) {
      // the softmax Jacobian of the entropy's weight gradient needs sum_k e_k raw_k with raw_k = -(slog_k + sum_j w_j W_jk) / ns:
      // summed the other way round, sum_j [e_j slog_j + w_j sum_k e_k W_jk], it needs no finished raw_k.  lane = j, the
      // three waves split the k range
      const int part = wave - 5, k0 = (K * part) / 3, k1 = (K * (part + 1)) / 3;
      double term = 0.0;
      if (lane < K) {
        const double* rj = recs + (size_t)lane * RE;
        double d0 = 0.0, d1 = 0.0;
        int k = k0;
        for (; k + 1 < k1; k += 2) {
          d0 = fma(ee[k], rj[2 + 2 * D + k], d0);
          d1 = fma(ee[k + 1], rj[2 + 2 * D + k + 1], d1);
        }
        if (k < k1) d0 = fma(ee[k], rj[2 + 2 * D + k], d0);
        term = pack[ml.o_w + lane] * (d0 + d1);
        if (part == 0) term = fma(ee[lane], rj[0], term);
      }
      term = fm::wave_sum_dpp(term);
      if (lane == 0) red[6 * SW + part] = term;
    }
    if (a.has_bnd) {
      loss = fm::wave_sum_dpp(loss);
      if (lane == 0) red[SW + wave] = loss;
    }
    stamp(t, 5);
    __syncthreads();
    stamp(t, 6);
    // stage 2: per entry of theta, dF = entropy-free part + Jacobian(entropy gradient) (entmc_vbmc.py:114-130) and the
    // Adam update with the box clamp (minimize_adam.py:89-105)
    {
      const double Gv = red[0], lossv = a.has_bnd ? sumw(red + SW) : 0.0;
      const double sm_s = o_w ? red[2 * SW] : 1.0;       // sum_k e_k
      const double pm_dot = o_w ? red[3 * SW] : 0.0;     // sum_k e_k gw_k
      const double sm_dot = o_w ? -f.inv_ns * ((red[6 * SW] + red[6 * SW + 1]) + red[6 * SW + 2]) : 0.0;  // sum_k e_k raw_k
      const int sc0 = o_mu ? D * K : 0;
      auto out_sum = [&](int o) {
        double sum = (red2[0][o] + red2[1][o]) + (red2[2][o] + red2[3][o]);
        if (SW == 8) sum += (red2[4][o] + red2[5][o]) + (red2[6][o] + red2[7][o]);
        return sum;
      };
      if (tid == 0) {
        const double H = -out_sum(0) * f.inv_ns;
        const double y = -Gv - H + lossv;
        ywin[iter % BATCH] = y;
        if (g == 0) {
          double* y_out = a.y_tab + 3 * (size_t)iter;
          y_out[0] = y;
          y_out[1] = Gv;
          y_out[2] = H;
        }
      }
      double* x_row = a.x_tab + (size_t)iter * n;
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int i = u * NT + tid;
        if (i >= n) continue;
        double gr;  // dF_i
        if (o_mu && i < D * K) {
          const int k = i / D, d = i - k * D;
          double gm = 0.0;
          for (int sidx = 0; sidx < S; ++sidx) gm += crec[(size_t)(sidx * K + k) * RC + d];
          double gg = -gm;
          if (a.has_bnd) gg += dL[i];
          gr = gg - recs[k * RE + 1 + d];
        } else if (o_sg && i >= p_sg && i < p_sg + K) {
          const int k = i - p_sg;
          double gg = -gsg[k] * sg[k];
          if (a.has_bnd) {
            // the reference reshapes this block C-order (D,K) (:585-587); restated as-is
            double acc = 0.0;
            for (int d = 0; d < D; ++d) acc += dL[sc0 + d * K + k];
            gg += acc;
          }
          gr = gg - recs[k * RE + 1 + D] * sg[k];
        } else if (o_lm && i >= p_lm && i < p_lm + D) {
          const int d = i - p_lm;
          double gg = -glm[d] * lm[d];
          if (a.has_bnd) gg += bl[d];
          const double rr = out_sum(1 + d) * f.inv_ns * pack[ml.o_ilam + d];
          gr = gg - rr * lm[d];
        } else {
          const int k = i - p_w;
          const double e = ee[k];
          double gg = -e * pm_dot / (sm_s * sm_s) + e * gw[k] / sm_s;
          if (a.has_bnd) gg += dL[a.n_bnd - K + k];
          const double rr = -f.inv_ns * (recs[k * RE] + out_sum(1 + D + k));
          gr = gg + (e * sm_dot / (sm_s * sm_s) - e * rr / sm_s);
        }
        const double m = a.beta1 * r_m[u] + cst[0] * gr;
        const double v = a.beta2 * r_v[u] + cst[1] * (gr * gr);
        r_m[u] = m;
        r_v[u] = v;
        const double m_hat = m * c1, v_hat = v * c2;
        double x = theta[i] - step * m_hat / (sqrt(v_hat) + a.fudge);
        if (a.has_box) x = fmin(a.state[L.o_xub() + i], fmax(a.state[L.o_xlb() + i], x));  // (minimize_adam's lb / ub: not optimize_vp's path)
        theta[i] = x;
        if (g == 0) {
          if (f.stop_rule) st_wt(x_row + i, x);  // read back by every workgroup at the end of the batch (below)
          else x_row[i] = x;
        }
      }
      __syncthreads();
    }
    bool stop = false;
    if (f.stop_rule && (iter + 1) % BATCH == 0) {
      // the slope of a straight-line fit through the batch's objective values against its standard error (np.polyfit's:
      // residual sum / (n - 2) / sum t^2) and the distance between the mean iterates of the last two batches
      if (iter + 1 >= 2 * BATCH) {
        double part = 0.0;
        // the sums of the last two batches' iterates, added up in iteration order from the rows workgroup 0 has written
        // (complete before the flag of the iteration after them, i.e. before this iteration's gather; the last row is
        // this iteration's own update, the same bits in every workgroup)
        for (int i = tid; i < n; i += NT) {
          const double* col = a.x_tab + i;
          double xp = 0.0, xc = 0.0;
          for (int q = iter + 1 - 2 * BATCH; q <= iter - BATCH; ++q) xp += ld_wt(col + (size_t)q * n);
          for (int q = iter + 1 - BATCH; q < iter; ++q) xc += ld_wt(col + (size_t)q * n);
          xc += theta[i];
          const double dm = xc / BATCH - xp / BATCH;
          part += dm * dm / BATCH;
        }
        part = fm::wave_sum_dpp(part);
        if (lane == 0) red[7 * SW + wave] = part;
        __syncthreads();
        const double dx = sqrt(sumw(red + 7 * SW));
        double ty = 0.0, ys = 0.0, tt = 0.0;
        for (int k = 0; k < BATCH; ++k) {
          const double tk = k - 0.5 * (BATCH - 1);
          ty = fma(tk, ywin[k], ty);
          ys += ywin[k];
          tt = fma(tk, tk, tt);
        }
        const double slope = ty / tt, ym = ys / BATCH;
        double rs = 0.0;
        for (int k = 0; k < BATCH; ++k) {
          const double r = ywin[k] - ym - slope * (k - 0.5 * (BATCH - 1));
          rs = fma(r, r, rs);
        }
        const double c00 = rs / (BATCH - 2) / tt;
        const double err = sqrt(c00 + cst[3]), err_max = sqrt(c00 + cst[4]);
        stop = (dx < 0.001 && fabs(slope) < err_max) || (fabs(slope) < err && dx < 0.1);
      }
    }

    // ---- set_parameters + the pack of the next iterate ----
    stamp(t, 7);
    pack_waves(a, theta, aux, pack, tid);
    __syncthreads();
    stamp(t, 8);
    if (stop) {
      n_ran = t + 1;
      break;
    }
    }
  }
  int tid_e;  // (a last opaque copy: the write-back's addresses are not the start-up copies' carried across the loop)
  asm volatile("v_mov_b32 %0, %1" : "=v"(tid_e) : "v"(tid_launch));
  {
  const int tid = tid_e;
  if (g == 0 && tid == 0 && f.n_done) *f.n_done = n_ran;

  // the state the next batch (or vbmc_adam_end) starts from -- not if another workgroup has given up meanwhile (the host
  // redoes the batch from the state this launch started with, and restores it in case the flag is raised after this test)
  if (g == 0 && !(__hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 4)) {
    for (int i = tid; i < L.o_hyp(); i += NT) a.state[i] = sh[i];  // theta | aux
    for (int i = tid; i < ml.total; i += NT) a.mix[i] = pack[i];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int i = u * NT + tid;
      if (i < n) {
        a.state[L.o_m() + i] = r_m[u];
        a.state[L.o_v() + i] = r_v[u];
      }
    }
  }
  }
}

template <int DP>
int launch_fused(vbmc_ctx* ctx, hipStream_t st, const FusedArgs& f, size_t lds) {
  static size_t lds_limit[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (lds > 48 * 1024 && lds > lds_limit[dev & 63]) {
    HIP_TRY(ctx, hipFuncSetAttribute((const void*)adam_fused_kernel<DP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    lds_limit[dev & 63] = lds;
  }
  hipLaunchKernelGGL((adam_fused_kernel<DP>), dim3(f.n_ent + f.n_gp), dim3(64 * SW), lds, st, f);
  HIP_TRY(ctx, hipGetLastError());
  return 0;
}

int fused_dp(int D) { return D <= 2 ? 2 : D <= 4 ? 4 : D <= 6 ? 6 : D <= 8 ? 8 : D <= 10 ? 10 : D <= 12 ? 12 : D <= 16 ? 16 : D <= 20 ? 20 : 24; }

}  // namespace

namespace adam_dev {

// Fills the workgroup split and the LDS carve of f (f.a, f.N, f.rows set by the caller); returns the
// dynamic LDS bytes, or 0 when the fused loop does not apply to this shape.
size_t adam_fused_plan(FusedArgs& f) {
  const AdamDev& a = f.a;
  const int D = a.D, K = a.K, S = a.S, N = f.N;
  if (K > 64 || D > 24 || f.rows < 1 || a.n_theta > (D > 16 ? 3 : 2) * NT || N < 1) return 0;
  // Round 5: more than 64 antithetic rows per component are split over R <= 4 workgroups per component (every workgroup must
  // be resident: K R + n_gp <= CUs).  The caller (adam.hip) uses it up to 160 rows per component, where it still beats
  // four launches.  f.rows comes in as the rows per component and goes out as the rows per workgroup.
  f.rows_total = f.rows;
  int R = 1;
  if (f.rows > 64) {
    const int cus = f.cus > 0 ? f.cus : 256;
    const int gp_min = S * K < 32 ? S * K : 32;
    const int r_max = (cus - 8 - gp_min) / K < 4 ? (cus - 8 - gp_min) / K : 4;
    R = f.rows > 128 ? r_max : (r_max < 2 ? r_max : 2);  // (as many slices as fit: shorter row loops)
    if (R < 1 || (f.rows + R - 1) / R > 512) return 0;
    f.rows = (f.rows + R - 1) / R;
  }
  const int DP = fused_dp(D);
  const int RE = 2 + 2 * D + K, RG = 1 + 2 * D, RC = 2 * D + 4;
  f.n_ent = K * R;
  f.n_gp = S * K < 128 ? S * K : 128;
  if (R > 1) {
    const int cus = f.cus > 0 ? f.cus : 256;
    if (f.n_gp > cus - 8 - f.n_ent) f.n_gp = cus - 8 - f.n_ent;  // (a few CUs' margin: every workgroup must become resident)
    if (f.n_gp < 1) return 0;
  }
  size_t o = (size_t)a.lay.o_res() + (size_t)a.n_bnd + 2 * (size_t)K + 2 * (size_t)D;  // theta | aux | hyp | phase B scratch
  auto take = [&](size_t cnt) {
    const size_t at = o;
    o += (cnt + 1) & ~(size_t)1;  // 16-byte granules
    return (int)at;
  };
  o = (o + 1) & ~(size_t)1;
  f.o_pack = take(a.ml.total);
  f.o_ee = take(K);
  f.o_eps = take((size_t)f.rows * D);
  f.o_out = take(RG);  // a GP block's sums (the entropy workgroups store their record entries straight from registers)
  // the entropy workgroups' reduction scratch lies over what they do not use in phase A: the gathered
  // records (dead until the gather) and the GP workgroups' arrays
  f.o_recs = take((size_t)K * RE + (size_t)S * K * RC);
  f.o_part = f.o_recs;
  f.o_gp = take((size_t)2 * D + N + SW + 4);
  f.o_alpha = take((size_t)S * N);
  size_t need_part = (size_t)SW * (2 * D + 1) * K + (size_t)SW * 64 + SW + 2;
  f.part_waves = SW;
  // dynamic LDS: the CU's 160 KB less the kernel's static arrays (8 984 B in every build: -Rpass-analysis=kernel-resource-usage;
  // the 154 KB of rounds 3-4 overshot that by 2.8 KB -- a plan between 151 and 154 KB would have failed at launch)
  constexpr size_t LDS_MAX = (160 - 9) * 1024;
  if (D > 16 && (f.o_recs + need_part) * sizeof(double) > LDS_MAX) {  // (see the kernel: two half-rounds through half the scratch)
    f.part_waves = SW / 2;
    need_part = (size_t)(SW / 2) * (2 * D + 1) * K + (size_t)SW * 64 + SW + 2;
  }
  (void)DP;
  // X^T (D N doubles: 64 KB at D = 20, N = 400) stays in LDS when everything fits; else the GP workgroups read it from
  // memory in their two passes (round 5: D = 20, K = 50 fits N <= 270 with it and N ~ 2000 without; D = 10, K = 50: 900 / ~5000)
  const size_t o_before = o;
  f.o_xt = take((size_t)D * N);
  if (o - (size_t)f.o_recs < need_part) o = (size_t)f.o_recs + need_part;
  if (o * sizeof(double) > LDS_MAX) {
    f.o_xt = -1;
    o = o_before;
    if (o - (size_t)f.o_recs < need_part) o = (size_t)f.o_recs + need_part;
  }
  const size_t bytes = o * sizeof(double);
  static const bool dbg = getenv("VBMC_FUSED_PLAN_DEBUG") != nullptr;  // measurement aid: the carve on stderr
  if (dbg)
    fprintf(stderr, "[vbmc] fused plan D=%d K=%d S=%d N=%d rows=%d/%d R=%d: state %d pack %d recs %d gp %d alpha %d xt %d part_waves %d need_part %zu -> %zu B (limit %zu)\n",
            D, K, S, N, f.rows, f.rows_total, R, (int)a.lay.o_res(), f.o_pack, f.o_recs, f.o_gp, f.o_alpha, f.o_xt, f.part_waves, need_part, bytes, LDS_MAX);
  return bytes <= LDS_MAX ? bytes : 0;
}

int adam_fused_launch(vbmc_ctx* ctx, hipStream_t st, const FusedArgs& f, size_t lds) {
  switch (fused_dp(f.a.D)) {
    case 2: return launch_fused<2>(ctx, st, f, lds);
    case 4: return launch_fused<4>(ctx, st, f, lds);
    case 6: return launch_fused<6>(ctx, st, f, lds);
    case 8: return launch_fused<8>(ctx, st, f, lds);
    case 10: return launch_fused<10>(ctx, st, f, lds);
    case 12: return launch_fused<12>(ctx, st, f, lds);
    case 16: return launch_fused<16>(ctx, st, f, lds);
    case 20: return launch_fused<20>(ctx, st, f, lds);
    default: return launch_fused<24>(ctx, st, f, lds);
  }
}

}  // namespace adam_dev
