// The GP expected-log-joint block (s, k) -- reference vbmc/variational_optimization.py:1400-1465:
//     res[(s*K+k)*(1+2D) + it],  it = 0       : sum_n z_n alpha_n
//                                it = 1..D    : sum_n delta_nd   z_n alpha_n
//                                it = D+1..2D : sum_n delta_nd^2 z_n alpha_n
//     z_n = exp(lnnf - 1/2 sum_d delta_nd^2),  delta_nd = (mu_dk - X_nd)/tau_dk,
//     tau_dk = sqrt(sigma_k^2 lambda_d^2 + ell_d^2);  optionally Z[s][k][n] = z_n.
// One 256-thread workgroup per (s, k); dynamic LDS of glj_block_lds(D, N) bytes.  Shared by the
// prep launch (prep.hip) and, in the polled host-driven step, by the finish launch (entropy.hip),
// where the 7 us latency chain of these blocks runs beside the reduction instead of in front of
// the entropy kernel.
#pragma once
#include "common.h"
#include "fastmath.h"

inline size_t glj_block_lds(int D, int N) { return sizeof(double) * ((size_t)2 * D + N + 4 + 1); }
// Round 6: with X^T in LDS behind those arrays (PrepArgs::x_lds).  A block is a chain of memory latencies -- X^T is read
// twice, in 5 + 8 dependent groups of loads at N = 800, D = 20 -- and a rider workgroup of the entropy launch works
// through four to five items one after the other: staged ONCE per workgroup (all loads in flight) both passes of every
// item read LDS.  What it buys is small (config 3: the GP word 1.5-2 us earlier, S = 8 step 99.2 -> 97.3 us, the two-launch
// optimiser iteration 92.4 -> 91.6): beside two entropy waves per SIMD the riders are slowed by issue contention more than by
// their loads.  Used where it needs no LDS the launch does not have anyway (the riders) or little (prep.hip).
__host__ __device__ inline size_t glj_x_off(int D, int N) { return ((size_t)2 * D + N + 4 + 1 + 1) & ~(size_t)1; }  // doubles, 16-byte aligned
inline size_t glj_block_lds_x(int D, int N) { return sizeof(double) * (glj_x_off(D, N) + (size_t)D * N); }

// all 256 threads: X^T (D x N) into LDS behind a block's arrays; 32 loads in flight per thread and round
__device__ __forceinline__ void glj_stage_x(const PrepArgs& a, double* lds) {
  const int D = a.ml.D, n = D * a.N, tid = threadIdx.x;
  double* sX = lds + glj_x_off(D, a.N);
  constexpr int U = 32;
  for (int base = 0; base < n; base += U * 256) {
    double r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = base + u * 256 + tid;
      r[u] = a.XT[i < n ? i : n - 1];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = base + u * 256 + tid;
      if (i < n) sX[i] = r[u];
    }
  }
  __syncthreads();
}

// a.mix / a.res already advanced to this candidate; b = s * K + k.  XL: X^T was staged in LDS (glj_stage_x).
template <bool XL = false>
__device__ __forceinline__ void glj_block(const PrepArgs& a, int b, double* lds) {
  const int D = a.ml.D, K = a.ml.K;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int s = b / K, k = b - s * K;
  const int N = a.N;
  double* sItau = lds;             // [D]
  double* sMu = sItau + D;         // [D]
  double* sZa = sMu + D;           // [N]
  double* sPart = sZa + N;         // [4]
  double* sMisc = sPart + 4;       // [1]
  const double* XT = XL ? lds + glj_x_off(D, N) : a.XT;  // (compile-time: LDS or global addressing, never flat)
  const double* h = a.hyp + (size_t)s * a.P;
  const double sigk = a.mix[a.ml.o_sig + k];
  if (tid < 64) {
    // wave 0: 1/tau_d, mu_dk and lnnf = 2 hyp[D] + sum_d (hyp[d] - log tau_d)
    double term = 0.0;
    for (int d = tid; d < D; d += 64) {
      const double ell = fm::exp2_fast(0x1.71547652b82fep+0 * h[d]);  // exp(h_d)
      const double lam = a.mix[a.ml.o_lam + d];
      const double tau2 = sigk * sigk * lam * lam + ell * ell;
      sItau[d] = fm::rsqrt_fast(tau2);
      sMu[d] = a.mix[a.ml.o_mu + k * D + d];
      term += h[d] - 0.5 * fm::log_fast(tau2);
    }
    term = fm::wave_sum_dpp(term);
    if (tid == 0) sMisc[0] = 2.0 * h[D] + term;
  }
  __syncthreads();
  const double lnnf = sMisc[0];
  // X is read through its transpose XT[d][n]: consecutive threads take consecutive points of one
  // dimension (coalesced) -- the strided row reads of rounds 1-2 were this block's latency chain
  // (13 us at N = 800, D = 20, on the critical path of config 5's step)
  for (int nb = 0; nb < N; nb += 4 * 256) {  // four points per thread and pass, four dimensions per step: 16 loads in flight
    int nn[4];
    double d2[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      nn[p] = min(nb + tid + 256 * p, N - 1);
      d2[p] = 0.0;
    }
    for (int d0 = 0; d0 < D; d0 += 4) {
      double x[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double* col = XT + (size_t)min(d0 + u, D - 1) * N;
#pragma unroll
        for (int p = 0; p < 4; ++p) x[p][u] = col[nn[p]];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (d0 + u < D) {
          const double m = sMu[d0 + u], it = sItau[d0 + u];
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            const double dl = (m - x[p][u]) * it;
            d2[p] = fma(dl, dl, d2[p]);
          }
        }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int n = nb + tid + 256 * p;
      if (n < N) {
        const double z = fm::exp2_fast(0x1.71547652b82fep+0 * (lnnf - 0.5 * d2[p]));  // exp(.)
        sZa[n] = z * a.alpha[(size_t)s * N + n];
        if (a.Z) a.Z[((size_t)s * K + k) * N + n] = z;
      }
    }
  }
  __syncthreads();
  double* out = a.res + ((size_t)s * K + k) * (1 + 2 * D);
  const bool sig = a.done.flag != nullptr;
  const bool dev = sig && a.done.dev != 0;  // a.res and the flag are device memory, read by a workgroup of this launch (DoneSignal)
  const bool staged = (sig && a.done.host_out != nullptr) || dev;  // a.res is device memory; the last block ships it
  auto put = [&](double* p, double v) {
    if (staged) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // write-through (sc1): read by another workgroup
    else if (sig) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // write-through to pinned memory
    else *p = v;
  };
  {
    double acc = 0.0;
    for (int n = tid; n < N; n += 256) acc += sZa[n];
    acc = fm::wave_sum_dpp(acc);
    if (lane == 0) sPart[wave] = acc;
  }
  if (a.want_grad) {
    // thread = (slice ns of the points, dimension slot ds): every dimension's two sums advance
    // side by side (independent loads, one short shuffle reduction over the 16 slices) instead
    // of one block-wide reduction per dimension
    const int ns = tid & 15, ds = tid >> 4;
    for (int d = ds; d < D; d += 16) {
      const double m = sMu[d], itau = sItau[d];
      double au = 0.0, at = 0.0;
      // sixteen points per step, their loads issued together (indices clamped, weights zeroed past
      // the end): a step costs one memory latency instead of sixteen -- the loop was this block's
      // time (N / 16 dependent round trips per dimension: 20 us at N = 800, D = 20)
      constexpr int PS = 16;
      for (int n0 = ns; n0 < N; n0 += 16 * PS) {
        double x[PS], za[PS];
#pragma unroll
        for (int u = 0; u < PS; ++u) {
          const int n = min(n0 + 16 * u, N - 1);
          x[u] = XT[(size_t)d * N + n];
          za[u] = n0 + 16 * u < N ? sZa[n] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < PS; ++u) {
          const double dl = (m - x[u]) * itau;
          const double t = dl * za[u];
          au += t;
          at = fma(dl, t, at);
        }
      }
      au = fm::row16_sum_dpp(au);
      at = fm::row16_sum_dpp(at);
      if (ns == 0) {
        put(out + 1 + d, au);
        put(out + 1 + D + d, at);
      }
    }
  }
  if (sig) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this thread's results have been acknowledged
  __syncthreads();
  if (tid == 0) {
    put(out, (sPart[0] + sPart[1]) + (sPart[2] + sPart[3]));
    if (sig) {
      __builtin_amdgcn_s_waitcnt(0x0F70);
      const bool last = __hip_atomic_fetch_add(a.done.cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.n_glj * a.batch - 1;
      sMisc[0] = last ? 1.0 : 0.0;
    }
  }
  if (!sig) return;
  __syncthreads();
  if (sMisc[0] == 0.0) return;
  // last block to count: every block's sums are in memory (drained write-through stores)
  if (staged && !dev) {
    staged_copy_to_host(a.res, a.done.host_out, a.done.host_n);
    __syncthreads();
  }
  if (tid == 0) {
    __hip_atomic_store(a.done.cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (dev) __hip_atomic_store(a.done.flag, a.done.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_store(a.done.flag, a.done.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#ifdef GLJ_STAMP
    GLJ_STAMP();
#endif
  }
}
