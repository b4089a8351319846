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

#ifndef GLJ_T
#define GLJ_T(i)  // measurement aid (prep.hip with -DPREP_TIMES): phase stamps of a block
#endif
// doubles of a block's own arrays: the two-pass form's [1/tau (D) | mu (D) | z alpha (N) | 4 | 1] or the one-pass form's
// [1/tau (24) | mu (24) | 2 | 16 partial sums of each of 49 items], whichever is larger
constexpr int GLJ_1P_DMAX = 24;                                                   // the one-pass form's largest D
constexpr int GLJ_1P_OWN = 2 * GLJ_1P_DMAX + 2 + 16 * (2 * GLJ_1P_DMAX + 1);      // 834 doubles
__host__ __device__ inline size_t glj_own(int D, int N) {
  const size_t two_pass = (size_t)2 * D + N + 4 + 1;
  return two_pass > (size_t)GLJ_1P_OWN ? two_pass : (size_t)GLJ_1P_OWN;
}
inline size_t glj_block_lds(int D, int N) { return sizeof(double) * glj_own(D, N); }
// Round 6: with X^T in LDS behind those arrays (PrepArgs::x_lds).  A block is a chain of memory latencies -- X^T is read
// twice, in 5 + 8 dependent groups of loads at N = 800, D = 20 -- and a rider workgroup of the entropy launch works
// through four to five items one after the other: staged ONCE per workgroup (all loads in flight) both passes of every
// item read LDS.  What it buys is small (config 3: the GP word 1.5-2 us earlier, S = 8 step 99.2 -> 97.3 us, the two-launch
// optimiser iteration 92.4 -> 91.6): beside two entropy waves per SIMD the riders are slowed by issue contention more than by
// their loads.  Used where it needs no LDS the launch does not have anyway (the riders) or little (prep.hip).
__host__ __device__ inline size_t glj_x_off(int D, int N) { return (glj_own(D, N) + 1) & ~(size_t)1; }  // doubles, 16-byte aligned
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

// How a block's results are stored: plain, write-through to device memory another workgroup reads, or write-through to
// pinned host memory (DoneSignal)
struct GljPut {
  bool staged, sig;
  __device__ __forceinline__ void operator()(double* p, double v) const {
    if (staged) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // write-through (sc1): read by another workgroup
    else if (sig) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // write-through to pinned memory
    else *p = v;
  }
};

// wave 0: 1/tau_d, mu_dk (entries [D, DPAD) zeroed) and lnnf = 2 hyp[D] + sum_d (hyp[d] - log tau_d) -> sMisc[0]
__device__ __forceinline__ void glj_setup(const PrepArgs& a, int s, int k, double* sItau, double* sMu, double* sMisc, int DPAD) {
  const int D = a.ml.D, tid = threadIdx.x;
  if (tid < 64) {
    const double* h = a.hyp + (size_t)s * a.P;
    const double sigk = a.mix[a.ml.o_sig + k];
    double term = 0.0;
    for (int d = tid; d < DPAD; d += 64) {
      if (d < D) {
        const double ell = fm::exp2_fast(0x1.71547652b82fep+0 * h[d]);  // exp(h_d)
        const double lam = a.mix[a.ml.o_lam + d];
        const double tau2 = sigk * sigk * lam * lam + ell * ell;
        sItau[d] = fm::rsqrt_fast(tau2);
        sMu[d] = a.mix[a.ml.o_mu + k * D + d];
        term += h[d] - 0.5 * fm::log_fast(tau2);
      } else {
        sItau[d] = 0.0;
        sMu[d] = 0.0;
      }
    }
    term = fm::wave_sum_dpp(term);
    if (tid == 0) sMisc[0] = 2.0 * h[D] + term;
  }
}

// ---- the two-pass form (rounds 1-5; D > 24 still): z_n alpha_n through LDS, X^T read twice ----
template <bool XL>
__device__ __forceinline__ void glj_sums_2p(const PrepArgs& a, int s, int k, double* lds, const GljPut& put) {
  const int D = a.ml.D, K = a.ml.K;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N = a.N;
  double* sItau = lds;             // [D]
  double* sMu = sItau + D;         // [D]
  double* sZa = sMu + D;           // [N]
  double* sPart = sZa + N;         // [4]
  double* sMisc = sPart + 4;       // [1]
  const double* XT = XL ? lds + glj_x_off(D, N) : a.XT;  // (compile-time: LDS or global addressing, never flat)
  glj_setup(a, s, k, sItau, sMu, sMisc, D);
  __syncthreads();
  const double lnnf = sMisc[0];
  // X is read through its transpose XT[d][n]: consecutive threads take consecutive points of one
  // dimension (coalesced) -- the strided row reads of rounds 1-2 were this block's latency chain
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
      // the end): a step costs one memory latency instead of sixteen
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
  __syncthreads();
  if (tid == 0) put(out, (sPart[0] + sPart[1]) + (sPart[2] + sPart[3]));
}

// ---- the one-pass form (round 6, D <= 24): X^T read ONCE.  Thread = point: the D coordinates of P points per thread and
// round are requested together (P DP <= 40 loads in flight -- and the first round's before the block's set-up, whose own
// loads they do not depend on), z_n from them, and the 1 + 2D sums advance in registers; one reduction at the end (a DPP
// row sum per item, sixteen partials through LDS, a thread per item adds them in a fixed order).  The two-pass form went
// through X^T twice in 5 + 8 dependent groups of loads at N = 800, D = 20: a 13 us chain in front of config 5's entropy
// kernel and 7 us at config 3's shape in every rider item; this one is a set-up, ceil(N / 256 P) round trips and the
// reduction. ----
template <int DP, int P, bool XL>
__device__ __forceinline__ void glj_sums_1p(const PrepArgs& a, int s, int k, double* lds, const GljPut& put) {
  const int D = a.ml.D, K = a.ml.K;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N = a.N;
  double* sItau = lds;                      // [24], zero beyond D
  double* sMu = sItau + GLJ_1P_DMAX;        // [24]
  double* sMisc = sMu + GLJ_1P_DMAX;        // [2]
  double* sRed = sMisc + 2;                 // [1 + 2D][16]
  const double* XT = XL ? lds + glj_x_off(D, N) : a.XT;  // (compile-time: LDS or global addressing, never flat)
  if constexpr (!XL) asm volatile("" : "+s"(XT));  // (opaque per call: a caller's loop over items does not carry the DP column bases across it)
  const double* al = a.alpha + (size_t)s * N;
  double x[P][DP], av[P];
  auto request = [&](int base) {
    // (the DP column bases are formed anew per request -- opaque copy of the pointer --: carried from the first request to
    // the later rounds' they are 2 DP scalar registers held across the block, and inside the wave-split entropy kernel,
    // whose registers are allotted over all of its code, that put 25 reloads of spilled scalar registers into pass 2 of
    // the BATCH loop: tools/ws_hot_blocks.py)
    const double* XTq = XT;
    if constexpr (!XL) asm volatile("" : "+s"(XTq));
#pragma unroll
    for (int p = 0; p < P; ++p) {
      if (p > 0 && base + 256 * p >= N) break;  // (wave-uniform: a small N pays for the point slots it fills)
      const int n = base + 256 * p + tid;
      // (a wave-uniform base and a 32-bit byte offset per lane: the scalar-base form of the load, one offset register per
      // point instead of an address pair per load)
      const unsigned off = (unsigned)min(n, N - 1) * 8u;
#pragma unroll
      for (int d = 0; d < DP; ++d) {
        const char* col = (const char*)(XTq + (size_t)min(d, D - 1) * N);
        x[p][d] = *(const double*)(col + off);
      }
      av[p] = *(const double*)((const char*)al + off);
    }
  };
  GLJ_T(0);
  request(0);
  glj_setup(a, s, k, sItau, sMu, sMisc, DP);
  __syncthreads();
  GLJ_T(1);
  const double lnnf = sMisc[0];
  double acc0 = 0.0, U[DP], T[DP];
#pragma unroll
  for (int d = 0; d < DP; ++d) U[d] = T[d] = 0.0;
  for (int base = 0;;) {
    // (1/tau and mu are read from LDS where they are used: hoisted out of this loop -- the compiler's choice without the
    // opaque zero -- they were 4 DP more registers)
    int zoff;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zoff));
    const double* qMu = sMu + zoff;
    const double* qItau = sItau + zoff;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      if (p > 0 && base + 256 * p >= N) break;
      const int n = base + 256 * p + tid;
      double d2 = 0.0;
#pragma unroll
      for (int d = 0; d < DP; ++d) {
        const double dl = (qMu[d] - x[p][d]) * qItau[d];  // (0 beyond D: 1/tau is stored as 0 there)
        x[p][d] = dl;
        d2 = fma(dl, dl, d2);
      }
      const double z = fm::exp2_fast(0x1.71547652b82fep+0 * (lnnf - 0.5 * d2));  // exp(.)
      const double za = n < N ? z * av[p] : 0.0;
      if (a.Z && n < N) a.Z[((size_t)s * K + k) * N + n] = z;
      acc0 += za;
      if (a.want_grad) {
#pragma unroll
        for (int d = 0; d < DP; ++d) {
          const double t = x[p][d] * za;
          U[d] += t;
          T[d] = fma(x[p][d], t, T[d]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);  // (point by point)
    }
    base += 256 * P;
    if (base >= N) break;
    request(base);
  }
  GLJ_T(2);
  // item it: 0 = sum z alpha, 1 + d, 1 + D + d; partial (wave, row of 16 lanes) at sRed[it * 16 + wave * 4 + row]
  const int slot = wave * 4 + (lane >> 4);
  const bool first = (lane & 15) == 0;
  {
    const double r = fm::row16_sum_dpp(acc0);
    if (first) sRed[slot] = r;
  }
  if (a.want_grad) {
#pragma unroll
    for (int d = 0; d < DP; ++d) {
      const double ru = fm::row16_sum_dpp(U[d]), rt = fm::row16_sum_dpp(T[d]);
      if (first && d < D) {
        sRed[(1 + d) * 16 + slot] = ru;
        sRed[(1 + D + d) * 16 + slot] = rt;
      }
      __builtin_amdgcn_sched_barrier(0);  // (one dimension's pair at a time: interleaved, the 2 DP reductions' temporaries were the kernel's register peak)
    }
  }
  __syncthreads();
  GLJ_T(3);
  const int NI = a.want_grad ? 1 + 2 * D : 1;
  if (tid < NI) {
    const double* r = sRed + tid * 16;
    double v = 0.0;
#pragma unroll
    for (int w = 0; w < 4; ++w) v += (r[4 * w] + r[4 * w + 1]) + (r[4 * w + 2] + r[4 * w + 3]);
    put(a.res + ((size_t)s * K + k) * (1 + 2 * D) + tid, v);
  }
}

// a.mix / a.res already advanced to this candidate; b = s * K + k.  XL: X^T was staged in LDS (glj_stage_x).
// [DMIN, DMAX]: the D the caller's launch can see (a kernel built for one padded D instantiates one form only).
// LEAN: one point per thread and round from D = 13 (a rider inside an entropy kernel built for two waves per SIMD).
template <bool XL = false, int DMAX = 32, int DMIN = 1, bool LEAN = false>
__device__ __forceinline__ void glj_block(const PrepArgs& a, int b, double* lds) {
  const int D = a.ml.D, K = a.ml.K;
  const int tid = threadIdx.x;
  const int s = b / K, k = b - s * K;
  const bool sig = a.done.flag != nullptr;
  const bool dev = sig && a.done.dev != 0;  // a.res and the flag are device memory, read by a workgroup of this launch (DoneSignal)
  const bool staged = (sig && a.done.host_out != nullptr) || dev;  // a.res is device memory; the last block ships it
  const GljPut put{staged, sig};
#ifndef GLJ_P20
#define GLJ_P20 2  // points per thread and round in the build for D = 17 .. 20 (measurement aid)
#endif
  if constexpr (DMAX > 24) {
    if (DMIN > 24 || D > 24) {
      glj_sums_2p<XL>(a, s, k, lds, put);
      goto tail;
    }
  }
  if constexpr (DMAX > 20 && DMIN <= 24) {
    if (DMIN > 20 || D > 20) {
      glj_sums_1p<24, 1, XL>(a, s, k, lds, put);
      goto tail;
    }
  }
  if constexpr (DMAX > 16 && DMIN <= 20) {
    if (DMIN > 16 || D > 16) {
      glj_sums_1p<20, LEAN ? 1 : GLJ_P20, XL>(a, s, k, lds, put);
      goto tail;
    }
  }
  if constexpr (DMAX > 12 && DMIN <= 16) {
    if (DMIN > 12 || D > 12) {
      glj_sums_1p<16, LEAN ? 1 : 2, XL>(a, s, k, lds, put);
      goto tail;
    }
  }
  if constexpr (DMAX > 8 && DMIN <= 12) {
    if (DMIN > 8 || D > 8) {
      glj_sums_1p<12, LEAN ? 1 : 2, XL>(a, s, k, lds, put);
      goto tail;
    }
  }
  if constexpr (DMAX > 4 && DMIN <= 8) {
    if (DMIN > 4 || D > 4) {
      glj_sums_1p<8, 4, XL>(a, s, k, lds, put);
      goto tail;
    }
  }
  if constexpr (DMIN <= 4) glj_sums_1p<4, 4, XL>(a, s, k, lds, put);
tail:
  // the one-pass form's [1/tau | mu] are 48 doubles, the two-pass form's at most 2 D = 64 in front of its N-sized array
  double* const sMisc = lds + 2 * GLJ_1P_DMAX;
  GLJ_T(4);
  if (sig) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this thread's results have been acknowledged
  __syncthreads();
  GLJ_T(5);
  if (!sig) return;
  if (tid == 0) {
    const bool last = __hip_atomic_fetch_add(a.done.cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.n_glj * a.batch - 1;
    sMisc[0] = last ? 1.0 : 0.0;
  }
  __syncthreads();
  GLJ_T(6);
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
