// The entropy reduction's block body, shared by entmc_finish_kernel (entropy.hip) and the optimiser loop's tail launch
// (adam.hip).  Include after common.h / fastmath.h / philox.h (gen_slice_block).
#pragma once
#include "common.h"
#include "fastmath.h"
#include "philox.h"

// Finish: reduce the per-workgroup partial rows in a fixed order (bit-reproducible) and
// combine them into the raw accumulator vector
//     [H | mu (K blocks of D) | sigma (K) | lambda (D) | w (K)]
// (entmc_vbmc.py:80,98,102-112 with the 1/Ns and w_j factors applied).
// One wave per output element; lanes run over the (component j, chunk c) rows it sums.
// `raw` may be device memory or device-visible pinned host memory.
#ifdef FIN_TIMES_HERE
__device__ unsigned long long g_fin_times[4 + 3 * 64];  // [3] launch counter; per launch n % 64: start of block 0, publish, latest end
// phases of the LAST finish launch (tools/fin_phases.py): [0] earliest block start, [1] latest "sum formed", [2] latest
// "result store acknowledged", [3] the counting block knows it is last, [4] its staged copy is acknowledged, [5] flag stored,
// [6] latest start of a reduction block
__device__ unsigned long long g_fin_x[8];
#define FIN_X_MAX(i) do { if ((threadIdx.x & 63) == 0) atomicMax(&g_fin_x[i], wall_clock64()); } while (0)
#else
#define FIN_X_MAX(i) (void)0
#endif
// (the body of entmc_finish_kernel, entropy.hip; also the first blocks of the optimiser loop's tail launch, adam.hip)
__device__ __forceinline__ void entmc_finish_body(const double* __restrict__ partial, int chunks, int stride,
                                                  const double* __restrict__ mix, const MixLayout& ml, double inv_ns,
                                                  int want_grad, int mu_from_w, double* __restrict__ raw,
                                                  const GenSlice& gen, const DoneSignal& done) {
  if (done.cancel != nullptr && __hip_atomic_load(done.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == ~(uint64_t)0)
    return;  // armed evaluation that was cancelled (common.h ArmedEval)
  const int D = ml.D, K = ml.K;
  auto wave_sum = [](double v) { return fm::wave_sum_dpp(v); };
#ifdef FIN_TIMES_HERE
  if (blockIdx.x == 0 && threadIdx.x == 0) { g_fin_times[0] = wall_clock64(); g_fin_times[2] = 0; }
  struct EndStamp { __device__ ~EndStamp() { if (threadIdx.x == 0 && ((blockIdx.x & 127) == 0 || blockIdx.x + 8 >= gridDim.x)) atomicMax(&g_fin_times[2], wall_clock64()); } } end_stamp;
  // history: the slot of this launch is claimed by the publishing block (below); block 0's start and the
  // running end maximum are copied there by the last-numbered block, which is dispatched last
#endif
  {
    // spare workgroups after the reduction's own: a slice of the next draws (host-driven step, Adam loop)
    const int n_main = (1 + D * K + 2 * K + D + 3) / 4;
    if ((int)blockIdx.x >= n_main) {
      gen_slice_block(gen, blockIdx.x - n_main, threadIdx.x);
      return;
    }
  }
#ifdef FIN_TIMES_HERE
  if (threadIdx.x == 0) {
    atomicMin(&g_fin_x[0], wall_clock64());
    atomicMax(&g_fin_x[6], wall_clock64());
  }
#endif
  const double* w = mix + ml.o_w;
  const double* sig = mix + ml.o_sig;
  const double* ilam = mix + ml.o_ilam;
  const int n = 1 + D * K + 2 * K + D;
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (t >= n && !done.flag) return;  // (with a completion word every wave of the block meets at the barrier below)
  double v = 0.0;
  // Every long sum here runs over (row, chunk) pairs -- row = a component (k for the mean's Delta part, j elsewhere) --
  // with a coefficient that depends on the row alone.  Lane = row (row + 64, ... beyond 64 components), the chunks in an
  // inner loop with twelve loads in flight (eleven chunks at config 3: one round): the coefficient (up to three loads) is formed once per row and there is no
  // index arithmetic per term.  (Rounds 4-6 flattened the pairs over the lanes: a division and a remainder by run-time
  // divisors and three coefficient loads per TERM -- ~600 instructions per lane at K = 50, eleven chunks, and four
  // dependent rounds of them at K = 100, twenty chunks -- on the path between the entropy kernel and the completion word.)
  // Fixed order: bit-reproducible.  (32-bit element offsets: the launcher refuses a partial block of 2^31 elements or more)
#ifndef FIN_FOLD_U
#define FIN_FOLD_U 12
#endif
  constexpr int FOLD_U = FIN_FOLD_U;
  auto fold = [&](double acc, int rows, auto&& coef, auto&& base, auto&& step) {
    for (int r0 = 0; r0 < rows; r0 += 64) {
      const int r = r0 + lane;
      const int rc = min(r, rows - 1);
      const double c = r < rows ? coef(rc) : 0.0;  // (0 leaves acc unchanged)
      const unsigned b = base(rc);
      for (int c0 = 0; c0 < chunks; c0 += FOLD_U) {
        double x[FOLD_U];
#pragma unroll
        for (int u = 0; u < FOLD_U; ++u) x[u] = partial[b + (unsigned)(min(c0 + u, chunks - 1) * stride)];
#pragma unroll
        for (int u = 0; u < FOLD_U; ++u)
          if (c0 + u < chunks) acc = step(acc, c, x[u]);
      }
    }
    return acc;
  };
  const auto add_prod = [](double a, double c, double x) { return a + c * x; };
  const unsigned jstride = (unsigned)(chunks * stride);  // elements between two components' first partial rows
  if (t >= n) {
  } else if (t == 0) {
    v = fold(0.0, K, [&](int j) { return w[j]; }, [&](int j) { return (unsigned)j * jstride; },
             [](double a, double c, double x) { return a - c * x; });
    v = wave_sum(v) * inv_ns;
  } else if (want_grad) {
    int u = t - 1;
    if (u < D * K) {
      const int j = u / D, d = u - j * D;
      for (int c = lane; c < chunks; c += 64) v += partial[((int64_t)j * chunks + c) * stride + 1 + d];
      if (mu_from_w) {
        // wave-split kernel: the Delta part of the mean gradient comes from the W sums,
        //   sum_k w_k/sigma_k^2 (mu'_jd - mu'_kd) W_jk   (entropy_ws.hip, pass 2)
        const double* mup = mix + ml.o_mup;
        const double* is2 = mix + ml.o_is2;
        // lanes run over k within one partial row (coalesced), the chunks in the inner loop
        const double mjd = mup[j * D + d];
        v = fold(v, K, [&](int k) { return w[k] * is2[k] * (mjd - mup[k * D + d]); },
                 [&](int k) { return (unsigned)j * jstride + (unsigned)(2 + 2 * D + k); },
                 [](double a, double c, double x) { return fma(c, x, a); });
      }
      v = wave_sum(v) * w[j] * inv_ns * ilam[d];
    } else if ((u -= D * K) < K) {
      for (int c = lane; c < chunks; c += 64) v += partial[((int64_t)u * chunks + c) * stride + 1 + D];
      v = wave_sum(v) * w[u] * inv_ns;
    } else if ((u -= K) < D) {
      v = fold(0.0, K, [&](int j) { return w[j] * sig[j]; },
               [&](int j) { return (unsigned)j * jstride + (unsigned)(2 + D + u); }, add_prod);
      v = wave_sum(v) * inv_ns * ilam[u];
    } else {
      u -= D;
      double sl = 0.0;
      for (int c = lane; c < chunks; c += 64) sl += partial[((int64_t)u * chunks + c) * stride];
      double s = fold(0.0, K, [&](int j) { return w[j]; },
                      [&](int j) { return (unsigned)j * jstride + (unsigned)(2 + 2 * D + u); }, add_prod);
      s = wave_sum(s);
      sl = wave_sum(sl);
      v = -inv_ns * (sl + s);
    }
  }
  FIN_X_MAX(1);
  if (!done.flag) {
    if (lane == 0) raw[t] = v;
    return;
  }
  // `raw` is pinned host memory and the host polls `done.flag` instead of waiting for the stream.
  // No fence: a system-scope release would write back every dirty L2 line.  Instead each result
  // goes out as a write-through (sc0 sc1) store and its wave waits until the store has been
  // acknowledged; then the workgroup meets at a barrier and counts itself with ONE atomic (611
  // increments of a single word would take ~7 us: a word saturates at ~88 atomics per us), and the
  // last workgroup to count publishes the sequence number the same way (MI355X_MICROARCH.md,
  // hand-off with a drained sc1 payload and flag).
  const bool staged = done.host_out != nullptr;  // raw is device memory; the last workgroup ships it
  const bool dev = done.dev != 0;                // raw and the flag are device memory: another workgroup of this launch waits for them
  if (lane == 0 && t < n) {
    if (staged || dev) __hip_atomic_store(raw + t, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_store(raw + t, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  }
  FIN_X_MAX(2);
  __shared__ int s_last;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int n_main = (n + 3) / 4;
    if (done.sub != nullptr && n_main >= done.sub_min) {  // (DoneSignal::sub: one word takes ~88 increments per us)
      constexpr int NS = 16;
      const int g = (int)blockIdx.x % NS, n_g = (n_main - g + NS - 1) / NS;
      int* sc = done.sub + g * 64;
      bool last = false;
      if (__hip_atomic_fetch_add(sc, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_g - 1) {
        __hip_atomic_store(sc, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = __hip_atomic_fetch_add(done.cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == NS - 1;
      }
      s_last = last;
    } else {
      s_last = __hip_atomic_fetch_add(done.cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_main - 1;
    }
  }
  __syncthreads();
  if (!s_last) return;
#ifdef FIN_TIMES_HERE
  if (threadIdx.x == 0) g_fin_x[3] = wall_clock64();
#endif
  if (done.ident_dst && threadIdx.x == 255) {  // what this result block was computed from (DoneSignal)
    __hip_atomic_store(done.ident_dst, *done.ident_src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(done.ident_dst + 1, done.ident_seed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (!staged) __builtin_amdgcn_s_waitcnt(0x0F70);  // (the staged copy below drains them with its own stores)
  }
  if (staged) {  // one coalesced copy to the host instead of n single PCIe writes (DoneSignal)
    staged_copy_to_host(raw, done.host_out, done.host_n);
  }
  if (staged || done.ident_dst) __syncthreads();
  if (threadIdx.x == 0) {
#ifdef FIN_TIMES_HERE
    g_fin_x[4] = wall_clock64();
#endif
    __hip_atomic_store(done.cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (dev) __hip_atomic_store(done.flag, done.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_store(done.flag, done.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#ifdef FIN_TIMES_HERE
    g_fin_x[5] = wall_clock64();
    g_fin_times[1] = wall_clock64();
    const unsigned long long slot = g_fin_times[3]++ & 63;
    g_fin_times[4 + 3 * slot] = g_fin_times[0];
    g_fin_times[4 + 3 * slot + 1] = g_fin_times[1];
    g_fin_times[4 + 3 * slot + 2] = g_glj_stamp;  // (the GP word of the PREVIOUS launch: it is published later than this)
#endif
  }
}
