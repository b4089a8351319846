// Row j of the (j,k) constant table of the wave-split entropy kernel (entropy_ws.hip), by one 256-thread workgroup:
// T[j][k] = [Delta_jk (DP) | c0 | a | w | w/sigma_k^2 | pad pad], where c0 + a (sigma_j^2 |eps|^2 +- 2 sigma_j Delta.eps)
// is the log2 density of component k at mu_j +- sigma_j lambda eps:  a = -log2(e)/(2 sigma_k^2),  c0 = a |Delta|^2 + log2 c_k.
// Shared by the prep launch (prep.hip) and the optimiser loop's tail launch (adam.hip), where `mix` is the pack the
// workgroup has just made in its own LDS.  lds: K4 * DP doubles of scratch.
#pragma once
#include "common.h"

__device__ __forceinline__ void ws_table_row_block(const double* mix, const MixLayout& ml, int j, int DP, int K4,
                                                   double* table, double* lds) {
  const int D = ml.D, K = ml.K, tid = threadIdx.x, TS = DP + 6;
  const double* mup = mix + ml.o_mup;
  // all lanes on the (k, d) differences (coalesced row writes), then one lane per k on the tail
  double* sV2 = lds;  // [K4][DP] squared differences
  // (eight entries per thread and round, their loads requested together: one entry per round was a chain of K4 DP / 256
  // memory latencies -- the loads cannot move above the stores in front of them --, eight at K = 100, D = 20)
  // (the per-component constants of the row's tail are requested here as well, in front of the barrier: behind it they
  // were a second memory latency of this block -- and the block is the prep launch, between the host's go word and the
  // entropy kernel)
  const int kt = min(tid, K - 1);
  const double t_is2 = mix[ml.o_is2 + kt], t_w = mix[ml.o_w + kt], t_lrc = mix[ml.o_lrc + kt];
  constexpr int U = 8;
  for (int base = 0; base < K4 * DP; base += U * 256) {
    double vj[U], vk[U];
    // (slices wholly past the end are skipped by a wave-uniform test: a small table -- K = 50, D = 10: two entries per
    // thread -- must not pay eight entries' index arithmetic; this block is on the optimiser loop's chain too)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (base + u * 256 >= K4 * DP) break;
      const int idx = base + u * 256 + tid;
      const int k = idx / DP, d = idx - k * DP;
      vj[u] = mup[j * D + min(d, D - 1)];
      vk[u] = mup[min(k, K - 1) * D + min(d, D - 1)];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (base + u * 256 >= K4 * DP) break;
      const int idx = base + u * 256 + tid;
      const int k = idx / DP, d = idx - k * DP;
      if (idx < K4 * DP) {
        const double v = (d < D && k < K) ? (vj[u] - vk[u]) : 0.0;
        table[((size_t)j * K4 + k) * TS + d] = v;
        sV2[idx] = v * v;
      }
    }
  }
  __syncthreads();
  for (int k = tid; k < K4; k += 256) {
    double* row = table + ((size_t)j * K4 + k) * TS;
    double s = 0.0;
    for (int d = 0; d < DP; ++d) s += sV2[k * DP + d];
    if (k < K) {
      const bool pre = k == tid;  // (the first pass: K4 <= 256 in every build, so the only one)
      const double is2 = pre ? t_is2 : mix[ml.o_is2 + k];
      const double w = pre ? t_w : mix[ml.o_w + k];
      const double ak = -0.5 * 0x1.71547652b82fep+0 * is2;  // -log2(e) / (2 sigma_k^2)
      row[DP + 0] = fma(ak, s, pre ? t_lrc : mix[ml.o_lrc + k]);  // log2 density of component k at mu_j
      row[DP + 1] = ak;
      row[DP + 2] = w;
      row[DP + 3] = w * is2;
    } else {  // padding component: density exactly 0
      row[DP + 0] = -2000.0; row[DP + 1] = 0.0; row[DP + 2] = 0.0; row[DP + 3] = 0.0;
    }
    row[DP + 4] = 0.0;
    row[DP + 5] = 0.0;
  }
}
