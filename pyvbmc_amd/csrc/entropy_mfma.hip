// Monte-Carlo entropy, main kernel in its MATRIX-PIPE form -- reference entropy/entmc_vbmc.py:64-112;
// the same sums as entropy_ws.hip (same table, same partial rows, same finish kernel), for shapes
// where the 16 x 16 x 4 tile of v_mfma_f64_16x16x4_f64 pads little: D a multiple of 4 and K close
// below a multiple of 16 -- BASELINE config 5, D = 20, K = 100 (K -> 112: x1.12).
//
// Why only there.  FP64 matrix and vector instructions run on the same units at the same peak
// (tools/ubench_fp64.hip); what the matrix form buys is issue efficiency: the wave-split kernel
// <20,25> needs 512 registers, runs ONE wave per SIMD and issues a float64 instruction every ~6.9
// cycles where the pipe could take one every 4 (tools/ubench_ops.hip), while matrix instructions
// keep the pipe busy for 64 cycles each.  Its two D-deep contractions are 40 of its ~86 vector
// instructions per (sample pair, component).  At config 3 (D = 10, K = 50) the tile pads those
// products x1.54 / x1.66 and the form loses (profiles/r02_entropy_ablation.md); at config 5 the
// prototype measured x1.26 on contraction + rest (tools/ubench_mfma_entropy.hip,
// profiles/r03_entropy_ablation.md).
//
// Layout.  Workgroup = (component j, chunk of rows), four waves; a WAVE owns 16 rows of a 64-row
// batch and ALL K components (the wave-split kernel: 64 rows, a quarter of the components).
//   product 1   C'[k][row] = Delta_j[k][:] . e[row][:]      M = k (KTILES tiles of 16), N = 16 rows, D/4 steps
//       A: Delta[k = 16 kt + li][d = 4 s + lk]  (registers, loaded once per workgroup)
//       B: e[row = li][d = 4 s + lk]            (li = lane & 15, lk = lane >> 4)
//       C' element r of lane: k = 16 kt + lk + 4 r, row = li
//     so a lane holds 4 KTILES (row, k) pairs of ONE row: exponents, the two exp2, the density sums over
//     k (in-lane, then two cross-lane steps over lk), the normalised terms g -- all per lane;
//   product 2   Td[row][d] = sum_k gd[row][k] Delta_j[k][d]  M = 16 rows, N = d (tiles of 16), steps (kt, r)
//       A: gd[row = li][k = 16 kt + lk + 4 r]   = the lane's own pass-2 value: no transpose between the products
//       B: Delta[k = 16 kt + lk + 4 r][d = 16 nt + li]   (LDS)
//       C element r' of lane: row = lk + 4 r', d = 16 nt + li
// Delta_j and the per-k constants sit in LDS (the table row of component j as prep.hip wrote it,
// padding components with density exactly 0).  No barrier inside the batch loop: nothing crosses waves
// until the end-of-workgroup reduction.
#include <hip/hip_ext.h>

#include <cstdlib>

#include "adam_dev.h"
#include "common.h"
#include "entropy_args.h"
#include "fastmath.h"

namespace {

typedef double double4_t __attribute__((ext_vector_type(4)));
constexpr int WG = 256, WAVES = 4;

// 2^x: the entropy kernels' polynomial (fastmath.h VBMC_ENT_EXP2_COEFFS), coefficients as literals
__device__ __forceinline__ void exp2_pair(double x1, double x2, double& r1, double& r2) {
  const double t1 = __builtin_rint(x1), t2 = __builtin_rint(x2);
  const double f1 = x1 - t1, f2 = x2 - t2;
  const int n1 = (int)t1, n2 = (int)t2;
  constexpr int EN = VBMC_ENT_EXP2_N;
  constexpr double c[EN] = VBMC_ENT_EXP2_COEFFS;
  double p1 = c[EN - 1], p2 = c[EN - 1];
#pragma unroll
  for (int i = EN - 2; i >= 0; --i) {
    p1 = fma(p1, f1, c[i]);
    p2 = fma(p2, f2, c[i]);
  }
  p1 = fma(p1, f1, 1.0);
  p2 = fma(p2, f2, 1.0);
  r1 = __builtin_amdgcn_ldexp(p1, n1);
  r2 = __builtin_amdgcn_ldexp(p2, n2);
}
// sum over the four lanes li, li + 16, li + 32, li + 48 (the lk groups), result in all four
__device__ __forceinline__ double sum_lk(double v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// DP: D padded to a multiple of 4 (the products' depth / width); TDP: the padded D of the (j,k) table in
// memory (prep.hip: its rows are [Delta (TDP) | c0 | a | w | wis2 | pad pad]) -- equal except for D = 9, 10,
// whose table is 10 wide and whose products are 12 deep.
#if defined(MFMA_TIMES) && VBMC_MFMA_DP == 20
__device__ unsigned long long g_mfma_times[1024 * 4];  // per workgroup: start, table staged, batch loop end, reduction done
#define MF_STAMP(i) do { if (threadIdx.x == 0) g_mfma_times[(blockIdx.y * gridDim.x + blockIdx.x) * 4 + (i)] = wall_clock64(); } while (0)
#else
#define MF_STAMP(i) (void)0
#endif
// (not inlined: the entropy workgroups' code is laid out and scheduled as without it)
__device__ __noinline__ void ship_gp_sums(const double* src, double* dst, int n, uint64_t* flag, uint64_t seq) {
  staged_copy_to_host(src, dst, n);
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
template <int DP, int KTILES, int TDP = DP>
__global__ __launch_bounds__(WG, 1) void entmc_mfma_kernel(EntArgs a, const double* __restrict__ T) {
  constexpr int TS = DP + 6, TTS = TDP + 6, NS = DP / 4, NT = (DP + 15) / 16, KP = 16 * KTILES;
  static_assert(DP % 4 == 0, "D padded to a multiple of 4");
  extern __shared__ double dyn[];
  double* sT = dyn;              // [KP][TS]: the table row of component j
  double* sMu = sT + KP * TS;    // [WAVES][DP]
  double* sWk = sMu + WAVES * DP;  // [WAVES][KP]
  __shared__ double sLam[WAVES][NT * 16];
  __shared__ double sLog[WAVES];
  // Adam loop (adam.hip): grid row 0 is not an entropy row -- its first workgroup computes the
  // entropy-free part of the iteration's gradient beside the entropy workgroups (adam_dev.h), as in
  // entropy_ws.hip
  if (a.extra != nullptr && blockIdx.y == 0) {
    if (blockIdx.x == 0) {
      const adam_dev::AdamDev& pa = *(const adam_dev::AdamDev*)a.extra;
      if (a.extra_lds > 0) adam_dev::adam_pre_body<true, false, 8>(pa, dyn, &sLam[0][0], pa.pre);
      else adam_dev::adam_pre_body<false, false, 8>(pa, nullptr, &sLam[0][0], pa.pre);
    }
    return;
  }
  if (a.cancel != nullptr && __hip_atomic_load(a.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == ~(uint64_t)0) return;
  if (__builtin_expect(a.ship_flag != nullptr && blockIdx.y == gridDim.y - 1, 0)) {
    // the ship row (entropy_args.h): the prep launch's GP sums to pinned memory, then their word
    if (blockIdx.x == 0) ship_gp_sums(a.ship_src, a.ship_dst, a.ship_n, a.ship_flag, a.ship_seq);
    return;
  }
  const int D = a.ml.D, K = a.ml.K, K4 = ws_table_rows(K);  // table rows per component (entropy_args.h)
  const int j = a.extra != nullptr ? blockIdx.y - 1 : blockIdx.y, chunk = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const double sig_j = a.mix[a.ml.o_sig + j];
  const double sj2 = sig_j * sig_j, two_sj = 2.0 * sig_j;
  MF_STAMP(0);
  {
    const double* Tj = T + (size_t)j * K4 * TTS;
    // (round 6: sixteen loads in flight per thread -- in-kernel stamps put this copy of 23 KB at 3.6 us per workgroup, one
    // load waited for at a time, of the ~66 us a workgroup of config 5's launch lives)
    constexpr int U = 16;
    for (int base = 0; base < KP * TS; base += U * WG) {
      double v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = base + u * WG + tid;
        const int k = i / TS, c = i - k * TS;
        // LDS row [Delta (DP, zero beyond TDP) | c0 | a | w | wis2 | pad pad]; beyond K4: a component of density exactly 0
        const int cs = c < TDP ? c : c < DP ? -1 : TDP + (c - DP);
        const bool ld = i < KP * TS && k < K4 && cs >= 0;
        const double x = Tj[ld ? (size_t)k * TTS + cs : 0];
        v[u] = ld ? x : (k >= K4 && c == DP ? -2000.0 : 0.0);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = base + u * WG + tid;
        if (i < KP * TS) sT[i] = v[u];
      }
    }
  }
  __syncthreads();
  MF_STAMP(1);
  double a1[KTILES][NS];
#pragma unroll
  for (int kt = 0; kt < KTILES; ++kt)
#pragma unroll
    for (int s = 0; s < NS; ++s) a1[kt][s] = sT[(16 * kt + li) * TS + 4 * s + lk];

  double slog = 0.0, mu_acc[NS], lam_acc[NT], Wacc[KTILES][4];
#pragma unroll
  for (int s = 0; s < NS; ++s) mu_acc[s] = 0.0;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) lam_acc[nt] = 0.0;
#pragma unroll
  for (int kt = 0; kt < KTILES; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) Wacc[kt][r] = 0.0;

  const int rows_per_wg = a.rg * 64;
  const double* epj = a.eps + (int64_t)j * a.eps_rows * D;
  for (int it = 0; it < a.rg; ++it) {
    const int64_t row0 = (int64_t)chunk * rows_per_wg + it * 64 + wave * 16;
    const int64_t rowA = row0 + li;
    const bool valid = rowA < a.row_count;
    // ---- this lane's normals: row li, dimensions 4 s + lk ----
    double eb[NS], e2 = 0.0;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      eb[s] = (valid && 4 * s + lk < D) ? epj[rowA * D + 4 * s + lk] : 0.0;
      e2 = fma(eb[s], eb[s], e2);
    }
    const double b = sj2 * sum_lk(e2);  // sigma_j^2 |eps|^2 of row li
    // The table does not depend on the batch: left alone the compiler hoists its ~170 LDS reads per
    // lane out of the batch loop into registers and spills.  An opaque zero offset ties them to
    // this iteration (the same device as entropy_ws.hip's scalar table loads).
    int zoff;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zoff));
    const double* sTb = sT + zoff;
    // ---- product 1 and pass 1, one k-tile at a time: the lane's pairs (row li, k = 16 kt + lk + 4 r) ----
    // One wave per SIMD: nothing else hides an LDS read's latency, so the per-k constants of tile kt + 1
    // are requested BEFORE the matrix instructions and the exp2 chains of tile kt (LDS results return in
    // order; the scheduling barriers keep the requests where they are written).
    double rp[KTILES][4], rm[KTILES][4], qp = 0.0, qm = 0.0;
    double cur[4][3], nxt[4][3];  // [r][c0 | a | w]
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int q = 0; q < 3; ++q) cur[r][q] = sTb[(lk + 4 * r) * TS + DP + q];
#pragma unroll
    for (int kt = 0; kt < KTILES; ++kt) {
      if (kt + 1 < KTILES) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int q = 0; q < 3; ++q) nxt[r][q] = sTb[(16 * (kt + 1) + lk + 4 * r) * TS + DP + q];
      }
      __builtin_amdgcn_sched_barrier(0);
      double4_t c = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < NS; ++s) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[kt][s], eb[s], c, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double sp = fma(two_sj, c[r], b), sm = fma(-two_sj, c[r], b);
        exp2_pair(fma(cur[r][1], sp, cur[r][0]), fma(cur[r][1], sm, cur[r][0]), rp[kt][r], rm[kt][r]);
        qp = fma(cur[r][2], rp[kt][r], qp);
        qm = fma(cur[r][2], rm[kt][r], qm);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 3; ++q) cur[r][q] = nxt[r][q];
    }
    // pass 2's first tile: requested before the cross-lane sums and the logarithm
    double w2c[4], b2c[4][NT], w2n[4], b2n[4][NT];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double* tk = sTb + (lk + 4 * r) * TS;
      w2c[r] = tk[DP + 3];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b2c[r][nt] = tk[min(16 * nt + li, TS - 1)];
    }
    __builtin_amdgcn_sched_barrier(0);
    qp = sum_lk(qp);
    qm = sum_lk(qm);
    if (lk < 2) {  // log q: the lk = 0 lanes take the + samples, lk = 1 the - samples
      const double lq = fm::log_fast(lk == 0 ? qp : qm);
      slog += valid ? lq : 0.0;
    }
    // ---- pass 2: normalised terms (entropy_ws.hip, pass 2); every gd goes straight into product 2,
    // Td[row][d] = sum_k gd[row][k] Delta[k][d], as its A operand ----
    const double ip = valid ? fm::rcp_fast(qp) : 0.0, im = valid ? fm::rcp_fast(qm) : 0.0;
    double sgs = 0.0, sgd = 0.0;
    double4_t td2[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) td2[nt] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kt = 0; kt < KTILES; ++kt) {
      if (kt + 1 < KTILES) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double* tk = sTb + (16 * (kt + 1) + lk + 4 * r) * TS;
          w2n[r] = tk[DP + 3];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) b2n[r][nt] = tk[min(16 * nt + li, TS - 1)];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double t2 = rm[kt][r] * im;
        const double ts = fma(rp[kt][r], ip, t2), td = fma(rp[kt][r], ip, -t2);  // (entropy_ws.hip pass 2)
        Wacc[kt][r] += ts;
        sgs = fma(ts, w2c[r], sgs);
        const double gd = td * w2c[r];
        sgd += gd;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          // (columns d >= DP of the last tile read the row's constants instead of zeros: their outputs,
          // Td[.][d >= DP], are never used -- the lam sums below take d < D only)
          td2[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(gd, b2c[r][nt], td2[nt], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        w2c[r] = w2n[r];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b2c[r][nt] = b2n[r][nt];
      }
    }
    const double cs = sig_j * sum_lk(sgs), cd = sig_j * sum_lk(sgd);  // of row li, in all four lk lanes
#pragma unroll
    for (int s = 0; s < NS; ++s) mu_acc[s] = fma(eb[s], cd, mu_acc[s]);
    // lam += e_d (e_d cs + Td_d) over the rows, in product 2's layout: row = lk + 4 r, d = 16 nt + li
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rl = lk + 4 * r;
      const double csr = __shfl(cs, rl, 64);
      const int64_t rowC = row0 + rl;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int d = 16 * nt + li;
        const double ec = (rowC < a.row_count && d < D) ? epj[rowC * D + d] : 0.0;
        lam_acc[nt] = fma(ec, fma(ec, csr, td2[nt][r]), lam_acc[nt]);
      }
    }
  }

  MF_STAMP(2);
  // ---- workgroup reduction -> the partial row [Slog | mu (D) | sig | lam (D) | W (K)] ----
  {
    const double v = fm::wave_sum_dpp(slog);
    if (lane == 0) sLog[wave] = v;
  }
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const double v = fm::row16_sum_dpp(mu_acc[s]);  // over the rows li
    if (li == 0) sMu[wave * DP + 4 * s + lk] = v;
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const double v = sum_lk(lam_acc[nt]);  // over the row groups lk
    if (lk == 0) sLam[wave][16 * nt + li] = v;
  }
#pragma unroll
  for (int kt = 0; kt < KTILES; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double v = fm::row16_sum_dpp(Wacc[kt][r]);
      if (li == 0) sWk[wave * KP + 16 * kt + lk + 4 * r] = v;
    }
  __syncthreads();
  MF_STAMP(3);
  double* out = a.partial + ((int64_t)j * a.chunks + chunk) * a.stride;
  for (int t = tid; t < a.stride; t += WG) {
    double v = 0.0;
    if (t == 0) {
      for (int wv = 0; wv < WAVES; ++wv) v += sLog[wv];
    } else if (t <= D) {
      for (int wv = 0; wv < WAVES; ++wv) v += sMu[wv * DP + t - 1];
    } else if (t == D + 1) {
      for (int d = 0; d < D; ++d)
        for (int wv = 0; wv < WAVES; ++wv) v += sLam[wv][d];
    } else if (t < 2 * D + 2) {
      for (int wv = 0; wv < WAVES; ++wv) v += sLam[wv][t - (D + 2)];
    } else {
      for (int wv = 0; wv < WAVES; ++wv) v += sWk[wv * KP + t - (2 * D + 2)];
    }
    out[t] = v;
  }
}

template <int DP, int KTILES, int TDP = DP>
void launch_mfma(hipStream_t st, const EntArgs& a, const double* d_table, hipEvent_t e0, hipEvent_t e1) {
  constexpr int TS = DP + 6, KP = 16 * KTILES;
  size_t lds = sizeof(double) * ((size_t)KP * TS + WAVES * DP + WAVES * KP);
  const bool extra_row = a.extra != nullptr;
  if (extra_row && sizeof(double) * (size_t)a.extra_lds > lds) lds = sizeof(double) * (size_t)a.extra_lds;
  auto kern = entmc_mfma_kernel<DP, KTILES, TDP>;
  static size_t lds_limit[64] = {};  // per device: function attributes are per device
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (lds > 32 * 1024 && lds > lds_limit[dev & 63]) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    lds_limit[dev & 63] = lds;
  }
  hipExtLaunchKernelGGL(kern, dim3(a.chunks, a.ml.K + (extra_row ? 1 : 0) + (a.ship_flag != nullptr ? 1 : 0)), dim3(WG), (std::uint32_t)lds, st, e0, e1, 0u, a,
                        d_table);
}

}  // namespace

// One translation unit per padded D (build.py: -DVBMC_MFMA_DP=16 / 20 / 24 / 32), each with the k-tile counts 3..8.
#ifndef VBMC_MFMA_DP
#error "compile with -DVBMC_MFMA_DP=<padded D: 16, 20, 24 or 32>"
#endif
#define VBMC_CAT2(a, b) a##b
#define VBMC_CAT(a, b) VBMC_CAT2(a, b)
void VBMC_CAT(launch_entmc_mfma_dp, VBMC_MFMA_DP)(hipStream_t st, const EntArgs& a, int ktiles, const double* d_table, hipEvent_t e0,
                                                   hipEvent_t e1) {
#if VBMC_MFMA_DP == 12
  if (a.ml.D <= 10) {  // the table of D = 9, 10 is 10 wide (entropy.hip padded_d)
    switch (ktiles) {
      case 3: launch_mfma<12, 3, 10>(st, a, d_table, e0, e1); break;
      case 4: launch_mfma<12, 4, 10>(st, a, d_table, e0, e1); break;
      case 5: launch_mfma<12, 5, 10>(st, a, d_table, e0, e1); break;
      case 6: launch_mfma<12, 6, 10>(st, a, d_table, e0, e1); break;
      case 7: launch_mfma<12, 7, 10>(st, a, d_table, e0, e1); break;
      default: launch_mfma<12, 8, 10>(st, a, d_table, e0, e1); break;
    }
    return;
  }
#endif
  switch (ktiles) {
    case 3: launch_mfma<VBMC_MFMA_DP, 3>(st, a, d_table, e0, e1); break;
    case 4: launch_mfma<VBMC_MFMA_DP, 4>(st, a, d_table, e0, e1); break;
    case 5: launch_mfma<VBMC_MFMA_DP, 5>(st, a, d_table, e0, e1); break;
    case 6: launch_mfma<VBMC_MFMA_DP, 6>(st, a, d_table, e0, e1); break;
    case 7: launch_mfma<VBMC_MFMA_DP, 7>(st, a, d_table, e0, e1); break;
    default: launch_mfma<VBMC_MFMA_DP, 8>(st, a, d_table, e0, e1); break;
  }
}

#if VBMC_MFMA_DP == 20
void launch_entmc_mfma_dp12(hipStream_t, const EntArgs&, int, const double*, hipEvent_t, hipEvent_t);
void launch_entmc_mfma_dp16(hipStream_t, const EntArgs&, int, const double*, hipEvent_t, hipEvent_t);
void launch_entmc_mfma_dp24(hipStream_t, const EntArgs&, int, const double*, hipEvent_t, hipEvent_t);
void launch_entmc_mfma_dp32(hipStream_t, const EntArgs&, int, const double*, hipEvent_t, hipEvent_t);

// Shapes this form is built for: resident draws, value + gradient, no GP-sums grid row (the optimiser
// loop's pre row is supported); D padded to 12 (tables 10 or 12 wide), 16, 20, 24 or 32; K within 12
// components below a multiple of 16 (the exp2 work is padded with the tile) and at least 36; and only
// where the wave-split kernel would run one wave per SIMD -- the regime this form exists for
// (D <= 10: from K = 81, the wave-split kernel's <10, 20> build is still quick: 29 against 38 us at K = 72;
// tools/mfma_probe.py: 1.09-1.28x at D = 10, K = 96-128; 1.06-1.34x at D = 12, K = 56-128; 1.3-1.85x at D = 16; 1.1-1.35x at D = 20;
// 1.6-2.1x at D = 24; 2.2-3.5x at D = 32).  In the two-waves regime it loses or ties (BASELINE config 3,
// D = 10, K = 50: 105 against 88 us, tools/mfma_c3_probe.py with VBMC_MFMA_ANY=1 -- K = 50 pads the exp2
// work itself by 28 %).
static int mfma_ktiles(int K4) {
  const int kt = (K4 + 15) / 16;
  return (kt >= 3 && kt <= 8 && 16 * kt - K4 <= 12) ? kt : 0;
}
bool entmc_mfma_applies(const EntArgs& a, int DP) {
  const int K4 = ((a.ml.K + 3) / 4) * 4;
  static const bool any_regime = [] { const char* e = getenv("VBMC_MFMA_ANY"); return e && e[0] == '1'; }();  // experiments
  const int dp4 = DP == 10 ? 12 : DP;
  return (dp4 == 12 || DP == 16 || DP == 20 || DP == 24 || DP == 32) && mfma_ktiles(K4) != 0 &&
         (any_regime || (ws_min_waves(DP, ws_ktmax_for(a.ml.K), true) == 1 && (DP != 10 || ws_ktmax_for(a.ml.K) >= 25))) && a.want_grad && a.eps_mode != VBMC_EPS_PHILOX &&
         a.eps != nullptr && a.gp_items == 0;
}

void launch_entmc_mfma(hipStream_t st, const EntArgs& a, int DP, const double* d_table, hipEvent_t e0, hipEvent_t e1) {
  const int kt = mfma_ktiles(((a.ml.K + 3) / 4) * 4);
  switch (DP) {
    case 10:
    case 12: launch_entmc_mfma_dp12(st, a, kt, d_table, e0, e1); break;
    case 16: launch_entmc_mfma_dp16(st, a, kt, d_table, e0, e1); break;
    case 20: launch_entmc_mfma_dp20(st, a, kt, d_table, e0, e1); break;
    case 24: launch_entmc_mfma_dp24(st, a, kt, d_table, e0, e1); break;
    default: launch_entmc_mfma_dp32(st, a, kt, d_table, e0, e1); break;
  }
}
#endif

#if defined(MFMA_TIMES) && VBMC_MFMA_DP == 20
extern "C" int vbmc_debug_mfma_times(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mfma_times), sizeof(unsigned long long) * n);
}
#endif
