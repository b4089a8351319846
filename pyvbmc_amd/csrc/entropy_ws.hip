// Monte-Carlo entropy, main kernel -- reference entropy/entmc_vbmc.py:64-112.
//
// Per evaluated (sample, component) pair the work is ~3D FMAs + one exp, all float64.
// On gfx950 the FP64 vector FMA and the FP64 MFMA share one pipe and one peak
// (78.6 TFLOP/s; measured 70 / 78.8 with NO overlap between the two,
// tools/ubench_fp64.hip), so recasting the K x D distance block as a matrix product
// buys nothing: this is a VALU kernel whose job is to spend as few issue slots per
// pair as possible and to keep the VALU fed.
//
//   * antithetic samples x+- = mu_j +- sigma_j lambda o eps share the dot product
//     Delta_k . eps of their squared distances (D FMAs for two samples);
//   * each exp() is evaluated ONCE.  The K densities of a sample are needed twice
//     (for q = sum_k w_k r_k and, once q is known, for sum_n r_k/q), so they stay in
//     registers; to make that fit, the K components are split over the 4 waves of a
//     workgroup ("ws" = wave-split):
//                 lane = antithetic-pair row (64 rows per batch),
//                 wave w owns components k = w, w+4, w+8, ...  (KT = ceil(K/4)).
//     Everything indexed by k is therefore wave-uniform: the per-(j,k) constants
//     (Delta_jk row, log2 density at the component mean, exponent scale, weights) come from a
//     small precomputed table through SCALAR loads and feed the FMAs as SGPR operands
//     -- the inner loop issues no vector-memory or LDS instruction at all;
//   * only q+ and q- cross waves (one LDS exchange + barrier per 64-row batch); every
//     gradient accumulator is linear in the per-wave partial sums and is reduced once
//     per workgroup;
//   * exp is exp2 with log2(e), -1/(2 sigma_k^2) and log2 of the normalisation folded
//     into one FMA; log is an atanh series (fastmath.h);
//   * the Delta part of the mean gradient, summed over the rows, only needs the W sums
//     the weight gradient already collects, so it is formed by the finish kernel.
// Output: the same per-workgroup partial rows as entropy.hip, reduced by its kernels
// (entmc_finish_kernel, mu_from_w = 1).
#include <hip/hip_ext.h>

#include "adam_dev.h"
#include "common.h"
#include "entropy_args.h"
#include "fastmath.h"
#include "glj_block.h"

// exp2 with its polynomial coefficients held in VGPRs (frees SGPRs for table rows); the polynomial: fastmath.h,
// VBMC_ENT_EXP2_COEFFS (degree 8, 1.07e-12 -- the accuracy argument is there).
#ifndef WS_EXP2_N
#define WS_EXP2_N VBMC_ENT_EXP2_N
#endif
constexpr int EN = WS_EXP2_N;  // coefficients beside the constant 1: the polynomial's degree
__device__ __forceinline__ double exp2_vc(double x, const double (&c)[EN]) {
  const double t = __builtin_rint(x);
  const double f = x - t;
  const int n = (int)t;
  double p = c[EN - 1];
#pragma unroll
  for (int i = EN - 2; i >= 0; --i) p = fma(p, f, c[i]);
  p = fma(p, f, 1.0);
  return __builtin_amdgcn_ldexp(p, n);
}
// two independent exp2 evaluations with their Horner chains interleaved step by step
// (a dependent v_fma_f64 chain alone leaves the FP64 pipe half idle)
__device__ __forceinline__ void exp2_vc2(double x1, double x2, const double (&c)[EN], double& r1,
                                         double& r2) {
  const double t1 = __builtin_rint(x1), t2 = __builtin_rint(x2);
  const double f1 = x1 - t1, f2 = x2 - t2;
  const int n1 = (int)t1, n2 = (int)t2;
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
#if WS_EXP2_N == VBMC_ENT_EXP2_N
__device__ const double kExp2C[EN] = VBMC_ENT_EXP2_COEFFS;
#elif WS_EXP2_N == 10  // rounds 1-3 (4.1e-16); kept for the A/B in profiles/r04_exp2_degree.md
__device__ const double kExp2C[10] = {0x1.62e42fefa3a19p-1, 0x1.ebfbdff82c598p-3, 0x1.c6b08d703ce49p-5, 0x1.3b2ab6fba1ddap-7, 0x1.5d87fe9d7a584p-10, 0x1.430913096fd9fp-13, 0x1.ffcb54062e698p-17, 0x1.62bfd47773353p-20, 0x1.b675bca4eeebbp-24, 0x1.e6063f7217bc6p-28};
#elif WS_EXP2_N == 9  // 3.8e-14
__device__ const double kExp2C[9] = {0x1.62e42fefa39f7p-1, 0x1.ebfbdff8149f2p-3, 0x1.c6b08d7044119p-5, 0x1.3b2ab72b175eep-7, 0x1.5d87fe908f88ap-10, 0x1.43088e257f341p-13, 0x1.ffcb76789860fp-17, 0x1.63ef969a64d3cp-20, 0x1.b6571de2f2351p-24};
#endif

#include "philox.h"

#ifndef VBMC_DP
#error "compile with -DVBMC_DP=<padded D>"
#endif

namespace {

constexpr int WG = 256;
constexpr int WAVES = WG / 64;

__device__ __forceinline__ double wave_sum(double v) { return fm::wave_sum_dpp(v); }

// The (j,k) table rows [Delta_jk (DP) | a | c | lrc | w | wis2 | pad] are written by prep.hip, always
// 4 KTMAX of them per component j (padding components have zero density, entropy_args.h): the loops
// over a wave's components carry no guards.
// Per-lane state is ~(4 DP + 3 KTMAX) doubles with gradients.  Up to ~95 it fits the 256
// registers of 2 waves/SIMD; beyond that one wave per SIMD with the 512-register budget
// (AGPRs as spill space) beats spilling to scratch memory.
// (ws_min_waves: entropy_args.h -- the host uses the same rule to count free workgroup slots)
// the smallest D this translation unit's kernels see (padded_d: D pads up to the next entry of VBMC_WS_DPS); glj_block.h
// instantiates the one form of the GP block that range needs
constexpr int ws_dmin(int dp) { return dp <= 12 ? dp - 1 : dp == 16 ? 13 : dp == 20 ? 17 : dp == 24 ? 21 : 25; }

// End-of-workgroup reduction: every wave holds 1 + 2 DP + KT per-lane accumulators whose 64 lanes have to be added up.
// Round 6 form: NQ DPP steps add each accumulator up inside groups of 2^NQ neighbouring lanes (independent across the
// accumulators: they pipeline), the 64 / 2^NQ group sums of every (wave, item) go through LDS as one row [NV (+1)], ONE
// thread per row adds them up in a fixed order, and the output stage adds the four waves.  NQ is the smallest step count
// whose buffer fits the kernel's occupancy (one step, 32 values per row, 36 KB at D_p = 10, KT = 13).
// Measured at BASELINE config 3's shape (profiles/r06_notes.md): rounds 3-5 laid all 64 lanes down (rows of 65, in two
// half-size passes to bound the buffer, a thread per row): 3.1 us per workgroup behind its last batch at EVERY problem
// size by in-kernel stamps, 3.2 us of the 64 us launch by ablation; NQ = 1 / 2 / 3: 62.5 / 62.7 / 63.0 us per launch
// against 64.0-64.5, 12.1 / 12.3 / 12.9 against 13.6 us at 1 024 samples per component.  Full DPP wave reductions
// (~20 dependent instructions per accumulator) remain the fallback where no buffer fits.
// The buffer lives in the dynamic LDS region (which the Adam loop's pre workgroup uses instead, adam_dev.h).
constexpr int ws_epi_cap_doubles(int dp, int ktmax, bool grad) { return (ws_min_waves(dp, ktmax, grad) >= 2 ? 44 : 100) * 1024 / 8; }
// DPP steps in front of the LDS pass (0: no buffer fits -- DPP reductions all the way)
constexpr int ws_epi_nq(int dp, int ktmax, bool grad) {
  if (!grad) return 0;
  const int ni = 1 + 2 * dp + ktmax;
  for (int nq = 1; nq <= 3; ++nq)
    if (WAVES * ni * ((64 >> nq) + 1) <= ws_epi_cap_doubles(dp, ktmax, grad)) return nq;
  return 0;
}
// the buffer's size in doubles (0 = none)
constexpr int ws_epi_doubles(int dp, int ktmax, bool grad) {
  const int nq = ws_epi_nq(dp, ktmax, grad);
  return nq == 0 ? 0 : WAVES * (1 + 2 * dp + ktmax) * ((64 >> nq) + 1);
}

#if defined(WS_TIMES) && VBMC_DP == 10
__device__ unsigned long long g_ws_times[1024 * 4];  // per workgroup: start, batch loop start, batch loop end, end
#define WS_STAMP(i) do { if (threadIdx.x == 0) g_ws_times[(blockIdx.y * gridDim.x + blockIdx.x) * 4 + (i)] = wall_clock64(); } while (0)
#else
#define WS_STAMP(i) (void)0
#endif
// STAGE (resident draws only; the host chooses, launch_one): how a batch's 64 rows of draws -- one contiguous block of
// 512 D bytes -- reach the waves.  1: D == D_p and 16-byte aligned (every even D): LDS-direct 16-byte loads one batch
// ahead; 2: any D: the same with rows D doubles apart in LDS (16-byte loads when the block starts 16-byte aligned, else
// 2 D instructions of 256 bytes);
// 0: each wave fetches its rows itself with guarded loads (in-line Philox builds: unused)
constexpr int ws_dprev(int dp) { return dp <= 12 ? dp - 2 : dp == 16 ? 12 : dp == 20 ? 16 : dp == 24 ? 20 : 24; }
template <int DP, int KTMAX, bool GRAD, bool PHILOX, int STAGE = 0>
__global__ __launch_bounds__(WG, ws_min_waves(DP, KTMAX, GRAD)) void entmc_ws_kernel(
    EntArgs a, const double* __restrict__ T) {
  constexpr int TS = DP + 6;
  __shared__ double sQ[2][WAVES][2][64];          // q partials, double-buffered by batch parity
  constexpr int GB = 4;                           // Philox mode: batches generated per round
  // Philox mode: their normals [batch][d][row].  Resident draws: the rows of this batch and of the next one as they
  // lie in memory ([2][64 rows][DP]), filled one batch ahead by LDS-direct loads (ws_fill below)
  __shared__ __attribute__((aligned(16))) double sE[PHILOX ? GB : 2][DP][64];
  __shared__ double sRed[WAVES][2 * DP + 1];
  extern __shared__ double dyn[];                 // [epilogue transpose buffer | sW[K4]]
  constexpr int EPI = ws_epi_doubles(DP, KTMAX, GRAD);
  double* sW = dyn + EPI;

  // Adam loop (adam.hip): grid row 0 is not an entropy row -- its first workgroup computes the
  // entropy-free part of the iteration's gradient beside the entropy workgroups (adam_dev.h)
  constexpr bool EXTRA_ROW = GRAD && !PHILOX;
  const bool span = a.sp.cus > 0;  // (span mode: a one-dimensional grid, entropy_args.h WsSpan)
  if constexpr (EXTRA_ROW) {
    // (span mode: the one workgroup behind the entropy parts and the GP row -- it finds a free slot at once, the
    // plan leaves some: entropy_args.h)
    if (a.extra != nullptr && (span ? (int)blockIdx.x == a.sp.n_parts() + (a.gp_items > 0 ? a.gp_wgs : 0) : blockIdx.y == 0)) {
      if (span || blockIdx.x == 0) {
        const adam_dev::AdamDev& pa = *(const adam_dev::AdamDev*)a.extra;
        if (a.gp_items > 0 && a.gp.done.dev) {
          // the GP sums this workgroup finalises are made by workgroups of this very launch (dispatched in front of
          // it): wait for their word, then drop this CU's cached lines (hand-off with drained write-through stores
          // and an agent-scope flag).  Bounded: after 20 ms the status word says so and the host gives the run up.
          if (threadIdx.x == 0) {
            const unsigned long long t0 = wall_clock64();
            while (__hip_atomic_load(a.gp.done.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.gp.done.seq) {
              if (wall_clock64() - t0 > 2000000ull) {
                atomicOr(pa.status, 8);
                break;
              }
              __builtin_amdgcn_s_sleep(8);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          }
          __syncthreads();
        }
        if (a.extra_lds > 0) adam_dev::adam_pre_body<true>(pa, dyn, &sRed[0][0], pa.pre);
        else adam_dev::adam_pre_body<false>(pa, nullptr, &sRed[0][0], pa.pre);
      }
      return;
    }
  }
  // armed evaluation that the host cancelled (common.h ArmedEval): nothing to do
  if (a.cancel != nullptr && __hip_atomic_load(a.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == ~(uint64_t)0) return;
  // Host-driven step (api_elbo.hip): the LAST workgroups of the grid work through the GP expected-log-joint
  // items in the workgroup slots this launch's entropy parts leave free (entropy_args.h)
  if (a.gp_items > 0 && (span ? (int)blockIdx.x >= a.sp.n_parts() && (int)blockIdx.x < a.sp.n_parts() + a.gp_wgs : blockIdx.y == gridDim.y - 1)) {
    const int first = span ? (int)blockIdx.x - a.sp.n_parts() : (int)blockIdx.x, step = span ? a.gp_wgs : (int)gridDim.x;
    if (a.gp.x_lds) {  // X^T staged once for all of this workgroup's items (glj_block.h)
      if (first < a.gp_items) glj_stage_x(a.gp, dyn);
      for (int it = first; it < a.gp_items; it += step) {
        glj_block<true, DP, ws_dmin(DP), true>(a.gp, it, dyn);
        __syncthreads();
      }
    } else {
      for (int it = first; it < a.gp_items; it += step) {
        glj_block<false, DP, ws_dmin(DP), true>(a.gp, it, dyn);
        __syncthreads();
      }
    }
    return;
  }
  const int D = a.ml.D;
  constexpr int KT = KTMAX, K4 = KT * 4;
  // ---- this workgroup's batches: [g_lo, g_hi) of the component-major batch list, cut into stretches inside one
  // component.  Chunk mode: one stretch (component j = blockIdx.y, batches [chunk rg, chunk rg + rg) of its nb). ----
  int64_t g_lo, g_hi;
  int nb, part = 0, chunk_slot = 0, j_chunk = 0;
  int nbv;  // slots per component in the list that is cut: its batches, then (span mode) the padding slots
  if (span) {
    nb = a.sp.nb;
    nbv = a.sp.nbv();
    part = a.sp.part_of_block((int)blockIdx.x);
    g_lo = a.sp.lo(part);
    g_hi = a.sp.lo(part + 1);
  } else {
    int j = (EXTRA_ROW && a.extra != nullptr) ? blockIdx.y - 1 : blockIdx.y, chunk = blockIdx.x;
    // Scalar-cache locality.  With two workgroups resident per CU and a grid of one round, workgroups
    // b and b + CUs of a launch share a CU (tools/probes/placement.hip); in grid order they would read
    // table rows j and j + CUs / chunks -- every CU (and the neighbour it shares its 16 KB scalar cache
    // with) then sweeps distinct 6.6 KB rows 32 times each and the rows evict each other.  Hand the
    // (j, chunk) items out so that co-resident workgroups take neighbouring items, i.e. the same j.
    if (a.pair_cus > 0) {
      const int total = a.ml.K * a.chunks, b = j * a.chunks + chunk;
      const int paired = total - a.pair_cus;  // workgroups of the second round = pairs (b, b + pair_cus)
      if (paired > 0 && b < 2 * a.pair_cus) {
        const int m = b < a.pair_cus ? b : b - a.pair_cus;
        const int item = m < paired ? 2 * m + (b >= a.pair_cus ? 1 : 0) : paired + m;
        j = item / a.chunks;
        chunk = item - j * a.chunks;
      }
    }
    nb = (int)((a.row_count + 63) >> 6);
    nbv = nb;
    chunk_slot = chunk;
    j_chunk = j;
    const int64_t i0 = (int64_t)chunk * a.rg;
    g_lo = (int64_t)j * nb + (i0 < nb ? i0 : nb);
    g_hi = (int64_t)j * nb + (i0 + a.rg < nb ? i0 + a.rg : nb);
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  WS_STAMP(0);
  constexpr bool eps_exact = !PHILOX && STAGE != 0;

  double ec[EN];
  {
    int vz;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vz));  // opaque per-lane zero: keeps the loads vector loads
#pragma unroll
    for (int i = 0; i < EN; ++i) ec[i] = kExp2C[i + vz];
  }
  WS_STAMP(1);
  bool first_stretch = true;
  for (int64_t g = g_lo; g < g_hi || (first_stretch && !span);) {  // (chunk mode: an empty chunk still writes its zero row)
  first_stretch = false;
  const int j = span ? (int)(g / nbv) : j_chunk;
  const int ib0 = (int)(g - (int64_t)j * nbv);  // first batch of the stretch within component j (or a padding slot >= nb)
  if (span && ib0 >= nb) {                       // padding behind component j: no work (chunk mode: an empty chunk falls through with n_it = 0 and writes its zero row)
    g = (int64_t)(j + 1) * nbv;
    continue;
  }
  const int n_it = g < g_hi ? (int)((g_hi - g < nb - ib0) ? g_hi - g : nb - ib0) : 0;  // its batches
  const int slot = span ? part - a.sp.part_of((int64_t)j * nbv) : chunk_slot;    // its partial row among component j's
  const double sig_j = a.mix[a.ml.o_sig + j];
  const double sj2 = sig_j * sig_j;
  const double two_sj = 2.0 * sig_j;
  const double* Tj = T + (size_t)j * K4 * TS;
  double slog_acc = 0.0;
  double mu_acc[DP], lam_acc[DP], Wacc[KTMAX];
#pragma unroll
  for (int d = 0; d < DP; ++d) mu_acc[d] = lam_acc[d] = 0.0;
#pragma unroll
  for (int kk = 0; kk < KTMAX; ++kk) Wacc[kk] = 0.0;

  // LDS-direct copy of batch ib's rows into sE[buf]: D_p / 2 instructions of 1 KB, dealt over the waves.  The compiler
  // does not count these loads: the wave drains them (vmcnt) in front of the batch's barrier, one batch after issue.
  auto ws_fill = [&](int ib, int buf) {
    if constexpr (!PHILOX && STAGE == 1) {
      const char* base = (const char*)(a.eps + ((int64_t)j * a.eps_rows + ((int64_t)ib << 6)) * DP);
      const int64_t left = (a.row_count - ((int64_t)ib << 6)) * (DP * 8) - 16;  // last 16 bytes of the slice, relative
      const unsigned last = (unsigned)(left < 64 * DP * 8 - 16 ? left : 64 * DP * 8 - 16);
#pragma unroll
      for (int c = 0; c < DP / 2; ++c) {
        if ((c & 3) != wave) continue;
        unsigned voff = (unsigned)(c * 1024 + lane * 16);
        voff = voff < last ? voff : last;
        const unsigned dst = (unsigned)(uintptr_t)&sE[buf][0][0] + (unsigned)(c * 1024);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(base), "s"(dst) : "memory");
      }
    } else if constexpr (!PHILOX && STAGE == 2) {
      const char* base = (const char*)(a.eps + ((int64_t)j * a.eps_rows + ((int64_t)ib << 6)) * D);
      const int64_t left = (a.row_count - ((int64_t)ib << 6)) * (int64_t)(D * 8);  // bytes of the slice from here on
      const int blk = 64 * D * 8;
      // 16-byte loads need the block's start AND its end on 16-byte boundaries (a lane's 16 bytes land at ITS LDS slot
      // wherever they were read from: the clamped tail below cannot deliver a trailing half)
      if ((((uintptr_t)base) & 15) == 0 && (left >= blk || (left >= 16 && (left & 15) == 0))) {
        const unsigned last = (unsigned)((left < blk ? left : blk) - 16);
        for (int c = wave; c < (D + 1) / 2; c += WAVES) {  // 1 KB per instruction
          unsigned voff = (unsigned)(c * 1024 + lane * 16);
          voff = voff < last ? voff : last;
          const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(uintptr_t)&sE[buf][0][0] + (unsigned)(c * 1024)));
          asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(base), "s"(dst) : "memory");
        }
      } else {
        const unsigned last = (unsigned)((left < blk ? left : blk) - 4);
        for (int c = wave; c < 2 * D; c += WAVES) {  // 256 bytes per instruction
          unsigned voff = (unsigned)(c * 256 + lane * 4);
          voff = voff < last ? voff : last;
          const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(uintptr_t)&sE[buf][0][0] + (unsigned)(c * 256)));
          asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" : : "v"(voff), "s"(base), "s"(dst) : "memory");
        }
      }
    }
  };
  (void)ws_fill;
  if (eps_exact && n_it > 0) {
    ws_fill(ib0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  // log q of a row is taken once per LOGB batches: the mantissas of the q are multiplied up and their exponents
  // added (six instructions per batch instead of the ~30 of a logarithm on the two waves that carry it)
  constexpr int LOGB = 8;
  double lmant = 1.0;
  int lexp = 0;

  for (int it = 0; it < n_it; ++it) {
    const int64_t i_loc = ((int64_t)(ib0 + it) << 6) + lane;
    const bool valid = i_loc < a.row_count;

    // ---- this row's D standard normals ----
    double e[DP];
#pragma unroll
    for (int d = 0; d < DP; ++d) e[d] = 0.0;
    if (!PHILOX) {
      if constexpr (eps_exact) {
        // D == D_p (every even D): the batch's 64 rows are one contiguous block of 512 D_p bytes that the four waves
        // copied into LDS while the previous batch was computed (no per-dimension guards, the load latency off the
        // wave's path, every row fetched once per workgroup instead of once per wave); rows beyond the slice hold
        // copies of its last 16 bytes -- they only have to be finite, every sum they enter is masked (valid) below
        if (it + 1 < n_it) ws_fill(ib0 + it + 1, (it + 1) & 1);
        if constexpr (STAGE == 1) {
          const double2* rp = (const double2*)&sE[it & 1][0][0] + lane * (DP / 2);
#pragma unroll
          for (int d = 0; d < DP / 2; ++d) {
            const double2 v = rp[d];
            e[2 * d] = v.x;
            e[2 * d + 1] = v.y;
          }
        } else {
          // rows D doubles apart: D_p reads whatever D is (what lies behind a row is never used: dimensions >= D are set
          // to zero; only those above the next smaller padded width can be >= D at all)
          const double* rp = &sE[it & 1][0][0] + lane * D;
#pragma unroll
          for (int d = 0; d < DP; ++d) e[d] = rp[d];
#pragma unroll
          for (int d = ws_dprev(DP) + 1; d < DP; ++d) e[d] = d < D ? e[d] : 0.0;
        }
      } else if (valid) {
        const double* rp = a.eps + ((int64_t)j * a.eps_rows + i_loc) * D;
#pragma unroll
        for (int d = 0; d < DP; ++d)
          if (d < D) e[d] = rp[d];
      }
    } else {
      // Every GB-th batch the four waves generate the Philox blocks (four normals each, philox.h)
      // of the next GB batches together: (batch, block) slots are dealt round-robin over the waves,
      // so each block is generated exactly once per workgroup and the waves stay balanced.  LDS
      // hands the normals to all waves.
      constexpr int NP = DP / 2;
      const int nbk = (D + 3) / 4;  // blocks actually needed
      if ((it % GB) == 0) {
        const int nb = min(GB, n_it - it);
        __syncthreads();  // readers of the previous round are done with sE
        for (int q = wave; q < nb * nbk; q += WAVES) {
          const int bq = q / nbk, blk = q - bq * nbk;
          const int64_t il = ((int64_t)(ib0 + it + bq) << 6) + lane;
          const uint64_t grow = (uint64_t)j * (uint64_t)a.n_half + (uint64_t)(a.row_begin + il);
          double z[4];
          philox_normal_quad(grow, (uint32_t)blk, a.seed, z);
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (4 * blk + c < DP) sE[bq][4 * blk + c][lane] = z[c];
        }
        __syncthreads();
      }
      (void)NP;
#pragma unroll
      for (int d = 0; d < DP; ++d)
        if (valid && d < D) e[d] = sE[it % GB][d][lane];
    }
    double e2 = 0.0;
#pragma unroll
    for (int d = 0; d < DP; ++d) e2 = fma(e[d], e[d], e2);
    const double b = sj2 * e2;

    // The wave's table rows do not depend on the batch; left alone the compiler hoists all
    // of them out of the batch loop into spilled SGPRs (hundreds of v_readlane).  An opaque
    // zero offset ties the loads to this iteration so they stay streaming scalar loads.
    int zoff;
    asm volatile("s_mov_b32 %0, 0" : "=s"(zoff));
    const double* Tw = Tj + (size_t)wave * TS + zoff;

    // ---- pass 1 over this wave's components: densities (cached) and partial q ----
    // The table row of component kk+1 is requested (scalar loads -> SGPRs) as soon as row kk
    // has landed, so its latency hides behind the ~45 VALU instructions of component kk.
    // Scalar loads return out of order, hence the explicit "wait, then issue" sequence.
    double qp = 0.0, qm = 0.0;
    double rp_[KTMAX], rm_[KTMAX];
    constexpr int NR1 = DP + 3;  // Delta, c0, a, w
    double cur[NR1], nxt[NR1];
    {
      const double* tr = Tw;
#pragma unroll
      for (int i = 0; i < NR1; ++i) cur[i] = tr[i];
    }
#pragma unroll
    for (int kk = 0; kk < KTMAX; ++kk) {
      {
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): row kk is in SGPRs
        __builtin_amdgcn_sched_barrier(0);
        if (kk + 1 < KTMAX) {
          const double* tn = Tw + (size_t)(4 * (kk + 1)) * TS;  // wave-uniform
#pragma unroll
          for (int i = 0; i < NR1; ++i) nxt[i] = tn[i];
        }
        __builtin_amdgcn_sched_barrier(0);
        double c = 0.0;
#pragma unroll
        for (int d = 0; d < DP; ++d) c = fma(cur[d], e[d], c);
        const double sp = fma(two_sj, c, b);   // sigma_j^2 |eps|^2 +- 2 sigma_j Delta.eps
        const double sm = fma(-two_sj, c, b);
        double rp, rm;  // norm_j1 of the reference for the + and - sample
        exp2_vc2(fma(cur[DP + 1], sp, cur[DP + 0]), fma(cur[DP + 1], sm, cur[DP + 0]), ec, rp, rm);
        rp_[kk] = rp;
        rm_[kk] = rm;
        qp = fma(cur[DP + 2], rp, qp);
        qm = fma(cur[DP + 2], rm, qm);
#pragma unroll
        for (int i = 0; i < NR1; ++i) cur[i] = nxt[i];
      }
    }
    // ---- q = sum over the 4 waves ----
    const int buf = it & 1;
    sQ[buf][wave][0][lane] = qp;
    sQ[buf][wave][1][lane] = qm;
    if constexpr (eps_exact) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the next batch's rows
    __syncthreads();
    qp = (sQ[buf][0][0][lane] + sQ[buf][1][0][lane]) + (sQ[buf][2][0][lane] + sQ[buf][3][0][lane]);
    qm = (sQ[buf][0][1][lane] + sQ[buf][1][1][lane]) + (sQ[buf][2][1][lane] + sQ[buf][3][1][lane]);

    // log q: wave 0 takes the + samples, wave 1 the - samples
    if (wave < 2) {
      const double qv = valid ? (wave == 0 ? qp : qm) : 1.0;
      lmant *= __builtin_amdgcn_frexp_mant(qv);  // [1/2, 1) each: eight of them stay far from underflow; q = 0 -> 0 -> -inf
      lexp += __builtin_amdgcn_frexp_exp(qv);
      if ((it & (LOGB - 1)) == LOGB - 1 || it + 1 == n_it) {
        slog_acc += fma((double)lexp, 0x1.62e42fefa39efp-1, fm::log_fast(lmant));
        lmant = 1.0;
        lexp = 0;
      }
    }
    if (GRAD) {
      // ---- pass 2: with 1/q known, every gradient sum is accumulated already normalised.
      // lsum_d/q of the two samples (entmc_vbmc.py:93-106), u_dk = Delta_dk +- s e_d:
      //   lp_d = sum_k gp_k (Delta_dk + s e_d),  lm_d = sum_k gm_k (Delta_dk - s e_d),
      //   gp_k = w_k r+_k / (sigma_k^2 q+), gm_k likewise;
      //   mu  += lp + lm       = sum_k (gp+gm) Delta_dk + s e_d sum_k (gp-gm)
      //          the first sum, added over the rows, is sum_k wis2_k Delta_dk W_jk with
      //          W_jk = sum_rows (r+_k/q+ + r-_k/q-) -- exactly what Wacc collects for the
      //          weight gradient, so the finish kernel forms it once per (j, d) and the
      //          inner loop only keeps the e_d part
      //   lam += (lp - lm) e_d = e_d [ sum_k (gp-gm) Delta_dk + s e_d sum_k (gp+gm) ]
      const double ip = valid ? fm::rcp_fast(qp) : 0.0;
      const double im = valid ? fm::rcp_fast(qm) : 0.0;
      double sgs = 0.0, sgd = 0.0;
      double Td[DP];
#pragma unroll
      for (int d = 0; d < DP; ++d) Td[d] = 0.0;
      constexpr int NR2 = DP + 4;  // Delta, (c0, a, w skipped), wis2
      double c2[NR2], n2[NR2];
      {
        const double* tr = Tw;
#pragma unroll
        for (int d = 0; d < DP; ++d) c2[d] = tr[d];
        c2[DP + 3] = tr[DP + 3];
      }
#pragma unroll
      for (int kk = 0; kk < KTMAX; ++kk) {
        {
          __builtin_amdgcn_s_waitcnt(0xC07F);
          __builtin_amdgcn_sched_barrier(0);
          if (kk + 1 < KTMAX) {
            const double* tn = Tw + (size_t)(4 * (kk + 1)) * TS;
#pragma unroll
            for (int d = 0; d < DP; ++d) n2[d] = tn[d];
            n2[DP + 3] = tn[DP + 3];
          }
          __builtin_amdgcn_sched_barrier(0);
          // norm_j1 / q of the two samples, their sum and difference: three instructions (t1 is never rounded on its own)
          const double t2 = rm_[kk] * im;
          const double ts = fma(rp_[kk], ip, t2), td = fma(rp_[kk], ip, -t2);
          Wacc[kk] += ts;
          sgs = fma(ts, c2[DP + 3], sgs);  // sum_k (gp + gm),  g = w_k / sigma_k^2 * norm_j1 / q
          const double gd = td * c2[DP + 3];
          sgd += gd;
#pragma unroll
          for (int d = 0; d < DP; ++d) Td[d] = fma(gd, c2[d], Td[d]);
#pragma unroll
          for (int d = 0; d < DP; ++d) c2[d] = n2[d];
          c2[DP + 3] = n2[DP + 3];
        }
      }
      const double cs = sig_j * sgs, cd = sig_j * sgd;
#pragma unroll
      for (int d = 0; d < DP; ++d) {
        mu_acc[d] = fma(e[d], cd, mu_acc[d]);
        lam_acc[d] = fma(e[d], fma(e[d], cs, Td[d]), lam_acc[d]);
      }
    }
  }

  WS_STAMP(2);
  // ---- workgroup reduction ----
  if constexpr (EPI > 0) {
    constexpr int NQ = ws_epi_nq(DP, KTMAX, GRAD);  // DPP steps: groups of 2 / 4 / 8 lanes
    constexpr int GL = 1 << NQ, NV = 64 / GL;        // values per accumulator and wave that go through LDS
    constexpr int NI = 1 + 2 * DP + KTMAX, RS = NV + 1;
    static_assert(NQ >= 1 && NQ <= 3 && WAVES * NI * RS == EPI, "epilogue buffer");
    auto group = [](double v) {
      v += fm::dpp_get<0xB1, 0xf>(v);                          // quad_perm [1,0,3,2]
      if constexpr (NQ >= 2) v += fm::dpp_get<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
      if constexpr (NQ >= 3) v += fm::dpp_get<0x141, 0xf>(v);  // row_half_mirror
      return v;
    };
    {
      double* mine = dyn + (size_t)wave * NI * RS + (lane / GL);  // mine[item * RS]
      const bool wr = (lane & (GL - 1)) == 0;
      double v = group(slog_acc);
      if (wr) mine[0] = v;
#pragma unroll
      for (int d = 0; d < DP; ++d) {
        v = group(mu_acc[d]);
        if (wr) mine[(1 + d) * RS] = v;
        v = group(lam_acc[d]);
        if (wr) mine[(1 + DP + d) * RS] = v;
      }
#pragma unroll
      for (int kk = 0; kk < KTMAX; ++kk) {
        v = group(Wacc[kk]);
        if (wr) mine[(1 + 2 * DP + kk) * RS] = v;
      }
    }
    __syncthreads();
    for (int r = tid; r < WAVES * NI; r += WG) {
      const int wv = r / NI, it = r - wv * NI;
      const double* row = dyn + (size_t)r * RS;
      double x[NV];
#pragma unroll
      for (int l = 0; l < NV; ++l) x[l] = row[l];
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
      for (int l = 0; l < NV; l += 4) {
        s0 += x[l];
        s1 += x[l + 1];
        s2 += x[l + 2];
        s3 += x[l + 3];
      }
      const double sum = (s0 + s1) + (s2 + s3);
      if (it <= 2 * DP) {
        sRed[wv][it] = sum;
      } else {
        const int kk = it - 1 - 2 * DP;
        sW[4 * kk + wv] = sum;
      }
    }
    __syncthreads();
  } else {
    {
      const double v = wave_sum(slog_acc);
      if (lane == 0) sRed[wave][0] = v;
    }
    if (GRAD) {
#pragma unroll
      for (int d = 0; d < DP; ++d) {
        const double vmu = wave_sum(mu_acc[d]);
        const double vlam = wave_sum(lam_acc[d]);
        if (lane == 0) {
          sRed[wave][1 + d] = vmu;
          sRed[wave][1 + DP + d] = vlam;
        }
      }
#pragma unroll
      for (int kk = 0; kk < KTMAX; ++kk)
        {
          const double v = wave_sum(Wacc[kk]);
          if (lane == 0) sW[4 * kk + wave] = v;
        }
    }
    __syncthreads();
  }

  WS_STAMP(3);
  double* out = a.partial + ((int64_t)j * a.chunks + slot) * a.stride;
  for (int t = tid; t < a.stride; t += WG) {
    double v = 0.0;
    if (t == 0) {
      for (int wv = 0; wv < WAVES; ++wv) v += sRed[wv][0];
    } else if (GRAD) {
      if (t <= D) {
        for (int wv = 0; wv < WAVES; ++wv) v += sRed[wv][t];
      } else if (t == D + 1) {
        for (int d = 0; d < D; ++d)
          for (int wv = 0; wv < WAVES; ++wv) v += sRed[wv][1 + DP + d];
      } else if (t < 2 * D + 2) {
        const int d = t - (D + 2);
        for (int wv = 0; wv < WAVES; ++wv) v += sRed[wv][1 + DP + d];
      } else {
        v = sW[t - (2 * D + 2)];
      }
    }
    out[t] = v;
  }
  if (span) {
    // the part that ends component j zeroes the rows of j that no part writes (the finish kernel sums R of them)
    if (ib0 + n_it == nb)
      for (int t = tid; t < (a.chunks - 1 - slot) * a.stride; t += WG) out[a.stride + t] = 0.0;
    __syncthreads();  // sRed / sW are free again
  }
  g += n_it;
  if (!span) break;
  }
}

template <int DP, int KTMAX>
void launch_one(hipStream_t st, const EntArgs& a_in, const double* d_table, hipEvent_t e0, hipEvent_t e1) {
  EntArgs a = a_in;
  const int K = a.ml.K;
  const int K4 = 4 * KTMAX;  // the table carries zero-density padding rows up to 4 * KTMAX (entropy_args.h)
  size_t lds = sizeof(double) * ((size_t)K4 + ws_epi_doubles(DP, KTMAX, a.want_grad != 0));
  const bool philox = a.eps_mode == VBMC_EPS_PHILOX;
  const bool extra_row = a.extra != nullptr && a.want_grad && !philox;  // see EXTRA_ROW in the kernel
  const dim3 grid = a.sp.cus > 0 ? dim3(a.sp.n_parts() + (a.gp_items > 0 ? a.gp_wgs : 0) + (extra_row ? 1 : 0))
                                 : dim3(a.chunks, K + (extra_row ? 1 : 0) + (a.gp_items > 0 ? 1 : 0));
  const dim3 block(WG);
  if (extra_row && sizeof(double) * (size_t)a.extra_lds > lds) lds = sizeof(double) * (size_t)a.extra_lds;
  if (a.gp_items > 0 && glj_block_lds(a.ml.D, a.gp.N) > lds) lds = glj_block_lds(a.ml.D, a.gp.N);
  // the GP riders read X^T from LDS where that needs no more dynamic LDS than the launch has anyway (its occupancy is the
  // entropy workgroups': config 3's 36 KB hold N = 400, D = 10)
  {
    static const bool x_lds_on = [] {
      const char* e = getenv("VBMC_GLJ_X_LDS");  // measurement aid: 0 = X^T always from memory
      return !(e && e[0] == '0');
    }();
    a.gp.x_lds = (x_lds_on && a.gp_items > 0 && glj_block_lds_x(a.ml.D, a.gp.N) <= lds) ? 1 : 0;
  }
  int dev = 0;
  if (lds > 32 * 1024) (void)hipGetDevice(&dev);
  // (the raised dynamic-LDS limit is set once per instantiation and size)
#define VBMC_LAUNCH_WS(G, P, X)                                                                           \
  do {                                                                                                    \
    auto kern = entmc_ws_kernel<DP, KTMAX, G, P, X>;                                                      \
    static size_t lds_limit[64] = {};  /* per device: function attributes are per device */              \
    if (lds > 32 * 1024 && lds > lds_limit[dev & 63]) {                                                   \
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      lds_limit[dev & 63] = lds;                                                                          \
    }                                                                                                     \
    hipExtLaunchKernelGGL(kern, grid, block, (std::uint32_t)lds, st, e0, e1, 0u, a, d_table);             \
  } while (0)
  // resident draws arrive through LDS one batch ahead: 16-byte loads for full-width aligned rows, 4-byte loads otherwise
  const bool exact = !philox && a.ml.D == DP && (DP % 2) == 0 && ((uintptr_t)a.eps & 15) == 0;
  if (a.want_grad) {
    if (philox) VBMC_LAUNCH_WS(true, true, 0); else if (exact) VBMC_LAUNCH_WS(true, false, 1); else VBMC_LAUNCH_WS(true, false, 2);
  } else {
    if (philox) VBMC_LAUNCH_WS(false, true, 0); else if (exact) VBMC_LAUNCH_WS(false, false, 1); else VBMC_LAUNCH_WS(false, false, 2);
  }
#undef VBMC_LAUNCH_WS
}

}  // namespace

#define VBMC_CAT2(a, b) a##b
#define VBMC_CAT(a, b) VBMC_CAT2(a, b)

// one exported launcher per padded D; picks the smallest register-array size that holds KT.
// d_table must hold K * ws_table_rows(K) * (DP+6) doubles.
void VBMC_CAT(launch_entmc_ws_dp, VBMC_DP)(hipStream_t st, const EntArgs& a, const double* d_table, hipEvent_t e0,
                                            hipEvent_t e1) {
  switch (ws_ktmax_for(a.ml.K)) {
    case 4: launch_one<VBMC_DP, 4>(st, a, d_table, e0, e1); break;
    case 8: launch_one<VBMC_DP, 8>(st, a, d_table, e0, e1); break;
    case 10: launch_one<VBMC_DP, 10>(st, a, d_table, e0, e1); break;
    case 13: launch_one<VBMC_DP, 13>(st, a, d_table, e0, e1); break;
    case 16: launch_one<VBMC_DP, 16>(st, a, d_table, e0, e1); break;
    case 20: launch_one<VBMC_DP, 20>(st, a, d_table, e0, e1); break;
    case 25: launch_one<VBMC_DP, 25>(st, a, d_table, e0, e1); break;
    default: launch_one<VBMC_DP, 32>(st, a, d_table, e0, e1); break;
  }
}

#if defined(WS_TIMES) && VBMC_DP == 10
extern "C" int vbmc_debug_ws_times(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ws_times), sizeof(unsigned long long) * n);
}
#endif
