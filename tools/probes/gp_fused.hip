// GP.predict for a batch of points as ONE launch (gpyreg third party: SURVEY Appendix A; consumer
// acquisition_functions/abstract_acq_fcn.py:79-97):
//     K*_mn = sf^2 exp(-1/2 |(x*_m - X_n) / ell|^2),   fmu_m = mean(x*_m) + sum_n K*_mn alpha_n,
//     fs2_m = max(0, sf^2 - |(sW o K*_m)^T L^-1|^2) (+ noise)            (Cholesky samples)
//
// Rounds 1-3 ran three launches: K* written to HBM (26 MB at M = 8192, N = 400) by one, read back
// by the variance product, whose per-tile partial sums a third one folded: 76.7 MB of traffic for
// ~2 MB of algorithmic bytes and two kernel boundaries (57.5 us of kernels, 65.9 us between events).
// Here a workgroup owns 32 points and ALL columns of T = (sW o K*) L^-1, and the contraction index
// n runs in the OUTER loop, 32 training points (one panel) at a time:
//   * the panel's 32 x 32 block of sW o K* is made on the spot from a per-panel RECORD prepared once
//     per GP update (predict_pack_kernel: the 32 training points' centred, length-scaled coordinates,
//     their squared norms, alpha and sW, zero beyond N) that arrives by LDS-direct loads two panels
//     ahead: cross term on the FP64 matrix cores (the reference's centred form,
//     abstract_acq_fcn.py:195-222), exp2 with folded constants, laid down in LDS in the matrix
//     instruction's operand layout (16-byte chunk q of row r in slot q ^ (r & 15): conflict-free fragment
//     reads, as predict_var_dma_kernel).  The block of panel p + 1 is made inside the first step of
//     panel p, between its matrix instructions: no barrier of its own, its latencies in their shadow.
//     K* never exists as a whole, in LDS or in memory;
//   * the panel's rows of L^-1 (zero padded: GpState::d_LinvP), one 32 x 64 block per column tile
//     c >= n / 64 (upper triangular), stream through SIX 16 KB stages by global_load_lds_dwordx4,
//     four or five blocks ahead of the matrix instructions -- with one workgroup of four waves per CU
//     nothing else hides the ~1 us round trip of a load (a two-stage ring ran 72 us); two blocks are
//     consumed per barrier (four independent accumulator chains, half the synchronisations);
//   * all of a wave's accumulators (its 16 rows x 32 columns of every column tile: 56 doubles per
//     lane at N <= 448) stay in registers until the end: squared row sums, the mean's sum_n K* alpha_n,
//     the mean function, then fs2 and fmu are written.  No partial sums, no finish launch.
// M = 8192: 256 workgroups, one per CU, one round.  N > 448, non-Cholesky samples and batches of
// <= 32 points keep the three-launch path (gp.hip).
// (Built with -mllvm -amdgpu-mfma-vgpr-form: the accumulators stay in VGPRs; the default moved every
// tile through 16 AGPRs around its matrix instructions, 32 copies per block.)
#include <type_traits>

#include "common.h"
#include "fastmath.h"

#ifdef FUSED_TIMES
__device__ unsigned long long g_fused_times[8 * 1024];
extern "C" int vbmc_debug_fused_times(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fused_times), sizeof(unsigned long long) * n);
}
#define FSTAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 1024) g_fused_times[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
__device__ unsigned long long g_fused_times2[8 * 1024];
extern "C" int vbmc_debug_fused_times2(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fused_times2), sizeof(unsigned long long) * n);
}
#define FSTAMP2(i) do { if (threadIdx.x == 64 && blockIdx.x < 1024) g_fused_times2[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define FSTAMP(i) (void)0
#define FSTAMP2(i) (void)0
#endif
namespace {

constexpr int FR = 32;            // points per workgroup
constexpr int FTS = 64;           // columns per tile
constexpr int FDK = 32;           // panel depth
constexpr int FCT = 7;            // column tiles at most (N <= 448)
constexpr int FKS = 34;           // row stride of the points' scaled coordinates in LDS (= 2 mod 4)
constexpr int F_STAGE = FDK * FTS * 8;  // bytes of one block of L^-1
constexpr int F_NS = 6;           // stages
constexpr int F_RECSLOT = 12288;  // LDS slot of a panel record (8 960 bytes at D = 32, loaded as 3 x 4 KB); three of them: a ring
typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ const char* uniform_ptr_f(const void* p) {
  const uint64_t v = (uint64_t)p;
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32));
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
  return (const char*)(((uint64_t)hi << 32) | (uint64_t)lo);
}

// LDS carve (bytes)
constexpr int O_ST = 0;                            // F_NS stages
constexpr int O_KS = O_ST + F_NS * F_STAGE;        // [2][32 rows][256 B]: a panel's block of sW o K*
constexpr int O_RC = O_KS + 2 * FR * 256;          // [3] panel records (slot 2 first holds the points' coordinates)
constexpr int O_RD = O_RC + 3 * F_RECSLOT;         // [32] |a|^2, [32][2] row sums, [32][2] means
constexpr int F_LDS = O_RD + (32 + 64 + 64) * 8;

// Panel records, once per GP update: rec[s][p] = [b (32 x KD) | |b|^2 (32) | alpha (32) | sW (32)], KD = 4 ceil(D / 4),
// b_nd = (X_nd - centre_d) / ell_d, everything 0 for n >= N or d >= D.
__global__ __launch_bounds__(256) void predict_pack_kernel(const double* __restrict__ X, const double* __restrict__ alpha_all,
                                                           const double* __restrict__ sW_all, const double* __restrict__ hyp_all,
                                                           const double* __restrict__ cen, int P, int N, int D, int KD, int npan,
                                                           double* __restrict__ rec_all) {
  const int p = blockIdx.x, s = blockIdx.y;
  const int recsz = 32 * KD + 96;
  double* rec = rec_all + ((size_t)s * npan + p) * recsz;
  const double* hyp = hyp_all + (size_t)s * P;
  const int tid = threadIdx.x, r = tid >> 3, q = tid & 7;
  const int n = p * 32 + r;
  double b2 = 0.0;
  for (int d = q; d < KD; d += 8) {
    const double b = (n < N && d < D) ? (X[(size_t)n * D + d] - cen[d]) * exp(-hyp[d]) : 0.0;
    rec[r * KD + d] = b;
    b2 = fma(b, b, b2);
  }
  b2 += fm::dpp_get<0xB1, 0xf>(b2);
  b2 += fm::dpp_get<0x4E, 0xf>(b2);
  b2 += fm::dpp_get<0x141, 0xf>(b2);
  if (q == 0) {
    rec[32 * KD + r] = b2;
    rec[32 * KD + 32 + r] = n < N ? alpha_all[(size_t)s * N + n] : 0.0;
    rec[32 * KD + 64 + r] = n < N ? sW_all[(size_t)s * N + n] : 0.0;
  }
}

__global__ __launch_bounds__(256, 1) void predict_fused_kernel(
    const double* __restrict__ rec_all, const double* __restrict__ xs, const double* __restrict__ hyp_all,
    const double* __restrict__ cen, const double* __restrict__ smeta, const double* __restrict__ LinvP_all, int P, int N,
    int D, int64_t M, int lda, int mean_kind, int add_noise, double* __restrict__ fmu, double* __restrict__ fs2,
    int64_t ldo) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
  const int smp = blockIdx.y;
  const double* hyp = hyp_all + (size_t)smp * P;
  const double* LinvP = LinvP_all + (size_t)smp * lda * lda;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mt = wave & 1, ch = wave >> 1;  // row tile (16 rows); column half (32 of a tile's 64 columns / 16 of a panel's 32)
  const int li = lane & 15, lk = lane >> 4;
  const int64_t m0 = (int64_t)blockIdx.x * FR;
  const int nct = lda / FTS;
  const int np = (N + FDK - 1) / FDK;  // panels of the contraction
  const int DQ = (D + 3) / 4, KD = 4 * DQ;
  const int recsz = 32 * KD + 96;
  const double* rec_s = rec_all + (size_t)smp * (lda / FDK) * recsz;
  double* sA2 = (double*)(fsm + O_RD);
  double* sS = sA2 + 32;   // [32][2]
  double* sF = sS + 64;    // [32][2]

  // ---- LDS-direct loads.  Every wave issues the same number of them at the same places: its vmcnt
  // then counts one and the same queue, and the waits below are written against that sequence. ----
  const unsigned lds0 = (unsigned)(uintptr_t)fsm;
  auto dma16 = [&](unsigned voff, const char* base, unsigned dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(base), "s"(dst) : "memory");
  };
  // block (panel p, column tile c) of L^-1: instruction j of wave w fills rows 2 (4 w + j) + (lane >> 5), chunk lane & 31
  unsigned voffB[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) voffB[j] = (unsigned)(2 * (4 * wave + j) + (lane >> 5)) * (unsigned)lda * 8u + (unsigned)((lane & 31) * 16);
  auto issue_block = [&](int p, int c, int stage) {
    const char* bg = uniform_ptr_f(LinvP + (size_t)p * FDK * lda + (size_t)c * FTS);
#pragma unroll
    for (int j = 0; j < 4; ++j) dma16(voffB[j], bg, lds0 + (unsigned)(O_ST + stage * F_STAGE + wave * 4096 + j * 1024));
  };
  // the record of panel p into ring slot p % 3: nrw instructions per wave (1 KB each; 1 at D <= 12, else 2 or 3),
  // addresses beyond the record clamped to its last chunk (those lanes' LDS bytes lie beyond it, inside the slot)
  const unsigned rec_bytes = (unsigned)recsz * 8u;
  const int nrw = (int)((rec_bytes + 4095u) / 4096u);
  auto issue_rec = [&](int p) {
    const char* base = uniform_ptr_f(rec_s + (size_t)p * recsz);
    for (int j = 0; j < nrw; ++j) {
      const unsigned off = (unsigned)((wave * nrw + j) * 1024 + lane * 16);
      dma16(min(off, rec_bytes - 16u), base, lds0 + (unsigned)(O_RC + (p % 3) * F_RECSLOT + (wave * nrw + j) * 1024));
    }
  };
  // wait until at most `younger` of this wave's loads are outstanding (they complete in order: everything older has
  // then landed); the immediate operand comes from a short ladder, rounding down is safe
  auto wait_loads = [&](int younger) {
    if (younger >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (younger >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (younger >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (younger >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (younger >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (younger >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  // ---- prologue ----
  FSTAMP(0);
  issue_rec(0);
  if (np > 1) issue_rec(1);
  // the sequence of blocks: panel p = 0 .. np-1, column tiles c = p / 2 .. nct-1
  int dp = 0, dc = 0, n_issued = 0;  // cursor of the block loads
  auto issue_next = [&]() {
    if (dp < np) {
      issue_block(dp, dc, n_issued % F_NS);
      ++n_issued;
      if (++dc == nct) {
        ++dp;
        dc = dp >> 1;
      }
    }
  };
#pragma unroll 1
  for (int i = 0; i < F_NS - 1; ++i) issue_next();
  // the points: centred, length-scaled coordinates (ring slot 2 is free until panel 0 asks for record 2)
  double* sAm = (double*)(fsm + O_RC + 2 * F_RECSLOT);
  {
    const int r = tid >> 3, q = tid & 7;
    const int64_t m = m0 + r;
    double a2 = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int d = q + 8 * i;
      if (d < KD) {
        const double a = (d < D && m < M) ? (xs[m * D + d] - cen[d]) * exp(-hyp[d]) : 0.0;
        sAm[r * FKS + d] = a;
        a2 = fma(a, a, a2);
      }
    }
    a2 += fm::dpp_get<0xB1, 0xf>(a2);   // quad_perm [1,0,3,2]
    a2 += fm::dpp_get<0x4E, 0xf>(a2);   // quad_perm [2,3,0,1]
    a2 += fm::dpp_get<0x141, 0xf>(a2);  // row_half_mirror: the other quad of the eight
    if (q == 0) sA2[r] = a2;
  }
  const double l2sf2 = 2.0 * hyp[D] * 0x1.71547652b82fep+0;  // log2(sf^2)
  const double cexp = -0.5 * 0x1.71547652b82fep+0;           // -log2(e) / 2
  FSTAMP(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // records 0, 1 and the first blocks
  __syncthreads();
  FSTAMP(2);
  // this wave's operands of the distance product (rows mt 16 + li) and its rows' squared norms, for the whole launch
  double afr[8], a2r[4];
#pragma unroll
  for (int kq = 0; kq < 8; ++kq) afr[kq] = kq < DQ ? sAm[(mt * 16 + li) * FKS + 4 * kq + lk] : 0.0;
#pragma unroll
  for (int r = 0; r < 4; ++r) a2r[r] = sA2[mt * 16 + lk + 4 * r];
  double facc[4] = {0.0, 0.0, 0.0, 0.0};  // sum_n K* alpha_n of rows mt 16 + lk + 4 r over this wave's columns of every panel
  // the 32 x 32 block of sW o K* of panel q, from its record: wave = (row tile mt, column tile ch)
  auto make_block = [&](int q) {
    const double* rc = (const double*)(fsm + O_RC + (q % 3) * F_RECSLOT);
    double4_t dd = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kq = 0; kq < 8; ++kq)
      if (kq < DQ) dd = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[kq], rc[(ch * 16 + li) * KD + 4 * kq + lk], dd, 0, 0, 0);
    const int col = ch * 16 + li;
    const double b2 = rc[32 * KD + col], al = rc[32 * KD + 32 + col], sw = rc[32 * KD + 64 + col];
    unsigned char* ks = fsm + O_KS + (q & 1) * (FR * 256);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = mt * 16 + lk + 4 * r;
      const double d2 = fmax(fma(-2.0, dd[r], a2r[r] + b2), 0.0);
      const double kv = fm::exp2_fast(fma(cexp, d2, l2sf2));
      facc[r] = fma(kv, al, facc[r]);  // (alpha and sW are 0 beyond N)
      *(double*)(ks + row * 256 + (((col >> 1) ^ (row & 15)) << 4) + ((col & 1) << 3)) = kv * sw;
    }
  };
  make_block(0);
  __syncthreads();  // (also: everybody has taken its operands out of ring slot 2)
  FSTAMP(3);

  double4_t acc[FCT][2];
#pragma unroll
  for (int c = 0; c < FCT; ++c) acc[c][0] = acc[c][1] = (double4_t){0.0, 0.0, 0.0, 0.0};
  int step = 0;  // blocks consumed so far
  unsigned offA[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) offA[t] = (unsigned)((mt * 16 + li) * 256 + (((4 * t + lk) ^ li) << 4));
  const unsigned offB = (unsigned)(2 * lk * 512 + (16 * ch + li) * 16);
  int rm_next = 0;  // blocks issued before the record of panel p + 1 went out (record 1: in the prologue, before all of them)

#pragma unroll 1
  for (int p = 0; p < np; ++p) {
    if (p == 1) FSTAMP(6);
    if (p == 6) FSTAMP(7);
    const int c0 = p >> 1;
    double2_t fa[4];
    bool first = true;
#pragma unroll
    for (int c = 0; c < FCT; ++c) {
      if (c >= c0 && c < nct && ((c - c0) & 1) == 0) {
        // ---- one synchronisation: tiles c and c + 1 of panel p ----
        const bool two = c + 1 < nct;
        // what must have landed: the block(s) of this step -- the blocks issued after them are younger -- and, at a
        // panel's first step, the record of panel p + 1 (it went out a panel ago, before the blocks counted from
        // rm_next on).  Lower bounds of "younger": a record load in between only makes the wait longer.
        const bool probe = (p == 2 && c == 3);
        if (probe) FSTAMP2(0);
        int younger = 4 * (n_issued - step - (two ? 2 : 1));
        if (first && p + 1 < np) {
          const int yr = 4 * (n_issued - rm_next);
          younger = yr < younger ? yr : younger;
        }
#ifndef FUSED_ABL_NOWAIT
        wait_loads(younger);
        __syncthreads();
#endif
        if (probe) FSTAMP2(1);
        // refill the stages the previous step has freed (never one of the two about to be read): the addresses now,
        // the load instructions one behind every group of matrix instructions below (issued in a row in front of them
        // they cost the step ~0.6 us: an LDS-direct load holds the wave while it issues)
        const char* ibg[2] = {nullptr, nullptr};
        unsigned idst[2] = {0u, 0u};
        {
          const int free_stages = F_NS - (n_issued - step);
#pragma unroll
          for (int u = 0; u < 2; ++u)
            if (free_stages > u && dp < np) {
              ibg[u] = uniform_ptr_f(LinvP + (size_t)dp * FDK * lda + (size_t)dc * FTS);
              idst[u] = lds0 + (unsigned)(O_ST + (n_issued % F_NS) * F_STAGE + wave * 4096);
              ++n_issued;
              if (++dc == nct) {
                ++dp;
                dc = dp >> 1;
              }
            }
        }
        if (first) {
          rm_next = n_issued;                // (the record of panel p + 2 goes out now: the blocks from here on are younger)
          if (p + 2 < np) issue_rec(p + 2);  // (its ring slot held record p - 1: last read a panel ago)
          // this wave's fragments of the panel's block: rows mt 16 + li, k = 8 t + 2 lk + {0, 1}
          const unsigned char* sa = fsm + O_KS + (p & 1) * (FR * 256);
#pragma unroll
          for (int t = 0; t < 4; ++t) fa[t] = *(const double2_t*)(sa + offA[t]);
        }
        const unsigned char* sb0 = fsm + O_ST + (step % F_NS) * F_STAGE + offB;
        const unsigned char* sb1 = fsm + O_ST + ((step + 1) % F_NS) * F_STAGE + offB;
        double2_t fb[2][2][2];  // [buffer][block][h]
        auto load = [&](int t, int buf) {
          fb[buf][0][0] = *(const double2_t*)(sb0 + (8 * t) * 512);
          fb[buf][0][1] = *(const double2_t*)(sb0 + (8 * t + 1) * 512);
          if (two) {
            fb[buf][1][0] = *(const double2_t*)(sb1 + (8 * t) * 512);
            fb[buf][1][1] = *(const double2_t*)(sb1 + (8 * t + 1) * 512);
          }
        };
        const int c1 = c + 1 < FCT ? c + 1 : c;  // (c = 6 has no second tile: `two` is false there)
        if (probe) FSTAMP2(2);
        load(0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int cur = t & 1;
          if (t < 3) load(t + 1, cur ^ 1);
          __builtin_amdgcn_sched_barrier(0);
          if (probe) FSTAMP2(3 + t);
#ifdef FUSED_ABL_NOMFMA
          acc[c][0][0] += fa[t][0] * fb[cur][0][0][0] + fb[cur][0][1][1];
          if (two) acc[c1][0][0] += fa[t][1] * fb[cur][1][0][0] + fb[cur][1][1][1];
#else
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            acc[c][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[t][h], fb[cur][0][h][0], acc[c][0], 0, 0, 0);
            acc[c][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[t][h], fb[cur][0][h][1], acc[c][1], 0, 0, 0);
            if (two) {
              acc[c1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[t][h], fb[cur][1][h][0], acc[c1][0], 0, 0, 0);
              acc[c1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[t][h], fb[cur][1][h][1], acc[c1][1], 0, 0, 0);
            }
          }
#endif
          if (ibg[0]) dma16(voffB[t], ibg[0], idst[0] + (unsigned)(t * 1024));
          if (ibg[1]) dma16(voffB[t], ibg[1], idst[1] + (unsigned)(t * 1024));
          __builtin_amdgcn_sched_barrier(0);
          // the next panel's block of sW o K*, in the shadow of this step's matrix instructions (its record landed
          // before the barrier above; the barrier of the next step publishes the block)
#ifndef FUSED_ABL_NOBLOCK
          if (first && t == 0 && p + 1 < np) make_block(p + 1);
#endif
        }
        if (probe) FSTAMP2(7);
        step += two ? 2 : 1;
        first = false;
      }
    }
  }
  FSTAMP(4);
  // ---- end: squared row sums over all column tiles, the mean, the outputs ----
  double sacc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int c = 0; c < FCT; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sacc[r] = fma(acc[c][0][r], acc[c][0][r], sacc[r]);
      sacc[r] = fma(acc[c][1][r], acc[c][1][r], sacc[r]);
    }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const double v = fm::row16_sum_dpp(sacc[r]);
    const double f = fm::row16_sum_dpp(facc[r]);
    if (li == 0) {
      sS[(mt * 16 + lk + 4 * r) * 2 + ch] = v;
      sF[(mt * 16 + lk + 4 * r) * 2 + ch] = f;
    }
  }
  __syncthreads();
  if (tid < FR && m0 + tid < M) {
    const int64_t m = m0 + tid;
    const double sf2 = exp(2.0 * hyp[D]);
    const double add = add_noise ? exp(2.0 * hyp[D + 1]) * smeta[3 * smp + 1] : 0.0;
    fs2[(size_t)smp * ldo + m] = fmax(sf2 - (sS[tid * 2] + sS[tid * 2 + 1]), 0.0) + add;
    // mean function at x* (variational_optimization.py:1383-1392 layout)
    double mean = 0.0;
    const double* hm = hyp + D + 2;
    if (mean_kind == VBMC_MEAN_CONST) mean = hm[0];
    if (mean_kind == VBMC_MEAN_NEGQUAD) {
      mean = hm[0];
      for (int d = 0; d < D; ++d) {
        const double t = (xs[m * D + d] - hm[1 + d]) * exp(-hm[1 + D + d]);
        mean -= 0.5 * t * t;
      }
    }
    fmu[(size_t)smp * ldo + m] = mean + (sF[tid * 2] + sF[tid * 2 + 1]);
  }
  FSTAMP(5);
}

}  // namespace

size_t gp_predict_pack_elems(int S, int N, int D) {
  const int KD = 4 * ((D + 3) / 4);
  return (size_t)S * (size_t)(predict_ld(N) / FDK) * (size_t)(32 * KD + 96);
}

// the panel records of the fused kernel, once per vbmc_set_gp (after X, alpha, sW, hyp and the centre are on the device)
int launch_predict_pack(vbmc_ctx* ctx) {
  GpState& g = ctx->gp;
  if (!g.d_Bpack || g.D > 32) return 0;
  const int KD = 4 * ((g.D + 3) / 4), npan = predict_ld(g.N) / FDK;
  hipLaunchKernelGGL(predict_pack_kernel, dim3(npan, g.S), dim3(256), 0, ctx->stream, (const double*)g.d_X,
                     (const double*)g.d_alpha, (const double*)g.d_sW, (const double*)g.d_hyp, (const double*)g.d_xc, g.P, g.N,
                     g.D, KD, npan, g.d_Bpack);
  HIP_TRY(ctx, hipGetLastError());
  return 0;
}

// All S samples of a batch in one launch; false when the shape is not the fused kernel's (the caller keeps the
// three-launch path): more than 32 points, every sample a Cholesky one, N <= 448, D <= 32.
bool gp_predict_fused_applies(const vbmc_ctx* ctx, int64_t M) {
  const GpState& g = ctx->gp;
  if (!ctx->opt_predict_fused || M <= 32 || g.N > FCT * FTS || g.D > 32 || !g.d_LinvP || !g.d_Bpack) return false;
  for (int s = 0; s < g.S; ++s)
    if (!g.L_chol[s]) return false;
  return true;
}

int launch_gp_predict_fused(vbmc_ctx* ctx, int64_t M, const double* d_xs, int add_noise, double* d_fmu, double* d_fs2,
                            int64_t ld) {
  const GpState& g = ctx->gp;
  const int lda = predict_ld(g.N);
  static bool lds_set[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!lds_set[dev & 63]) {
    HIP_TRY(ctx, hipFuncSetAttribute((const void*)predict_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS));
    lds_set[dev & 63] = true;
  }
  hipLaunchKernelGGL(predict_fused_kernel, dim3((unsigned)((M + FR - 1) / FR), g.S), dim3(256), F_LDS, ctx->stream,
                     (const double*)g.d_Bpack, d_xs, (const double*)g.d_hyp, (const double*)g.d_xc, (const double*)g.d_smeta,
                     (const double*)g.d_LinvP, g.P, g.N, g.D, M, lda, g.mean_kind, add_noise, d_fmu, d_fs2, ld);
  HIP_TRY(ctx, hipGetLastError());
  return 0;
}
