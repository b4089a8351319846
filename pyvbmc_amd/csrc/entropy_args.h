// Kernel argument block shared by the entropy kernels (entropy.hip, entropy_ws.hip).
#pragma once
#include "common.h"

struct EntArgs {
  const double* mix;
  MixLayout ml;
  const double* eps;   // resident draws [K][eps_rows][D] or nullptr
  int64_t eps_rows;    // rows resident per component (== row_count of the ctx slice)
  int64_t n_half;      // antithetic pairs per component in the whole job
  int64_t row_begin;   // first row of this ctx's slice
  int64_t row_count;   // rows of this ctx's slice
  uint64_t seed;
  int eps_mode;
  int want_grad;
  double* partial;     // [K][chunks][stride]
  int chunks;          // workgroups per component
  int stride;          // 2 + 2D + K : [Slog | mu(D) | sig | lam(D) | W(K)]
  int rg;              // wave-split kernel: 64-row batches per workgroup
  // wave-split kernel only: when set, the launch gets one extra grid row whose first workgroup
  // runs adam_dev::adam_pre_body on this argument block (device memory) -- see adam_dev.h
  const void* extra = nullptr;
  int extra_lds = 0;  // doubles of dynamic LDS that workgroup may use (0: it works from global memory)
  // wave-split kernel only: gp_items > 0 appends ONE MORE grid row (the last) whose `chunks`
  // workgroups work through the GP expected-log-joint items 0 .. gp_items-1 (glj_block.h) with the
  // argument block `gp`.  Dispatched behind every entropy workgroup, they take the workgroup slots
  // the entropy grid leaves free (the host checks that there are enough, api_elbo.hip), so the GP
  // sums -- and the host's share of them -- are done long before the entropy kernel ends.
  int gp_items = 0;
  PrepArgs gp;
  // wave-split kernel only: > 0 = number of CUs; (j, chunk) items are then handed to workgroups so
  // that the two workgroups of a CU read the same table row (entropy_ws.hip).  Set by entmc_plan for
  // one-round grids of the 2-waves/SIMD builds.
  int pair_cus = 0;
  // armed evaluation (common.h ArmedEval): every workgroup returns at once when *cancel == ~0
  const uint64_t* cancel = nullptr;
};

// register-array size (components per wave) the wave-split launcher picks, and the waves per SIMD
// its build runs at (= resident workgroups per CU: a workgroup is one wave on each SIMD).
// The (j,k) table always carries 4 * ws_ktmax_for(K) rows per component j (ws_table_rows): the rows
// beyond K are components of density exactly 0, so the kernel's loops run over whole register arrays
// with no per-component guards.  (Rounds 1-2 padded to 4 ceil(K/4) only and kept a guarded variant for
// ceil(K/4) < KTMAX; that variant ran 2-3.5x slower per evaluated pair than the guard-free one --
// K = 48 against K = 50 at D = 10: 3 699 against 1 592 ps per pair, tools/ws_k_probe.py -- far more
// than the padding costs.)
constexpr int ws_ktmax_for(int K) {
  const int KT = (K + 3) / 4;
  return KT <= 4 ? 4 : KT <= 8 ? 8 : KT <= 10 ? 10 : KT <= 13 ? 13 : KT <= 16 ? 16 : KT <= 20 ? 20 : KT <= 25 ? 25 : 32;
}
constexpr int ws_table_rows(int K) { return 4 * ws_ktmax_for(K); }
constexpr int ws_min_waves(int dp, int ktmax, bool grad) {
#ifdef VBMC_WS_FORCE_WAVES
  return VBMC_WS_FORCE_WAVES;
#else
  return (grad && 4 * dp + 3 * ktmax > 95) ? 1 : 2;
#endif
}

struct EntPlan {
  EntArgs a;
  bool pregen_hit = false;  // the draws come from a speculative generation (entmc_pregen)
  bool ws = false;
  int DP = 0;
  double inv_ns = 0.0;
  double* table = nullptr;  // ws only
};

// padded-D instantiations of the wave-split kernel (entropy_ws.hip), one translation
// unit each; d_table: K * ws_table_rows(K) * (dp+6) doubles of scratch for the (j,k) table
#define VBMC_WS_DPS(X) X(2) X(4) X(6) X(8) X(10) X(12) X(16) X(20) X(24) X(32)
// e0 / e1 (may be null): HIP events that take the start / stop timestamps of the dispatch itself
// (hipExtLaunchKernel) -- unlike hipEventRecord they put no barrier packet between dependent kernels
#define VBMC_DECL_WS(dp) \
  void launch_entmc_ws_dp##dp(hipStream_t st, const EntArgs& a, const double* d_table, hipEvent_t e0, hipEvent_t e1);
VBMC_WS_DPS(VBMC_DECL_WS)
#undef VBMC_DECL_WS

// matrix-pipe form for shapes the 16 x 16 x 4 tile pads little (entropy_mfma.hip: D = 20, K up to 112 -- BASELINE
// config 5); same table, same partial rows
bool entmc_mfma_applies(const EntArgs& a, int DP);
void launch_entmc_mfma(hipStream_t st, const EntArgs& a, int DP, const double* d_table, hipEvent_t e0, hipEvent_t e1);

// small sample counts: lane = component (entropy_small.hip); same table, same partial rows
bool entmc_small_applies(const EntArgs& a, int DP);
void launch_entmc_small(hipStream_t st, const EntArgs& a, int DP, const double* d_table);
