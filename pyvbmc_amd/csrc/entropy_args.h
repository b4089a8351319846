// Kernel argument block shared by the entropy kernels (entropy.hip, entropy_ws.hip).
#pragma once
#include "common.h"

// Span mode of the wave-split kernel: which 64-row batches a workgroup works through.
//
// The SIMD issues the OLDER of two ready waves first.  With two workgroups resident per CU (the 2-waves/SIMD builds)
// the first-dispatched one therefore runs as if it had the CU to itself and the other only fills the issue slots it
// leaves: given equal shares the first ends after 2/3 of the kernel and the second finishes alone at the one-wave
// cadence (tools/ws_times.py: 52 against 79 us at BASELINE config 3).  So the shares are made unequal on purpose: the
// `front` workgroup of a CU (block p < cus, dispatched in the first round) takes `front` per mille of the CU's batches,
// the `filler` (block cus + p, which the dispatcher places beside it) the rest, and both end together.  Batches are
// numbered g over components j and cut into cus + pb consecutive parts
//     A_0 B_0 A_1 B_1 ... A_pb-1 B_pb-1 A_pb ... A_cus-1        (units p >= pb have no filler: their second slot is left
// to the GP-sum workgroups of the host-driven step; cus - pb = ceil(S K / 5) of them, at least ten: entmc_plan) whose lengths are proportional to the weights front / 1000 - front,
// so every CU ends at the same time whatever K and the row count are.  A part that crosses a component boundary is
// worked through as two stretches with a partial row each; component j's rows are those of the parts that overlap it,
// in part order (slot = part - first part of j), R per component, the unused ones zeroed by the part that ends j.
// Ending a component costs its part a second end-of-workgroup reduction (~3 us, about one batch of the front
// workgroup): the list that is cut is therefore the VIRTUAL one in which every component is followed by `pad` slots
// of no work -- the part that ends a component gets that much less of everything else.
// Everything is static arithmetic on (T, nb, cus, pb, front): results are bit-reproducible.
struct WsSpan {
  int cus = 0;      // > 0: span mode
  int pb = 0;       // units that have a filler part
  int front = 0;    // weight of a front part, per mille of a full unit
  int nb = 0;       // batches per component
  int pad = 0;      // virtual slots behind every component (the price of ending it)
  int R = 0;        // partial rows per component
  int64_t T = 0;    // K * (nb + pad): slots of the virtual list; component j's batches are [j (nb + pad), j (nb + pad) + nb)
  __host__ __device__ int nbv() const { return nb + pad; }
  __host__ __device__ int n_parts() const { return cus + pb; }
  __host__ __device__ int64_t W() const { return (int64_t)pb * 1000 + (int64_t)(cus - pb) * front; }
  // cumulative weight in front of part u; cw(n_parts()) = W()
  __host__ __device__ int64_t cw(int u) const {
    return u < 2 * pb ? (int64_t)(u >> 1) * 1000 + (int64_t)(u & 1) * front : (int64_t)pb * 1000 + (int64_t)(u - 2 * pb) * front;
  }
  __host__ __device__ int64_t lo(int u) const { return cw(u) * T / W(); }  // first slot of part u; lo(n_parts()) = T
  // the part that holds slot g: the largest u with lo(u) <= g, i.e. cw(u) T < (g + 1) W
  __host__ __device__ int part_of(int64_t g) const {
    const int64_t c = ((g + 1) * W() - 1) / T;
    int u;
    if (c < (int64_t)pb * 1000) {
      const int p = (int)(c / 1000);
      u = 2 * p + ((c - (int64_t)p * 1000) >= front ? 1 : 0);
    } else {
      u = 2 * pb + (int)((c - (int64_t)pb * 1000) / front);
    }
    return u < n_parts() ? u : n_parts() - 1;
  }
  // block -> part: blocks [0, cus) are the front parts of units 0 .. cus-1, blocks [cus, cus + pb) the fillers
  __host__ __device__ int part_of_block(int b) const { return b < cus ? (b < pb ? 2 * b : 2 * pb + (b - pb)) : 2 * (b - cus) + 1; }
};

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
  WsSpan sp;          // wave-split kernel: span mode (above); then chunks = sp.R and rg = the longest part
  int gp_wgs = 0;     // span mode: workgroups of the GP row (blocks behind the entropy parts)
  // armed evaluation (common.h ArmedEval): every workgroup returns at once when *cancel == ~0
  const uint64_t* cancel = nullptr;
  // matrix-pipe kernel in the host-driven step (round 6): the GP sums of the prep launch IN FRONT of this one lie in device
  // memory (ship_src, complete at the kernel boundary).  The launch gets one more grid row, the last, whose first workgroup
  // copies them to pinned memory (ship_dst) and publishes ship_seq to ship_flag -- the 32 KB over PCIe and its
  // acknowledgement (6 us as the last GP block of the prep launch: between the go word and this kernel's start) then run
  // beside the entropy workgroups, in one of the workgroup slots the grid leaves free.
  const double* ship_src = nullptr;
  double* ship_dst = nullptr;
  int ship_n = 0;
  uint64_t* ship_flag = nullptr;
  uint64_t ship_seq = 0;
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
  bool gp_in_ws = false;    // span mode: the plan left workgroup slots for the GP sums (a.gp_wgs of them)
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
struct vbmc_ctx;
bool entmc_uses_mfma(const vbmc_ctx* ctx, const EntPlan& p);  // entmc_launch_main's choice for this plan (entropy.hip)
void launch_entmc_mfma(hipStream_t st, const EntArgs& a, int DP, const double* d_table, hipEvent_t e0, hipEvent_t e1);

// small sample counts: lane = component (entropy_small.hip); same table, same partial rows
bool entmc_small_applies(const EntArgs& a, int DP);
void launch_entmc_small(hipStream_t st, const EntArgs& a, int DP, const double* d_table);
