// Monte-Carlo entropy for SMALL sample counts -- reference entropy/entmc_vbmc.py:64-112.
//
// The optimiser's default is ns_ent = 100 K^(2/3) samples in total
// (advanced_vbmc_options.ini:43): 28 per component at K = 50, i.e. 14 antithetic rows.  The
// wave-split kernel (entropy_ws.hip: lane = row, waves split the components) then runs with 14 of
// its 64 lanes busy and is pure latency (~11 us).  Here the roles are swapped:
//
//     workgroup = component j,  lane = component k (K <= 64),  the four waves split the rows.
//
// Everything indexed by k -- the (j,k) table row written by prep.hip -- is loaded ONCE into the
// lane's registers; everything indexed by the row (its D normals) is wave-uniform: the component's
// block of normals is staged in LDS with one coalesced batch of loads and read back by broadcast.
// The only cross-lane traffic per row is the two density sums q+ and q- (DPP, fastmath.h).
// All gradient sums are kept per lane (i.e. per k) and reduced over lanes once at the end:
//     A_d(k) = sum_rows e_d   gd_k,   B_d(k) = sum_rows e_d^2 gs_k,   W(k) = sum_rows (t+ + t-)
//     t+- = r+-_k / q+-,   gs_k = wis2_k (t+ + t-),   gd_k = wis2_k (t+ - t-),
// from which the workgroup's partial row (same layout and meaning as entropy_ws.hip, so
// entmc_finish_kernel with mu_from_w = 1 is unchanged) is
//     mu_d  = sigma_j sum_k A_d(k)
//     lam_d = sigma_j sum_k B_d(k) + sum_k Delta_kd A_d(k),   sig = sum_d lam_d,   W_k.
// Used when the draws come from memory, K <= 64, at most 16 rows per component (the kernel handles up
// to 64; above 16 the wave-split kernel is as fast or faster, entmc_small_applies) and D <= 16.
#include <cstdlib>

#include "adam_dev.h"
#include "common.h"
#include "entropy_args.h"
#include "fastmath.h"

namespace {

__device__ __forceinline__ double wave_sum(double v) { return fm::wave_sum_dpp(v); }

constexpr int SW = 4;  // waves per workgroup

template <int DP, bool GRAD>
__global__ __launch_bounds__(64 * SW) void entmc_small_kernel(EntArgs a, const double* __restrict__ T) {
  extern __shared__ double sh[];  // the component's normals [rows][D], then the cross-wave partials
                                  // [SW][NACC][64] (or: the pre workgroup's LDS)
  __shared__ double red[24];
  constexpr int TS = DP + 6;
  constexpr int NACC = GRAD ? 2 * DP + 2 : 1;  // slog | W | A_d | B_d
  // Adam loop (adam.hip): grid row 0 is not an entropy row -- see entropy_ws.hip
  if (GRAD && a.extra != nullptr && blockIdx.y == 0) {
    const adam_dev::AdamDev& pa = *(const adam_dev::AdamDev*)a.extra;
    if (a.extra_lds > 0) adam_dev::adam_pre_body<true>(pa, sh, red, pa.pre);
    else adam_dev::adam_pre_body<false>(pa, nullptr, red, pa.pre);
    return;
  }
  const int D = a.ml.D, K = a.ml.K;
  const int K4 = ws_table_rows(K);  // rows of the (j,k) table per component (entropy_args.h)
  const int j = (GRAD && a.extra != nullptr) ? blockIdx.y - 1 : blockIdx.y;
  const int tid = threadIdx.x, k = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool live = k < K;

  const double sig_j = a.mix[a.ml.o_sig + j];
  const double sj2 = sig_j * sig_j, two_sj = 2.0 * sig_j;
  // this lane's table row [Delta_jk (DP) | c0 | a | w | w/sigma_k^2 | pad pad]
  double dl[DP], c0 = -2000.0, ak = 0.0, wk = 0.0, wis2 = 0.0;
#pragma unroll
  for (int d = 0; d < DP; ++d) dl[d] = 0.0;
  if (live) {
    const double* row = T + ((size_t)j * K4 + k) * TS;
#pragma unroll
    for (int d = 0; d < DP; ++d) dl[d] = row[d];
    c0 = row[DP + 0];
    ak = row[DP + 1];
    wk = row[DP + 2];
    wis2 = row[DP + 3];
  }
  double slog = 0.0, W = 0.0, A[DP], B[DP];
#pragma unroll
  for (int d = 0; d < DP; ++d) A[d] = B[d] = 0.0;

  // the component's normals go to LDS in one coalesced batch (issued together with the table row
  // above): the row loop then has no global-memory round trip of its own
  const int rows = (int)a.row_count;
  {
    const double* src = a.eps + (int64_t)j * a.eps_rows * D;
    for (int i = tid; i < rows * D; i += 64 * SW) sh[i] = src[i];
  }
  __syncthreads();
  for (int i = wave; i < rows; i += SW) {
    // the row's normals: one LDS address for the whole wave (broadcast read)
    const double* rp = sh + i * D;
    double e[DP], e2 = 0.0;
#pragma unroll
    for (int d = 0; d < DP; ++d) {
      e[d] = (d < D) ? rp[d] : 0.0;
      e2 = fma(e[d], e[d], e2);
    }
    const double b = sj2 * e2;
    double c = 0.0;
#pragma unroll
    for (int d = 0; d < DP; ++d) c = fma(dl[d], e[d], c);
    const double sp = fma(two_sj, c, b), sm = fma(-two_sj, c, b);  // sigma_j^2 |eps|^2 +- 2 sigma_j Delta.eps
    const double r1 = fm::exp2_fast(fma(ak, sp, c0)), r2 = fm::exp2_fast(fma(ak, sm, c0));
    const double qp = wave_sum(wk * r1), qm = wave_sum(wk * r2);
    slog += fm::log_fast(qp) + fm::log_fast(qm);
    if (GRAD) {
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
  }

  // ---- cross-wave sums, then the lane (= k) sums of wave 0 ----
  double* part = sh + ((rows * D + 7) & ~7);
  double* mine = part + (size_t)wave * NACC * 64;
  mine[k] = slog;
  if (GRAD) {
    mine[64 + k] = W;
#pragma unroll
    for (int d = 0; d < DP; ++d) {
      mine[(2 + d) * 64 + k] = A[d];
      mine[(2 + DP + d) * 64 + k] = B[d];
    }
  }
  __syncthreads();
  if (wave != 0) return;
  auto total = [&](int item) {
    double v = 0.0;
#pragma unroll
    for (int wv = 0; wv < SW; ++wv) v += part[((size_t)wv * NACC + item) * 64 + k];
    return v;
  };
  double* out = a.partial + (int64_t)j * a.chunks * a.stride;  // chunks == 1
  const double s_all = total(0);  // every lane of a wave carries the same slog
  if (k == 0) out[0] = s_all;
  if (GRAD) {
    const double Wk = total(1);
    if (live) out[2 + 2 * D + k] = Wk;
    double sig = 0.0;
#pragma unroll
    for (int d = 0; d < DP; ++d) {
      if (d < D) {
        const double Ad = total(2 + d), Bd = total(2 + DP + d);
        const double mu = sig_j * wave_sum(Ad);
        const double lam = wave_sum(fma(sig_j, Bd, dl[d] * Ad));
        sig += lam;
        if (k == 0) {
          out[1 + d] = mu;
          out[2 + D + d] = lam;
        }
      }
    }
    if (k == 0) out[1 + D] = sig;
  }
}

template <int DP>
void launch_small(hipStream_t st, const EntArgs& a, const double* d_table) {
  const bool extra_row = a.extra != nullptr && a.want_grad;
  const dim3 grid(1, a.ml.K + (extra_row ? 1 : 0)), block(64 * SW);
  const int nacc = a.want_grad ? 2 * DP + 2 : 1;
  size_t lds = sizeof(double) * ((size_t)SW * nacc * 64 + (size_t)a.row_count * a.ml.D + 8);
  if (extra_row && sizeof(double) * (size_t)a.extra_lds > lds) lds = sizeof(double) * (size_t)a.extra_lds;
  if (a.want_grad) {
    if (lds > 32 * 1024) {  // raised once per device and size (function attributes are per device)
      static size_t lds_limit[64] = {};
      int dev = 0;
      (void)hipGetDevice(&dev);
      if (lds > lds_limit[dev & 63]) {
        (void)hipFuncSetAttribute((const void*)entmc_small_kernel<DP, true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        lds_limit[dev & 63] = lds;
      }
    }
    hipLaunchKernelGGL((entmc_small_kernel<DP, true>), grid, block, lds, st, a, d_table);
  } else {
    hipLaunchKernelGGL((entmc_small_kernel<DP, false>), grid, block, lds, st, a, d_table);
  }
}

}  // namespace

bool entmc_small_applies(const EntArgs& a, int DP) {
  static const int max_rows = [] {
    // measured in the optimiser loop (K = 50, D = 10, per iteration): this kernel 31.0 us at 14 rows
    // per component, 31.8 at 24, 37.5 at 50, 40.6 at 64; the wave-split kernel 31.3-31.5 us at all of them
    const char* e = getenv("VBMC_SMALL_MAX_ROWS");  // experiments
    return e ? atoi(e) : 16;
  }();
  return a.eps_mode != VBMC_EPS_PHILOX && a.eps != nullptr && a.ml.K <= 64 && a.row_count <= max_rows && DP <= 16 &&
         a.chunks == 1;
}

void launch_entmc_small(hipStream_t st, const EntArgs& a, int DP, const double* d_table) {
  switch (DP) {
    case 2: launch_small<2>(st, a, d_table); break;
    case 4: launch_small<4>(st, a, d_table); break;
    case 6: launch_small<6>(st, a, d_table); break;
    case 8: launch_small<8>(st, a, d_table); break;
    case 10: launch_small<10>(st, a, d_table); break;
    case 12: launch_small<12>(st, a, d_table); break;
    default: launch_small<16>(st, a, d_table); break;
  }
}
