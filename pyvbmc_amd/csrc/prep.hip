// "prep" launch of one ELBO evaluation: two independent small jobs share ONE grid so
// they run side by side in front of the long entropy kernel:
//   blocks [0, n_table)          : row j of the (j,k) constant table the wave-split
//                                  entropy kernel reads through scalar loads (entropy_ws.hip)
//   blocks [n_table, n_table+S*K): the GP expected-log-joint sums of component k under GP
//                                  sample s -- reference vbmc/variational_optimization.py:1400-1465:
//        res[(s*K+k)*(1+2D) + it],  it = 0     : sum_n z_n alpha_n
//                                   it = 1..D  : sum_n delta_nd   z_n alpha_n
//                                   it = D+1..2D: sum_n delta_nd^2 z_n alpha_n
//        z_n = exp(lnnf - 1/2 sum_d delta_nd^2),  delta_nd = (mu_dk - X_nd)/tau_dk,
//        tau_dk = sqrt(sigma_k^2 lambda_d^2 + ell_d^2);  optionally Z[s][k][n] = z_n.
//      The host turns these into G, dG (api_gp.hip: glj_finalize).  (Block body: glj_block.h.)
#include <cstdlib>

#include "common.h"
#include "fastmath.h"
#include "philox.h"
#ifdef PREP_TIMES
__device__ unsigned long long g_glj_phase[8 * 1024];
#define GLJ_T(i) do { if (threadIdx.x == 0 && blockIdx.x < 1024) g_glj_phase[8 * blockIdx.x + (i)] = wall_clock64(); } while (0)
#endif
#include "glj_block.h"
#include "ws_table.h"

namespace {

__device__ __forceinline__ double wave_sum(double v) {
  return fm::wave_sum_dpp(v);
}

#ifdef FIN_TIMES
__device__ unsigned long long g_prep_times[2 + 64];  // [0] launch counter, [2 + n % 64] start of block 0 of launch n
#endif
#ifdef PREP_TIMES
// measurement aid (tools/prep_times.py): start and end of every workgroup of the LAST prep launch, wall-clock ticks
__device__ unsigned long long g_prep_blocks[2 * 1024];
struct PrepStamp {
  int b;
  __device__ PrepStamp() : b((int)blockIdx.x) {
    if (threadIdx.x == 0 && b < 1024 && blockIdx.y == 0) g_prep_blocks[2 * b] = wall_clock64();
  }
  __device__ ~PrepStamp() {
    __syncthreads();
    if (threadIdx.x == 0 && b < 1024 && blockIdx.y == 0) g_prep_blocks[2 * b + 1] = wall_clock64();
  }
};
#endif
__global__ __launch_bounds__(256) void elbo_prep_kernel(PrepArgs a) {
  extern __shared__ double lds[];
#ifdef PREP_TIMES
  PrepStamp stamp_;
#endif
#ifdef FIN_TIMES
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_prep_times[2 + (g_prep_times[0]++ & 63)] = wall_clock64();
#endif
  const int tid = threadIdx.x;
  if (a.gen.n_blocks > 0 && (int)blockIdx.x >= a.n_table + a.n_glj && (int)blockIdx.x < a.n_table + a.n_glj + a.gen.n_blocks) {
    // ---- draw-generation block (philox.h): depends on the seed only, never waits for theta ----
    gen_slice_block(a.gen, blockIdx.x - a.n_table - a.n_glj, tid);
    return;
  }
  if (a.go) {
    // ---- armed launch: wait for the host's theta (the pack in a.mix and the go word) ----
    __shared__ int s_go;
    if (tid == 0) {
      const unsigned long long t0 = wall_clock64();
      int st = 0;
      for (;;) {
        const uint64_t v = __hip_atomic_load(a.go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (v == a.go_seq) { st = 1; break; }
        if (v == ~(uint64_t)0) break;
        if (wall_clock64() - t0 > a.go_timeout) {  // (2 ms by default: twice the host's limit) give up
          if (blockIdx.x == 0 && blockIdx.y == 0) {
            __hip_atomic_store(a.go, ~(uint64_t)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(a.dead, a.go_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          }
          break;
        }
        __builtin_amdgcn_s_sleep(2);  // (a poll every ~0.06 us: the go word's arrival is on the step's critical path)
      }
      s_go = st;
    }
    __syncthreads();
    if (!s_go) return;
  }
  if (a.mix_copy && blockIdx.x == gridDim.x - 1) {
    // ---- copy block: the pack for the launches behind this one (+ its checksum, DoneSignal) ----
    // (sixteen loads in flight per thread and round: one load per round -- the compiler cannot move a load above the
    // store in front of it -- made this block a chain of n / 256 memory latencies, 18 at K = 100, D = 20: it was the
    // prep launch's duration there, 16 us in front of config 5's entropy kernel)
    uint64_t ck = 0;
    constexpr int U = 16;
    const int n = a.mix_copy_n;
    for (int base = 0; base < n; base += U * 256) {
      double v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (base + u * 256 >= n) break;  // (wave-uniform)
        v[u] = a.mix[min(base + u * 256 + tid, n - 1)];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (base + u * 256 >= n) break;
        const int i = base + u * 256 + tid;
        if (i < n) {
          a.mix_copy[i] = v[u];
          ck += pack_ck_term(v[u], (uint32_t)i);
        }
      }
    }
    if (a.ident_out) {
      // (sixteen LDS words, sixteen threads each: 256 atomic additions to ONE word are executed one after the other --
      // ~0.8 us of this block, which is the longest of the prep launch at config 3; integer sums: any order gives the same bits)
      __shared__ unsigned long long s_ck[16];
      if (tid < 16) s_ck[tid] = 0;
      __syncthreads();
      atomicAdd(&s_ck[tid & 15], (unsigned long long)ck);
      __syncthreads();
      if (tid == 0) {
        unsigned long long t = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += s_ck[i];
        *a.ident_out = t;
      }
    }
    return;
  }
  a.mix += (size_t)blockIdx.y * a.mix_stride;  // batched sieve: one candidate mixture per grid.y
  a.res += (size_t)blockIdx.y * a.res_stride;

  if ((int)blockIdx.x < a.n_table) {
    // ---- table row block: T[j][k] = [Delta_jk (DP) | c0 | a | w | w/sigma_k^2 | pad pad], where
    // c0 + a (sigma_j^2 |eps|^2 +- 2 sigma_j Delta.eps) is the log2 density of component k at
    // mu_j +- sigma_j lambda eps:  a = -log2(e)/(2 sigma_k^2),  c0 = a |Delta|^2 + log2 c_k ----
    ws_table_row_block(a.mix, a.ml, (int)blockIdx.x, a.DP, a.K4, a.table, lds);
    return;
  }


  // ---- GP expected-log-joint block (s,k): glj_block.h ----
  if (a.x_lds) {
    glj_stage_x(a, lds);
    glj_block<true>(a, blockIdx.x - a.n_table, lds);
  } else {
    glj_block<false>(a, blockIdx.x - a.n_table, lds);
  }
}

}  // namespace

#ifdef PREP_TIMES
extern "C" int vbmc_debug_prep_blocks(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prep_blocks), sizeof(unsigned long long) * 2048);
}
extern "C" int vbmc_debug_glj_phases(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_glj_phase), sizeof(unsigned long long) * 8192);
}
#endif
#ifdef FIN_TIMES
extern "C" int vbmc_debug_prep_times(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prep_times), sizeof(unsigned long long) * 66);
}
#endif
int launch_prep(vbmc_ctx* ctx, const PrepArgs& a) { return launch_prep_on(ctx, ctx->stream, a); }

int launch_prep_on(vbmc_ctx* ctx, hipStream_t stream, const PrepArgs& a_in) {
  PrepArgs a = a_in;
  const int D = a.ml.D;
  const int gblocks = a.n_glj;
  const int grid = a.n_table + gblocks + a.gen.n_blocks + (a.mix_copy ? 1 : 0);
  if (grid <= 0) return 0;
  size_t lds = 0;
  a.x_lds = 0;
  if (gblocks > 0) {
    lds = glj_block_lds(D, a.N);
    if (lds > 150 * 1024) return vbmc_fail(ctx, VBMC_E_UNSUP, "gp_log_joint: N=%d too large", a.N);
    // X^T through LDS (glj_block.h) where it is small (<= 40 KB: N = 400 at D = 10) and the launch is only its few
    // latency-bound blocks: the launch's dynamic LDS is the same for all of its workgroups, and a grid that also generates
    // draws (~1e4 throughput blocks when no speculative generation hit) must keep many of them per CU.  (Measured at
    // config 5's shape, 128 KB per block: the step 172.9 -> 174-177 us -- staging 100 x 128 KB costs more than the 13 us
    // chain of dependent loads it replaces; not used there.)
    static const bool x_lds_on = [] {
      const char* e = getenv("VBMC_GLJ_X_LDS");  // measurement aid: 0 = X^T always from memory
      return !(e && e[0] == '0');
    }();
    if (x_lds_on && a.gen.n_blocks == 0 && a.batch == 1 && glj_block_lds_x(D, a.N) <= 40 * 1024) {
      a.x_lds = 1;
      lds = glj_block_lds_x(D, a.N);
    }
  }
  if (a.n_table > 0) {
    const size_t lt = sizeof(double) * (size_t)a.K4 * a.DP;
    if (lt > 150 * 1024) return vbmc_fail(ctx, VBMC_E_UNSUP, "entropy table: K=%d D=%d too large", a.ml.K, D);
    lds = lt > lds ? lt : lds;
  }
  if (lds > 64 * 1024)
    HIP_TRY(ctx, hipFuncSetAttribute((const void*)elbo_prep_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(elbo_prep_kernel, dim3(grid, a.batch), dim3(256), lds, stream, a);
  HIP_TRY(ctx, hipGetLastError());
  return 0;
}
