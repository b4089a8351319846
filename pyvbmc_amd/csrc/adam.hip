// the stochastic optimiser's inner loop kept on the device.
//
// The reference's minimize_adam (vbmc/minimize_adam.py:84-105) calls the objective
// _neg_elcbo(theta, gp, vp, beta, NsK, compute_grad=True, theta_bnd=...)
// (vbmc/variational_optimization.py:238-249) once per iteration and updates theta on the
// host: one host<->device round trip, one synchronisation and ~50 us of host arithmetic
// per iteration.  Here one iteration is four launches on ONE stream and no
// synchronisation, ordered by the stream alone (no events: a record between two dependent
// kernels opens a ~6 us gap on MI355X, a cross-queue wait up to 13 us):
//     elbo_prep_kernel      entropy table rows + GP sums        (+ last third of the draws)
//  -> entmc_ws_kernel       Monte-Carlo entropy sums, and as one extra grid row the "pre"
//                           workgroup (adam_dev.h): everything of dF that does not need the
//                           entropy -- GP-sum finalisation (host twin: api_gp.hip glj_finalize),
//                           soft bounds + weight penalty (api_elbo.hip), their Jacobians
//  -> entmc_finish_kernel   reduction of the entropy partials   (+ first fifth of the next draws)
//  [-> all-reduce]          multi-rank only
//  -> adam_step_kernel      entropy Jacobians (api_entropy.hip entropy_pack), Adam update with
//                           the box clamp, the in-place max-shift of the eta tail
//                           (variational_optimization.py:1082-1085 -- minimize_adam's x IS the
//                           array _neg_elcbo shifts), set_parameters with the lambda
//                           renormalisation (variational_posterior.py:680-759) and the mixture
//                           pack of the next iterate           (+ about half of the next draws)
// The Philox draws of iteration i + 1 are produced by spare workgroups of the three short
// launches, during which the GPU is otherwise idle.  The host only decides when to stop (every
// batch_size = 20 iterations, as the reference does) from the y_tab / x_tab rows it copies back.
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "adam_dev.h"
#include "common.h"
#include "fastmath.h"
#include "entropy_args.h"
#include "philox.h"
#include "glj_block.h"
#include "finish_body.h"
#include "ws_table.h"

using namespace adam_dev;

namespace {

// stand-alone launch of the pre workgroup, for entropy kernels without the extra row
template <bool LDS>
__global__ __launch_bounds__(256) void adam_pre_kernel(AdamDev a) {
  extern __shared__ double sh[];
  __shared__ double red[16];
  adam_pre_body<LDS, false, 8>(a, sh, red, a.pre);
}

// ---------------------------------------------------------------------------
// adam_step_kernel (one workgroup, main stream): the only work between the entropy reduction and
// the next iteration's table launch.  dF = pre + Jacobian(entropy gradient) (entmc_vbmc.py:114-130),
// the Adam update with the box clamp (minimize_adam.py:89-105), then set_parameters + mixture pack
// of the new iterate.  Every array that is read exactly once (entropy sums, pre, m, v, box) goes
// from global memory straight to registers in one batch of independent loads; only theta and the
// mixture attributes, which the pack re-reads across barriers, live in LDS.
template <bool LDS>
__global__ __launch_bounds__(256) void adam_step_kernel(AdamDev a, int do_step, GenSlice gen) {
  extern __shared__ double sh[];
  __shared__ double red[16];
  if (blockIdx.x > 0) {
    // this kernel is one workgroup of latency chains on an otherwise idle GPU: the spare
    // workgroups of its launch generate a slice of the next iteration's draws meanwhile
    gen_slice_block(gen, blockIdx.x - 1, threadIdx.x);
    return;
  }
  const int D = a.D, K = a.K, tid = threadIdx.x, n = a.n_theta;
  const AdamLayout& L = a.lay;
  double* theta = LDS ? sh : a.state + L.o_theta();
  double* aux = LDS ? sh + n : a.state + L.o_aux();
  double* ee = LDS ? sh + n + L.n_aux : a.ee;
  const double* raw = a.state + L.o_raw();
  const bool o_mu = a.mask & 1, o_sg = a.mask & 2, o_lm = a.mask & 4, o_w = a.mask & 8;
  const int p_sg = o_mu ? D * K : 0, p_lm = p_sg + (o_sg ? K : 0), p_w = n - K;
  const int f_w = 1 + D * K + K + D;  // weight block of the entropy accumulator

  constexpr int U = 4;
  double r_raw[U], r_pre[U], r_m[U], r_v[U], r_lo[U], r_hi[U];
  double rw = 0.0;
  int iter = 0;
  if (do_step) {
    iter = a.iter_base[0] + a.it_off;
    if (o_w && tid < K) rw = raw[f_w + tid];  // K <= 256 per pass below
  }
  if (LDS) {
    const int cnt = L.o_hyp();  // theta | aux: eight loads in flight per thread and round
    constexpr int UC = 8;
    for (int base = 0; base < cnt; base += 256 * UC) {
      double r[UC];
#pragma unroll
      for (int u = 0; u < UC; ++u) r[u] = a.state[min(base + u * 256 + tid, cnt - 1)];
#pragma unroll
      for (int u = 0; u < UC; ++u) {
        const int i = base + u * 256 + tid;
        if (i < cnt) sh[i] = r[u];
      }
    }
  }
  // which accumulator entry / Jacobian scale belongs to theta index i
  auto raw_index = [&](int i) -> int {
    if (o_mu && i < D * K) return 1 + i;
    if (o_sg && i >= p_sg && i < p_sg + K) return 1 + D * K + (i - p_sg);
    if (o_lm && i >= p_lm && i < p_lm + D) return 1 + D * K + K + (i - p_lm);
    return f_w + (i - p_w);
  };
  auto load_chunk = [&](int base) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = base + u * 256 + tid;
      const bool in = i < n;
      r_raw[u] = in ? raw[raw_index(i)] : 0.0;
      r_pre[u] = in ? a.pre[i] : 0.0;
      r_m[u] = in ? a.state[L.o_m() + i] : 0.0;
      r_v[u] = in ? a.state[L.o_v() + i] : 0.0;
      r_lo[u] = (in && a.has_box) ? a.state[L.o_xlb() + i] : 0.0;
      r_hi[u] = (in && a.has_box) ? a.state[L.o_xub() + i] : 0.0;
    }
  };
  if (do_step) load_chunk(0);
  // minimize_adam.py:92-98 (evaluated while the loads above are in flight)
  // beta^(i+1) and exp(-(i+1)/decay) as exp2 of host-prepared logarithms (pow() is ~200 dependent
  // instructions; the powers are accurate to ~1e-15 relative for any iteration count in use)
  const double it1 = (double)(iter + 1);
  const double c1 = 1.0 / (1.0 - fm::exp2_fast(it1 * a.l2_beta1));
  const double c2 = 1.0 / (1.0 - fm::exp2_fast(it1 * a.l2_beta2));
  const double step = a.master_min + (a.master_max - a.master_min) * fm::exp2_fast(-it1 * a.l2e_over_decay);
  double* x_row = a.x_tab + (size_t)iter * n;
  double* y_out = a.y_tab + 3 * (size_t)iter;
  __syncthreads();

  if (do_step) {
    const double* sg = aux + K * D;
    const double* lm = sg + K;
    const double* eta = lm + D + K;
    // ---- softmax Jacobian of the entropy's weight gradient: needs two sums over k ----
    double sm_s = 1.0, sm_dot = 0.0;
    if (o_w) {
      double ps = 0.0, pd = 0.0;
      for (int k = tid; k < K; k += 256) {
        const double e = fm::exp2_fast(0x1.71547652b82fep+0 * eta[k]);
        ee[k] = e;
        ps += e;
        pd += e * (k == tid ? rw : raw[f_w + k]);
      }
      ps = wave_sum(ps);
      pd = wave_sum(pd);
      if ((tid & 63) == 0) {
        red[tid >> 6] = ps;
        red[4 + (tid >> 6)] = pd;
      }
      __syncthreads();
      sm_s = (red[0] + red[1]) + (red[2] + red[3]);
      sm_dot = (red[4] + red[5]) + (red[6] + red[7]);
    }
    if (tid == 0) {
      const double G = a.pre[n], loss = a.pre[n + 1], H = raw[0];
      y_out[0] = -G - H + loss;
      y_out[1] = G;
      y_out[2] = H;
    }
    for (int base = 0; base < n; base += 256 * U) {
      if (base > 0) load_chunk(base);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = base + u * 256 + tid;
        if (i >= n) continue;
        // ---- dF: pre + Jacobian of the entropy part of F = -G - H ----
        double g;
        if (o_mu && i < D * K) {
          g = r_pre[u] - r_raw[u];
        } else if (o_sg && i >= p_sg && i < p_sg + K) {
          g = r_pre[u] - r_raw[u] * sg[i - p_sg];
        } else if (o_lm && i >= p_lm && i < p_lm + D) {
          g = r_pre[u] - r_raw[u] * lm[i - p_lm];
        } else {
          const double e = ee[i - p_w];
          g = r_pre[u] + (e * sm_dot / (sm_s * sm_s) - e * r_raw[u] / sm_s);
        }
        // ---- Adam update (minimize_adam.py:89-105) ----
        const double m = a.beta1 * r_m[u] + (1.0 - a.beta1) * g;
        const double v = a.beta2 * r_v[u] + (1.0 - a.beta2) * (g * g);
        a.state[L.o_m() + i] = m;
        a.state[L.o_v() + i] = v;
        const double m_hat = m * c1, v_hat = v * c2;
        double x = theta[i] - step * m_hat / (sqrt(v_hat) + a.fudge);
        if (a.has_box) x = fmin(r_hi[u], fmax(r_lo[u], x));
        theta[i] = x;
        x_row[i] = x;
      }
    }
    __syncthreads();
  }

  pack_from_theta(a, theta, aux, red, a.mix);
  if (LDS) {  // write the mirrored, modified arrays back
    __syncthreads();
    for (int i = tid; i < L.o_hyp(); i += 256) a.state[i] = sh[i];  // theta | aux
  }
}

// ---------------------------------------------------------------------------
// adam_tail_kernel: everything between two entropy launches in ONE launch (the two-launch iteration, below).
//   blocks [0, n_main)            the entropy reduction (finish_body.h), results written through to the state block; the
//                                 last block to count publishes the iteration's sequence number at agent scope
//   blocks [n_main, n_main + K)   block j waits for that word, then does -- every one of them, identically -- what
//                                 adam_step_kernel does (Jacobians, dF, Adam update, set_parameters, the pack: in LDS) and
//                                 from that pack row j of the entropy kernel's (j,k) table.  Block 0 of them (the writer)
//                                 also puts theta, the attributes, m, v, the iterate's row of x_tab / y_tab and the pack into
//                                 memory -- m and v once every block has read the old ones (one counter), so nobody can
//                                 read an updated moment
//   the rest                      the next iteration's draws
// A launch boundary costs ~1.5 us and the hand-off inside the launch ~2; the four-launch iteration paid four boundaries and
// ran the reduction, the step and the table one behind the other with a kernel's fill and drain each.
struct TailArgs {
  const double* partial = nullptr;
  int chunks = 0, stride = 0, mu_from_w = 0;
  double inv_ns = 0.0;
  DoneSignal red;             // dev mode: counter, flag, sequence number of the reduction
  int DP = 0, K4 = 0;
  double* table = nullptr;
  unsigned long long* rd_cnt = nullptr;  // table blocks that are through with the old moments (monotonic)
  unsigned long long rd_target = 0;
  unsigned long long timeout = 2000000;  // wall-clock ticks (100 MHz): 20 ms
  GenSlice gen;
  unsigned long long* times = nullptr;  // optional [8] stamps of the writer block (VBMC_TAIL_TIMES=1)
};
#define TAIL_STAMP(i) do { if (t.times && writer && tid == 0) t.times[i] = wall_clock64(); } while (0)

__global__ __launch_bounds__(256) void adam_tail_kernel(AdamDev a, TailArgs t) {
  extern __shared__ double sh[];
  __shared__ double red[16];
  __shared__ int s_ok;
  const int D = a.D, K = a.K, tid = threadIdx.x, n = a.n_theta;
  const AdamLayout& L = a.lay;
  const int n_main = (L.n_raw + 3) / 4;
  if ((int)blockIdx.x < n_main + K) __builtin_amdgcn_s_setprio(3);  // the chain's blocks before the draws' on every SIMD they share
  if ((int)blockIdx.x < n_main) {
    entmc_finish_body(t.partial, t.chunks, t.stride, a.mix, a.ml, t.inv_ns, 1, t.mu_from_w, a.state + L.o_raw(), GenSlice(), t.red);
    return;
  }
  const int tb = (int)blockIdx.x - n_main;
  if (tb >= K) {
    gen_slice_block(t.gen, tb - K, tid);
    return;
  }
  const bool writer = tb == 0;
  TAIL_STAMP(0);
  // ---- wait for the reduction (bounded) ----
  if (tid == 0) {
    const unsigned long long t0 = wall_clock64();
    int ok = 1;
    while (__hip_atomic_load(t.red.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != t.red.seq) {
      if (wall_clock64() - t0 > t.timeout) {
        ok = 0;
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
    s_ok = ok;
  }
  __syncthreads();
  if (!s_ok) {
    if (tid == 0) atomicOr(a.status, 8);
    return;
  }
  TAIL_STAMP(1);
  // LDS: theta n | aux | ee K | pack | table scratch K4 DP | new m, new v (writer)
  double* theta = sh;
  double* aux = sh + n;
  double* ee = aux + L.n_aux;
  double* pack = ee + K;
  double* tsc = pack + a.ml.total;
  double* new_m = tsc + t.K4 * t.DP;
  double* new_v = new_m + n;
  // the raw vector was written through by the reduction's blocks of this launch: read past this CU's L1 (agent-scope loads;
  // an acquire fence instead costs ~2 us on the chain); everything else read here was written by earlier launches
  const double* raw_p = a.state + L.o_raw();
  auto raw = [&](int i) { return __hip_atomic_load(raw_p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  const bool o_mu = a.mask & 1, o_sg = a.mask & 2, o_lm = a.mask & 4, o_w = a.mask & 8;
  const int p_sg = o_mu ? D * K : 0, p_lm = p_sg + (o_sg ? K : 0), p_w = n - K;
  const int f_w = 1 + D * K + K + D;

  constexpr int U = 4;
  double r_raw[U], r_pre[U], r_m[U], r_v[U], r_lo[U], r_hi[U];
  double rw = 0.0;
  const int iter = a.iter_base[0] + a.it_off;
  if (o_w && tid < K) rw = raw(f_w + tid);
  for (int i = tid; i < L.o_hyp(); i += 256) sh[i] = a.state[i];  // theta | aux
  auto raw_index = [&](int i) -> int {
    if (o_mu && i < D * K) return 1 + i;
    if (o_sg && i >= p_sg && i < p_sg + K) return 1 + D * K + (i - p_sg);
    if (o_lm && i >= p_lm && i < p_lm + D) return 1 + D * K + K + (i - p_lm);
    return f_w + (i - p_w);
  };
  auto load_chunk = [&](int base) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = base + u * 256 + tid;
      const bool in = i < n;
      r_raw[u] = in ? raw(raw_index(i)) : 0.0;
      r_pre[u] = in ? a.pre[i] : 0.0;
      r_m[u] = in ? a.state[L.o_m() + i] : 0.0;
      r_v[u] = in ? a.state[L.o_v() + i] : 0.0;
      r_lo[u] = (in && a.has_box) ? a.state[L.o_xlb() + i] : 0.0;
      r_hi[u] = (in && a.has_box) ? a.state[L.o_xub() + i] : 0.0;
    }
  };
  load_chunk(0);
  const double it1 = (double)(iter + 1);
  const double c1 = 1.0 / (1.0 - fm::exp2_fast(it1 * a.l2_beta1));
  const double c2 = 1.0 / (1.0 - fm::exp2_fast(it1 * a.l2_beta2));
  const double step = a.master_min + (a.master_max - a.master_min) * fm::exp2_fast(-it1 * a.l2e_over_decay);
  double* x_row = a.x_tab + (size_t)iter * n;
  double* y_out = a.y_tab + 3 * (size_t)iter;
  __syncthreads();
  TAIL_STAMP(2);
  {
    const double* sg = aux + K * D;
    const double* lm = sg + K;
    const double* eta = lm + D + K;
    double sm_s = 1.0, sm_dot = 0.0;
    if (o_w) {
      double ps = 0.0, pd = 0.0;
      for (int k = tid; k < K; k += 256) {
        const double e = fm::exp2_fast(0x1.71547652b82fep+0 * eta[k]);
        ee[k] = e;
        ps += e;
        pd += e * (k == tid ? rw : raw(f_w + k));
      }
      ps = wave_sum(ps);
      pd = wave_sum(pd);
      if ((tid & 63) == 0) {
        red[tid >> 6] = ps;
        red[4 + (tid >> 6)] = pd;
      }
      __syncthreads();
      sm_s = (red[0] + red[1]) + (red[2] + red[3]);
      sm_dot = (red[4] + red[5]) + (red[6] + red[7]);
    }
    if (writer && tid == 0) {
      const double G = a.pre[n], loss = a.pre[n + 1], H = raw(0);
      y_out[0] = -G - H + loss;
      y_out[1] = G;
      y_out[2] = H;
    }
    for (int base = 0; base < n; base += 256 * U) {
      if (base > 0) load_chunk(base);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = base + u * 256 + tid;
        if (i >= n) continue;
        double g;
        if (o_mu && i < D * K) {
          g = r_pre[u] - r_raw[u];
        } else if (o_sg && i >= p_sg && i < p_sg + K) {
          g = r_pre[u] - r_raw[u] * sg[i - p_sg];
        } else if (o_lm && i >= p_lm && i < p_lm + D) {
          g = r_pre[u] - r_raw[u] * lm[i - p_lm];
        } else {
          const double e = ee[i - p_w];
          g = r_pre[u] + (e * sm_dot / (sm_s * sm_s) - e * r_raw[u] / sm_s);
        }
        const double m = a.beta1 * r_m[u] + (1.0 - a.beta1) * g;
        const double v = a.beta2 * r_v[u] + (1.0 - a.beta2) * (g * g);
        if (writer) {
          new_m[i] = m;
          new_v[i] = v;
        }
        const double m_hat = m * c1, v_hat = v * c2;
        double x = theta[i] - step * m_hat / (sqrt(v_hat) + a.fudge);
        if (a.has_box) x = fmin(r_hi[u], fmax(r_lo[u], x));
        theta[i] = x;
        if (writer) x_row[i] = x;
      }
    }
    __syncthreads();
  }
  // this block is through with the old moments (every load above has been consumed)
  if (tid == 0) __hip_atomic_fetch_add(t.rd_cnt, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  TAIL_STAMP(3);

  pack_from_theta(a, theta, aux, red, pack);
  __syncthreads();
  TAIL_STAMP(4);
  ws_table_row_block(pack, a.ml, tb, t.DP, t.K4, t.table, tsc);
  if (!writer) return;
  TAIL_STAMP(5);
  // ---- the writer: the new iterate goes to memory, the moments once every block has read the old ones ----
  for (int i = tid; i < a.ml.total; i += 256) a.mix[i] = pack[i];  // (read by the reduction blocks only: they are done)
  if (tid == 0) {
    const unsigned long long t0 = wall_clock64();
    int ok = 1;
    while (__hip_atomic_load(t.rd_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < t.rd_target) {
      if (wall_clock64() - t0 > t.timeout) {
        ok = 0;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    s_ok = ok;
  }
  __syncthreads();
  if (!s_ok) {
    if (tid == 0) atomicOr(a.status, 8);
    return;
  }
  TAIL_STAMP(6);
  for (int i = tid; i < L.o_hyp(); i += 256) a.state[i] = sh[i];  // theta | aux
  for (int i = tid; i < n; i += 256) {
    a.state[L.o_m() + i] = new_m[i];
    a.state[L.o_v() + i] = new_v[i];
  }
  TAIL_STAMP(7);
}

__global__ void set_int_kernel(int* p, int v) { *p = v; }

}  // namespace

// ---------------------------------------------------------------------------
struct AdamState {
  bool active = false;
  int n_theta = 0, n_bnd = 0, max_iter = 0, iter = 0, mask = 0;
  int64_t ns = 0, row_begin = 0, row_count = 0;
  int eps_mode = 0;
  uint64_t seed = 0;
  double master_min = 0, master_max = 0, master_decay = 0;
  double tol_con = 0, w_thresh = 0, w_pen = 0;
  bool has_box = false, has_bnd = false;
  AdamLayout lay;
  bool pre_lds = false, step_lds = false;  // which kernels keep their working set in LDS
  size_t pre_lds_bytes = 0, step_lds_bytes = 0;
  double* d_buf = nullptr;
  size_t d_cap = 0;
  int* d_status = nullptr;
  // carve of d_buf
  double *state = nullptr, *work = nullptr, *ee = nullptr, *pre = nullptr, *x_tab = nullptr, *y_tab = nullptr;
  AdamDev* d_args = nullptr;  // the argument block in device memory, for the entropy launch's pre row
  // Philox mode: the draws of iteration i + 1 are generated by spare workgroups of iteration i's
  // finish and step launches and of iteration i + 1's prep launch -- the three short launches
  // during which the GPU is otherwise idle (~26 us per iteration at BASELINE config 3, about what
  // generating 5e6 normals takes).  One buffer suffices: the entropy kernel that reads it runs
  // between prep and finish.  No second stream, no events: a record between two dependent
  // kernels opens a ~6 us gap on MI355X and a cross-queue wait up to 13 us.
  bool pregen = false;
  size_t n_eps = 0;          // doubles per iteration
  double* d_eps1 = nullptr;  // [K][row_count][D]
  size_t eps_cap = 0;        // doubles allocated
  double eps_have = 0.0;     // the share [0, eps_have) of the NEXT iteration's draws that earlier launches already put into
                             // the buffer: 0 at the start of a run (and after a batch in the one-launch form), f_step behind a
                             // four-launch iteration (its finish and step launches), 1 behind a two-launch iteration
  // the fused loop (adam_fused.hip): one launch per batch at the reference's own sample counts
  bool fused = false;
  int fused_gave_up = 0;  // batches redone as four launches per iteration after a bounded wait ran out
  FusedArgs fz;
  size_t fused_lds = 0;
  double* d_xch = nullptr;
  size_t xch_cap = 0;
  unsigned long long* d_flags = nullptr;  // [512]
  // the two-launch iteration (adam_tail_kernel): its own (j,k) table, which the tail launch of iteration i fills for
  // iteration i + 1, and the words its hand-offs use -- [0] GP-sum word, [1] reduction word, [2] readers' counter,
  // ints at [8]: GP-sum counter, [9]: reduction counter
  double* d_table = nullptr;
  size_t table_cap = 0;
  bool table_valid = false;   // d_table holds the rows of the current iterate
  unsigned long long* d_sync = nullptr;
  unsigned long long sync_seq = 0, tail_launches = 0;
  size_t tail_lds_bytes = 0;
  bool tail_ok = false;
  int last_form = 0;          // launches per iteration of the last batch (2 or 4)
};

// How much dynamic LDS the pre workgroup may ask of the entropy launch it rides in (its working set then sits in LDS: ~10 us
// against 25-30 from global memory).  The launch's dynamic LDS is the same for all of its workgroups: where the entropy
// kernel runs two workgroups per CU the pre workgroup's share must leave room for both (60 KB: rounds 4-5); the builds that
// run ONE workgroup per CU -- the matrix-pipe form and the wave-split instantiations at one wave per SIMD, i.e. the large
// K D shapes whose working set is the largest -- have the CU's LDS to themselves (round 6: config 5's own shape, D = 20,
// K = 100, N = 800, went from 33 to ~15 us per entropy launch of the four-launch iteration).
static size_t pre_lds_limit(const vbmc_ctx* ctx, const EntPlan& plan) {
  if (!plan.ws) return 0;
  if (plan.a.sp.cus == 0 && !entmc_small_applies(plan.a, plan.DP) && ctx->opt_entmc_mfma && entmc_mfma_applies(plan.a, plan.DP))
    return 158 * 1024;  // (its static arrays are 1 056 B)
  return ws_min_waves(plan.DP, ws_ktmax_for(ctx->K), true) >= 2 ? 60 * 1024 : 110 * 1024;  // (the widest build holds 38 KB of static LDS)
}

static size_t fused_backup_len(const AdamDev& a) { return (size_t)a.lay.o_hyp() + 2 * (size_t)a.n_theta + (size_t)a.ml.total; }

// A fused launch that gave up waiting may have got as far as its write-back in workgroup 0 (the others can run out of
// time during the very last iteration): put back the state it started from -- workgroup 0 saved it behind the exchange
// records before its first iteration -- so that the batch can be redone from the same point.
static int fused_restore(vbmc_ctx* ctx, AdamState* st) {
  const AdamDev& a = st->fz.a;
  const AdamLayout& L = a.lay;
  const double* b = st->fz.backup;
  const size_t nh = (size_t)L.o_hyp(), n = (size_t)a.n_theta;
  HIP_TRY(ctx, hipMemcpyAsync(a.state, b, sizeof(double) * nh, hipMemcpyDeviceToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(a.state + L.o_m(), b + nh, sizeof(double) * n, hipMemcpyDeviceToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(a.state + L.o_v(), b + nh + n, sizeof(double) * n, hipMemcpyDeviceToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(a.mix, b + nh + 2 * n, sizeof(double) * (size_t)a.ml.total, hipMemcpyDeviceToDevice, ctx->stream));
  return 0;
}

static AdamState* adam_of(vbmc_ctx* ctx) {
  if (!ctx->adam) ctx->adam = new AdamState();
  return (AdamState*)ctx->adam;
}

void adam_free(vbmc_ctx* ctx) {
  AdamState* st = (AdamState*)ctx->adam;
  if (!st) return;
  if (st->d_buf) (void)hipFree(st->d_buf);
  if (st->d_status) (void)hipFree(st->d_status);
  if (st->d_eps1) (void)hipFree(st->d_eps1);
  if (st->d_args) (void)hipFree(st->d_args);
  if (st->d_xch) (void)hipFree(st->d_xch);
  if (st->d_flags) (void)hipFree(st->d_flags);
  if (st->d_table) (void)hipFree(st->d_table);
  if (st->d_sync) (void)hipFree(st->d_sync);
  delete st;
  ctx->adam = nullptr;
}

static void launch_step(const AdamState& st, hipStream_t sm, const AdamDev& a, int do_step, const GenSlice& gen) {
  const dim3 grid(1 + gen.n_blocks);
  if (st.step_lds) hipLaunchKernelGGL(adam_step_kernel<true>, grid, dim3(256), st.step_lds_bytes, sm, a, do_step, gen);
  else hipLaunchKernelGGL(adam_step_kernel<false>, grid, dim3(256), 0, sm, a, do_step, gen);
}

static void launch_pre(const AdamState& st, hipStream_t sm, const AdamDev& a) {
  if (st.pre_lds) hipLaunchKernelGGL(adam_pre_kernel<true>, dim3(1), dim3(256), st.pre_lds_bytes, sm, a);
  else hipLaunchKernelGGL(adam_pre_kernel<false>, dim3(1), dim3(256), 0, sm, a);
}

static void fill_dev(const vbmc_ctx* ctx, const AdamState& st, AdamDev& a) {
  const GpState& g = ctx->gp;
  a.ml = ctx->ml;
  a.D = ctx->D;
  a.K = ctx->K;
  a.S = g.S;
  a.P = g.P;
  a.mean_kind = g.mean_kind;
  a.mask = st.mask;
  a.n_theta = st.n_theta;
  a.n_bnd = st.n_bnd;
  a.has_box = st.has_box;
  a.has_bnd = st.has_bnd;
  a.lay = st.lay;
  a.mix = ctx->d_mix;
  a.state = st.state;
  a.work = st.work;
  a.ee = st.ee;
  a.pre = st.pre;
  a.tol_con = st.tol_con;
  a.w_thresh = st.w_thresh;
  a.w_pen = st.w_pen;
  a.fudge = std::sqrt(2.220446049250313e-16);  // sqrt(np.spacing(1))
  a.beta1 = 0.9;
  a.beta2 = 0.999;
  a.c_norm = 1.0 / std::pow(2.0 * M_PI, 0.5 * ctx->D);
  a.l2_beta1 = std::log2(a.beta1);
  a.l2_beta2 = std::log2(a.beta2);
  a.l2e_over_decay = 1.4426950408889634 / st.master_decay;
  a.master_min = st.master_min;
  a.master_max = st.master_max;
  a.master_decay = st.master_decay;
  a.iter_base = st.d_status + 1;
  a.it_off = 0;
  a.x_tab = st.x_tab;
  a.y_tab = st.y_tab;
  a.status = st.d_status;
}

extern "C" int vbmc_adam_begin(vbmc_ctx* ctx, const double* theta0, int n_theta,
                               const vbmc_elbo_opts* opts, const double* lb, const double* ub,
                               int max_iter, double master_min, double master_max,
                               double master_decay) {
  if (!ctx || !theta0 || !opts || max_iter < 1) return VBMC_E_ARG;
  NEED_DEVICE(ctx);
  if (!ctx->mix_set) return vbmc_fail(ctx, VBMC_E_ARG, "adam_begin: mixture (D,K) not set");
  if (!ctx->gp.set) return vbmc_fail(ctx, VBMC_E_ARG, "adam_begin: GP not set");
  if (ctx->gp.D != ctx->D) return vbmc_fail(ctx, VBMC_E_ARG, "adam_begin: GP/mixture D mismatch");
  if (opts->ns_per_comp <= 0 || (opts->ns_per_comp & 1))
    return vbmc_fail(ctx, VBMC_E_UNSUP,
                     "adam_begin: the stochastic optimiser needs an even ns_per_comp > 0");
  if (!opts->compute_grad) return vbmc_fail(ctx, VBMC_E_ARG, "adam_begin: compute_grad must be set");
  if ((lb == nullptr) != (ub == nullptr)) return vbmc_fail(ctx, VBMC_E_ARG, "adam_begin: lb/ub must both be given");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (ctx->spec.armed) spec_disarm(ctx);  // launches waiting for a theta would hold CUs the loop's workgroups need
  const int D = ctx->D, K = ctx->K, S = ctx->gp.S;
  const int mask = opts->optimize_mask;
  const int need = ((mask & 1) ? D * K : 0) + ((mask & 2) ? K : 0) + ((mask & 4) ? D : 0) +
                   ((mask & 8) ? K : 0);
  if (n_theta != need || n_theta < 1)
    return vbmc_fail(ctx, VBMC_E_ARG, "adam_begin: theta length %d does not match D=%d K=%d mask=%d",
                     n_theta, D, K, mask);
  for (int i = 0; i < n_theta; ++i)
    if (!std::isfinite(theta0[i])) return vbmc_fail(ctx, VBMC_E_NONFINITE, "theta has a non-finite entry");
  AdamState* st = adam_of(ctx);
  st->active = false;
  ctx->pack_valid = false;  // the loop rewrites d_mix on the device; vbmc_adam_end re-syncs the host copies
  st->has_bnd = opts->bnd_lb && opts->bnd_ub;
  st->n_bnd = st->has_bnd ? opts->n_bnd : 0;
  if (st->has_bnd) {
    const int want = ((mask & 1) ? D * K : 0) + ((mask & 6) ? D * K : 0) + ((mask & 8) ? K : 0);
    if (st->n_bnd != want)
      return vbmc_fail(ctx, VBMC_E_ARG, "adam_begin: bounds length %d != %d", st->n_bnd, want);
  }
  st->has_box = lb != nullptr;
  st->n_theta = n_theta;
  st->max_iter = max_iter;
  st->iter = 0;
  st->mask = mask;
  st->ns = opts->ns_per_comp;
  st->eps_mode = opts->eps_mode;
  st->seed = opts->seed;
  st->master_min = master_min;
  st->master_max = master_max;
  st->master_decay = master_decay;
  st->tol_con = opts->tol_con;
  st->w_thresh = opts->weight_threshold;
  st->w_pen = opts->weight_penalty;
  const int64_t n_half = st->ns / 2;
  st->row_begin = opts->row_begin;
  st->row_count = opts->row_count;
  if (st->row_count < 0) {
    st->row_begin = n_half * ctx->rank / ctx->world;
    st->row_count = n_half * (ctx->rank + 1) / ctx->world - st->row_begin;
  }
  if (st->row_begin < 0 || st->row_begin + st->row_count > n_half)
    return vbmc_fail(ctx, VBMC_E_ARG, "adam_begin: bad row slice");
  if (st->eps_mode == VBMC_EPS_RESIDENT &&
      (!ctx->d_eps || ctx->eps_K != K || ctx->eps_D != D || ctx->eps_n_half != n_half ||
       ctx->eps_row_begin != st->row_begin || ctx->eps_rows != st->row_count))
    return vbmc_fail(ctx, VBMC_E_ARG, "adam_begin: resident eps does not match the request");

  // one allocation, carved: the state block, scratch, x_tab, y_tab
  AdamLayout& L = st->lay;
  L.n = n_theta;
  L.n_aux = (int)aux_len(D, K);
  L.n_hyp = S * ctx->gp.P;
  L.n_raw = raw_len(D, K);
  L.n_res = S * K * (1 + 2 * D);
  L.n_bnd = st->n_bnd;
  const size_t n_work = work_len(D, K, S, st->n_bnd);
  {
    // LDS plan: each kernel keeps its working set in LDS when it fits
    // (beyond it the kernels work from global memory.  158 KB: the CU's 160 KB less the static arrays of the kernels that
    // host these working sets -- adam_pre_kernel / adam_step_kernel 128 B, the matrix-pipe entropy kernel 1 056 B; round 6:
    // was 150 KB, 3 KB short of config 5's own shape at S = 1 -- D = 20, K = 100: 153.5 KB)
    const size_t cap = 158 * 1024 / sizeof(double);
    const size_t n_pre = (size_t)L.o_blb() + n_work, n_step = (size_t)L.o_hyp() + K;
    st->pre_lds = n_pre <= cap;
    st->step_lds = n_step <= cap;
    st->pre_lds_bytes = sizeof(double) * n_pre;
    st->step_lds_bytes = sizeof(double) * n_step;
    if (st->pre_lds && st->pre_lds_bytes > 64 * 1024)
      HIP_TRY(ctx, hipFuncSetAttribute((const void*)adam_pre_kernel<true>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)st->pre_lds_bytes));
    if (st->step_lds && st->step_lds_bytes > 64 * 1024)
      HIP_TRY(ctx, hipFuncSetAttribute((const void*)adam_step_kernel<true>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)st->step_lds_bytes));
  }
  const size_t total = (size_t)L.end() + n_work + (size_t)K + (size_t)n_theta + 2 +
                       (size_t)max_iter * n_theta + 3 * (size_t)max_iter + 64;
  int rc = ensure_dev(ctx, &st->d_buf, &st->d_cap, total);
  if (rc) return rc;
  if (!st->d_status) HIP_TRY(ctx, hipMalloc((void**)&st->d_status, 4 * sizeof(int)));  // flag, iteration base
  if (!st->d_args) HIP_TRY(ctx, hipMalloc((void**)&st->d_args, sizeof(AdamDev)));
  st->state = st->d_buf;
  st->work = st->state + L.end();
  st->ee = st->work + n_work;
  st->pre = st->ee + K;
  st->x_tab = st->pre + n_theta + 2;
  st->y_tab = st->x_tab + (size_t)max_iter * n_theta;

  hipStream_t sm = ctx->stream;
  double* sb = st->state;
  HIP_TRY(ctx, hipMemsetAsync(sb, 0, sizeof(double) * L.end(), sm));  // m = v = 0
  HIP_TRY(ctx, hipMemcpyAsync(sb + L.o_theta(), theta0, sizeof(double) * n_theta, hipMemcpyHostToDevice, sm));
  HIP_TRY(ctx, hipMemsetAsync(st->d_status, 0, 4 * sizeof(int), sm));
  // attributes of the blocks theta does not carry start from the ctx mixture
  std::vector<double> aux(L.n_aux);
  memcpy(aux.data(), ctx->mu.data(), sizeof(double) * K * D);
  memcpy(aux.data() + K * D, ctx->sigma.data(), sizeof(double) * K);
  memcpy(aux.data() + K * D + K, ctx->lambd.data(), sizeof(double) * D);
  memcpy(aux.data() + K * D + K + D, ctx->w.data(), sizeof(double) * K);
  memcpy(aux.data() + K * D + 2 * K + D, ctx->eta.data(), sizeof(double) * K);
  HIP_TRY(ctx, hipMemcpyAsync(sb + L.o_aux(), aux.data(), sizeof(double) * L.n_aux, hipMemcpyHostToDevice, sm));
  HIP_TRY(ctx, hipMemcpyAsync(sb + L.o_hyp(), ctx->gp.d_hyp, sizeof(double) * L.n_hyp, hipMemcpyDeviceToDevice, sm));
  if (st->has_bnd) {
    HIP_TRY(ctx, hipMemcpyAsync(sb + L.o_blb(), opts->bnd_lb, sizeof(double) * st->n_bnd, hipMemcpyHostToDevice, sm));
    HIP_TRY(ctx, hipMemcpyAsync(sb + L.o_bub(), opts->bnd_ub, sizeof(double) * st->n_bnd, hipMemcpyHostToDevice, sm));
  }
  if (st->has_box) {
    HIP_TRY(ctx, hipMemcpyAsync(sb + L.o_xlb(), lb, sizeof(double) * n_theta, hipMemcpyHostToDevice, sm));
    HIP_TRY(ctx, hipMemcpyAsync(sb + L.o_xub(), ub, sizeof(double) * n_theta, hipMemcpyHostToDevice, sm));
  }
  HIP_TRY(ctx, hipStreamSynchronize(sm));  // `aux` is pageable stack-adjacent memory
  ctx->pack_in_flight = false;
  AdamDev a;
  fill_dev(ctx, *st, a);
  HIP_TRY(ctx, hipMemcpyAsync(st->d_args, &a, sizeof(AdamDev), hipMemcpyHostToDevice, sm));
  HIP_TRY(ctx, hipStreamSynchronize(sm));  // `a` is a stack object
  launch_step(*st, sm, a, 0, GenSlice());
  HIP_TRY(ctx, hipGetLastError());
  // draws generated ahead by spare workgroups (Philox mode, unless switched off or > 32 GiB)
  {
    st->n_eps = (size_t)K * (size_t)st->row_count * D;
    st->pregen = st->eps_mode == VBMC_EPS_PHILOX && st->n_eps > 0 && st->n_eps <= ((size_t)1 << 32);
    st->eps_have = 0.0;
    if (st->pregen && st->eps_cap < st->n_eps) {
      if (st->d_eps1) HIP_TRY(ctx, hipFree(st->d_eps1));
      st->d_eps1 = nullptr;
      st->eps_cap = 0;
      HIP_TRY(ctx, hipMalloc((void**)&st->d_eps1, sizeof(double) * st->n_eps));
      st->eps_cap = st->n_eps;
    }
  }
  // the two-launch iteration (adam_tail_kernel): words of its hand-offs, LDS plan
  {
    if (!st->d_sync) HIP_TRY(ctx, hipMalloc((void**)&st->d_sync, 16 * sizeof(unsigned long long)));
    HIP_TRY(ctx, hipMemsetAsync(st->d_sync, 0, 16 * sizeof(unsigned long long), sm));
    st->sync_seq = st->tail_launches = 0;
    st->table_valid = false;
    const int DPp = [&] { const int dps[] = {2, 4, 6, 8, 10, 12, 16, 20, 24, 32}; for (int dp : dps) if (D <= dp) return dp; return 0; }();
    const size_t n_tab = (size_t)ws_table_rows(K) * (size_t)DPp;
    st->tail_lds_bytes = sizeof(double) * ((size_t)L.o_hyp() + K + (size_t)ctx->ml.total + n_tab + 2 * (size_t)n_theta);
    st->tail_ok = DPp > 0 && K <= 128 && st->tail_lds_bytes <= 150 * 1024;
    if (st->tail_ok && st->tail_lds_bytes > 48 * 1024)
      HIP_TRY(ctx, hipFuncSetAttribute((const void*)adam_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)st->tail_lds_bytes));
    if (st->tail_ok) {
      const size_t need_t = (size_t)K * (size_t)ws_table_rows(K) * (size_t)(DPp + 6);
      if (st->table_cap < need_t) {
        if (st->d_table) HIP_TRY(ctx, hipFree(st->d_table));
        st->d_table = nullptr;
        st->table_cap = 0;
        HIP_TRY(ctx, hipMalloc((void**)&st->d_table, sizeof(double) * need_t));
        st->table_cap = need_t;
      }
    }
  }
  // the fused loop where its shape applies (adam_fused.hip)
  st->fused = false;
  st->fused_gave_up = 0;
  if (ctx->opt_adam_fused && ctx->world == 1 && st->row_begin == 0 && st->row_count == n_half) {
    FusedArgs& f = st->fz;
    f = FusedArgs();
    f.a = a;
    f.N = ctx->gp.N;
    f.rows = (int)st->row_count;
    f.cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
    // (up to 160 rows per component: beyond that the lane-per-component row loop of the fused kernel loses to the wave-split
    // kernel's 64 rows per instruction -- K = 50, D = 10, NsK = 130 / 256 / 1 024 / 4 096: 22.5 / 27.0 / 38.0 / 100 us per
    // iteration against 33.3 / 33.3 / 34.5 / 46.8 with four launches, tools/adam_d20_probe.py)
    const size_t lds = st->row_count <= 160 ? adam_fused_plan(f) : 0;
    if (lds) {
      const size_t rt = (size_t)f.n_ent * (2 + 2 * D + K) + (size_t)S * K * (2 * D + 4);  // (n_ent = K R: R planes of entropy records)
      // (behind the records: the copy of the state a launch starts from, restored if it gives up waiting)
      rc = ensure_dev(ctx, &st->d_xch, &st->xch_cap, 2 * rt + fused_backup_len(a));
      if (rc) return rc;
      if (!st->d_flags) HIP_TRY(ctx, hipMalloc((void**)&st->d_flags, 512 * sizeof(unsigned long long)));
      f.XT = ctx->gp.d_XT;
      f.alpha = ctx->gp.d_alpha;
      f.eps_mode = st->eps_mode;
      f.eps = ctx->d_eps;
      f.eps_rows = st->row_count;
      f.n_half = n_half;
      f.row_begin = st->row_begin;
      f.seed = st->seed;
      f.inv_ns = 1.0 / (double)st->ns;
      f.xch = st->d_xch;
      f.backup = st->d_xch + 2 * rt;
      f.flags = st->d_flags;
      st->fused_lds = lds;
      st->fused = true;
    }
  }
  st->active = true;
  return VBMC_OK;
}

// One batch of iterations [i0, i0 + n_iters): four launches per iteration on one stream,
//   prep (table rows + GP sums) -> entropy (+ the pre row) -> finish [-> all-reduce] -> step,
// ordered by the stream alone.
static int enqueue_batch(vbmc_ctx* ctx, AdamState* st, int i0, int n_iters, bool multi) {
  hipStream_t sm = ctx->stream;
  AdamDev a;
  fill_dev(ctx, *st, a);
  const bool use_gen = st->pregen;
  // shares of the next iteration's draws given to the finish / step / prep launches, about
  // proportional to how long each of them leaves the GPU idle (5.5 / 12 / 8.5 us)
  // (a sweep of the split moved the iteration time by < 1 %)
  const double f_fin = 0.20, f_step = 0.67;
  auto slice = [&](int seed_off, double f0, double f1) {
    return make_gen_slice(st->d_eps1, ctx->K, ctx->D, st->row_count, st->ns / 2, st->row_begin,
                          st->seed + (uint64_t)seed_off, st->d_status + 1, f0, f1);
  };
  // The two-launch iteration where it gains (measured, config 3's shape: 20 000 samples per component 97.5 -> 93 us per
  // iteration; 8 192: 61 -> 64; 4 096: 48 -> 52 -- below ~40 us of entropy kernel the GP sums and the pre workgroup in its
  // launch outlast the entropy parts, and the hand-off inside the tail launch costs what the two boundaries it replaces
  // cost): jobs of at least 250 000 antithetic rows.  Option "adam_tail" = 2 forces it wherever its shape applies (tests).
  const bool try_tail = ctx->opt_adam_tail && !multi && st->tail_ok &&
                        (ctx->opt_adam_tail == 2 || (int64_t)ctx->K * st->row_count >= 250000);
  // GP items per workgroup of the entropy launch: four where the entropy parts run long enough (>= ~60 us) to cover four
  // items (28 us) and the pre workgroup behind them (10 us from LDS) -- 2 / 3 / 4 / 5 items: 91.6 / 91.2 / 90.4 / 96.0 us per
  // iteration at config 3 --, one below that
  static const int gp_per_slot_env = [] {
    const char* e = getenv("VBMC_TAIL_GP_PER_SLOT");  // measurement aid
    return e ? atoi(e) : 0;
  }();
  const int tail_gp_per_slot = gp_per_slot_env > 0 ? gp_per_slot_env : ((int64_t)ctx->K * st->row_count >= 250000 ? 4 : 1);
  for (int it = 0; it < n_iters; ++it) {
    PrepArgs pa;
    glj_fill_prep(ctx, 1, st->state + st->lay.o_res(), nullptr, pa);
    EntPlan plan;
    int rc = entmc_plan(ctx, st->ns, use_gen ? VBMC_EPS_RESIDENT : st->eps_mode, st->seed + (uint64_t)(i0 + it),
                        st->row_begin, st->row_count, 1, plan, try_tail ? pa.n_glj : 0, true, try_tail ? tail_gp_per_slot : 0);
    if (rc) return rc;
    // ---- two launches per iteration (adam_tail_kernel): the wave-split kernel in span mode with the GP sums and the
    // pre workgroup in its free slots, then everything up to the next table in one launch ----
    if (try_tail && plan.ws && plan.a.sp.cus > 0 && plan.gp_in_ws && pa.n_glj > 0 &&
        (use_gen || st->eps_mode == VBMC_EPS_RESIDENT)) {
      if (use_gen) {
        plan.a.eps = st->d_eps1;
        plan.a.eps_rows = st->row_count;
      }
      plan.table = st->d_table;
      if (!st->table_valid || (use_gen && st->eps_have < 1.0)) {
        // first iteration of a run (or after an iteration in another form): the table of the current iterate and the
        // part of its draws that is not there yet -- everything at the start of a run, [f_step, 1) behind a four-launch
        // iteration (whose finish and step launches made [0, f_step): ADVICE r05) --, by a prep launch of their own
        PrepArgs pt;
        entmc_fill_prep(ctx, plan, pt);
        if (st->table_valid) pt.n_table = 0;
        if (use_gen && st->eps_have < 1.0) pt.gen = slice(it, st->eps_have, 1.0);
        rc = launch_prep(ctx, pt);
        if (rc) return rc;
      }
      const unsigned long long seq = ++st->sync_seq;
      PrepArgs gp = pa;
      gp.n_table = 0;
      gp.gen = GenSlice();
      gp.mix = ctx->d_mix;
      gp.done = DoneSignal();
      gp.done.cnt = (int*)(st->d_sync + 8);
      gp.done.flag = (uint64_t*)st->d_sync;
      gp.done.seq = seq;
      gp.done.dev = 1;
      plan.a.gp = gp;
      plan.a.gp_items = gp.n_glj;
      plan.a.extra = st->d_args;
      plan.a.extra_lds = (st->pre_lds && st->pre_lds_bytes <= pre_lds_limit(ctx, plan)) ? (int)(st->pre_lds_bytes / sizeof(double)) : 0;
      rc = entmc_launch_main(ctx, plan);
      if (rc) return rc;
      TailArgs t;
      t.partial = plan.a.partial;
      t.chunks = plan.a.chunks;
      t.stride = plan.a.stride;
      t.mu_from_w = 1;
      t.inv_ns = plan.inv_ns;
      t.red.cnt = (int*)(st->d_sync + 8) + 2;
      t.red.flag = (uint64_t*)(st->d_sync + 1);
      t.red.seq = seq;
      t.red.dev = 1;
      t.DP = plan.DP;
      t.K4 = ws_table_rows(ctx->K);
      t.table = st->d_table;
      t.rd_cnt = st->d_sync + 2;
      t.rd_target = (unsigned long long)ctx->K * (++st->tail_launches);
      static const int gen_mode = [] {
        const char* e = getenv("VBMC_TAIL_GEN");  // measurement aid: 0 = no draws (timing only: wrong values), 2 = a launch of their own behind the tail launch
        return e ? atoi(e) : 1;
      }();
      if (use_gen && gen_mode == 1) t.gen = slice(it + 1, 0.0, 1.0);
      a.it_off = it;
      static const bool want_times = [] {
        const char* e = getenv("VBMC_TAIL_TIMES");  // measurement aid: the writer block's stamps of the batch's last iteration to stderr
        return e && e[0] == '1';
      }();
      static unsigned long long* d_tt = nullptr;
      if (want_times) {
        if (!d_tt) HIP_TRY(ctx, hipMalloc((void**)&d_tt, 16 * sizeof(unsigned long long)));
        t.times = d_tt;
      }
      const int n_main = (raw_len(ctx->D, ctx->K) + 3) / 4;
      hipLaunchKernelGGL(adam_tail_kernel, dim3(n_main + ctx->K + t.gen.n_blocks), dim3(256), st->tail_lds_bytes, sm, a, t);
      if (use_gen && gen_mode == 2) {
        rc = launch_eps_gen(ctx, sm, slice(it + 1, 0.0, 1.0));
        if (rc) return rc;
      }
      if (want_times && it == n_iters - 1) {
        unsigned long long tt[8];
        HIP_TRY(ctx, hipStreamSynchronize(sm));
        HIP_TRY(ctx, hipMemcpy(tt, d_tt, sizeof(tt), hipMemcpyDeviceToHost));
        fprintf(stderr, "tail launch, writer block, us: (%.0f) start -> word %.2f | loads %.2f | update %.2f | pack %.2f | table row %.2f | pack out + readers %.2f | write-back %.2f | total %.2f\n",
                0.0, (tt[1] - tt[0]) * 0.01, (tt[2] - tt[1]) * 0.01, (tt[3] - tt[2]) * 0.01, (tt[4] - tt[3]) * 0.01, (tt[5] - tt[4]) * 0.01,
                (tt[6] - tt[5]) * 0.01, (tt[7] - tt[6]) * 0.01, (tt[7] - tt[0]) * 0.01);
      }
      st->table_valid = true;
      st->eps_have = (use_gen && gen_mode != 0) ? 1.0 : 0.0;
      st->last_form = 2;
      ctx->last_plan[0] = 6;  // vbmc_last_entmc_plan: the wave-split kernel in span mode inside the two-launch iteration
      continue;
    }
    st->table_valid = false;
    st->last_form = 4;
    if (use_gen) {
      plan.a.eps = st->d_eps1;
      plan.a.eps_rows = st->row_count;
      // this iteration's remaining draws (all of them when no earlier launch began the buffer, none behind a two-launch
      // iteration, whose tail launch made them all)
      if (st->eps_have < 1.0) pa.gen = slice(it, st->eps_have, 1.0);
      st->eps_have = f_step;  // (what this iteration's finish and step launches below put in for the next one)
    }
    entmc_fill_prep(ctx, plan, pa);
    rc = launch_prep(ctx, pa);
    if (rc) return rc;
    // the entropy-free part of dF: an extra row of the entropy launch when that kernel has one
    // (wave-split, draws from memory), otherwise a launch of its own in front of it
    const bool pre_row = plan.ws && plan.a.eps_mode != VBMC_EPS_PHILOX;
    if (pre_row) {
      plan.a.extra = st->d_args;
      // its working set in LDS when the launch's entropy workgroups still fit beside it (pre_lds_limit)
      plan.a.extra_lds = (st->pre_lds && st->pre_lds_bytes <= pre_lds_limit(ctx, plan)) ? (int)(st->pre_lds_bytes / sizeof(double)) : 0;
    } else {
      launch_pre(*st, sm, a);
    }
    rc = entmc_launch_main(ctx, plan);
    if (rc) return rc;
    const GenSlice g_fin = use_gen ? slice(it + 1, 0.0, f_fin) : GenSlice();
    rc = entmc_launch_finish(ctx, plan, st->state + st->lay.o_raw(), &g_fin);
    if (rc) return rc;
    if (multi) {
      rc = comm_allreduce_sum(ctx, st->state + st->lay.o_raw(), raw_len(ctx->D, ctx->K));
      if (rc) return rc;
    }
    a.it_off = it;
    launch_step(*st, sm, a, 1, use_gen ? slice(it + 1, f_fin, f_step) : GenSlice());
  }
  HIP_TRY(ctx, hipGetLastError());
  return 0;
}

extern "C" int vbmc_adam_run(vbmc_ctx* ctx, int n_iters, double* y_tab_out, double* x_tab_out,
                             double* G_out, double* H_out) {
  if (!ctx || n_iters < 0) return VBMC_E_ARG;
  NEED_DEVICE(ctx);
  AdamState* st = (AdamState*)ctx->adam;
  if (!st || !st->active) return vbmc_fail(ctx, VBMC_E_ARG, "adam_run: vbmc_adam_begin not called");
  if (st->iter + n_iters > st->max_iter)
    return vbmc_fail(ctx, VBMC_E_ARG, "adam_run: %d + %d iterations exceed max_iter %d", st->iter,
                     n_iters, st->max_iter);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int n = st->n_theta;
  static const bool force_coll = [] {
    const char* e = getenv("VBMC_FORCE_COLLECTIVE");
    return e && e[0] == '1';
  }();
  const bool multi = ctx->comm != nullptr && (ctx->world > 1 || force_coll);
  struct TimingOff {  // no per-kernel event pairs inside the loop; restored on every exit path
    vbmc_ctx* c;
    bool was;
    explicit TimingOff(vbmc_ctx* c_) : c(c_), was(c_->timing) { c->timing = false; }
    ~TimingOff() { c->timing = was; }
  } timing_off(ctx);
  const int i0 = st->iter;
  int rc = 0;
  int status = 0;
  std::vector<double> y3(3 * (size_t)n_iters);
  // (second pass only when the one-launch form gave up waiting: the state in memory is then still that of the start of
  // the batch -- workgroup 0 writes it back at the very end -- so the batch is simply run again as four launches per
  // iteration, and the rest of the optimisation too)
  for (int attempt = 0; attempt < 2; ++attempt) {
  if (st->fused && !multi && n_iters > 0) {
    FusedArgs f = st->fz;
    f.i0 = i0;
    f.n_iters = n_iters;
    f.test_absent = ctx->opt_adam_fused == 2 ? 1 : 0;
    f.rel_acq = ctx->opt_adam_fused == 3 ? 1 : 0;
    HIP_TRY(ctx, hipMemsetAsync(st->d_flags, 0, 512 * sizeof(unsigned long long), ctx->stream));
    static const bool want_times = [] {
      const char* e = getenv("VBMC_FUSED_TIMES");  // measurement aid: phase stamps of two workgroups to stderr
      return e && e[0] == '1';
    }();
    unsigned long long* d_times = nullptr;
    if (want_times) {
      HIP_TRY(ctx, hipMalloc((void**)&d_times, sizeof(unsigned long long) * 2 * 64 * 16));
      HIP_TRY(ctx, hipMemsetAsync(d_times, 0, sizeof(unsigned long long) * 2 * 64 * 16, ctx->stream));
      f.times = d_times;
    }
    rc = adam_fused_launch(ctx, ctx->stream, f, st->fused_lds);
    ctx->last_plan[0] = 4;  // vbmc_last_plan: the fused loop
    // (the iterate moves without the other forms' side buffers following it: their table and any draws made ahead are stale)
    st->table_valid = false;
    st->eps_have = 0.0;
    if (want_times && !rc) {
      std::vector<unsigned long long> tt(2 * 64 * 16);
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      HIP_TRY(ctx, hipMemcpy(tt.data(), d_times, sizeof(unsigned long long) * tt.size(), hipMemcpyDeviceToHost));
      (void)hipFree(d_times);
      const int nt = n_iters < 64 ? n_iters : 64;
      for (int w = 0; w < 2; ++w) {
        double acc[9] = {}, sub[4] = {};
        int cnt = 0;
        for (int t = 2; t + 1 < nt; ++t, ++cnt) {
          const unsigned long long* r = &tt[((size_t)w * 64 + t) * 16];
          for (int p = 0; p < 8; ++p) acc[p] += (double)(r[p + 1] - r[p]) * 0.01;
          acc[8] += (double)(tt[((size_t)w * 64 + t + 1) * 16] - r[0]) * 0.01;
          if (w == 0) { sub[0] += (double)(r[9] - r[0]) * 0.01; sub[1] += (double)(r[10] - r[9]) * 0.01; sub[2] += (double)(r[11] - r[10]) * 0.01; sub[3] += (double)(r[1] - r[11]) * 0.01; }
        }
        if (cnt > 0)
          fprintf(stderr, "fused loop, %s workgroup, us: phase A %.2f | drain %.2f | wait %.2f | gather %.2f | raw %.2f | pre %.2f | step %.2f | pack %.2f | iteration %.2f\n",
                  w == 0 ? "entropy" : "GP-sum", acc[0] / cnt, acc[1] / cnt, acc[2] / cnt, acc[3] / cnt, acc[4] / cnt, acc[5] / cnt,
                  acc[6] / cnt, acc[7] / cnt, acc[8] / cnt);
        if (cnt > 0 && w == 0)
          fprintf(stderr, "   phase A of the entropy workgroup: draws + table row %.2f | row loop %.2f | wave sums %.2f | record %.2f\n",
                  sub[0] / cnt, sub[1] / cnt, sub[2] / cnt, sub[3] / cnt);
      }
    }
  } else {
    hipLaunchKernelGGL(set_int_kernel, dim3(1), dim3(1), 0, ctx->stream, st->d_status + 1, i0);
    rc = enqueue_batch(ctx, st, i0, n_iters, multi);
  }
  if (rc) return rc;
  HIP_TRY(ctx, hipGetLastError());
  if (n_iters > 0) {
    HIP_TRY(ctx, hipMemcpyAsync(y3.data(), st->y_tab + 3 * (size_t)i0, sizeof(double) * 3 * n_iters,
                                hipMemcpyDeviceToHost, ctx->stream));
    if (x_tab_out)
      HIP_TRY(ctx, hipMemcpyAsync(x_tab_out, st->x_tab + (size_t)i0 * n, sizeof(double) * (size_t)n_iters * n,
                                  hipMemcpyDeviceToHost, ctx->stream));
  }
  HIP_TRY(ctx, hipMemcpyAsync(&status, st->d_status, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, stream_wait(ctx));
  if (!(status & 4) || !st->fused) break;
  st->fused = false;
  st->fused_gave_up++;
  rc = fused_restore(ctx, st);
  if (rc) return rc;
  HIP_TRY(ctx, hipMemsetAsync(st->d_status, 0, sizeof(int), ctx->stream));
  }
  st->iter = i0 + n_iters;
  for (int it = 0; it < n_iters; ++it) {
    if (y_tab_out) y_tab_out[it] = y3[3 * (size_t)it];
    if (G_out) G_out[it] = y3[3 * (size_t)it + 1];
    if (H_out) H_out[it] = y3[3 * (size_t)it + 2];
  }
  if (status & 4) {  // (cannot happen: the second pass does not wait for anybody)
    st->active = false;
    return vbmc_fail(ctx, VBMC_E_HIP, "adam_run: a workgroup of the fused loop did not publish within 20 ms");
  }
  if (status & 8) {
    st->active = false;
    return vbmc_fail(ctx, VBMC_E_HIP, "adam_run: a hand-off inside the two-launch iteration did not arrive within 20 ms");
  }
  if (status) {
    st->active = false;
    return vbmc_fail(ctx, VBMC_E_NONFINITE, "adam_run: an iterate became non-finite");
  }
  return VBMC_OK;
}

// The whole optimisation in one call where the one-launch form applies: the workgroups apply the reference's stopping
// rule themselves every 20 iterations (minimize_adam.py:107-140), so the host is not in the loop at all.
extern "C" int vbmc_adam_run_auto(vbmc_ctx* ctx, int max_iters, double tol_fun, int* n_done, double* y_tab_out,
                                  double* x_tab_out, double* G_out, double* H_out) {
  if (!ctx || max_iters < 0 || !n_done) return VBMC_E_ARG;
  NEED_DEVICE(ctx);
  AdamState* st = (AdamState*)ctx->adam;
  if (!st || !st->active) return vbmc_fail(ctx, VBMC_E_ARG, "adam_run_auto: vbmc_adam_begin not called");
  if (st->iter + max_iters > st->max_iter)
    return vbmc_fail(ctx, VBMC_E_ARG, "adam_run_auto: %d + %d iterations exceed max_iter %d", st->iter, max_iters, st->max_iter);
  static const bool force_coll = [] {
    const char* e = getenv("VBMC_FORCE_COLLECTIVE");
    return e && e[0] == '1';
  }();
  const bool multi = ctx->comm != nullptr && (ctx->world > 1 || force_coll);
  // the kernel's stopping rule compares the mean iterate of a batch with the previous batch's, which it accumulates
  // itself from iteration 0 on: a run resumed after vbmc_adam_run iterations has no previous-batch mean to compare with
  if (!st->fused || multi || max_iters == 0 || st->iter != 0) return VBMC_W_NOT_FUSED;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int n = st->n_theta, i0 = st->iter;
  FusedArgs f = st->fz;
  f.i0 = i0;
  f.n_iters = max_iters;
  f.stop_rule = 1;
  f.tol_fun = tol_fun;
  f.n_done = st->d_status + 2;
  f.test_absent = ctx->opt_adam_fused == 2 ? 1 : 0;
  f.rel_acq = ctx->opt_adam_fused == 3 ? 1 : 0;
  HIP_TRY(ctx, hipMemsetAsync(st->d_flags, 0, 512 * sizeof(unsigned long long), ctx->stream));
  int rc = adam_fused_launch(ctx, ctx->stream, f, st->fused_lds);
  if (rc) return rc;
  ctx->last_plan[0] = 4;
  int word[3] = {0, 0, 0};
  HIP_TRY(ctx, hipMemcpyAsync(word, st->d_status, 3 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, stream_wait(ctx));
  if (word[0] & 4) {  // gave up waiting: nothing was written back, the caller runs its batches (four launches per iteration)
    st->fused = false;
    st->fused_gave_up++;
    rc = fused_restore(ctx, st);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemsetAsync(st->d_status, 0, sizeof(int), ctx->stream));
    return VBMC_W_NOT_FUSED;
  }
  const int done = word[2];
  if (done < 1 || done > max_iters) return vbmc_fail(ctx, VBMC_E_HIP, "adam_run_auto: bad iteration count %d", done);
  std::vector<double> y3(3 * (size_t)done);
  HIP_TRY(ctx, hipMemcpyAsync(y3.data(), st->y_tab + 3 * (size_t)i0, sizeof(double) * 3 * done, hipMemcpyDeviceToHost, ctx->stream));
  if (x_tab_out)
    HIP_TRY(ctx, hipMemcpyAsync(x_tab_out, st->x_tab + (size_t)i0 * n, sizeof(double) * (size_t)done * n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, stream_wait(ctx));
  st->iter = i0 + done;
  *n_done = done;
  for (int it = 0; it < done; ++it) {
    if (y_tab_out) y_tab_out[it] = y3[3 * (size_t)it];
    if (G_out) G_out[it] = y3[3 * (size_t)it + 1];
    if (H_out) H_out[it] = y3[3 * (size_t)it + 2];
  }
  if (word[0]) {
    st->active = false;
    return vbmc_fail(ctx, VBMC_E_NONFINITE, "adam_run_auto: an iterate became non-finite");
  }
  return VBMC_OK;
}

extern "C" int vbmc_adam_end(vbmc_ctx* ctx, double* theta_out, double* mu_KxD, double* sigma_K,
                             double* lambd_D, double* w_K, double* eta_K, int* iterations) {
  if (!ctx) return VBMC_E_ARG;
  NEED_DEVICE(ctx);
  AdamState* st = (AdamState*)ctx->adam;
  if (!st || !st->active) return vbmc_fail(ctx, VBMC_E_ARG, "adam_end: no optimisation in progress");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int D = ctx->D, K = ctx->K;
  const size_t n_aux = aux_len(D, K);
  std::vector<double> aux(n_aux), th(st->n_theta);
  HIP_TRY(ctx, hipMemcpyAsync(aux.data(), st->state + st->lay.o_aux(), sizeof(double) * n_aux, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(th.data(), st->state + st->lay.o_theta(), sizeof(double) * st->n_theta, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, stream_wait(ctx));
  // the device pack in d_mix is the mixture of the last iterate: make the host copies agree
  ctx->mu.assign(aux.begin(), aux.begin() + K * D);
  ctx->sigma.assign(aux.begin() + K * D, aux.begin() + K * D + K);
  ctx->lambd.assign(aux.begin() + K * D + K, aux.begin() + K * D + K + D);
  ctx->w.assign(aux.begin() + K * D + K + D, aux.begin() + K * D + 2 * K + D);
  ctx->eta.assign(aux.begin() + K * D + 2 * K + D, aux.end());
  ctx->exp_eta_valid = false;
  ctx->pack_valid = true;
  if (theta_out) memcpy(theta_out, th.data(), sizeof(double) * st->n_theta);
  if (mu_KxD) memcpy(mu_KxD, ctx->mu.data(), sizeof(double) * K * D);
  if (sigma_K) memcpy(sigma_K, ctx->sigma.data(), sizeof(double) * K);
  if (lambd_D) memcpy(lambd_D, ctx->lambd.data(), sizeof(double) * D);
  if (w_K) memcpy(w_K, ctx->w.data(), sizeof(double) * K);
  if (eta_K) memcpy(eta_K, ctx->eta.data(), sizeof(double) * K);
  if (iterations) *iterations = st->iter;
  st->active = false;
  return VBMC_OK;
}
