// Internal declarations shared by the translation units of libvbmc_hip.so.
// Not part of the ABI (that is include/vbmc_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <chrono>

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/vbmc_hip.h"

struct ncclComm;

// Device-resident copy of the mixture, laid out for the kernels.
// One contiguous allocation of doubles; offsets below in units of double.
struct MixLayout {
  int D = 0, K = 0;
  int o_mup = 0;    // [K][D]  mu_dk / lambda_d   (means in lambda-scaled coordinates)
  int o_mu = 0;     // [K][D]  mu_dk
  int o_is2 = 0;    // [K]     1 / sigma_k^2
  int o_wc = 0;     // [K]     w_k * nconst / sigma_k^D   (nconst = (2pi)^(-D/2) / prod lambda)
  int o_rc = 0;     // [K]     nconst / sigma_k^D
  int o_lrc = 0;    // [K]     log2(nconst / sigma_k^D)
  int o_sig = 0;    // [K]     sigma_k
  int o_w = 0;      // [K]     w_k
  int o_lam = 0;    // [D]     lambda_d
  int o_ilam = 0;   // [D]     1 / lambda_d
  int total = 0;
  void plan(int D_, int K_) {
    D = D_;
    K = K_;
    int o = 0;
    o_mup = o; o += K * D;
    o_mu = o; o += K * D;
    o_is2 = o; o += K;
    o_wc = o; o += K;
    o_rc = o; o += K;
    o_lrc = o; o += K;
    o_sig = o; o += K;
    o_w = o; o += K;
    o_lam = o; o += D;
    o_ilam = o; o += D;
    total = o;
  }
};

struct GpState {
  bool set = false;
  int N = 0, D = 0, S = 0, P = 0, mean_kind = 0;
  std::vector<double> hyp;       // S x P (host copy)
  std::vector<int32_t> L_chol;   // S
  std::vector<double> sn2_eff;   // S  (1 / sW[0]^2)
  std::vector<double> sn2_mult;  // S
  double* d_X = nullptr;      // N x D
  double* d_XT = nullptr;     // D x N: the same, transposed (glj_block.h reads a dimension of consecutive points)
  double* d_alpha = nullptr;  // S x N
  double* d_L = nullptr;      // S x N x N
  double* d_Linv = nullptr;   // S x N x N : inverse of the upper Cholesky factor (L_chol samples)
  double* d_LinvP = nullptr;  // S x ld x ld, ld = predict_ld(N): the same, zero padded (predict_var_dma_kernel)
  double* d_sW = nullptr;     // S x N
  double* d_hyp = nullptr;    // S x P
  double* d_xc = nullptr;     // D : column means of X (centre of the pairwise-distance expansion)
  double* d_smeta = nullptr;  // S x 3 : (L_chol as 0/1, sn2_mult, 1/sn2_eff) per sample, for kernels batched over s
  size_t cap_X = 0, cap_XT = 0, cap_alpha = 0, cap_L = 0, cap_Linv = 0, cap_LinvP = 0, cap_sW = 0, cap_hyp = 0, cap_xc = 0, cap_smeta = 0;
  std::vector<double> h_small;  // host source of the xc / smeta uploads
  std::vector<double> h_XT;     // host source of the d_XT upload
};

// per-sample results of the GP expected log joint on the host (api_gp.hip glj_finalize)
struct GljHost {
  std::vector<double> G;        // S
  std::vector<double> I_sk;     // S x K
  std::vector<double> mu;       // S x (K*D)   d/dmu, 'F' order (K blocks of D)
  std::vector<double> sigma;    // S x K       pre-Jacobian
  std::vector<double> lambd;    // S x D
  std::vector<double> w;        // S x K  (= I_sk)
};

// host scratch of one fused evaluation, kept in the context so that the hot call allocates nothing
struct ElboScratch {
  GljHost glj;
  std::vector<double> dG, dH, dF, dFb, mu, sg, lm, wg, ext, ln_sigma, ln_lambd, dL, wpen, jw;
};

struct vbmc_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev[12] = {};  // pairs: (0,1) entmc, (2,3) glj, (4,5) pdf, (6,7) predict, (8,9) elbo, (10,11) predict's variance product
  bool ev_valid[6] = {false, false, false, false, false, false};
  std::string err;
  hipDeviceProp_t prop;

  // mixture (host copies are authoritative for finalisation arithmetic)
  bool mix_set = false;
  int D = 0, K = 0;
  std::vector<double> mu, sigma, lambd, w, eta;  // mu is K x D
  MixLayout ml;
  double* d_mix = nullptr;
  // Armed evaluation (api_elbo.hip): the three launches of the NEXT host-driven evaluation are queued
  // while the current one runs; their prep kernel waits on a control word in host-written device
  // memory, so when theta arrives the host only writes the pack and that word (no launch latency).
  struct ArmedEval {
    bool armed = false;
    bool keep = false;        // set while vbmc_neg_elcbo itself runs (its inner calls must not disarm)
    uint64_t seq = 0;         // completion sequence number the armed launches will publish
    uint64_t seed = 0;        // the Philox seed they were planned for
    int n_theta = 0, mask = 0, grad_flags = 0, eps_mode = 0;
    int64_t ns_per_comp = 0, row_begin = 0, row_count = 0;
    // state to restore when the armed launches are cancelled instead of run
    int gen_cur_before = 0;
    bool ahead_before_valid = false;
    uint64_t ahead_before_seed = 0;
    int ahead_before_buf = 0;
    double ahead_before_frac = 1.0;
    uint64_t hits = 0, cancels = 0, late = 0, ident_checked = 0, ident_bad = 0, lost = 0;  // see vbmc_armed_stats
    bool ident = false;       // its launches carry an identity (DoneSignal)
    double limit_ms = 1.0;  // use it within this time of arming (>= 1 ms, 2.5 x the last evaluation's duration); the device waits twice as long
    std::chrono::steady_clock::time_point t_armed;  // the device gives up after 2 ms: the host does not use an armed evaluation older than 1 ms
  } spec;
  uint64_t* d_ctl = nullptr;   // fine-grained device memory, 8 words: go / cancel word of evaluation seq is [seq & 7]
  double* d_stage = nullptr;   // device staging of the results the polled step hands to the host (DoneSignal)
  size_t d_stage_cap = 0;
  double* d_mix_fg = nullptr;  // host-writable (fine-grained) device memory: the host-driven step writes the pack here itself
  std::vector<double> mu_scratch;
  std::vector<double> t2m_mu, t2m_sg, t2m_lm, t2m_w, t2m_eta;  // vbmc_theta_to_mixture's working copies  // vbmc_set_mixture_dk's transposed means
  double* d_acq_fg = nullptr;  // the same kind of memory for a small acquisition batch's points (api_acq.hip); 256 x 33 doubles
  bool acq_fg_failed = false;
  uint64_t acq_seq = 0;        // sequence number of the acquisition completion word (h_done[7])
  size_t d_mix_fg_cap = 0;
  bool mix_fg_failed = false;
  size_t d_mix_cap = 0;

  // Philox draws generated ahead of the entropy kernel (fused objective): two buffers, so the
  // draws of evaluation i+1 can be generated behind evaluation i's finish kernel while the host
  // is busy with result i (api_elbo.hip); `ahead` says what the speculative buffer holds
  double* d_epsgen[2] = {nullptr, nullptr};
  size_t d_epsgen_cap[2] = {0, 0};
  int gen_cur = 0;  // buffer the current evaluation reads
  struct AheadDraws {
    bool valid = false;
    uint64_t seed = 0;
    int K = 0, D = 0, buf = 0;
    int64_t rows = 0, n_half = 0, row_begin = 0;
    double frac = 1.0;  // the part [0, frac) of the items is (being) generated; the consumer's prep launch adds the rest
  } ahead;
  // completion word of the fused objective: the finish kernel's last result wave stores the
  // evaluation's sequence number into pinned host memory and the host polls it -- the spare
  // workgroups of the same launch go on generating the next evaluation's draws meanwhile
  uint64_t* h_done = nullptr;   // pinned, device-visible
  uint64_t* hd_done = nullptr;  // its device-side address
  int* d_done_cnt = nullptr;    // result waves finished so far (reset by the last one)
  int* d_done_sub = nullptr;    // 16 sub-counters of the finish launch, 64 ints apart (DoneSignal::sub)
  uint64_t done_seq = 0;

  // resident antithetic half draws: [K][eps_rows][D]
  double* d_eps = nullptr;
  size_t d_eps_cap = 0;
  int eps_K = 0, eps_D = 0;
  int64_t eps_rows = 0, eps_row_begin = 0, eps_n_half = 0;

  // scratch (grown on demand)
  double* d_scratch = nullptr;
  size_t d_scratch_cap = 0;
  double* d_out = nullptr;  // small result vectors
  size_t d_out_cap = 0;
  double* d_ptick = nullptr;  // predict: one arrival ticket (int) per 64-point row tile and GP sample (gp.hip, the fused finish)
  size_t d_ptick_cap = 0;
  double* h_pinned = nullptr;  // pinned host staging for results (also written directly by kernels)
  double* h_eps = nullptr;     // pinned host buffer the reference-stream draws are generated into (vbmc_set_eps_numpy)
  size_t h_eps_cap = 0;
  size_t h_pinned_cap = 0;
  double* h_pack = nullptr;    // pinned source of the mixture pack upload
  size_t h_pack_cap = 0;
  bool pack_in_flight = false;   // a mixture-pack upload was queued and the stream not waited for since
  bool pack_valid = false;     // d_mix holds the pack of the host copies (mu, sigma, lambd, w)
  bool defer_mix_upload = false;  // set_mixture_host packs but leaves the copy to upload_packed_mixture
  double* hp_dev = nullptr;    // device-side address of h_pinned (cached: the query is an API call)
  int timing = 0;              // 1: record the HIP event pair around the dominant kernel (vbmc_set_timing); 2: also the pair INSIDE predict
  double host_us[5] = {0, 0, 0, 0, 0};  // see vbmc_last_host_us
  double step_marks[4] = {0, 0, 0, 0};  // see vbmc_last_step_marks
  int gp_where = 0;  // where the GP sums of the launches issued last run: 0 prep launch, 1 finish launch, 2 entropy launch
  // vbmc_set_option switches (defaults from the environment at context creation)
  int opt_entmc_valu = 0;   // 1: always the generic entropy kernel
  int opt_gp_ship = 1;      // host-driven step: the prep launch's GP sums are shipped to the host by a workgroup of the matrix-pipe entropy launch (EntArgs::ship_*)
  int opt_entmc_mfma = 1;   // the matrix-pipe form of the entropy kernel where its shape applies (entropy_mfma.hip)
  int opt_elbo_pregen = 1;  // Philox draws generated ahead of the entropy kernel
  int opt_elbo_ahead = 1;   // ... and those of seed+1 speculatively behind the finish kernel
  int opt_predict_dma = 1;  // predict's variance product through the LDS-direct kernel (batches on Cholesky samples)
  int opt_predict_fused = 1;  // ... with predict's finish in its epilogue: 0 never, 1 for one-round product grids, 2 always (gp.hip)
  int opt_arm_late_test = 0;  // test hook: n > 0 = the n-th use of an armed evaluation from now takes the late-go recovery path
  int opt_acq_poll = 1;       // small acquisition batches: points written by the CPU, results polled (api_acq.hip)
  void (*release_cb)(void*) = nullptr;  // vbmc_set_release_callback
  void* release_cb_user = nullptr;
  int opt_adam_fused = 1;     // the optimiser loop as one launch per batch where its shape applies (adam_fused.hip)
  int opt_adam_tail = 1;      // the optimiser loop at large sample counts as two launches per iteration (adam.hip adam_tail_kernel)
  int opt_ident_test = 0;     // test hook: n > 0 = the n-th identity check from now fails (the recovery path runs)
  bool ident_retry = false;   // inside the re-evaluation after a failed identity check
  int opt_elbo_arm = 1;     // queue the next host-driven evaluation's launches ahead of its theta (armed evaluation): 0 never, 1 while
                            // this context is the only one on its device in the process, 2 always
  bool counted = false;     // this context is in the per-device count of live contexts (ctx.hip)
  int opt_ws_span = 1;      // entropy kernel: span mode (entropy_args.h WsSpan: front / filler parts sized to end together); 0 = equal chunks
  int opt_ws_front = 0;     // span mode: the front workgroup's share of a CU's batches, per mille (0 = the built-in value)
  int opt_ws_pad = -1;      // span mode: padding slots behind every component (-1 = the built-in value)
  int opt_mix_bar = 1;      // host-driven step: pack written by the CPU into device memory (no upload launch), GP sums in the finish launch
  double* h_pack_dev = nullptr;     // device-side address of h_pack ...
  double* h_pack_dev_of = nullptr;  // ... valid for this h_pack
  // exp(eta) and its sum, shared by the three softmax Jacobians of an evaluation
  std::vector<double> exp_eta;
  double exp_eta_sum = 0.0;
  bool exp_eta_valid = false;
  int last_plan[4] = {-1, 0, 0, 0};  // see vbmc_last_entmc_plan
  size_t last_raw_off = 0;           // see vbmc_last_elbo_raw: offset into h_pinned, length (0: none)
  int last_raw_n = 0;

  GpState gp;

  ncclComm* comm = nullptr;
  int rank = 0, world = 1;

  // host arrays the device GP came from, re-checksummed by vbmc_neg_elcbo while it waits (vbmc_set_gp_watch)
  std::vector<const double*> gp_watch_ptrs;
  std::vector<int64_t> gp_watch_lens;
  uint64_t gp_watch_ck = 0;

  ElboScratch elbo;      // api_elbo.hip
  void* adam = nullptr;  // device-resident optimiser state (adam.hip)
  void* acq_is = nullptr;  // resident importance-sampling state of AcqFcnVIQR / IMIQR (api_acq_is.hip)
  bool host_bound = false;    // vbmc_ctx_create narrowed the calling thread's affinity to the device's NUMA node (ctx.hip)
  int host_cpus = 0;          // CPUs in that thread's affinity set after context creation (0: not looked at)
  void* randn_dev = nullptr;  // buffers of the device-side NumPy stream (device_randn.hip)
  int randn_last_reused = 0;  // the last device pass found its window in the pass before (device_randn.hip)
  int opt_randn_dev = 1;      // vbmc_set_eps_numpy: the reference's stream generated on the device (0: on the host cores + PCIe)
};
int vbmc_live_contexts_on(int device);  // ctx.hip
void adam_free(vbmc_ctx* ctx);
void acq_is_free(vbmc_ctx* ctx);
void randn_dev_free(vbmc_ctx* ctx);
// device_randn.hip: the next n values of NumPy's legacy randn stream into d_out (device memory), state advanced as
// vbmc_mt19937_randn does; VBMC_W_NOT_FUSED = not this path's size (the caller uses the host generator)
int randn_device(vbmc_ctx* ctx, uint32_t* key, int* pos, int* has_gauss, double* gauss, double* d_out, int64_t n);

// error helpers -----------------------------------------------------------
int vbmc_fail(vbmc_ctx* ctx, int code, const char* fmt, ...);
extern thread_local std::string g_create_err;

#define HIP_TRY(ctx, call)                                                            \
  do {                                                                                \
    hipError_t e_ = (call);                                                           \
    if (e_ != hipSuccess)                                                             \
      return vbmc_fail((ctx), VBMC_E_HIP, "%s failed: %s (%s:%d)", #call,             \
                       hipGetErrorString(e_), __FILE__, __LINE__);                    \
  } while (0)

// Entry points that launch kernels refuse a host-only context (device_id -1).
// cancel an armed evaluation (api_elbo.hip): its queued kernels return at once, host state is restored
void spec_disarm(vbmc_ctx* ctx);
#define NEED_DEVICE(ctx)                                                                     \
  do {                                                                                       \
    if ((ctx)->device < 0)                                                                   \
      return vbmc_fail((ctx), VBMC_E_NODEV,                                                  \
                       "host-only context: this entry point needs a gfx950 device "          \
                       "(libvbmc_hip has no CPU fallback)");                                 \
    if ((ctx)->spec.armed && !(ctx)->spec.keep) spec_disarm(ctx);                            \
  } while (0)

void write_mixture_pack(const MixLayout& ml, const double* mu_KxD, const double* sigma,
                        const double* lambd, const double* w, double* p);
int theta_to_arrays(int D, int K, const double* theta, int n_theta, int optimize_mask, double* mu,
                    double* sg, double* lm, double* w, double* eta);
int upload_packed_mixture(vbmc_ctx* ctx);
double* write_pack_to_device(vbmc_ctx* ctx);
int set_mixture_host(vbmc_ctx* ctx, int D, int K, const double* mu_KxD, const double* sigma_K,
                     const double* lambd_D, const double* w_K, const double* eta_K, bool skip_if_same);

// buffer helpers (ctx.hip) --------------------------------------------------
int ensure_dev(vbmc_ctx* ctx, double** p, size_t* cap, size_t n_doubles);
int ensure_pinned(vbmc_ctx* ctx, size_t n_doubles);
// csrc/host_randn.hip: vbmc_mt19937_randn with progress(user, m) = "out[0 .. m) is final", called from
// the calling thread while the worker threads still write the rest
int randn_with_progress(uint32_t* key, int* pos, int* has_gauss, double* gauss, double* out, int64_t n, int n_threads,
                        void (*progress)(void*, int64_t), void* user);

// raw-vector length of the entropy accumulator
static inline int raw_len(int D, int K) { return 1 + D * K + 2 * K + D; }

// arguments of the prep launch (prep.hip): table rows for the wave-split entropy kernel
// and/or the GP expected-log-joint sums
// A slice of the draw buffer eps[K][rows][D] for other kernels' spare workgroups to fill: the
// (row, block) items [item_begin, item_begin + item_count) of the K * rows * ceil(D/4) items -- one
// Philox block = up to four normals of a row (philox.h) -- 256 per workgroup.  The values
// depend on (seed, row, block) only, so any kernel may generate any slice.
// completion signalling of entmc_finish_kernel (all null/0: none)
struct DoneSignal {
  int* cnt = nullptr;        // device counter of finished result waves
  // optional, entmc_finish_kernel: sixteen sub-counters, 256 B apart.  One word takes ~88 atomic increments per us, and at
  // K = 100, D = 20 the reduction has 556 workgroups that arrive together: 6 us of the 14 between the entropy kernel and the
  // completion word.  From 256 workgroups on, workgroup b counts on sub-counter b % 16 and the last of each on `cnt`.
  int* sub = nullptr;
  int sub_min = 256;         // workgroups from which the sub-counters are used
  uint64_t* flag = nullptr;  // device-visible pinned word that receives `seq` when all are done
  uint64_t seq = 0;
  // Staged hand-over (optional): the kernel's result pointer is then DEVICE memory, and the last
  // workgroup to count copies host_n doubles from there to host_out (pinned) in coalesced stores
  // before it publishes.  One 8-byte store per result straight to pinned memory is one PCIe write
  // each -- ~1700 of them per evaluation queue up for 9-14 us in front of the completion word.
  double* host_out = nullptr;
  int host_n = 0;
  // armed evaluation: the finish kernel returns at once when *cancel == ~0 (ArmedEval)
  const uint64_t* cancel = nullptr;
  // Self-identifying results (optional): next to the sequence number the publishing workgroup hands
  // the host WHAT was evaluated -- ident_dst[0] = *ident_src, the checksum the prep launch's copy
  // block took of the mixture pack it actually read (pack_checksum below), ident_dst[1] = ident_seed,
  // the Philox seed these launches were planned for.  The host compares both with what it sent
  // before it accepts the result block (api_elbo.hip): a result computed on a stale pack or by the
  // launches of another evaluation cannot pass for this one's.
  const uint64_t* ident_src = nullptr;
  uint64_t* ident_dst = nullptr;
  uint64_t ident_seed = 0;
  // Device-side hand-over (optimiser loop, adam.hip): results AND flag are device memory and the reader is another
  // workgroup of the same launch -- write-through (agent-scope) stores, drained, then the flag at agent scope; no host copy
  int dev = 0;
};

// Order-independent 64-bit checksum of a block of doubles: sum_i bits(v_i) * (odd_i) mod 2^64.  Every
// multiplier is odd, hence invertible: any change of a single element changes the sum.  The same
// terms are added up by the host (over its pinned pack) and by the 256 threads of the prep launch's
// copy block (over the pack as the device reads it).
__host__ __device__ inline uint64_t pack_ck_term(double v, uint32_t i) {
  uint64_t b;
  __builtin_memcpy(&b, &v, 8);
  return b * (0x9E3779B97F4A7C15ull + 2ull * i);
}
inline uint64_t pack_checksum(const double* p, size_t n) {
  uint64_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  size_t i = 0;
  for (; i + 4 <= n; i += 4) {
    s0 += pack_ck_term(p[i], (uint32_t)i);
    s1 += pack_ck_term(p[i + 1], (uint32_t)i + 1);
    s2 += pack_ck_term(p[i + 2], (uint32_t)i + 2);
    s3 += pack_ck_term(p[i + 3], (uint32_t)i + 3);
  }
  for (; i < n; ++i) s0 += pack_ck_term(p[i], (uint32_t)i);
  return (s0 + s1) + (s2 + s3);
}

#ifdef __HIPCC__
// The staged hand-over's copy (DoneSignal): n doubles from device memory written by other workgroups
// (write-through stores, already drained) to pinned host memory, by the 256 threads of one
// workgroup.  Eight loads are in flight per thread before the first store (a load -> store chain
// per element costs one memory latency each); sc1 loads read past this XCD's L2.
__device__ __forceinline__ void staged_copy_to_host(const double* __restrict__ src, double* __restrict__ dst, int n) {
  // (indices are clamped, not predicated: threads past the end repeat element n-1 -- the same value to
  // the same address -- so the loop body has no branch and the compiler keeps all eight loads, then
  // all eight write-through stores, in flight instead of draining the queue around every one)
  for (int base = 0; base < n; base += 256 * 8) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = min(base + u * 256 + (int)threadIdx.x, n - 1);
      v[u] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = min(base + u * 256 + (int)threadIdx.x, n - 1);
      __hip_atomic_store(dst + i, v[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): acknowledged
}
#endif

struct GenSlice {
  double* eps = nullptr;
  int K = 0, D = 0;
  int nb = 0;              // ceil(D / 4): Philox blocks per row
  uint32_t nb_magic = 0, rows_magic = 0;  // min(floor(2^32 / d), 2^32 - 1) for the two 32-bit divisions (philox.h gen_div32)
  int64_t rows = 0;        // rows per component held in eps (this rank's slice of the n_half pairs)
  int64_t n_half = 0, row_begin = 0;
  uint64_t seed = 0;
  const int* seed_add = nullptr;  // optional device-side addend (the Adam loop's iteration base)
  int64_t item_begin = 0, item_count = 0;
  int n_blocks = 0;        // ceil(item_count / 256)
};

struct PrepArgs {
  const double* mix = nullptr;
  MixLayout ml;
  // optional: one more workgroup copies mix[0 .. mix_copy_n) to mix_copy (the host-driven step
  // hands the pack over in host-written device memory; the later launches read the ordinary copy)
  double* mix_copy = nullptr;
  int mix_copy_n = 0;
  uint64_t* ident_out = nullptr;  // optional: that workgroup also stores pack_checksum(mix[0 .. mix_copy_n)) here (DoneSignal)
  // armed evaluation (ArmedEval): every workgroup first waits until *go == go_seq (the host has
  // written the pack) -- or leaves at once when it reads ~0 (cancelled); after ~2 ms without either
  // it cancels by itself (*go = ~0 for the launches behind it, *dead = go_seq for the host)
  uint64_t* go = nullptr;
  uint64_t go_seq = 0;
  uint64_t go_timeout = 200000;  // wall-clock ticks (100 MHz) the launch waits for the go word
  uint64_t* dead = nullptr;
  // table part (n_table = K blocks, or 0)
  int n_table = 0, DP = 0, K4 = 0;
  double* table = nullptr;
  // GP part (n_glj = S*K blocks, or 0)
  int n_glj = 0, N = 0, P = 0, want_grad = 0;
  int batch = 1;              // candidates (grid.y); candidate b uses mix + b*mix_stride, res + b*res_stride
  size_t mix_stride = 0, res_stride = 0;
  const double* X = nullptr;
  const double* XT = nullptr;  // [D][N]
  int x_lds = 0;               // glj_block.h: the launch's dynamic LDS holds X^T behind the block's own arrays (glj_block_lds_x): staged once per workgroup
  const double* alpha = nullptr;
  const double* hyp = nullptr;
  double* res = nullptr;  // [S][K][1+2D]  (device or device-visible pinned host memory)
  double* Z = nullptr;    // optional [S][K][N]
  // draw-generation part (gen.n_blocks workgroups, or 0): Philox/Box-Muller normals of the entropy
  // kernel -- throughput work that fills the GPU while the few table / GP blocks above sit in
  // their latency chains
  GenSlice gen;
  // optional completion word of the GP part (res must then be pinned host memory): the GP blocks
  // store their sums write-through, drain, and the last one to count publishes done.seq, so the
  // host can finalise G / dG while the entropy kernel runs
  DoneSignal done;
};
// the slice [frac_begin, frac_end) of the K * rows * ceil(D/4) draw items of eps[K][rows][D]
GenSlice make_gen_slice(double* eps, int K, int D, int64_t rows, int64_t n_half, int64_t row_begin,
                        uint64_t seed, const int* seed_add, double frac_begin, double frac_end);
// wait for everything queued on the ctx stream (also: the pinned mixture pack is free again)
inline hipError_t stream_wait(vbmc_ctx* ctx) {
  if (ctx->spec.armed) spec_disarm(ctx);  // (launches waiting for a theta would make this wait last their time-out)
  const hipError_t e = hipStreamSynchronize(ctx->stream);
  if (e == hipSuccess) ctx->pack_in_flight = false;
  return e;
}
int launch_prep(vbmc_ctx* ctx, const PrepArgs& a);                           // on ctx->stream
int launch_prep_on(vbmc_ctx* ctx, hipStream_t stream, const PrepArgs& a);

// staged launch of the Monte-Carlo entropy (entropy.hip)
struct EntPlan;
// gp_items: GP expected-log-joint items the caller would like this launch to carry in spare workgroup slots (the plan
// says whether it does: EntPlan::gp_in_ws); allow_span = false keeps the chunk grid (no caller does: the optimiser loop's pre
// workgroup is the LAST block of a span-mode launch and finds one of the slots the plan leaves free)
// gp_per_slot > 0 (the optimiser loop): that many items per workgroup and one more free slot, for its pre workgroup
int entmc_plan(vbmc_ctx* ctx, int64_t ns_per_comp, int eps_mode, uint64_t seed, int64_t row_begin,
               int64_t row_count, int want_grad, EntPlan& p, int gp_items = 0, bool allow_span = true, int gp_per_slot = 0);
void entmc_fill_prep(const vbmc_ctx* ctx, const EntPlan& p, PrepArgs& a);
int entmc_pregen(vbmc_ctx* ctx, EntPlan& p, PrepArgs& a);

int entmc_launch_main(vbmc_ctx* ctx, const EntPlan& p);
// `gen`: optional slice of draws for spare workgroups of the finish launch to generate
int entmc_launch_finish(vbmc_ctx* ctx, const EntPlan& p, double* raw_out, const GenSlice* gen = nullptr,
                        const DoneSignal* done = nullptr);
// multi-GPU step: hand the all-reduced raw vector (device memory) to the host and publish done.seq;
// spare workgroups generate `gen` (entropy.hip)
int entmc_launch_publish(vbmc_ctx* ctx, const double* d_raw, const DoneSignal& done, const GenSlice* gen);
// the slice of seed+1's draws the finish launch's spare workgroups should generate (n_blocks == 0: none)
GenSlice entmc_ahead_slice(vbmc_ctx* ctx, const EntPlan& p);
void glj_fill_prep(const vbmc_ctx* ctx, int want_grad, double* res, double* Z, PrepArgs& a);

// kernels' host launchers (one per .hip file) -------------------------------
// entropy
int launch_entmc(vbmc_ctx* ctx, int64_t ns_per_comp, int eps_mode, uint64_t seed,
                 int64_t row_begin, int64_t row_count, int want_grad, double* d_raw);
int launch_entlb(vbmc_ctx* ctx, double* d_res);  // writes raw entlb terms
// the in-line Philox/Box-Muller draws of the entropy kernel, written to eps[K][rows][D] by a
// kernel of their own on `st` (same values: counter = global row, pair; key = seed [+ *seed_add,
// a device-side int, when given])
int launch_eps_gen(vbmc_ctx* ctx, hipStream_t st, const GenSlice& g);
// mixture pdf
int launch_mixture_pdf(vbmc_ctx* ctx, int64_t n, const double* d_x, int log_flag,
                       int grad_flag, double df, double* d_y, double* d_dy);
int launch_mixture_pdf_on(vbmc_ctx* ctx, const double* d_pack, const MixLayout& ml, int64_t n,
                          const double* d_x, int log_flag, double* d_y);
// gp
int launch_gp_log_joint(vbmc_ctx* ctx, int want_grad, double* d_res, double* d_Z);
int launch_gp_var(vbmc_ctx* ctx, const double* d_Z, double* d_V, double* d_Q);
int launch_trinv(vbmc_ctx* ctx);
// leading dimension of the padded operands of predict's variance product, and the size of the
// K* scratch of a batch of mb points (rows rounded up to whole 64-row tiles)
inline size_t align32(size_t n) { return (n + 31) & ~(size_t)31; }
inline int predict_ld(int N) { return (N + 63) / 64 * 64; }
inline size_t predict_ks_elems(int S, int64_t mb, int N) {
  return (size_t)S * (size_t)((mb + 63) / 64 * 64) * (size_t)predict_ld(N);
}
int launch_gp_predict_products(vbmc_ctx* ctx, int64_t M, const double* d_xs, double* d_Ks, double* d_part,
                               const void* fin = nullptr, bool* fin_done = nullptr);
int launch_gp_predict_all(vbmc_ctx* ctx, int64_t M, const double* d_xs, double* d_Ks, double* d_part,
                          int add_noise, double* d_fmu, double* d_fs2, int64_t ld);
// c[n][m] = |a_n - b_m|^2 (centred expansion, cross term on the FP64 matrix cores), optional
// row-wise argmin; d_cen[D] = the centre subtracted from both sets
int launch_sq_dist(vbmc_ctx* ctx, const double* d_a, int64_t n, const double* d_b, int m, int D,
                   const double* d_cen, double* d_c, double* d_pmin, int64_t* d_argmin);

// host finalisation of the GP expected log joint (api_gp.hip)
void glj_finalize(const vbmc_ctx* ctx, const double* res, int want_grad, GljHost& o);
// packs dG for one sample (or the average) into out; returns its length
int glj_pack(vbmc_ctx* ctx, const double* mu, const double* sg, const double* lm,
             const double* wg, int grad_flags, int jacobian_flag, double* out);
// comm
int comm_allreduce_sum(vbmc_ctx* ctx, double* d_buf, int n);

// host finalisation (api_entropy.hip) -------------------------------------
void softmax_jacobian_apply(vbmc_ctx* ctx, const double* g, double* out);  // J_w(ctx->eta) @ g
int entropy_pack(vbmc_ctx* ctx, double H, const double* mu, const double* sg,
                 const double* lm, const double* wg, int grad_flags, int jacobian_flag,
                 double* H_out, double* dH_out);
