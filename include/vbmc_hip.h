/*
 * vbmc_hip.h -- C ABI of libvbmc_hip.so: the MI355X (gfx950) implementation of
 * PyVBMC's ELBO-evaluation hot path.
 *
 * The reference (acerbilab/pyvbmc) is pure Python and has no FFI; the boundary
 * it offers for this path is a set of Python callables.  Each entry point below
 * names the reference callable whose arithmetic it replaces (paths relative to
 * /root/reference/pyvbmc).  The Python mirror of those callables lives in
 * pyvbmc_amd/ and reaches this library through ctypes only.
 *
 * Conventions
 *   - every pointer argument is CALLER-OWNED HOST memory, contiguous, float64 /
 *     int32 / int64 as typed; the library copies in and out.  No torch types.
 *   - matrices are row-major.  `mu_KxD` is K rows of D means, i.e. exactly
 *     theta[:D*K] / mu.ravel(order="F") of the reference's (D,K) array
 *     (variational_posterior/variational_posterior.py:653-676).
 *   - return value: 0 = ok, <0 = error (VBMC_E_*); text via vbmc_last_error().  One entry point
 *     has a positive "do it again" code (VBMC_W_GP_CHANGED), one a positive "not here" code (VBMC_W_NOT_FUSED).
 *     No C++ exception crosses the boundary.
 *   - a vbmc_ctx owns one device, its HIP streams, device scratch and (optionally)
 *     one RCCL communicator.  A ctx is not thread-safe; distinct ctxs are
 *     independent.
 *   - grad_flags bit0..3 = gradients wrt (mu, sigma, lambda, w), the reference's
 *     4-tuple `grad_flags` (entropy/entmc_vbmc.py:9, entlb_vbmc.py:8).
 *   - all arithmetic is float64.
 */
#ifndef VBMC_HIP_H
#define VBMC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vbmc_ctx vbmc_ctx;

enum {
  VBMC_OK = 0,
  VBMC_E_ARG = -1,     /* bad argument / state (e.g. mixture not set)      */
  VBMC_E_HIP = -2,     /* HIP runtime error                                 */
  VBMC_E_RCCL = -3,    /* RCCL error                                        */
  VBMC_E_NODEV = -4,   /* no usable gfx950 device                           */
  VBMC_E_UNSUP = -5,   /* combination the reference raises NotImplemented on */
  VBMC_E_NONFINITE = -6, /* non-finite input where the path needs finite    */
  VBMC_E_NOMEM = -7,    /* host allocation failed (vbmc_mt19937_randn)       */
  VBMC_W_GP_CHANGED = 1, /* vbmc_neg_elcbo only: the watched GP arrays changed (vbmc_set_gp_watch);
                           the outputs were computed on the GP of the last vbmc_set_gp: discard them,
                           upload the GP again and repeat the call                               */
  VBMC_W_NOT_FUSED = 2   /* vbmc_adam_run_auto: this run does not have the one-launch form (shape, ranks,
                           or a launch that gave up waiting); nothing was done: use vbmc_adam_run in batches.
                           vbmc_mt19937_randn_dev: the request is not the device generator's; the state is
                           untouched: draw with vbmc_mt19937_randn                                        */
};

/* GP mean functions understood by the path
 * (vbmc/variational_optimization.py:1383-1392). */
enum { VBMC_MEAN_ZERO = 0, VBMC_MEAN_CONST = 1, VBMC_MEAN_NEGQUAD = 2 };

/* Source of the standard-normal draws of the Monte-Carlo entropy. */
enum {
  VBMC_EPS_RESIDENT = 0, /* draws uploaded with vbmc_set_eps (parity mode: the
                            reference's np.random.randn stream, entmc_vbmc.py:67) */
  VBMC_EPS_PHILOX = 1    /* Philox4x32-10 + Box-Muller generated on the device, fresh
                            per call from `seed` (throughput mode): by extra blocks
                            of the prep launch, by spare workgroups of the optimiser
                            loop's short launches one iteration ahead, or inside the
                            entropy kernel -- the same values                        */
};

/* ---- library / context ------------------------------------------------- */

/* ABI version of this header (bumped on any signature change). */
int vbmc_abi_version(void);

/* Number of visible HIP devices (0 on a box without a GPU; never fails hard). */
int vbmc_device_count(int* n_out);

/* Create a context on HIP device `device_id`.  Fails with VBMC_E_NODEV when no
 * GPU is present -- there is no CPU fallback behind this ABI.
 * device_id == -1 creates a HOST-ONLY context: it can hold the mixture
 * (vbmc_set_mixture / vbmc_theta_to_mixture) and run the host finalisation
 * (vbmc_entmc_finalize); every entry point that would launch a kernel returns
 * VBMC_E_NODEV on it.  It exists so the sharded reduce->finalise path can be
 * exercised by multi-process CPU tests. */
int vbmc_ctx_create(int device_id, vbmc_ctx** out);
void vbmc_ctx_destroy(vbmc_ctx* ctx);

/* Host placement.  The polled evaluation is a latency chain over PCIe, and on a two-socket host it is 2.5 us (3 %) per
 * evaluation shorter from the device's own NUMA node (csrc/ctx.hip bind_host_thread).  vbmc_ctx_create therefore narrows
 * the CALLING THREAD's CPU affinity to the CPUs local to the device (sysfs local_cpulist of its PCI function) before it
 * allocates its pinned buffers.  It only removes CPUs from the thread's current set and leaves a set alone that is
 * already inside the node or entirely outside it; VBMC_HOST_AFFINITY=0 in the environment disables it.  This reports what
 * happened: *bound_out = 1 when this context narrowed the set, *n_cpus_out = CPUs in the thread's set afterwards
 * (0: the device's CPU list could not be read).  No counterpart in the reference (a NumPy program). */
int vbmc_host_affinity(const vbmc_ctx* ctx, int* bound_out, int* n_cpus_out);

/* Last error text of `ctx` (or of the failed vbmc_ctx_create when ctx==NULL). */
const char* vbmc_last_error(const vbmc_ctx* ctx);

/* Device facts for the bench harness: name (NUL-terminated, truncated to
 * name_len), compute units, max clock kHz, global memory bytes. */
int vbmc_device_info(const vbmc_ctx* ctx, char* name, int name_len, int* cu_count,
                     int* clock_khz, uint64_t* hbm_bytes);

/* Block until everything queued on the ctx has finished (bench bracketing). */
int vbmc_synchronize(vbmc_ctx* ctx);

/* Switch the HIP event pair around the dominant kernels on or off (default: off).
 * A record between two dependent kernels puts a barrier packet into the queue, which
 * costs ~6 us per record on MI355X -- measurement harnesses switch it on for the
 * launches they want timed, production callers leave it off.
 * on = 2 additionally records the pair around gp_predict's variance product (which = 5 below): those two records sit
 * BETWEEN predict's launches and lengthen the interval which = 3 reports by their own ~12 us, so a harness reads
 * which = 3 at on = 1 and which = 5 at on = 2. */
int vbmc_set_timing(vbmc_ctx* ctx, int on);

/* Duration in milliseconds of the most recent TIMED launch (vbmc_set_timing) of the
 * dominant kernel of the given entry point, from HIP events on the ctx's own stream.
 * which: 0 = entmc main kernel, 1 = gp_log_joint, 2 = mixture pdf,
 *        3 = gp_predict (all its launches), 4 = whole last vbmc_neg_elcbo device section,
 *        5 = gp_predict's variance product kernel alone -- recorded at vbmc_set_timing(ctx, 2) ONLY, and that pass runs
 *            the product WITHOUT the finish in its epilogue (three launches where production runs two): it measures
 *            the product kernel, not the production configuration; a predict at level 0 / 1 invalidates the record
 *            (VBMC_E_ARG "no timed launch recorded", never a stale interval). */
int vbmc_last_kernel_ms(vbmc_ctx* ctx, int which, double* ms_out);

/* Host-side wall-clock breakdown (microseconds) of the most recent vbmc_neg_elcbo:
 * out[0] theta -> mixture + pack upload issue, out[1] kernel launches,
 * out[2] wait for the device, out[3] host finalisation, out[4] total.
 * Profiling aid for bench.py / DESIGN.md; no reference counterpart. */
int vbmc_last_host_us(const vbmc_ctx* ctx, double out[5]);
/* Where the two result blocks of the most recent vbmc_neg_elcbo arrived on the host's clock, microseconds from
 * the call's entry: out[0] the GP sums' completion word (the host finalises G / dG behind it, while the entropy
 * kernel still runs), out[1] the entropy's completion word (or the end of the stream wait); out[2] where the
 * GP sums ran: 0 the prep launch, 1 the finish launch, 2 spare workgroup slots of the entropy launch; out[3]
 * reserved (0).  Profiling aid for bench.py (S > 1 hyper-parameter samples). */
int vbmc_last_step_marks(const vbmc_ctx* ctx, double out[4]);
/* A function the library calls from inside vbmc_neg_elcbo once the evaluation's launches are released (the pack and
 * the go word written, or the launches queued) and before it starts waiting for the device: the caller's own
 * bookkeeping that depends on theta alone -- the reference's vp.set_parameters side effects, which the Python mirror
 * applies from the mu / sigma / lambda / w / eta outputs, already filled at that point -- then runs while the device
 * works instead of between two evaluations.  fn = NULL clears it.  No reference counterpart. */
int vbmc_set_release_callback(vbmc_ctx* ctx, void (*fn)(void*), void* user);

/* Per-context switches for tests and measurements (no reference counterpart).  Each starts
 * from the environment variable in brackets, read when the context is created.
 *   "entmc_kernel" [VBMC_ENTMC_KERNEL=valu -> 1]: 0 = pick the kernel by shape (default),
 *                  1 = always the generic thread-per-row kernel (on-device cross-check)
 *   "entmc_mfma"   1 = shapes the FP64 matrix tile pads little and the wave-split kernel runs one wave per SIMD on
 *                  (D > 10 or K > 80, K within 12 below a multiple of 16: BASELINE config 5) take the matrix-pipe form
 *                  of the entropy kernel (default), 0 = the wave-split kernel everywhere
 *   "gp_ship"      [VBMC_GP_SHIP]: 1 = in the host-driven step whose GP sums run in the prep launch in front of the matrix-pipe entropy
 *                  kernel, their copy to pinned memory and their completion word are issued by a workgroup of the entropy
 *                  launch (default: the prep launch then ends with the sums), 0 = by the prep launch's last GP block
 *   "acq_poll"     [VBMC_ACQ_POLL]: 1 = vbmc_acq_eval with at most 256 points has the CPU write the points into host-writable
 *                  device memory and polls a completion word for the results (default), 0 = copies + stream wait
 *   "adam_fused"   [VBMC_ADAM_FUSED]: 1 = vbmc_adam_run runs a batch of iterations as ONE launch where the shape allows
 *                  (one rank, K <= 64, D <= 24, <= 64 antithetic rows per component, LDS plan fits -- with X^T resident in
 *                  the GP workgroups' LDS, or read from memory where that does not fit; default),
 *                  0 = always four launches per iteration; 2 = test hook (the launch also waits for a workgroup
 *                  that does not exist, must give up by its 20 ms limit, and the batch is redone as four launches)
 *                  3 = the one-launch form with its exchange written as agent-scope release stores / acquire fences
 *                  instead of write-through records and relaxed flags (the documented fallback: the same results bit
 *                  for bit, +5 us per iteration)
 *   "elbo_pregen"  [VBMC_ELBO_PREGEN]: 1 = Philox draws generated ahead of the entropy
 *                  kernel (default), 0 = generated in-line by it; same values either way
 *   "elbo_ahead"   [VBMC_ELBO_AHEAD]: 1 = after a Philox evaluation with seed s the draws of
 *                  seed s+1 are generated speculatively, by spare workgroups of the finish launch,
 *                  while the host finalises (default), 0 = never
 *   "mix_bar"      [VBMC_MIX_BAR]: 1 = in the polled host-driven step the CPU writes the mixture
 *                  pack straight into (fine-grained) device memory and the GP sums ride in spare
 *                  workgroup slots of the entropy launch when they fit (default), 0 = upload kernel,
 *                  GP sums in the prep launch
 *   "predict_dma"  [VBMC_PREDICT_DMA]: 1 = gp_predict's variance product for batches of > 32
 *                  points on Cholesky samples runs in the LDS-direct kernel (default), 0 = the
 *                  plain 64 x 64-tile kernel (cross-check)
 *   "predict_fused" [VBMC_PREDICT_FUSED_FINISH]: gp_predict's third stage (fmu / fs2 from the partial sums)
 *                  in the LDS-direct product kernel's epilogue, by an arrival ticket per 64-point row tile --
 *                  two launches instead of three, bit-identical results: 1 = where the product grid is one
 *                  round of workgroups (default), 2 = always, 0 = never (the finish launch)
 *   "elbo_arm"     [VBMC_ELBO_ARM]: 1 = after a polled Philox evaluation the launches of the next one
 *                  (seed + 1, same shapes) are queued at once and wait on the device for the next
 *                  call's theta (default); any other use of the context cancels them.  Never after an
 *                  evaluation that took more than 1 ms, and the device-side wait is at most 5 ms.
 *   "arm_late_test" / "ident_test": test hooks -- the n-th use of an armed evaluation from now takes
 *                  the late-go recovery path / the n-th identity check of a result block from now
 *                  fails (the evaluation is repeated unarmed); see vbmc_armed_stats
 *   "ws_span"      [VBMC_WS_SPAN]: 1 = the wave-split entropy kernel in span mode (default): every CU's
 *                  first-dispatched workgroup takes "ws_front" per mille of the CU's batches, the second one
 *                  the rest, parts crossing component boundaries, so that all workgroups end together
 *                  (csrc/entropy_args.h WsSpan); 0 = equal chunks per component
 *   "ws_front"     [VBMC_WS_FRONT]: that share (0 = built-in per instantiation); "ws_pad" [VBMC_WS_PAD]:
 *                  batches a component's end is priced at when the parts are cut (-1 = built-in)
 * The results of an evaluation do not depend on any of these -- except that "ws_span" / "ws_front" /
 * "ws_pad" change how the Monte-Carlo rows are grouped into partial sums, i.e. the last bits of the
 * entropy (<= 1e-15 relative); for given values every evaluation is bit-reproducible.
 * Unknown key -> VBMC_E_ARG. */
int vbmc_set_option(vbmc_ctx* ctx, const char* key, int value);

/* Launch geometry of the most recent Monte-Carlo entropy of this ctx (vbmc_entmc,
 * vbmc_neg_elcbo, the optimiser loop): out[0] = kernel (0 generic, 1 wave-split on equal chunks, 2 small-
 * sample, 3 matrix-pipe form, 4 the fused optimiser loop: no entropy launch of its own, 5 wave-split in span
 * mode), out[1] = 64-row batches per workgroup (the wave-split kernel's batch loop count; span mode: the
 * longest part), out[2] = partial rows (workgroups) per component, out[3] = 1 if the draws were read from
 * HBM, 0 if generated in-line.  Lets the parity tests assert which code path they exercised. */
int vbmc_last_entmc_plan(const vbmc_ctx* ctx, int out[4]);

/* Host utility (no device, no ctx): the span-mode partition of the wave-split entropy kernel for cus CUs of
 * which pb have a filler part, front share `front` (per mille), nb batches per component, `pad` slots priced
 * per component end, K components -- part_lo[cus + pb + 1] = first slot of every part of the padded list
 * (part_lo[last] = K (nb + pad)), first_part[K] = the part that holds each component's first batch,
 * *rows_per_component = partial rows per component.  The CPU suite checks the partition's invariants. */
int vbmc_ws_span_layout(int cus, int pb, int front, int nb, int pad, int K, int64_t* part_lo, int* first_part,
                        int* rows_per_component);

/* Host utility (no device, no ctx): a 64-bit checksum over n blocks of doubles,
 * sum_a (sum_i bits(v_ai) * odd_i + len_a) * odd_a mod 2^64 -- any change of a single element changes
 * it.  The Python mirror keys its "is this GP already on the device" test on it (every posterior's
 * alpha and hyp in full: an in-place edit anywhere in them is seen; ~1 us + 0.05 us per KB). */
int vbmc_host_checksum(const double* const* ptrs, const int64_t* lens, int n, uint64_t* out);

/* Watch the HOST arrays the device GP was uploaded from: n blocks of doubles (every posterior's alpha
 * and hyp) whose vbmc_host_checksum was `expected` at upload time.  vbmc_neg_elcbo then recomputes
 * that checksum itself -- after it has released its launches, while the device works and the host
 * would only be polling -- and returns VBMC_W_GP_CHANGED when it differs.  This keeps the caller's
 * "is this GP still the one on the device" test (the reference overwrites GP records in place,
 * active_importance_sampling.py:207-209) off the path between two evaluations of the optimiser.
 * The caller guarantees the blocks stay allocated while the watch is set; n = 0 or the next
 * vbmc_set_gp clears it. */
int vbmc_set_gp_watch(vbmc_ctx* ctx, const double* const* ptrs, const int64_t* lens, int n, uint64_t expected);

/* Counters of the polled host-driven step (vbmc_neg_elcbo) since the context was created:
 * out[0] armed evaluations used, out[1] cancelled, out[2] of those: late go word (recovery path),
 * out[3] result blocks whose identity was checked -- the finish launch publishes, next to the
 * sequence number, the checksum of the mixture pack the device actually read and the Philox seed
 * its launches were planned for, and the host compares both with what it sent -- out[4] checks
 * that failed (the evaluation was repeated), out[5] evaluations whose completion word never came
 * (the armed launches had given up, or a stuck device) and that were repeated. */
int vbmc_armed_stats(const vbmc_ctx* ctx, uint64_t out[6]);

/* The raw (pre-Jacobian) entropy accumulator [H | mu (K x D) | sigma (K) | lambda (D) | w (K)] of
 * the most recent Monte-Carlo vbmc_neg_elcbo of this ctx -- the vector the sharded job all-reduces
 * (additive over disjoint row slices; with a communicator: the sum over the ranks).  n must be
 * 1 + D*K + 2K + D.  VBMC_E_ARG when the last evaluation had none.  Lets the parity tests add up
 * the slices of virtual ranks (row_begin/row_count of vbmc_elbo_opts) on one GPU. */
int vbmc_last_elbo_raw(const vbmc_ctx* ctx, double* out, int n);

/* ---- mixture state: VariationalPosterior attributes --------------------- */

/* Upload the mixture (variational_posterior.py:106-138: mu (D,K), sigma (1,K),
 * lambd (D,1), w (1,K), eta (1,K)).  Values are taken as given (no
 * renormalisation: that is set_parameters' job, see vbmc_theta_to_mixture).
 * Giving the values the device already holds is a no-op (no pack, no upload): pdf,
 * acquisition and sampling calls between two updates of the posterior pay nothing. */
int vbmc_set_mixture(vbmc_ctx* ctx, int D, int K, const double* mu_KxD,
                     const double* sigma_K, const double* lambd_D, const double* w_K,
                     const double* eta_K);

/* The same with the means in the reference's own layout, mu (D,K) row-major -- the array a VariationalPosterior
 * holds -- so that a caller in front of every pdf / acquisition call passes its attribute arrays as they are. */
int vbmc_set_mixture_dk(vbmc_ctx* ctx, int D, int K, const double* mu_DxK,
                        const double* sigma_K, const double* lambd_D, const double* w_K,
                        const double* eta_K);

/* Host-side restatement of VariationalPosterior.set_parameters (raw_flag=True)
 * (variational_posterior.py:680-759) for the fused objective: theta ->
 * (mu, sigma, lambd, w) with exp, softmax (max-shifted), lambda renormalised to
 * unit RMS, plus eta = theta[-K:] - max (variational_optimization.py:1082-1085).
 * optimize_mask bit0..3 = optimize_{mu,sigma,lambd,weights}; blocks whose bit is
 * clear are absent from theta and the current ctx mixture values are kept.
 * Outputs (each nullable) receive the new attribute values. */
int vbmc_theta_to_mixture(vbmc_ctx* ctx, const double* theta, int n_theta,
                          int optimize_mask, double* mu_KxD, double* sigma_K,
                          double* lambd_D, double* w_K, double* eta_K);

/* ---- a3: VariationalPosterior.pdf / log_pdf in the transformed space ---- */

/* y[n] = sum_k w_k N(x_n; mu_k, sigma_k^2 diag(lambda^2))  (df = +/-inf or 0),
 * multivariate-t (df > 0) or product-of-univariate-t (df < 0) tails
 * (variational_posterior.py:441-529); log_flag: log with 0 -> -inf (:531-541);
 * grad_flag: dy (n x D), for log_flag dy/y (:464-469,532-533).  grad with finite
 * non-zero df -> VBMC_E_UNSUP (the reference raises NotImplementedError :499,527).
 * Bounds masking / Jacobian of orig_flag=True stay on the host (parameter
 * transformer), as in the reference they are separate pre/post steps. */
int vbmc_mixture_pdf(vbmc_ctx* ctx, int64_t n, const double* x_nxD, int log_flag,
                     int grad_flag, double df, double* y_n, double* dy_nxD);

/* ---- a6: entropy/entmc_vbmc.py:6-134 ------------------------------------ */

/* Make the antithetic half-draws resident in HBM: eps_half is [K][n_half][D]
 * (for component j the rows the reference draws with randn(Ns//2, D),
 * entmc_vbmc.py:64-68).  `row_begin,row_count` select the slice of each
 * component's rows this ctx will process (sharding); pass 0,n_half
 * for the whole job. */
int vbmc_set_eps(vbmc_ctx* ctx, int K, int64_t n_half, int D, const double* eps_half,
                 int64_t row_begin, int64_t row_count);

/* The reference's draw stream on the host cores (no device, no ctx): the next `n` values
 * np.random.randn would return from NumPy's legacy global state -- MT19937 words, 53-bit
 * uniforms, polar method (the eps of entmc_vbmc.py:67) -- bit for bit, with the state advanced
 * exactly as NumPy would leave it.  `key[624], pos, has_gauss, gauss` are the fields of
 * np.random.get_state(legacy=True) (in/out).  n_threads <= 0: all host cores (at most 64).
 * The MT19937 recurrence runs on one thread; the polar method's attempts, each a pure
 * function of its word position, on all of them (csrc/host_randn.hip). */
int vbmc_mt19937_randn(uint32_t* key, int* pos, int* has_gauss, double* gauss, double* out,
                       int64_t n, int n_threads);

/* The same stream generated on the DEVICE (csrc/device_randn.hip: MT19937 with a GF(2) jump-ahead per workgroup, the
 * polar method's attempts as position-pure work items, a prefix sum over the accepted pairs) and copied to `out`.
 * Words, accept / reject decisions and the state handed back are bit-identical to np.random.randn's; the VALUES use a
 * double-double logarithm on the device and about 0.1 % of them differ from NumPy's (glibc's log) by 1-3 units in the
 * last place (neither logarithm is correctly rounded everywhere; histogram in the source).  What vbmc_set_eps_numpy
 * uses (without the copy) when the context holds all rows.  The caller's state is written only on success.  Returns
 * VBMC_W_NOT_FUSED, state untouched, when the request is not this path's (more than 8e8 words, or fewer accepted
 * attempts than pairs inside the 6-sigma word margin: ~1e-9 per call) -- draw with vbmc_mt19937_randn then.
 * Reference: entropy/entmc_vbmc.py:64-68. */
int vbmc_mt19937_randn_dev(vbmc_ctx* ctx, uint32_t* key, int* pos, int* has_gauss, double* gauss, double* out,
                           int64_t n);
/* *window_reused = 1 when the context's last device pass (vbmc_mt19937_randn_dev / vbmc_set_eps_numpy) found the 33 blocks
 * behind its incoming state in the word sequence of the pass before -- a caller that keeps drawing from the stream where
 * the last call left it -- and did not have to compute them (csrc/device_randn.hip). */
int vbmc_randn_dev_info(const vbmc_ctx* ctx, int* window_reused);
/* Host twins of that jump for the CPU tests: the MT19937 block that starts n_words (>= 1) words after key_in[0], by the
 * polynomial t^(n_words-1) mod the characteristic polynomial (found by Berlekamp-Massey, csrc/mt_jump.h); and the
 * polynomials t^(m stride - 1), m = 1 .. count, as [count][624] words. */
int vbmc_mt_jump_host(const uint32_t* key_in, uint64_t n_words, uint32_t* key_out);
int vbmc_mt_jump_polys(uint64_t stride_words, int count, uint32_t* out);

/* vbmc_mt19937_randn + vbmc_set_eps in one call: the next K*n_half*D values of NumPy's legacy
 * stream (= np.random.randn(n_half, D) for j = 0..K-1, the reference's draw order) are generated
 * into a pinned buffer the ctx keeps and rows [row_begin, +row_count) of every component are
 * uploaded as the resident draws.  The generator state advances by the whole job's values on
 * every rank.
 * WHICH GENERATOR RUNS, and what that means for parity: a single-rank context that holds all rows and asks for
 * >= 65 536 values draws ON THE DEVICE (vbmc_mt19937_randn_dev: state, words and accept / reject decisions
 * bit-identical to NumPy, ~0.1 % of the values 1-3 ulp off); a row shard, a multi-rank context, a smaller request, or
 * the device pass answering VBMC_W_NOT_FUSED draws on the host cores (vbmc_mt19937_randn: every value bit-identical).
 * F and dF computed from the two therefore agree to ~1e-15 relative, not bit for bit.  Strict parity switch:
 * vbmc_set_option(ctx, "randn_device", 0) or VBMC_RANDN_DEVICE=0 in the environment -- always the host generator.
 * The caller's state is written only when the draw succeeded. */
int vbmc_set_eps_numpy(vbmc_ctx* ctx, uint32_t* key, int* pos, int* has_gauss, double* gauss, int K,
                       int64_t n_half, int D, int64_t row_begin, int64_t row_count, int n_threads);

/* Monte-Carlo entropy and its reparameterisation gradient.
 *   ns_per_comp : the reference's (even-rounded) Ns = 2*n_half.
 *   eps_mode    : VBMC_EPS_RESIDENT or VBMC_EPS_PHILOX (`seed` used by the latter).
 *   row_begin,row_count : the antithetic-pair rows of every component this ctx
 *                 evaluates (0,n_half = all); the normaliser stays ns_per_comp.
 *   H, dH       : as the reference returns them -- dH packs only the enabled
 *                 blocks [mu 'F' | sigma | lambda | w] (entmc_vbmc.py:48-51,132),
 *                 Jacobians applied when jacobian_flag (entmc_vbmc.py:114-130).
 *   raw_out     : nullable; receives the un-Jacobianed accumulator vector
 *                 [H | mu (K blocks of D) | sigma (K) | lambda (D) | w (K)]
 *                 (length 1+D*K+2K+D) -- the vector the all-reduce sums.
 * If the ctx has a communicator (vbmc_comm_init) the raw vector is summed over
 * ranks with ONE ncclAllReduce before finalisation. */
int vbmc_entmc(vbmc_ctx* ctx, int64_t ns_per_comp, int eps_mode, uint64_t seed,
               int64_t row_begin, int64_t row_count, int grad_flags, int jacobian_flag,
               double* H, double* dH, double* raw_out);

/* The draws of VBMC_EPS_PHILOX mode themselves: out[K][row_count][D] = the standard normals the
 * entropy kernels use for rows [row_begin, +row_count) of every component of a job with n_half
 * antithetic pairs per component (counter-based: Philox4x32-10 keyed by seed on (global row, block),
 * four normals per block; csrc/philox.h, restated by oracle/philox_ref.py).  Generated by the same
 * kernel code as inside an evaluation.  For parity and distribution tests; no reference counterpart
 * (the reference draws np.random.randn, entropy/entmc_vbmc.py:64-68). */
int vbmc_philox_normals(vbmc_ctx* ctx, int K, int64_t n_half, int D, uint64_t seed, int64_t row_begin,
                        int64_t row_count, double* out);

/* Finalise a raw accumulator vector on the host (Jacobians + packing,
 * entmc_vbmc.py:114-132) with the ctx's current mixture: what every rank does
 * after the all-reduce.  Exposed so the sharded path can be tested without a GPU
 * collective. */
int vbmc_entmc_finalize(vbmc_ctx* ctx, const double* raw, int grad_flags,
                        int jacobian_flag, double* H, double* dH);

/* ---- a7: entropy/entlb_vbmc.py:6-180 ------------------------------------- */
int vbmc_entlb(vbmc_ctx* ctx, int grad_flags, int jacobian_flag, double* H, double* dH);

/* ---- GP posterior state: what gpyreg hands the path (a11) --------------- */

/* X (N x D); per GP sample s: hyp (P doubles: [log ell (D), log sf, noise (1),
 * mean (0 | 1 | 1+2D)]), alpha (N), L (N x N), L_chol, sn2_eff = 1/sW[0]^2
 * (variational_optimization.py:1394-1398), sW (N) and sn2_mult for predict. */
int vbmc_set_gp(vbmc_ctx* ctx, int N, int D, int S, int P, int mean_kind,
                const double* X_NxD, const double* hyp_SxP, const double* alpha_SxN,
                const double* L_SxNxN, const int32_t* L_chol_S, const double* sW_SxN,
                const double* sn2_mult_S);

/* ---- a8: vbmc/variational_optimization.py:1238-1606 _gp_log_joint ------- */

/* G, dG (enabled blocks; sigma/lambda/w blocks only under jacobian_flag,
 * :1528-1546), and when compute_var: varG, var_ss (:1578-1596), I_sk (S x K),
 * J_sjk (S x K x K).  avg_flag as the reference (:1578).  Outputs nullable.
 * With avg_flag=0 or S==1 semantics: G_out/dG_out hold per-sample values laid
 * out [S] and [n_dG][S] like the reference's arrays.  compute_var with any
 * gradient -> VBMC_E_UNSUP (reference raises :1303-1307); compute_var==2 too. */
int vbmc_gp_log_joint(vbmc_ctx* ctx, int grad_flags, int avg_flag, int jacobian_flag,
                      int compute_var, double* G, double* dG, double* varG,
                      double* var_ss, double* I_SxK, double* J_SxKxK);

/* ---- a12: gpyreg GP.predict (third party) ------------ */

/* Per sample: fmu = m(x*) + K*^T alpha; fs2 = max(0, sf^2 - ||L^-T (sW o K*)||^2)
 * (L_chol) or max(0, sf^2 + diag(K*^T L K*)).  separate_samples: outputs (M x S)
 * row-major; else (M): mean over s and mean_s fs2 + var_s(fmu, ddof=1)
 * (the law restated at acquisition_functions/abstract_acq_fcn.py:82-97).
 * add_noise: adds exp(2 hyp_noise)*sn2_mult to the variance. */
int vbmc_gp_predict(vbmc_ctx* ctx, int64_t M, const double* xs_MxD, int add_noise,
                    int separate_samples, double* fmu, double* fs2);

/* ---- a9: the fused objective _neg_elcbo (:991-1235) ---------------------- */

typedef struct {
  /* inputs */
  int64_t ns_per_comp;  /* reference `Ns` (per component); 0 -> entlb           */
  int eps_mode;         /* VBMC_EPS_*                                            */
  uint64_t seed;        /* Philox seed                                           */
  int compute_grad;     /* reference compute_grad                                */
  int optimize_mask;    /* bit0..3 = vp.optimize_{mu,sigma,lambd,weights}        */
  int64_t row_begin;    /* antithetic-pair rows of every component evaluated by  */
  int64_t row_count;    /* this ctx; row_count < 0 -> the rank's even share       */
  /* soft bounds (theta_bnd), all nullable together (:1195-1229)                 */
  const double* bnd_lb; /* length n_bnd                                          */
  const double* bnd_ub;
  int n_bnd;
  double tol_con;
  double weight_threshold;
  double weight_penalty;
} vbmc_elbo_opts;

/* One evaluation of F = -G - H (+ bound / weight penalties) and dF with a single
 * device round trip: theta -> mixture (vbmc_theta_to_mixture), G/dG kernel,
 * entropy kernels, optional all-reduce, host finalisation.  theta's eta tail is
 * max-shifted IN PLACE like the reference does to its caller (:1082-1085).
 * Outputs: F, dF (n_theta; untouched if !compute_grad), G, H; mixture outputs as
 * in vbmc_theta_to_mixture (nullable). */
int vbmc_neg_elcbo(vbmc_ctx* ctx, double* theta, int n_theta, const vbmc_elbo_opts* opts,
                   double* F, double* dF, double* G, double* H, double* mu_KxD,
                   double* sigma_K, double* lambd_D, double* w_K, double* eta_K);

/* The same call with its twelve arguments in ONE block (no reference counterpart: the reference's call is a Python call).
 * A ctypes foreign call converts every argument on every call -- 1.8 us for the thirteen of vbmc_neg_elcbo against 0.3 us for
 * two, on the path between two evaluations of a polled step -- so a binding that keeps its buffers fills the block once
 * and passes its address.  Fields exactly as vbmc_neg_elcbo's parameters. */
typedef struct vbmc_elbo_call {
  double* theta;
  int n_theta;
  const vbmc_elbo_opts* opts;
  double* F;
  double* dF;
  double* G;
  double* H;
  double* mu_KxD;
  double* sigma_K;
  double* lambd_D;
  double* w_K;
  double* eta_K;
} vbmc_elbo_call;
int vbmc_neg_elcbo_call(vbmc_ctx* ctx, const vbmc_elbo_call* c);

/* ---- the sieve's batch of candidate evaluations ---------- */

/* B candidate parameter vectors (rows of thetas_BxN) through the call _sieve makes per
 * candidate (vbmc/variational_optimization.py:775-787): Ns = 0 (lower-bound entropy), no
 * gradient, soft bounds as in `opts`; F_B[b] = -G_b - H_b + bound losses.  opts->ns_per_comp
 * must be 0 and opts->compute_grad 0 (VBMC_E_UNSUP otherwise).  Unlike vbmc_neg_elcbo this
 * neither changes the ctx mixture nor touches the theta rows.  G_B / H_B nullable.  Everything per
 * candidate -- theta -> mixture -> pack, the GP sums, the entropy, G, the bound losses -- runs on the
 * device (four launches for the whole batch); VBMC_E_NONFINITE names the first non-finite candidate. */
int vbmc_neg_elcbo_batch(vbmc_ctx* ctx, const double* thetas_BxN, int B, int n_theta,
                         const vbmc_elbo_opts* opts, double* F_B, double* G_B, double* H_B);

/* ---- the stochastic optimiser's loop, device resident ---- */

/* minimize_adam (vbmc/minimize_adam.py:8-146) specialised to the objective PyVBMC gives it,
 * vb_train_mc_fun = _neg_elcbo(theta, gp, vp0, beta, ns_ent_K, compute_grad=True,
 * theta_bnd=...) (vbmc/variational_optimization.py:238-249).  theta, the Adam moments and
 * the mixture stay on the device; one iteration is four kernel launches and no
 * synchronisation -- or, at the sample counts optimize_vp really uses (ns_ent = 100 K^(2/3) in total,
 * option_configs/advanced_vbmc_options.ini:43), a whole vbmc_adam_run is ONE launch of resident workgroups
 * that exchange their results once per iteration (option "adam_fused").  The early-stopping decision (minimize_adam.py:107-140) stays with the
 * caller, who sees y_tab / x_tab after every vbmc_adam_run -- the reference only tests it
 * every 20 iterations.
 *
 * vbmc_adam_begin: theta0[n_theta] start point (not modified); `opts` as for vbmc_neg_elcbo
 *   (ns_per_comp even and > 0, compute_grad set; with VBMC_EPS_PHILOX iteration i draws
 *   from seed + i; with VBMC_EPS_RESIDENT every iteration reuses the resident draws);
 *   lb/ub[n_theta] the optional box of minimize_adam (both NULL = unbounded); max_iter and
 *   master_* as in the reference.  Non-optimised blocks keep the ctx mixture's values.
 * vbmc_adam_run: the next n_iters iterations.  y_tab_out[n_iters] = objective values
 *   (minimize_adam's y_tab slice), x_tab_out[n_iters][n_theta] = iterates after each update
 *   (rows; the reference stores them as columns), G_out/H_out[n_iters] the two terms of the
 *   objective.  All nullable.  VBMC_E_NONFINITE if an iterate became non-finite.  (If a workgroup of the
 *   one-launch form does not publish its results within 20 ms, the launch gives up with the state of the start
 *   of the batch intact and this call runs the batch -- and the rest of the run -- as four launches per iteration.)
 * vbmc_adam_end: ends the run; the ctx mixture becomes that of the last iterate (outputs as
 *   in vbmc_theta_to_mixture, nullable; theta_out = last x with its eta tail max-shifted).
 * Between begin and end no other entry point of the same ctx may be called. */
int vbmc_adam_begin(vbmc_ctx* ctx, const double* theta0, int n_theta, const vbmc_elbo_opts* opts,
                    const double* lb, const double* ub, int max_iter, double master_min,
                    double master_max, double master_decay);
/* vbmc_adam_run_auto: the rest of the optimisation -- up to max_iters iterations -- in ONE call, where the run has the
 *   one-launch form (option "adam_fused") and has not run an iteration yet (the kernel's stopping rule needs the
 *   mean iterate of the previous batch, which it collects itself from iteration 0 on): the workgroups apply
 *   minimize_adam's stopping rule themselves (minimize_adam.py:107-140: every 20 iterations from the 40th on, the
 *   slope of a straight-line fit through the last 20 objective values against its standard error and tol_fun, and
 *   the distance between the mean iterates of the last two batches) and *n_done returns how many iterations ran;
 *   outputs as vbmc_adam_run for those.  Returns VBMC_W_NOT_FUSED (and does nothing) otherwise. */
int vbmc_adam_run_auto(vbmc_ctx* ctx, int max_iters, double tol_fun, int* n_done, double* y_tab_out,
                       double* x_tab_out, double* G_out, double* H_out);
int vbmc_adam_run(vbmc_ctx* ctx, int n_iters, double* y_tab_out, double* x_tab_out,
                  double* G_out, double* H_out);
int vbmc_adam_end(vbmc_ctx* ctx, double* theta_out, double* mu_KxD, double* sigma_K,
                  double* lambd_D, double* w_K, double* eta_K, int* iterations);

/* ---- acquisition evaluation, pairwise distances -- */

/* Closed-form acquisition functions of the reference. */
enum {
  VBMC_ACQ_STD = 0,     /* AcqFcn        acquisition_functions/acq_fcn.py:38-45          */
  VBMC_ACQ_LOG = 1,     /* AcqFcnLog     acquisition_functions/acq_fcn_log.py:43-52      */
  VBMC_ACQ_VANILLA = 2, /* AcqFcnVanilla acquisition_functions/acq_fcn_vanilla.py:38-42  */
  VBMC_ACQ_NOISY = 3    /* AcqFcnNoisy   acquisition_functions/acq_fcn_noisy.py:33-41    */
};

/* The device part of AbstractAcqFcn.__call__ (acquisition_functions/abstract_acq_fcn.py:
 * 80-131) for M points in transformed space: gp.predict(separate_samples=True) with the GP
 * of vbmc_set_gp, f_bar / var_tot over the hyper-parameter samples (:82-97), the density of
 * the ctx mixture (vp.pdf(Xs, orig_flag=False[, log_flag=True])), the formula selected by
 * `kind` with y_max = function_logger.y_max, the variance regularisation (:112-128; off when
 * tol_gp_var <= 0) and the clamp at -realmax (:130-131).  sn2_M: per-point observation noise,
 * VBMC_ACQ_NOISY only.  Integer-variable rounding (:77-79) and the hard-bound mask
 * (:133-139) involve the caller's parameter transformer and stay on the host.
 * f_bar_M / var_tot_M nullable. */
int vbmc_acq_eval(vbmc_ctx* ctx, int64_t M, const double* xs_MxD, int kind, double y_max,
                  double tol_gp_var, const double* sn2_M, double* acq_M, double* f_bar_M,
                  double* var_tot_M);

/* The importance-sampled acquisition functions for noisy targets, AcqFcnVIQR
 * (acquisition_functions/acq_fcn_viqr.py:30-160) and AcqFcnIMIQR (acq_fcn_imiqr.py:29-177).
 *
 * vbmc_acq_is_set uploads, once per active-sampling round, what the reference keeps in
 * optim_state["active_importance_sampling"] (vbmc/active_importance_sampling.py:262-306) for the
 * GP of vbmc_set_gp (S samples, N points): the importance points Xa (Na x D, or S x Na x D with
 * per_sample_xa -- IMIQR after MCMC), C_tmp[s] = (K+Sigma)^-1 K(X, Xa) resp. L K(X, Xa) (S x N x Na;
 * for IMIQR, which stores K_Xa_X instead, the caller forms it the same way), the predictive
 * variances f_s2 at Xa (Na x S as the reference stores them) and ln_weights (S x Na; NULL = VIQR's
 * constant weights).
 * vbmc_acq_is_eval evaluates the acquisition at M points of the transformed space: predictive
 * variance at the points, posterior cross-covariance with Xa, tau2 = C^2 / (f_s2 + sn2), s_pred,
 * log-sum-exp over Xa and over the GP samples; sn2_M = observation noise at the points
 * (_estimate_observation_noise), u = norm.ppf(quantile).  Integer rounding and the hard-bound mask
 * stay on the host, as for vbmc_acq_eval. */
int vbmc_acq_is_set(vbmc_ctx* ctx, int64_t Na, const double* Xa, int per_sample_xa,
                    const double* Ctmp_SxNxNa, const double* fs2a_NaxS, const double* lnw_SxNa);
int vbmc_acq_is_eval(vbmc_ctx* ctx, int64_t M, const double* xs_MxD, const double* sn2_M, double u,
                     double* acq_M, double* var_tot_M /* nullable: for the variance regularisation */);

/* AbstractAcqFcn._sq_dist (acquisition_functions/abstract_acq_fcn.py:195-222):
 * c[i][j] = max(|a_i - mu|^2 + |b_j - mu|^2 - 2 (a_i - mu).(b_j - mu), 0), mu the common
 * mean the reference subtracts first.  argmin_n (nullable) = np.argmin(c, axis=1), the
 * nearest-neighbour lookup of _estimate_observation_noise (:244-252); c_nxm nullable when
 * only the argmin is wanted.  D <= 32. */
int vbmc_sq_dist(vbmc_ctx* ctx, int64_t n, int64_t m, int D, const double* a_nxD,
                 const double* b_mxD, double* c_nxm, int64_t* argmin_n);

/* ---- sampling and its Monte-Carlo consumers -------------- */

/* VariationalPosterior.sample in transformed space, Gaussian components
 * (variational_posterior/variational_posterior.py:241-363 with orig_flag=False, df=inf):
 * x_n = mu_i + (lambda o z_n) sigma_i, i from the weights (np.random.choice, :316-319) or, with
 * balance_flag, split exactly by floor(w N) plus remainder draws (:296-313).  Draws come from
 * the counter-based Philox generator keyed by (seed, n) -- not NumPy's stream -- so sample n
 * is reproducible on its own; unlike the reference the balanced samples are NOT shuffled
 * (grouped by component).  x_NxD / comp_N nullable. */
int vbmc_mixture_sample(vbmc_ctx* ctx, int64_t N, uint64_t seed, int balance_flag, double* x_NxD,
                        int32_t* comp_N);
/* The same with multivariate Student-t tails of `df` degrees of freedom (:329-340, :345-353):
 * x_n = mu_i + lambda o z_n * t_n * sigma_i with t_n = (df/2) / sqrt(G_n), G_n ~ Gamma(df/2, scale
 * df/2) (Marsaglia-Tsang on Philox stream 4, one variate per sample).  df = +inf or 0: Gaussian
 * (vbmc_mixture_sample); df < 0 -> VBMC_E_ARG (numpy's gamma raises on a negative shape too). */
int vbmc_mixture_sample_t(vbmc_ctx* ctx, int64_t N, uint64_t seed, int balance_flag, double df,
                          double* x_NxD, int32_t* comp_N);

/* VariationalPosterior.kl_div, Monte-Carlo branch (gauss_flag=False, :1107-1126), between
 * the ctx mixture (vp1) and a second mixture over the same D: N balanced samples of each
 * (seed, seed+1), both densities, the reference's zero-density replacement rules, and
 * kl_out = max(0, [KL(vp1||vp2), KL(vp2||vp1)]).  Evaluated in transformed space, which equals
 * the reference's original-space value when both posteriors share one parameter transformer
 * (the Jacobians cancel in log q2 - log q1). */
int vbmc_kl_div_mc(vbmc_ctx* ctx, int64_t N, uint64_t seed, int K2, const double* mu2_KxD,
                   const double* sigma2_K, const double* lambd2_D, const double* w2_K,
                   double kl_out[2]);

/* ---- multi-GPU: one process per GPU, one collective ---------- */

/* 128-byte RCCL unique id, created on rank 0 and shipped to the other ranks by
 * the host launcher (file / env / any side channel). */
int vbmc_comm_unique_id(uint8_t id_out[128]);
/* Join the communicator: after this, vbmc_entmc / vbmc_neg_elcbo all-reduce the
 * raw entropy accumulator over `world` ranks. */
int vbmc_comm_init(vbmc_ctx* ctx, const uint8_t id[128], int rank, int world);
int vbmc_comm_destroy(vbmc_ctx* ctx);
/* Rank and size of the communicator as RCCL reports them (ncclCommUserRank / ncclCommCount);
 * 0 and 1 without a communicator.  bench.py prints the size as `comm_world` so a multi-GPU
 * line can be checked against the ranks that really took part. */
int vbmc_comm_info(vbmc_ctx* ctx, int* rank_out, int* world_out);
/* Device-side barrier + max over ranks of a host double (bench timing). */
int vbmc_comm_allreduce_max(vbmc_ctx* ctx, double* value_inout);
int vbmc_comm_barrier(vbmc_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* VBMC_HIP_H */
