// The fused objective: one call = one evaluation of the negative ELBO
// (reference vbmc/variational_optimization.py:991-1235 _neg_elcbo) with a single
// host<->device round trip.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cmath>
#include <cstring>

#include "common.h"
#include "entropy_args.h"

// _soft_bound_loss (:645-657)
static double soft_bound_loss(const std::vector<double>& x, const double* lb, const double* ub,
                              double tol_con, std::vector<double>* dy) {
  double y = 0.0;
  if (dy) dy->assign(x.size(), 0.0);
  for (size_t i = 0; i < x.size(); ++i) {
    if (x[i] >= lb[i] && x[i] <= ub[i]) continue;  // inside the bounds: no division at all
    const double ell = (ub[i] - lb[i]) * tol_con;
    if (x[i] < lb[i]) {
      const double t = (lb[i] - x[i]) / ell;
      y += 0.5 * t * t;
      if (dy) (*dy)[i] = (x[i] - lb[i]) / (ell * ell);
    }
    if (x[i] > ub[i]) {
      const double t = (x[i] - ub[i]) / ell;
      y += 0.5 * t * t;
      if (dy) (*dy)[i] = (x[i] - ub[i]) / (ell * ell);
    }
  }
  return y;
}

#ifdef FIN_TIMES
// debug build: host steady_clock stamps (ns) of the last 64 evaluations -- entry, prep launch issued,
// all launches issued, GP word seen, main word seen, exit -- and a clock pair for correlating them
// with the kernels' wall_clock64 stamps (tools/step_times.py)
static int64_t g_host_stamps[64][6];
static uint64_t g_host_n = 0;
static inline int64_t host_ns() {
  return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
namespace {
__global__ void clock_stamp_kernel(unsigned long long* out) {
  if (threadIdx.x == 0) __hip_atomic_store(out, (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace
extern "C" int vbmc_debug_host_stamps(int64_t* out) {
  memcpy(out, g_host_stamps, sizeof(g_host_stamps));
  return (int)(g_host_n & 63);
}
// one (host ns, gpu ticks) pair: the GPU stamp is taken between the two host reads returned
extern "C" int vbmc_debug_clock_pair(vbmc_ctx* ctx, int64_t* host_before, int64_t* host_after, unsigned long long* gpu) {
  volatile unsigned long long* f = (volatile unsigned long long*)(ctx->h_done + 6);
  *f = 0;
  (void)hipStreamSynchronize(ctx->stream);
  *host_before = host_ns();
  hipLaunchKernelGGL(clock_stamp_kernel, dim3(1), dim3(64), 0, ctx->stream, (unsigned long long*)(ctx->hd_done + 6));
  while (*f == 0) __builtin_ia32_pause();
  *host_after = host_ns();
  *gpu = *f;
  return 0;
}
#define HSTAMP(i) g_host_stamps[g_host_n & 63][i] = host_ns()
#else
#define HSTAMP(i) (void)0
#endif

// ---- armed evaluation (common.h ArmedEval) ----------------------------------------------------
// One control word per evaluation out of a ring of eight.  A word is reset when evaluation seq + 8
// is armed; by then the host has seen later evaluations complete, and the queue is in order, so
// the launches of evaluation seq -- run or cancelled -- have left it.  (With two words a cancelled
// evaluation whose launches had not started yet could read the reset word of its successor, run on
// stale inputs and overwrite the speculative draws: one wrong value in ~6 000 evaluations of a soak.)
static inline uint64_t* ctl_word(vbmc_ctx* ctx, uint64_t seq) { return ctx->d_ctl + (seq & 7); }

// Cancel the queued launches of an armed evaluation: they return at once (the prep kernel on the
// cancel value of its go word, the two behind it on the same word) and the host-side bookkeeping of
// the speculative draws goes back to what it was before arming.  Does not wait for the stream.
void spec_disarm(vbmc_ctx* ctx) {
  vbmc_ctx::ArmedEval& sp = ctx->spec;
  if (!sp.armed) return;
  *(volatile uint64_t*)ctl_word(ctx, sp.seq) = ~(uint64_t)0;
  __builtin_ia32_sfence();
  sp.armed = false;
  ++sp.cancels;
  // the armed finish launch would have generated the draws of seed + 1 into the other buffer; its
  // own draws (seed) are still where the previous evaluation put them
  ctx->gen_cur = sp.gen_cur_before;
  ctx->ahead.valid = sp.ahead_before_valid;
  ctx->ahead.seed = sp.ahead_before_seed;
  ctx->ahead.buf = sp.ahead_before_buf;
  ctx->ahead.frac = sp.ahead_before_frac;
}

extern "C" int vbmc_neg_elcbo(vbmc_ctx* ctx, double* theta, int n_theta,
                              const vbmc_elbo_opts* opts, double* F, double* dF, double* G,
                              double* H, double* mu_KxD, double* sigma_K, double* lambd_D,
                              double* w_K, double* eta_K) {
  if (!ctx || !theta || !opts) return VBMC_E_ARG;
  struct KeepArmed {  // inner calls must not cancel the armed evaluation this call may be about to use
    vbmc_ctx* c;
    explicit KeepArmed(vbmc_ctx* c_) : c(c_) { c->spec.keep = true; }
    ~KeepArmed() { c->spec.keep = false; }
  } keep_armed(ctx);
  NEED_DEVICE(ctx);
  if (!ctx->mix_set) return vbmc_fail(ctx, VBMC_E_ARG, "neg_elcbo: mixture (D,K) not set");
  if (!ctx->gp.set) return vbmc_fail(ctx, VBMC_E_ARG, "neg_elcbo: GP not set");
  if (ctx->gp.D != ctx->D) return vbmc_fail(ctx, VBMC_E_ARG, "neg_elcbo: GP/mixture D mismatch");
  using clk = std::chrono::steady_clock;
  const auto t_begin = clk::now();
  HSTAMP(0);
  auto us_since = [](clk::time_point a) {
    return std::chrono::duration<double, std::micro>(clk::now() - a).count();
  };
  const int D = ctx->D, K = ctx->K;
  const int mask = opts->optimize_mask;
  const bool o_mu = mask & 1, o_sg = mask & 2, o_lm = mask & 4, o_w = mask & 8;
  // theta -> mixture -> pinned pack now; its upload is issued further down, back to back with the
  // launches that wait for it (a copy issued here would sit finished in the queue while the
  // planning below runs: 4 us of idle GPU per evaluation)
  ctx->defer_mix_upload = true;
  int rc = vbmc_theta_to_mixture(ctx, theta, n_theta, mask, mu_KxD, sigma_K, lambd_D, w_K, eta_K);
  ctx->defer_mix_upload = false;
  if (rc) return rc;
  if (o_w) {
    // the reference shifts its caller's theta tail in place (:1082-1085)
    double* e = theta + (n_theta - K);
    double mx = e[0];
    for (int k = 1; k < K; ++k) mx = e[k] > mx ? e[k] : mx;
    for (int k = 0; k < K; ++k) e[k] -= mx;
  }
  const int grad_flags = opts->compute_grad ? mask : 0;
  const GpState& g = ctx->gp;
  const int S = g.S;
  const int st = 1 + 2 * D;
  const size_t n_res = (size_t)S * K * st;
  const int n_raw = raw_len(D, K);
  const bool mc = opts->ns_per_comp > 0;
  const bool lb_dev = !mc && K > 1;

  int64_t row_begin = opts->row_begin, row_count = opts->row_count;
  if (mc) {
    if (opts->ns_per_comp & 1)
      return vbmc_fail(ctx, VBMC_E_ARG, "neg_elcbo: ns_per_comp must be even");
    const int64_t n_half = opts->ns_per_comp / 2;
    if (row_count < 0) {
      row_begin = n_half * ctx->rank / ctx->world;
      row_count = n_half * (ctx->rank + 1) / ctx->world - row_begin;
    }
    if (row_begin < 0 || row_begin + row_count > n_half)
      return vbmc_fail(ctx, VBMC_E_ARG, "neg_elcbo: bad row slice");
    if (opts->eps_mode == VBMC_EPS_RESIDENT &&
        (!ctx->d_eps || ctx->eps_K != K || ctx->eps_D != D || ctx->eps_n_half != n_half ||
         ctx->eps_row_begin != row_begin || ctx->eps_rows != row_count))
      return vbmc_fail(ctx, VBMC_E_ARG, "neg_elcbo: resident eps does not match the request");
  }

  // Results land in pinned host memory: the GP sums are written there directly by the
  // prep kernel and, on one GPU, so is the raw entropy vector by the finish kernel (no
  // D2H copy).  With a communicator the raw vector goes through device memory for the
  // all-reduce and is copied back afterwards.
  rc = ensure_pinned(ctx, n_res + (size_t)n_raw);
  if (rc) return rc;
  double* hp_dev = ctx->hp_dev;  // device-side address of the pinned block
  double* res_out = hp_dev;
  double* raw_host = hp_dev + n_res;
  static const bool force_coll = [] {
    const char* e = getenv("VBMC_FORCE_COLLECTIVE");
    return e && e[0] == '1';
  }();
  const bool multi = ctx->comm != nullptr && (ctx->world > 1 || force_coll);
  double* raw_out = raw_host;
  if (multi && mc) {
    rc = ensure_dev(ctx, &ctx->d_out, &ctx->d_out_cap, (size_t)n_raw);
    if (rc) return rc;
    raw_out = ctx->d_out;
  }

  ctx->host_us[0] = us_since(t_begin);
  const auto t_launch = clk::now();
  // The host polls completion words instead of waiting for the stream (see below); the GP part gets
  // one of its own so that G / dG are finalised while the entropy kernel runs.  With Philox draws read
  // from the ahead buffers the next evaluation's draws are generated speculatively.  With a
  // communicator the raw entropy vector is all-reduced in-stream between the finish launch and a
  // small publish launch that hands it to the host (and generates the next draws); evaluations
  // are then never armed -- whether an armed evaluation is used or cancelled depends on host timing,
  // and every rank has to queue the same collectives.
  const bool can_poll = mc;
  double* stage = nullptr;  // device staging of both result blocks: [GP sums n_res | raw entropy n_raw]
  if (can_poll) {
    rc = ensure_dev(ctx, &ctx->d_stage, &ctx->d_stage_cap, n_res + (size_t)n_raw);
    if (rc) return rc;
    stage = ctx->d_stage;
  }
  vbmc_ctx::ArmedEval& sp = ctx->spec;
  // Not for long evaluations: what arming saves is ~10 us of launch latency, under 1 % of a step of a
  // millisecond, while a queued prep kernel that nobody releases keeps the device busy until its
  // time-out -- which every device-wide wait elsewhere in the process (another library's hipFree,
  // process teardown) then has to sit through.  So: no arming after an evaluation of more than
  // ARM_MAX_EVAL_US, and the wait is capped whatever the evaluation took.
  // And not on a device this context shares: a prep kernel that polls for its go word holds workgroup slots and makes
  // every device-wide wait of another user sit through its time-out.  By default (elbo_arm = 1) evaluations are armed
  // only while this is the process's only context on the device; elbo_arm = 2 arms regardless (a user who knows the GPU
  // is theirs although several contexts exist), 0 never.  Other processes cannot be seen from here: for them the
  // device-side wait is short -- ARM_DEVICE_WAIT_MS, 30 host turnarounds -- where rounds 2-4 waited 2 to 5 ms.
  constexpr double ARM_MAX_EVAL_US = 1000.0, ARM_LIMIT_MS = 0.3, ARM_DEVICE_WAIT_MS = 0.5;
  const bool arm_allowed = ctx->opt_elbo_arm >= 2 || (ctx->opt_elbo_arm == 1 && vbmc_live_contexts_on(ctx->device) <= 1);
  const bool arm_next = can_poll && !multi && arm_allowed && ctx->opt_mix_bar && !ctx->timing &&
                        opts->eps_mode == VBMC_EPS_PHILOX && ctx->host_us[4] <= ARM_MAX_EVAL_US;
  // The host's clock for an armed evaluation starts when it is queued, i.e. at the start of THIS evaluation; the armed
  // prep kernel's own clock starts when it starts to run, i.e. when this evaluation's launches are through.  The host
  // therefore uses an armed evaluation within (this evaluation's expected duration + ARM_LIMIT_MS) of arming -- the
  // optimiser comes back 20-100 us after the result -- and treats a go word written later than ARM_LATE_SLACK_MS after
  // that as late (drain and re-evaluate); the device gives up ARM_DEVICE_WAIT_MS after it started waiting.  Should this
  // evaluation end much earlier than the last one did, the device may give up first: it says so (the dead word) and the
  // evaluation is redone unarmed -- slower, never wrong.
  constexpr double ARM_LATE_SLACK_MS = 0.15;
  const double arm_limit_ms = 1e-3 * ctx->host_us[4] + ARM_LIMIT_MS;
  static_assert(ARM_LIMIT_MS + ARM_LATE_SLACK_MS < ARM_DEVICE_WAIT_MS, "the host's cut-off must come before the device's");

  // Plan and queue the launches of ONE evaluation with Philox seed `seed`.  spin = false: for this
  // call's theta (the pack is in ctx->h_pack); spin = true: armed -- the prep kernel waits for the
  // go word the NEXT call writes together with its pack.  Nothing here depends on theta.
  bool polled = false;
  uint64_t cur_seq = 0;
  bool ident_out_flag = false;  // the launches just issued carry an identity (set by issue)
  auto issue = [&](uint64_t seed, bool spin, bool& polled_out, uint64_t& seq_out) -> int {
    int rc2;
    PrepArgs pa;
    glj_fill_prep(ctx, grad_flags != 0, res_out, nullptr, pa);
    EntPlan plan;
    if (mc) {
      // (the GP sums may ride in this launch's spare workgroup slots: only in the polled single-GPU step, see below)
      // (the plan reserves the slots by the step's GP item count whether or not they ride -- only in the polled
      // single-GPU step with a CPU-written pack, below: the partition of the batches must not depend on that)
      rc2 = entmc_plan(ctx, opts->ns_per_comp, opts->eps_mode, seed, row_begin, row_count, grad_flags != 0, plan,
                       pa.n_glj);
      if (rc2) return rc2;
      entmc_fill_prep(ctx, plan, pa);
      rc2 = entmc_pregen(ctx, plan, pa);  // Philox draws generated by extra blocks of the prep launch
      if (rc2) return rc2;
    }
    const bool ahead_ok = can_poll && opts->eps_mode == VBMC_EPS_PHILOX && ctx->opt_elbo_ahead &&
                          plan.a.eps != nullptr && plan.a.eps == ctx->d_epsgen[ctx->gen_cur];
    if (spin && !(ahead_ok && plan.pregen_hit && plan.ws && !entmc_small_applies(plan.a, plan.DP)))
      return -1000;  // (not a shape to arm: the caller restores the bookkeeping)
    if (can_poll) {
      pa.done.cnt = ctx->d_done_cnt + 8;
      pa.done.flag = ctx->hd_done + 4;
      pa.done.seq = ++ctx->done_seq;  // the finish kernel publishes the same number to its own word
      // results leave through device memory: the last workgroup copies them to the pinned block in
      // coalesced stores (DoneSignal; single 8-byte stores are one PCIe write each)
      pa.done.host_out = res_out;
      pa.done.host_n = (int)n_res;
      pa.res = stage;
    }
    seq_out = ctx->done_seq;
    ident_out_flag = false;
    ctx->gp_where = 0;
    // Polled step: the CPU writes the pack into device memory itself (no upload launch; the prep
    // launch copies it on for the later kernels).
    PrepArgs gp_tail;
    double* fg = nullptr;
    if (can_poll && ctx->opt_mix_bar) fg = spin ? ctx->d_mix_fg : write_pack_to_device(ctx);
    if (spin && !fg) return -1000;
    const bool ident = fg != nullptr && mc;  // self-identifying results (DoneSignal): the copy block exists
    if (ident) pa.ident_out = (uint64_t*)(ctx->d_done_cnt + 12);
    if (fg) {
      // Where the GP sums run: in spare workgroup slots of the entropy launch when it has them (their 7 us
      // latency chain then runs beside the entropy kernel and the host finalises G / dG meanwhile), else in the
      // prep launch (measured at config 5, whose chunk grid is two rounds: prep placement 215 us, finish
      // placement 225 us -- the finish placement was dropped in round 4).
      // (Few slots carry them: beyond ~6 items per slot they would outlast the entropy kernel -- S = 4 hyper-parameter
      // samples took 186 us per step against 114 with the prep placement -- so larger S keeps the prep launch.)
      bool in_ws = false;
      if (plan.a.sp.cus > 0) {
        in_ws = plan.gp_in_ws && pa.n_glj > 0;  // span mode: the plan reserved the slots
      } else if (pa.n_glj > 0 && plan.ws && !entmc_small_applies(plan.a, plan.DP)) {
        const int cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
        const int slots = cus * ws_min_waves(plan.DP, ws_ktmax_for(K), grad_flags != 0);
        const int free_slots = slots - K * plan.a.chunks;
        in_ws = free_slots >= plan.a.chunks && pa.n_glj <= 6 * plan.a.chunks;
      }
      if (pa.n_glj > 0 && in_ws) {
        gp_tail = pa;
        gp_tail.n_table = 0;
        gp_tail.gen = GenSlice();
        gp_tail.mix = ctx->d_mix;
        pa.n_glj = 0;
        plan.a.gp = gp_tail;
        plan.a.gp_items = gp_tail.n_glj;
        ctx->gp_where = 2;
      }
      if (mc && pa.n_glj > 0 && pa.done.flag != nullptr && ctx->opt_gp_ship && entmc_uses_mfma(ctx, plan)) {
        // GP sums in the prep launch in front of the matrix-pipe kernel (config 5's shape): their hand-over to pinned memory
        // and the word ride in the entropy launch (EntArgs::ship_*): the prep launch ends with the sums, not with the 6 us
        // of its last block's PCIe copy
        plan.a.ship_src = pa.res;
        plan.a.ship_dst = pa.done.host_out;
        plan.a.ship_n = pa.done.host_n;
        plan.a.ship_flag = pa.done.flag;
        plan.a.ship_seq = pa.done.seq;
        pa.done = DoneSignal();  // plain stores to device memory, complete at the kernel boundary
      }
      pa.mix = fg;
      pa.mix_copy = ctx->d_mix;
      pa.mix_copy_n = ctx->ml.total;
    } else {
      rc2 = upload_packed_mixture(ctx);
      if (rc2) return rc2;
    }
    uint64_t* ctl = nullptr;
    if (spin) {
      ctl = ctl_word(ctx, seq_out);
      *(volatile uint64_t*)ctl = 0;  // (its previous user, eight evaluations back, has left the queue)
      __builtin_ia32_sfence();
      pa.go = ctl;
      pa.go_seq = seq_out;
      pa.go_timeout = (uint64_t)(ARM_DEVICE_WAIT_MS * 1e5);  // ticks of 10 ns
      pa.dead = ctx->hd_done + 5;
      plan.a.cancel = ctl;
    }
    HSTAMP(1);
    rc2 = launch_prep(ctx, pa);  // (j,k) table rows (+ GP sums, + the draws when they are not ahead), one launch
    if (rc2) return rc2;
    polled_out = false;
    if (mc) {
      rc2 = entmc_launch_main(ctx, plan);
      if (rc2) return rc2;
      // Philox draws: the next evaluation's (seed + 1) are generated by spare workgroups of this
      // evaluation's last launch, while the host finalises, returns and comes back with the next
      // theta.  The host therefore does not wait for the stream: the finish launch's last result
      // workgroup stores a sequence number into pinned memory and the host polls that word.
      GenSlice ahead_gen;
      DoneSignal done;
      if (can_poll) {
        if (ahead_ok) ahead_gen = entmc_ahead_slice(ctx, plan);
        done.cnt = ctx->d_done_cnt;
        done.sub = ctx->d_done_sub;
        {
          static const int sub_min = [] { const char* e = getenv("VBMC_FIN_SUB_MIN"); return e ? atoi(e) : 256; }();  // measurement aid
          done.sub_min = sub_min;
        }
        done.flag = ctx->hd_done;
        done.seq = seq_out;
        done.host_out = raw_out;
        done.host_n = n_raw;
        done.cancel = ctl;
        if (ident) {
          done.ident_src = (const uint64_t*)(ctx->d_done_cnt + 12);
          done.ident_dst = ctx->hd_done + 1;
          done.ident_seed = seed;
        }
        polled_out = true;
        ident_out_flag = ident;
      }
      // the speculative generation runs in spare workgroups of the finish launch itself (a launch of its own behind
      // it: +1.5-3 us per step; a stream of its own: +8-12 us -- both measured in round 2 and dropped in round 4)
      const GenSlice* gen_here = ahead_gen.n_blocks > 0 ? &ahead_gen : nullptr;
      if (!multi) {
        rc2 = entmc_launch_finish(ctx, plan, polled_out ? stage + n_res : raw_out, gen_here, polled_out ? &done : nullptr);
        if (rc2) return rc2;
      } else {
        // finish (this rank's rows) -> all-reduce -> publish (+ the next draws): raw_out is device memory
        rc2 = entmc_launch_finish(ctx, plan, raw_out, nullptr, nullptr);
        if (rc2) return rc2;
        rc2 = comm_allreduce_sum(ctx, raw_out, n_raw);
        if (rc2) return rc2;
        if (polled_out) {
          done.host_out = raw_host;
          rc2 = entmc_launch_publish(ctx, raw_out, done, gen_here);
          if (rc2) return rc2;
        } else {
          HIP_TRY(ctx, hipMemcpyAsync(ctx->h_pinned + n_res, raw_out, sizeof(double) * n_raw,
                                      hipMemcpyDeviceToHost, ctx->stream));
        }
      }
    } else if (lb_dev) {
      rc2 = launch_entlb(ctx, raw_host);
      if (rc2) return rc2;
    }
    return 0;
  };

  // What the result block must identify as (DoneSignal): the pack this call writes and its seed.
  // (taken further down, once the pack and the go word have left: it is compared when the results are back)
  uint64_t want_ck = 0;
  bool check_ident = false;
  // An armed evaluation planned for exactly this call?  Then its launches are already queued: write
  // the pack and the go word.  Otherwise cancel it (if any) and launch as usual.
  bool used_armed = false;
  if (sp.armed) {
    const bool match = can_poll && sp.seed == opts->seed && sp.n_theta == n_theta && sp.mask == mask &&
                       sp.grad_flags == grad_flags && sp.eps_mode == opts->eps_mode &&
                       sp.ns_per_comp == opts->ns_per_comp && sp.row_begin == row_begin && sp.row_count == row_count &&
                       !ctx->timing &&
                       std::chrono::duration<double, std::milli>(clk::now() - sp.t_armed).count() < sp.limit_ms;
    if (match) {
      memcpy(ctx->d_mix_fg, ctx->h_pack, sizeof(double) * (size_t)ctx->ml.total);
      __builtin_ia32_sfence();  // the pack before the go word (write-combined stores are not ordered otherwise)
      *(volatile uint64_t*)ctl_word(ctx, sp.seq) = sp.seq;
      __builtin_ia32_sfence();
      ctx->pack_valid = true;
      sp.armed = false;
      const bool force_late = ctx->opt_arm_late_test > 0 && --ctx->opt_arm_late_test == 0;  // test hook
      if (!force_late && std::chrono::duration<double, std::milli>(clk::now() - sp.t_armed).count() < sp.limit_ms + ARM_LATE_SLACK_MS) {
        ++sp.hits;
        used_armed = true;
        polled = true;
        cur_seq = sp.seq;
        check_ident = sp.ident;
        HSTAMP(1);
      } else {
        // This thread lost the CPU between the age check and the go word: the prep kernel's
        // time-out may have fired around the same moment.  Whatever ran is discarded: drain the
        // queue, clear the completion counters, put the speculative-draw bookkeeping back to what it
        // was before arming (still true: that evaluation only READ the draws it was planned on)
        // and evaluate as usual.
        *(volatile uint64_t*)ctl_word(ctx, sp.seq) = ~(uint64_t)0;
        __builtin_ia32_sfence();
        HIP_TRY(ctx, stream_wait(ctx));
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_done_cnt, 0, sizeof(int) * 16, ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_done_sub, 0, sizeof(int) * 16 * 64, ctx->stream));
        ctx->gen_cur = sp.gen_cur_before;
        ctx->ahead.valid = sp.ahead_before_valid;
        ctx->ahead.seed = sp.ahead_before_seed;
        ctx->ahead.buf = sp.ahead_before_buf;
        ctx->ahead.frac = sp.ahead_before_frac;
        ++sp.cancels;
        ++sp.late;
      }
    } else {
      spec_disarm(ctx);
    }
  }
  if (!used_armed) {
    rc = issue(opts->seed, false, polled, cur_seq);
    if (rc) return rc;
    check_ident = polled && ident_out_flag;
  }
  if (mc && can_poll && ctx->opt_mix_bar) want_ck = pack_checksum(ctx->h_pack, (size_t)ctx->ml.total);
  ctx->step_marks[2] = (double)ctx->gp_where;  // (of THIS evaluation's launches: issued just now, or armed by the previous call)
  // Arm the next evaluation (seed + 1, same shapes): its launches go into the queue now, behind this
  // one's, and wait for the next call's theta.
  if (polled && arm_next && ctx->d_ctl && ctx->d_mix_fg) {
    sp.gen_cur_before = ctx->gen_cur;
    sp.ahead_before_valid = ctx->ahead.valid;
    sp.ahead_before_seed = ctx->ahead.seed;
    sp.ahead_before_buf = ctx->ahead.buf;
    sp.ahead_before_frac = ctx->ahead.frac;
    const uint64_t seq_before = ctx->done_seq;
    bool p2 = false;
    uint64_t seq2 = 0;
    // The host's clock for this armed evaluation starts BEFORE its launches are queued: the prep
    // kernel's own time-out (2 x the limit, counted from when it starts to run) can then never fire
    // before the host's cut-off (1.5 x the limit from here).  Taken after issue() it could -- a host
    // thread that lost the CPU between the two for longer than the device waits came back, found the
    // evaluation "fresh", wrote a go word nobody was waiting for any more and polled for 5 s before
    // returning the PREVIOUS evaluation's block (found by the soak under a CPU hog, round 3).
    const auto t_arm0 = clk::now();
    const int rc2 = issue(opts->seed + 1, true, p2, seq2);
    if (rc2 == 0 && p2) {
      sp.armed = true;
      sp.ident = ident_out_flag;
      sp.limit_ms = arm_limit_ms;
      sp.t_armed = t_arm0;
      sp.seq = seq2;
      sp.seed = opts->seed + 1;
      sp.n_theta = n_theta;
      sp.mask = mask;
      sp.grad_flags = grad_flags;
      sp.eps_mode = opts->eps_mode;
      sp.ns_per_comp = opts->ns_per_comp;
      sp.row_begin = row_begin;
      sp.row_count = row_count;
    } else if (rc2 == -1000) {  // nothing was queued: undo the planning's bookkeeping
      ctx->done_seq = seq_before;
      ctx->gen_cur = sp.gen_cur_before;
      ctx->ahead.valid = sp.ahead_before_valid;
      ctx->ahead.seed = sp.ahead_before_seed;
      ctx->ahead.buf = sp.ahead_before_buf;
      ctx->ahead.frac = sp.ahead_before_frac;
    } else if (rc2 != 0) {
      return rc2;
    }
  }
  // the caller's theta-only bookkeeping (vbmc_set_release_callback): the device is at work from here on
  if (ctx->release_cb && !ctx->ident_retry) ctx->release_cb(ctx->release_cb_user);
  // ---- soft bounds and weight penalty (:1195-1229, _vp_bound_loss :537-606) ----
  // They depend on theta and the new mixture only: evaluated here, while the device works.
  ElboScratch& sc = ctx->elbo;  // reused vectors: the hot call does not allocate
  const bool has_bnd = opts->bnd_lb && opts->bnd_ub;
  double F_bnd = 0.0;
  std::vector<double>& dFb = sc.dFb;
  if (has_bnd && grad_flags) dFb.assign((size_t)n_theta, 0.0);
  if (has_bnd) {
    std::vector<double>& ext = sc.ext;
    ext.clear();
    int pos = 0;
    std::vector<double>&ln_sigma = sc.ln_sigma, &ln_lambd = sc.ln_lambd;
    ln_sigma.resize(K);
    ln_lambd.resize(D);
    if (o_mu) {
      ext.insert(ext.end(), theta, theta + D * K);
      pos = D * K;
    }
    if (o_sg) {
      for (int k = 0; k < K; ++k) ln_sigma[k] = theta[pos + k];
      pos += K;
    } else {
      for (int k = 0; k < K; ++k) ln_sigma[k] = std::log(ctx->sigma[k]);
    }
    if (o_lm) {
      for (int d = 0; d < D; ++d) ln_lambd[d] = theta[pos + d];
    } else {
      for (int d = 0; d < D; ++d) ln_lambd[d] = std::log(ctx->lambd[d]);
    }
    const int sc0 = (int)ext.size();
    if (o_sg || o_lm)
      for (int k = 0; k < K; ++k)
        for (int d = 0; d < D; ++d) ext.push_back(ln_lambd[d] + ln_sigma[k]);  // ravel('F')
    if (o_w) ext.insert(ext.end(), theta + (n_theta - K), theta + n_theta);
    if ((int)ext.size() != opts->n_bnd)
      return vbmc_fail(ctx, VBMC_E_ARG, "neg_elcbo: bounds length %d != %d", opts->n_bnd,
                       (int)ext.size());
    std::vector<double>& dL = sc.dL;
    const double L = soft_bound_loss(ext, opts->bnd_lb, opts->bnd_ub, opts->tol_con,
                                     grad_flags ? &dL : nullptr);
    F_bnd += L;
    if (grad_flags) {
      int q = 0;
      if (o_mu) {
        for (int i = 0; i < D * K; ++i) dFb[q + i] += dL[i];
        q += D * K;
      }
      if (o_sg || o_lm) {
        // the reference reshapes this block C-order (D,K) (:585-587); restated as-is
        if (o_sg) {
          for (int k = 0; k < K; ++k) {
            double a = 0.0;
            for (int d = 0; d < D; ++d) a += dL[sc0 + d * K + k];
            dFb[q + k] += a;
          }
          q += K;
        }
        if (o_lm) {
          for (int d = 0; d < D; ++d) {
            double a = 0.0;
            for (int k = 0; k < K; ++k) a += dL[sc0 + d * K + k];
            dFb[q + d] += a;
          }
          q += D;
        }
      }
      if (o_w)
        for (int k = 0; k < K; ++k) dFb[q + k] += dL[dL.size() - K + k];
    }
    if (o_w) {
      const double th = opts->weight_threshold, pen = opts->weight_penalty;
      double a = 0.0;
      std::vector<double>& wg = sc.wpen;
      wg.resize(K);
      for (int k = 0; k < K; ++k) {
        const bool small = ctx->w[k] < th;
        a += small ? ctx->w[k] : th;
        wg[k] = small ? pen : 0.0;
      }
      F_bnd += a * pen;
      if (grad_flags) {
        std::vector<double>& jw = sc.jw;
        jw.resize(K);
        softmax_jacobian_apply(ctx, wg.data(), jw.data());
        for (int k = 0; k < K; ++k) dFb[n_theta - K + k] += jw[k];
      }
    }
  }
  // The watched GP arrays (vbmc_set_gp_watch): checksummed here, with the launches released and the
  // device at work -- off the path between two evaluations.
  bool gp_changed = false;
  if (!ctx->gp_watch_ptrs.empty()) {
    uint64_t ck = 0;
    vbmc_host_checksum(ctx->gp_watch_ptrs.data(), ctx->gp_watch_lens.data(), (int)ctx->gp_watch_ptrs.size(), &ck);
    gp_changed = ck != ctx->gp_watch_ck;
  }
  ctx->host_us[1] = us_since(t_launch);
  HSTAMP(2);
  const auto t_wait = clk::now();
  // spin on a completion word (the GPU is <= ~100 us away).  0: the word arrived; 1: the armed prep
  // kernel reported that it gave up on this evaluation (its time-out: the launches behind it returned
  // at once and nothing will ever publish this sequence number); 2: no word after 5 s (a stuck device)
  auto spin_on = [&](const volatile uint64_t* f, uint64_t want) -> int {
    const volatile uint64_t* dead = ctx->h_done + 5;
    unsigned spins = 0;
    for (;;) {
      if (*f == want) break;
      if (*dead == want) return 1;
      __builtin_ia32_pause();
      if ((++spins & 0xFFFF) == 0 && us_since(t_wait) > 5e6) return 2;
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return 0;
  };
  // Whatever this evaluation's launches produced is discarded: drain the queue (that cancels what was
  // just armed), clear the completion counters, forget the speculative draws and evaluate again,
  // unarmed.  A result block is NEVER used unless its own completion word arrived and (when the
  // launches carry one) its identity matches.
  auto redo_unarmed = [&](const char* why) -> int {
    HIP_TRY(ctx, stream_wait(ctx));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_done_cnt, 0, sizeof(int) * 16, ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_done_sub, 0, sizeof(int) * 16 * 64, ctx->stream));
    ctx->ahead.valid = false;
    if (ctx->ident_retry) return vbmc_fail(ctx, VBMC_E_HIP, "neg_elcbo: %s, twice in a row", why);
    ctx->ident_retry = true;
    const int arm_was = ctx->opt_elbo_arm;
    ctx->opt_elbo_arm = 0;
    const int rc3 = vbmc_neg_elcbo(ctx, theta, n_theta, opts, F, dF, G, H, mu_KxD, sigma_K, lambd_D, w_K, eta_K);
    ctx->opt_elbo_arm = arm_was;
    ctx->ident_retry = false;
    ctx->spec.keep = true;  // (the inner call's guard cleared it; ours clears it again on return)
    return rc3;
  };
  // ---- host finalisation, GP part (as soon as the prep launch's sums have landed) ----
  GljHost& o = sc.glj;
  double Gv = 0.0;
  std::vector<double>&dGv = sc.dG, &dHv = sc.dH;
  auto finalize_gp = [&]() -> int {
    glj_finalize(ctx, ctx->h_pinned, grad_flags != 0, o);
    for (int s = 0; s < S; ++s) Gv += o.G[s];
    Gv /= S;
    dGv.assign((size_t)n_theta, 0.0);
    if (grad_flags) {
      // average the per-sample blocks (linear), then Jacobians
      std::vector<double>&mu = sc.mu, &sg = sc.sg, &lm = sc.lm, &wg = sc.wg;
      mu.assign((size_t)K * D, 0.0);
      sg.assign(K, 0.0);
      lm.assign(D, 0.0);
      wg.assign(K, 0.0);
      for (int s = 0; s < S; ++s) {
        for (int i = 0; i < K * D; ++i) mu[i] += o.mu[(size_t)s * K * D + i] / S;
        for (int k = 0; k < K; ++k) sg[k] += o.sigma[(size_t)s * K + k] / S;
        for (int d = 0; d < D; ++d) lm[d] += o.lambd[(size_t)s * D + d] / S;
        for (int k = 0; k < K; ++k) wg[k] += o.w[(size_t)s * K + k] / S;
      }
      const int n = glj_pack(ctx, mu.data(), sg.data(), lm.data(), wg.data(), grad_flags, 1, dGv.data());
      if (n != n_theta) return vbmc_fail(ctx, VBMC_E_ARG, "neg_elcbo: gradient length %d != n_theta %d", n, n_theta);
    }
    return 0;
  };
  bool gp_done = false;
  if (polled) {
    int st_w = spin_on(ctx->h_done + 4, cur_seq);
    HSTAMP(3);
    ctx->step_marks[0] = us_since(t_begin);
    if (st_w == 0) {
      rc = finalize_gp();  // overlaps the entropy kernel
      if (rc) return rc;
      gp_done = true;
      st_w = spin_on(ctx->h_done, cur_seq);
    }
    if (st_w != 0) {
      ++sp.lost;
      return redo_unarmed(st_w == 1 ? "the armed launches gave up before the go word reached them"
                                    : "no completion word from the device after 5 s");
    }
    if (check_ident) {
      // the block that carries this sequence number says what it was computed from
      const volatile uint64_t* id = ctx->h_done + 1;
      ++sp.ident_checked;
      const bool force_bad = ctx->opt_ident_test > 0 && --ctx->opt_ident_test == 0;  // test hook
      if (force_bad || id[0] != want_ck || id[1] != opts->seed) {
        // not this evaluation's result: a stale pack or another evaluation's launches
        ++sp.ident_bad;
        return redo_unarmed("the result block does not identify as this evaluation");
      }
    }
  } else {
    HIP_TRY(ctx, stream_wait(ctx));
  }
  ctx->pack_in_flight = false;  // the pack upload precedes everything waited for
  ctx->step_marks[1] = us_since(t_begin);
  if (!polled) ctx->step_marks[0] = ctx->step_marks[1];
  ctx->host_us[2] = us_since(t_wait);
  HSTAMP(4);
  const auto t_fin = clk::now();
  if (!gp_done) {
    rc = finalize_gp();
    if (rc) return rc;
  }
  dHv.assign((size_t)n_theta, 0.0);
  double Hv = 0.0;
  if (mc || lb_dev) {
    const double* r = ctx->h_pinned + n_res;
    entropy_pack(ctx, r[0], r + 1, r + 1 + D * K, r + 1 + D * K + K, r + 1 + D * K + K + D,
                 grad_flags, 1, &Hv, grad_flags ? dHv.data() : nullptr);
  } else {
    rc = vbmc_entlb(ctx, grad_flags, 1, &Hv, grad_flags ? dHv.data() : nullptr);  // K == 1 closed form
    if (rc) return rc;
  }
  ctx->last_raw_off = n_res;
  ctx->last_raw_n = mc ? n_raw : 0;
  double Fv = -Gv - Hv;
  std::vector<double>& dFv = sc.dF;
  dFv.assign((size_t)n_theta, 0.0);
  if (grad_flags)
    for (int i = 0; i < n_theta; ++i) dFv[i] = -dGv[i] - dHv[i];

  // soft bounds and weight penalty: computed while the device was busy (above)
  Fv += F_bnd;
  if (grad_flags && has_bnd)
    for (int i = 0; i < n_theta; ++i) dFv[i] += dFb[i];
  ctx->host_us[3] = us_since(t_fin);
  ctx->host_us[4] = us_since(t_begin);
#ifdef FIN_TIMES
  HSTAMP(5);
  ++g_host_n;
#endif
  if (F) *F = Fv;
  if (G) *G = Gv;
  if (H) *H = Hv;
  if (dF && grad_flags) memcpy(dF, dFv.data(), sizeof(double) * n_theta);
  return gp_changed ? VBMC_W_GP_CHANGED : VBMC_OK;
}

extern "C" int vbmc_neg_elcbo_call(vbmc_ctx* ctx, const vbmc_elbo_call* c) {
  if (!c) return VBMC_E_ARG;
  return vbmc_neg_elcbo(ctx, c->theta, c->n_theta, c->opts, c->F, c->dF, c->G, c->H, c->mu_KxD, c->sigma_K, c->lambd_D, c->w_K,
                        c->eta_K);
}

extern "C" int vbmc_armed_stats(const vbmc_ctx* ctx, uint64_t out[6]) {
  if (!ctx || !out) return VBMC_E_ARG;
  const vbmc_ctx::ArmedEval& sp = ctx->spec;
  out[0] = sp.hits;
  out[1] = sp.cancels;
  out[2] = sp.late;
  out[3] = sp.ident_checked;
  out[4] = sp.ident_bad;
  out[5] = sp.lost;
  return VBMC_OK;
}

extern "C" int vbmc_last_elbo_raw(const vbmc_ctx* ctx, double* out, int n) {
  if (!ctx || !out || ctx->last_raw_n <= 0 || n != ctx->last_raw_n || !ctx->h_pinned) return VBMC_E_ARG;
  memcpy(out, ctx->h_pinned + ctx->last_raw_off, sizeof(double) * (size_t)n);
  return VBMC_OK;
}

extern "C" int vbmc_set_release_callback(vbmc_ctx* ctx, void (*fn)(void*), void* user) {
  if (!ctx) return VBMC_E_ARG;
  ctx->release_cb = fn;
  ctx->release_cb_user = user;
  return VBMC_OK;
}

extern "C" int vbmc_last_step_marks(const vbmc_ctx* ctx, double out[4]) {
  if (!ctx || !out) return VBMC_E_ARG;
  for (int i = 0; i < 4; ++i) out[i] = ctx->step_marks[i];
  return VBMC_OK;
}

extern "C" int vbmc_last_host_us(const vbmc_ctx* ctx, double out[5]) {
  if (!ctx || !out) return VBMC_E_ARG;
  for (int i = 0; i < 5; ++i) out[i] = ctx->host_us[i];
  return VBMC_OK;
}
