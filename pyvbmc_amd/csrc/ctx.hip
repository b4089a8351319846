// Context, device memory and mixture/eps/GP state upload for libvbmc_hip.so.
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstring>

#include <sched.h>

#include <cctype>
#include <cstdio>

#include "common.h"

thread_local std::string g_create_err;

int vbmc_fail(vbmc_ctx* ctx, int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (ctx)
    ctx->err = buf;
  else
    g_create_err = buf;
  return code;
}

int ensure_dev(vbmc_ctx* ctx, double** p, size_t* cap, size_t n) {
  if (*cap >= n && *p) return 0;
  // never free under a running kernel: the stream is in-order, sync first
  if (*p) {
    HIP_TRY(ctx, stream_wait(ctx));
    HIP_TRY(ctx, hipFree(*p));
    *p = nullptr;
    *cap = 0;
  }
  size_t want = n + n / 4 + 64;
  HIP_TRY(ctx, hipMalloc((void**)p, want * sizeof(double)));
  *cap = want;
  return 0;
}

int ensure_pinned(vbmc_ctx* ctx, size_t n) {
  if (ctx->h_pinned_cap >= n && ctx->h_pinned) return 0;
  if (ctx->h_pinned) {
    HIP_TRY(ctx, stream_wait(ctx));
    HIP_TRY(ctx, hipHostFree(ctx->h_pinned));
    ctx->h_pinned = nullptr;
    ctx->h_pinned_cap = 0;
  }
  size_t want = n + n / 4 + 64;
  HIP_TRY(ctx, hipHostMalloc((void**)&ctx->h_pinned, want * sizeof(double), hipHostMallocDefault));
  ctx->h_pinned_cap = want;
  HIP_TRY(ctx, hipHostGetDevicePointer((void**)&ctx->hp_dev, ctx->h_pinned, 0));
  return 0;
}

// live contexts per device in this process (the armed evaluation is for a context that has its device to itself)
#include <atomic>
static std::atomic<int> g_live_ctx[64];
int vbmc_live_contexts_on(int device) { return device >= 0 && device < 64 ? g_live_ctx[device].load() : 2; }

static void options_from_env(vbmc_ctx* c) {
  const char* e = getenv("VBMC_ENTMC_KERNEL");
  c->opt_entmc_valu = (e && e[0] == 'v') ? 1 : 0;  // VBMC_ENTMC_KERNEL=valu
  e = getenv("VBMC_ELBO_PREGEN");
  c->opt_elbo_pregen = !(e && e[0] == '0');
  e = getenv("VBMC_ELBO_AHEAD");
  c->opt_elbo_ahead = !(e && e[0] == '0');
  e = getenv("VBMC_ELBO_ARM");
  c->opt_elbo_arm = e ? atoi(e) : 1;  // 0 off; 1 (default) when this context is the only one on its device; 2 always
  e = getenv("VBMC_ACQ_POLL");
  c->opt_acq_poll = !(e && e[0] == '0');
  e = getenv("VBMC_ADAM_FUSED");
  c->opt_adam_fused = e ? atoi(e) : 1;  // (3: the release / acquire form of its exchange)
  e = getenv("VBMC_RANDN_DEVICE");
  c->opt_randn_dev = e ? (e[0] == '0' ? 0 : e[0] == '2' ? 2 : 1) : 1;  // VBMC_RANDN_DEVICE=0: strict parity with np.random.randn (host generator)
  e = getenv("VBMC_ADAM_TAIL");
  c->opt_adam_tail = e ? atoi(e) : 1;  // (2: wherever its shape applies, whatever the job's size)
  e = getenv("VBMC_WS_SPAN");
  c->opt_ws_span = !(e && e[0] == '0');
  e = getenv("VBMC_WS_FRONT");
  if (e) c->opt_ws_front = atoi(e);
  e = getenv("VBMC_WS_PAD");
  if (e) c->opt_ws_pad = atoi(e);
  e = getenv("VBMC_MIX_BAR");
  c->opt_mix_bar = !(e && e[0] == '0');
  e = getenv("VBMC_PREDICT_DMA");
  c->opt_predict_dma = !(e && e[0] == '0');
  e = getenv("VBMC_PREDICT_FUSED_FINISH");
  if (e) c->opt_predict_fused = atoi(e);
  e = getenv("VBMC_GP_SHIP");
  c->opt_gp_ship = !(e && e[0] == '0');
}

extern "C" {

int vbmc_abi_version(void) { return 2; }

int vbmc_device_count(int* n_out) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    n = 0;
    (void)hipGetLastError();
  }
  if (n_out) *n_out = n;
  return VBMC_OK;
}

}  // extern "C"

// The host side of the polled step is a latency chain over PCIe (pack and go word out, completion word and results back,
// host-visible device memory written by the CPU): on the two-socket hosts of an 8 x MI355X node it matters which socket
// the calling thread runs on -- BASELINE config 3's step, same box, alternating: 85.3 us from the GPU's own NUMA node,
// 87.8 us from the other one, 85.7-89.5 us wherever the scheduler puts an unpinned thread (profiles/r06_notes.md).
// So a context NARROWS the calling thread's affinity to the CPUs local to its device
// (/sys/bus/pci/devices/<bus id>/local_cpulist), before it allocates its pinned buffers (first touch then lands them on
// that node too).  It only ever removes CPUs from the thread's current set -- a set the user already confined to the
// device's node, or to CPUs that are all on the other node, is left alone -- and VBMC_HOST_AFFINITY=0 switches it off.
// One process per GPU (the multi-GPU layout) thereby gets the usual per-rank NUMA binding without a launcher's help.
static void bind_host_thread(vbmc_ctx* ctx) {
  const char* e = getenv("VBMC_HOST_AFFINITY");
  if (e && e[0] == '0') return;
  char bus[64] = {0};
  if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus) - 1, ctx->device) != hipSuccess) {
    (void)hipGetLastError();
    return;
  }
  for (char* c = bus; *c; ++c) *c = (char)tolower((unsigned char)*c);
  char path[160];
  snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/local_cpulist", bus);
  FILE* f = fopen(path, "r");
  if (!f) return;
  char list[1024] = {0};
  const bool got = fgets(list, sizeof(list), f) != nullptr;
  fclose(f);
  if (!got) return;
  cpu_set_t local, cur, both;
  CPU_ZERO(&local);
  for (char* p = list; *p;) {  // "0-63,128-191"
    char* end = nullptr;
    const long a = strtol(p, &end, 10);
    if (end == p) break;
    long b = a;
    p = end;
    if (*p == '-') {
      b = strtol(p + 1, &end, 10);
      if (end == p + 1) break;
      p = end;
    }
    for (long c = a; c <= b && c < CPU_SETSIZE; ++c)
      if (c >= 0) CPU_SET((int)c, &local);
    while (*p == ',' || *p == ' ' || *p == '\n') ++p;
  }
  if (CPU_COUNT(&local) == 0 || sched_getaffinity(0, sizeof(cur), &cur) != 0) return;
  CPU_AND(&both, &cur, &local);
  const int n_both = CPU_COUNT(&both), n_cur = CPU_COUNT(&cur);
  ctx->host_cpus = n_cur;
  if (n_both == 0 || n_both == n_cur) return;  // all on the other node (the user's choice), or local already
  if (sched_setaffinity(0, sizeof(both), &both) == 0) {
    ctx->host_bound = true;
    ctx->host_cpus = n_both;
  }
}

extern "C" {

int vbmc_host_affinity(const vbmc_ctx* ctx, int* bound_out, int* n_cpus_out) {
  if (!ctx) return VBMC_E_ARG;
  if (bound_out) *bound_out = ctx->host_bound ? 1 : 0;
  if (n_cpus_out) *n_cpus_out = ctx->host_cpus;
  return VBMC_OK;
}

int vbmc_ctx_create(int device_id, vbmc_ctx** out) {
  if (!out) return vbmc_fail(nullptr, VBMC_E_ARG, "vbmc_ctx_create: out is NULL");
  *out = nullptr;
  if (device_id == -1) {
    // host-only context: mixture bookkeeping and host finalisation only (used by the
    // CPU tests of the sharded path); every kernel-launching entry point refuses it.
    vbmc_ctx* h = new vbmc_ctx();
    h->device = -1;
    options_from_env(h);
    *out = h;
    return VBMC_OK;
  }
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    return vbmc_fail(nullptr, VBMC_E_NODEV,
                     "no HIP device visible (%s); libvbmc_hip has no CPU fallback",
                     e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  }
  if (device_id < 0 || device_id >= n)
    return vbmc_fail(nullptr, VBMC_E_ARG, "device_id %d out of range [0,%d)", device_id, n);
  vbmc_ctx* ctx = new vbmc_ctx();
  ctx->device = device_id;
  options_from_env(ctx);
  e = hipSetDevice(device_id);
  if (e == hipSuccess) e = hipGetDeviceProperties(&ctx->prop, device_id);
  if (e == hipSuccess) bind_host_thread(ctx);  // (before the pinned allocations below: they follow the thread's node)
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
  for (int i = 0; i < 12 && e == hipSuccess; ++i) e = hipEventCreate(&ctx->ev[i]);
  if (e == hipSuccess) e = hipHostMalloc((void**)&ctx->h_done, 64, hipHostMallocDefault);
  if (e == hipSuccess) {
    for (int i = 0; i < 8; ++i) ctx->h_done[i] = 0;
    e = hipHostGetDevicePointer((void**)&ctx->hd_done, ctx->h_done, 0);
  }
  if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_done_cnt, 64);
  if (e == hipSuccess) e = hipMemset(ctx->d_done_cnt, 0, 64);
  if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_done_sub, 16 * 64 * sizeof(int));
  if (e == hipSuccess) e = hipMemset(ctx->d_done_sub, 0, 16 * 64 * sizeof(int));
  if (e != hipSuccess) {
    int rc = vbmc_fail(nullptr, VBMC_E_HIP, "context setup failed: %s", hipGetErrorString(e));
    delete ctx;
    return rc;
  }
  if (strncmp(ctx->prop.gcnArchName, "gfx950", 6) != 0) {
    int rc = vbmc_fail(nullptr, VBMC_E_NODEV, "device %d is %s; this library is built for gfx950 only",
                       device_id, ctx->prop.gcnArchName);
    vbmc_ctx_destroy(ctx);
    return rc;
  }
  if (device_id < 64) g_live_ctx[device_id].fetch_add(1);
  ctx->counted = true;
  *out = ctx;
  return VBMC_OK;
}

void vbmc_ctx_destroy(vbmc_ctx* ctx) {
  if (!ctx) return;
  if (ctx->device < 0) {
    delete ctx;
    return;
  }
  if (ctx->counted && ctx->device < 64) g_live_ctx[ctx->device].fetch_sub(1);
  (void)hipSetDevice(ctx->device);
  spec_disarm(ctx);
  if (getenv("VBMC_DEBUG_ARM")) fprintf(stderr, "[vbmc] armed evaluations: %llu used, %llu cancelled\n", (unsigned long long)ctx->spec.hits, (unsigned long long)ctx->spec.cancels);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->d_ctl) (void)hipFree(ctx->d_ctl);
  vbmc_comm_destroy(ctx);
  adam_free(ctx);
  acq_is_free(ctx);
  randn_dev_free(ctx);
  double* bufs[] = {ctx->d_mix, ctx->d_mix_fg, ctx->d_acq_fg, ctx->d_stage, ctx->d_eps, ctx->d_scratch, ctx->d_out, ctx->d_ptick, ctx->gp.d_X, ctx->gp.d_XT,
                    ctx->gp.d_alpha, ctx->gp.d_L, ctx->gp.d_Linv, ctx->gp.d_LinvP, ctx->gp.d_sW, ctx->gp.d_hyp,
                    ctx->gp.d_xc, ctx->gp.d_smeta};
  for (double* b : bufs)
    if (b) (void)hipFree(b);
  for (double* b : ctx->d_epsgen)
    if (b) (void)hipFree(b);
  if (ctx->h_done) (void)hipHostFree(ctx->h_done);
  if (ctx->d_done_cnt) (void)hipFree(ctx->d_done_cnt);
  if (ctx->d_done_sub) (void)hipFree(ctx->d_done_sub);
  if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
  if (ctx->h_eps) (void)hipHostFree(ctx->h_eps);
  if (ctx->h_pack) (void)hipHostFree(ctx->h_pack);

  for (hipEvent_t e : ctx->ev)
    if (e) (void)hipEventDestroy(e);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char* vbmc_last_error(const vbmc_ctx* ctx) {
  return ctx ? ctx->err.c_str() : g_create_err.c_str();
}

int vbmc_device_info(const vbmc_ctx* ctx, char* name, int name_len, int* cu_count,
                     int* clock_khz, uint64_t* hbm_bytes) {
  if (!ctx) return VBMC_E_ARG;
  if (ctx->device < 0) return vbmc_fail(const_cast<vbmc_ctx*>(ctx), VBMC_E_NODEV, "host-only context");
  if (name && name_len > 0) {
    snprintf(name, (size_t)name_len, "%s (%s)", ctx->prop.name, ctx->prop.gcnArchName);
  }
  if (cu_count) *cu_count = ctx->prop.multiProcessorCount;
  if (clock_khz) *clock_khz = ctx->prop.clockRate;
  if (hbm_bytes) *hbm_bytes = (uint64_t)ctx->prop.totalGlobalMem;
  return VBMC_OK;
}

int vbmc_synchronize(vbmc_ctx* ctx) {
  if (!ctx) return VBMC_E_ARG;
  if (ctx->device < 0) return VBMC_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, stream_wait(ctx));
  return VBMC_OK;
}

int vbmc_set_option(vbmc_ctx* ctx, const char* key, int value) {
  if (!ctx || !key) return VBMC_E_ARG;
  if (!strcmp(key, "arm_late_test")) {  // test hook of the armed evaluation's recovery path: must not cancel it
    ctx->opt_arm_late_test = value;
    return VBMC_OK;
  }
  if (!strcmp(key, "ident_test")) {  // test hook of the identity check's recovery path: must not cancel an armed evaluation either
    ctx->opt_ident_test = value;
    return VBMC_OK;
  }
  spec_disarm(ctx);
  if (!strcmp(key, "entmc_kernel")) ctx->opt_entmc_valu = value == 1;
  else if (!strcmp(key, "entmc_mfma")) ctx->opt_entmc_mfma = value != 0;
  else if (!strcmp(key, "gp_ship")) ctx->opt_gp_ship = value != 0;
  else if (!strcmp(key, "elbo_pregen")) ctx->opt_elbo_pregen = value != 0;
  else if (!strcmp(key, "elbo_ahead")) ctx->opt_elbo_ahead = value != 0;
  else if (!strcmp(key, "predict_dma")) ctx->opt_predict_dma = value != 0;
  else if (!strcmp(key, "predict_fused")) ctx->opt_predict_fused = value < 0 ? 0 : value > 2 ? 2 : value;
  else if (!strcmp(key, "mix_bar")) ctx->opt_mix_bar = value != 0;
  else if (!strcmp(key, "ws_span")) ctx->opt_ws_span = value != 0;
  else if (!strcmp(key, "ws_pad")) ctx->opt_ws_pad = value > 8 ? 8 : value;
  else if (!strcmp(key, "ws_front")) ctx->opt_ws_front = value < 0 ? 0 : value > 990 ? 990 : value;
  else if (!strcmp(key, "elbo_arm")) ctx->opt_elbo_arm = value < 0 ? 0 : value > 2 ? 2 : value;  // 1: only while this context is alone on its device; 2: always
  else if (!strcmp(key, "acq_poll")) ctx->opt_acq_poll = value != 0;
  else if (!strcmp(key, "randn_device")) ctx->opt_randn_dev = value < 0 ? 0 : value > 3 ? 3 : value;  // vbmc_set_eps_numpy: the NumPy stream on the device (device_randn.hip); 0 = strict parity (host generator, values bit-identical to np.random.randn); 2 = test hook: the device pass reports its margin exceeded; 3 = test hook: the window is always computed, never taken from the pass before
  else if (!strcmp(key, "adam_tail")) ctx->opt_adam_tail = value;  // the optimiser loop's two-launch iteration (adam.hip)
  else if (!strcmp(key, "adam_fused")) ctx->opt_adam_fused = value;  // 2: test hook, see FusedArgs::test_absent; 3: release / acquire flags (FusedArgs::rel_acq)
  else return vbmc_fail(ctx, VBMC_E_ARG, "vbmc_set_option: unknown key '%s'", key);
  return VBMC_OK;
}

int vbmc_set_gp_watch(vbmc_ctx* ctx, const double* const* ptrs, const int64_t* lens, int n, uint64_t expected) {
  if (!ctx || n < 0 || (n > 0 && (!ptrs || !lens))) return VBMC_E_ARG;
  ctx->gp_watch_ptrs.assign(ptrs, ptrs + n);
  ctx->gp_watch_lens.assign(lens, lens + n);
  ctx->gp_watch_ck = expected;
  return VBMC_OK;
}

int vbmc_host_checksum(const double* const* ptrs, const int64_t* lens, int n, uint64_t* out) {
  if (!ptrs || !lens || n < 0 || !out) return VBMC_E_ARG;
  uint64_t s = 0;
  for (int a = 0; a < n; ++a) {
    if (!ptrs[a] || lens[a] < 0) return VBMC_E_ARG;
    s += (pack_checksum(ptrs[a], (size_t)lens[a]) + (uint64_t)lens[a]) * (0xD1B54A32D192ED03ull + 2ull * (uint64_t)a);
  }
  *out = s;
  return VBMC_OK;
}

int vbmc_last_entmc_plan(const vbmc_ctx* ctx, int out[4]) {
  if (!ctx || !out) return VBMC_E_ARG;
  for (int i = 0; i < 4; ++i) out[i] = ctx->last_plan[i];
  return VBMC_OK;
}

int vbmc_set_timing(vbmc_ctx* ctx, int on) {
  if (!ctx) return VBMC_E_ARG;
  ctx->timing = on < 0 ? 0 : on > 2 ? 2 : on;
  return VBMC_OK;
}

int vbmc_last_kernel_ms(vbmc_ctx* ctx, int which, double* ms_out) {
  if (!ctx || which < 0 || which > 5 || !ms_out) return VBMC_E_ARG;
  NEED_DEVICE(ctx);
  if (!ctx->ev_valid[which])
    return vbmc_fail(ctx, VBMC_E_ARG, which == 5 ? "no timed launch recorded for 5 (gp_predict's product alone is timed at vbmc_set_timing(ctx, 2) only)"
                                                 : "no timed launch recorded for %d", which);
  HIP_TRY(ctx, hipEventSynchronize(ctx->ev[2 * which + 1]));
  float ms = 0.f;
  HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev[2 * which], ctx->ev[2 * which + 1]));
  *ms_out = (double)ms;
  return VBMC_OK;
}

}  // extern "C"

// The kernel-friendly mixture pack (MixLayout) of one mixture.
void write_mixture_pack(const MixLayout& ml, const double* mu_KxD, const double* sigma,
                        const double* lambd, const double* w, double* p) {
  const int D = ml.D, K = ml.K;
  double prod_lam = 1.0;
  for (int d = 0; d < D; ++d) prod_lam *= lambd[d];
  // nconst = 1 / (2 pi)^(D/2) / prod(lambda)   (entmc_vbmc.py:54-56)
  const double nconst = 1.0 / std::pow(2.0 * M_PI, 0.5 * D) / prod_lam;
  const double l2n = std::log2(nconst);
  for (int k = 0; k < K; ++k) {
    for (int d = 0; d < D; ++d) {
      p[ml.o_mu + k * D + d] = mu_KxD[(size_t)k * D + d];
      p[ml.o_mup + k * D + d] = mu_KxD[(size_t)k * D + d] / lambd[d];
    }
    const double s = sigma[k];
    // sigma^D by repeated squaring (D is a small integer): a few ulp from pow(), 20x cheaper;
    // this runs on the host inside every ELBO evaluation
    double sD = 1.0, b = s;
    for (int e = D; e > 0; e >>= 1) {
      if (e & 1) sD *= b;
      b *= b;
    }
    p[ml.o_is2 + k] = 1.0 / (s * s);
    p[ml.o_rc + k] = nconst / sD;
    p[ml.o_lrc + k] = l2n - D * std::log2(s);
    p[ml.o_wc + k] = w[k] * nconst / sD;
    p[ml.o_sig + k] = s;
    p[ml.o_w + k] = w[k];
  }
  for (int d = 0; d < D; ++d) {
    p[ml.o_lam + d] = lambd[d];
    p[ml.o_ilam + d] = 1.0 / lambd[d];
  }
}

// VariationalPosterior.set_parameters (raw) on plain arrays (variational_posterior.py:680-759)
// plus eta = theta[-K:] - max (variational_optimization.py:1082-1085).  In/out arrays start
// from the current attribute values; blocks whose optimize bit is clear are left alone.
int theta_to_arrays(int D, int K, const double* theta, int n_theta, int optimize_mask, double* mu,
                    double* sg, double* lm, double* w, double* eta) {
  const bool o_mu = optimize_mask & 1, o_sg = optimize_mask & 2, o_lm = optimize_mask & 4,
             o_w = optimize_mask & 8;
  const int need = (o_mu ? D * K : 0) + (o_sg ? K : 0) + (o_lm ? D : 0) + (o_w ? K : 0);
  if (n_theta != need) return -1;
  for (int i = 0; i < n_theta; ++i)
    if (!std::isfinite(theta[i])) return -2;
  int pos = 0;
  if (o_mu) {
    for (int i = 0; i < D * K; ++i) mu[i] = theta[i];
    pos = D * K;
  }
  if (o_sg) {
    for (int k = 0; k < K; ++k) sg[k] = std::exp(theta[pos + k]);
    pos += K;
  }
  if (o_lm)
    for (int d = 0; d < D; ++d) lm[d] = std::exp(theta[pos + d]);
  if (o_w) {
    const double* e = theta + (n_theta - K);
    double mx = e[0];
    for (int k = 1; k < K; ++k) mx = e[k] > mx ? e[k] : mx;
    for (int k = 0; k < K; ++k) {
      eta[k] = e[k] - mx;
      w[k] = std::exp(eta[k]);
    }
  }
  // lambda -> unit RMS, sigma absorbs it (variational_posterior.py:749-752)
  double s2 = 0.0;
  for (int d = 0; d < D; ++d) s2 += lm[d] * lm[d];
  const double nl = std::sqrt(s2 / D);
  for (int d = 0; d < D; ++d) lm[d] /= nl;
  for (int k = 0; k < K; ++k) sg[k] *= nl;
  if (o_w) {
    double ws = 0.0;
    for (int k = 0; k < K; ++k) ws += w[k];
    for (int k = 0; k < K; ++k) w[k] /= ws;
  }
  return 0;
}

// Host-side derivation of the kernel-friendly mixture pack (into the pinned staging buffer) ...
static int pack_mixture(vbmc_ctx* ctx) {
  const int D = ctx->D, K = ctx->K;
  ctx->ml.plan(D, K);
  const MixLayout& ml = ctx->ml;
  if (ctx->device < 0) return 0;  // host-only context keeps just the host copies
  if (ctx->h_pack_cap < (size_t)ml.total) {
    if (ctx->h_pack) {
      HIP_TRY(ctx, stream_wait(ctx));
      HIP_TRY(ctx, hipHostFree(ctx->h_pack));
      ctx->h_pack = nullptr;
    }
    const size_t want = (size_t)ml.total * 2 + 64;
    HIP_TRY(ctx, hipHostMalloc((void**)&ctx->h_pack, want * sizeof(double), hipHostMallocDefault));
    ctx->h_pack_cap = want;
  }
  if (ctx->pack_in_flight) {
    // The pinned pack may still be the source of the previous asynchronous upload.  Entry points
    // that wait for the stream anyway clear the flag (every ELBO evaluation does); this wait is
    // for back-to-back uploads only.  (No event: a record in front of the prep launch costs ~6 us.)
    HIP_TRY(ctx, stream_wait(ctx));
    ctx->pack_in_flight = false;
  }
  write_mixture_pack(ml, ctx->mu.data(), ctx->sigma.data(), ctx->lambd.data(), ctx->w.data(), ctx->h_pack);
  return ensure_dev(ctx, &ctx->d_mix, &ctx->d_mix_cap, (size_t)ml.total);
}

// ... and its upload: one asynchronous copy from pinned memory (pack_in_flight guards the buffer's
// reuse).  Split from the packing so that the fused objective can issue the copy and the launches
// that wait for it back to back, after all its planning (vbmc_neg_elcbo).
namespace {
// the pack upload as a kernel of our own: a kernel -> kernel dependency in one queue starts with no
// gap, a runtime copy -> kernel dependency costs ~4 us on this stack
__global__ void mix_upload_kernel(const double* __restrict__ src, double* __restrict__ dst, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}
}  // namespace

// The pack written by the CPU straight into (fine-grained, BAR-mapped) device memory: ~0.1 us for
// 10 KB of posted writes and no launch.  Returns the device address, or null when this stack does
// not offer such memory (the caller then uploads as usual).  The caller guarantees that no kernel
// reading the buffer is still in flight (the host-driven step has waited for the previous
// evaluation's completion word).
double* write_pack_to_device(vbmc_ctx* ctx) {
  if (ctx->device < 0 || ctx->mix_fg_failed) return nullptr;
  const size_t n = (size_t)ctx->ml.total;
  if (ctx->d_mix_fg_cap < n) {
    if (ctx->d_mix_fg) {
      (void)hipStreamSynchronize(ctx->stream);
      (void)hipFree(ctx->d_mix_fg);
      ctx->d_mix_fg = nullptr;
    }
    const size_t want = n * 2 + 64;
    int large_bar = 0;  // CPU stores into device memory need the whole of it behind the PCIe BAR
    if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, ctx->device) != hipSuccess || !large_bar) {
      (void)hipGetLastError();
      ctx->mix_fg_failed = true;
      return nullptr;
    }
    if (hipExtMallocWithFlags((void**)&ctx->d_mix_fg, want * sizeof(double), hipDeviceMallocFinegrained) != hipSuccess) {
      (void)hipGetLastError();
      ctx->d_mix_fg = nullptr;
      ctx->d_mix_fg_cap = 0;
      ctx->mix_fg_failed = true;
      return nullptr;
    }
    ctx->d_mix_fg_cap = want;
  }
  if (!ctx->d_ctl) {  // control words of the armed evaluation (common.h ArmedEval), same kind of memory
    if (hipExtMallocWithFlags((void**)&ctx->d_ctl, 64, hipDeviceMallocFinegrained) != hipSuccess) {
      (void)hipGetLastError();
      ctx->d_ctl = nullptr;
    } else {
      memset(ctx->d_ctl, 0, 64);
      __builtin_ia32_sfence();
    }
  }
  memcpy(ctx->d_mix_fg, ctx->h_pack, n * sizeof(double));
  __builtin_ia32_sfence();  // write-combined stores drained before the doorbell of the next launch
  ctx->pack_valid = true;   // (d_mix itself receives the pack from the prep launch's copy block)
  return ctx->d_mix_fg;
}

int upload_packed_mixture(vbmc_ctx* ctx) {
  if (ctx->device < 0) return 0;
  if (ctx->spec.armed) spec_disarm(ctx);  // (queued behind launches that wait for a theta it would wait with them)
  {
    // (a copy kernel of our own: hipMemcpyAsync from pinned memory measured slower on this stack)
    double* src = nullptr;
    if (ctx->h_pack_dev_of != ctx->h_pack) {
      HIP_TRY(ctx, hipHostGetDevicePointer((void**)&ctx->h_pack_dev, ctx->h_pack, 0));
      ctx->h_pack_dev_of = ctx->h_pack;
    }
    src = ctx->h_pack_dev;
    const int n = ctx->ml.total;
    hipLaunchKernelGGL(mix_upload_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, (const double*)src,
                       ctx->d_mix, n);
    HIP_TRY(ctx, hipGetLastError());
  }
  ctx->pack_in_flight = true;
  ctx->pack_valid = true;
  return 0;
}

int set_mixture_host(vbmc_ctx* ctx, int D, int K, const double* mu_KxD, const double* sigma_K,
                     const double* lambd_D, const double* w_K, const double* eta_K, bool skip_if_same) {
  if (D < 1 || K < 1) return vbmc_fail(ctx, VBMC_E_ARG, "set_mixture: bad D=%d K=%d", D, K);
  // The same mixture as the one already on the device (thousands of acquisition / pdf calls
  // between two updates of the posterior): nothing to pack or upload.  Only for the explicit
  // vbmc_set_mixture; the optimiser path (theta -> mixture) always packs and uploads, a new
  // theta every call being its normal case.
  if (skip_if_same && ctx->mix_set && ctx->pack_valid && ctx->D == D && ctx->K == K &&
      memcmp(ctx->mu.data(), mu_KxD, sizeof(double) * K * D) == 0 &&
      memcmp(ctx->sigma.data(), sigma_K, sizeof(double) * K) == 0 &&
      memcmp(ctx->lambd.data(), lambd_D, sizeof(double) * D) == 0 &&
      memcmp(ctx->w.data(), w_K, sizeof(double) * K) == 0) {
    if (eta_K) ctx->eta.assign(eta_K, eta_K + K);
    ctx->exp_eta_valid = false;
    return 0;
  }
  ctx->pack_valid = false;
  ctx->D = D;
  ctx->K = K;
  ctx->mu.assign(mu_KxD, mu_KxD + (size_t)K * D);
  ctx->sigma.assign(sigma_K, sigma_K + K);
  ctx->lambd.assign(lambd_D, lambd_D + D);
  ctx->w.assign(w_K, w_K + K);
  if (eta_K)
    ctx->eta.assign(eta_K, eta_K + K);
  else
    ctx->eta.assign((size_t)K, 0.0);
  ctx->exp_eta_valid = false;
  for (int k = 0; k < K; ++k)
    if (!(ctx->sigma[k] > 0.0) || !std::isfinite(ctx->sigma[k]))
      return vbmc_fail(ctx, VBMC_E_NONFINITE, "set_mixture: sigma[%d]=%g must be finite and > 0", k,
                       ctx->sigma[k]);
  for (int d = 0; d < D; ++d)
    if (!(ctx->lambd[d] > 0.0) || !std::isfinite(ctx->lambd[d]))
      return vbmc_fail(ctx, VBMC_E_NONFINITE, "set_mixture: lambd[%d]=%g must be finite and > 0", d,
                       ctx->lambd[d]);
  ctx->mix_set = true;
  int rc = pack_mixture(ctx);
  if (rc || ctx->defer_mix_upload) return rc;
  return upload_packed_mixture(ctx);
}

extern "C" {

int vbmc_set_mixture(vbmc_ctx* ctx, int D, int K, const double* mu_KxD, const double* sigma_K,
                     const double* lambd_D, const double* w_K, const double* eta_K) {
  if (!ctx || !mu_KxD || !sigma_K || !lambd_D || !w_K) return VBMC_E_ARG;
  if (ctx->device >= 0) HIP_TRY(ctx, hipSetDevice(ctx->device));
  return set_mixture_host(ctx, D, K, mu_KxD, sigma_K, lambd_D, w_K, eta_K, true);
}

int vbmc_set_mixture_dk(vbmc_ctx* ctx, int D, int K, const double* mu_DxK, const double* sigma_K,
                        const double* lambd_D, const double* w_K, const double* eta_K) {
  if (!ctx || !mu_DxK || !sigma_K || !lambd_D || !w_K || D < 1 || K < 1) return VBMC_E_ARG;
  if (ctx->device >= 0) HIP_TRY(ctx, hipSetDevice(ctx->device));
  // (D, K) -> K rows of D: the transposition the Python mirror used to make with NumPy on every pdf / acquisition call
  std::vector<double>& t = ctx->mu_scratch;
  t.resize((size_t)D * K);
  for (int d = 0; d < D; ++d)
    for (int k = 0; k < K; ++k) t[(size_t)k * D + d] = mu_DxK[(size_t)d * K + k];
  return set_mixture_host(ctx, D, K, t.data(), sigma_K, lambd_D, w_K, eta_K, true);
}

int vbmc_theta_to_mixture(vbmc_ctx* ctx, const double* theta, int n_theta, int optimize_mask,
                          double* mu_KxD, double* sigma_K, double* lambd_D, double* w_K,
                          double* eta_K) {
  if (!ctx || !theta) return VBMC_E_ARG;
  if (!ctx->mix_set) return vbmc_fail(ctx, VBMC_E_ARG, "theta_to_mixture: mixture (D,K) not set");
  if (ctx->device >= 0) HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int D = ctx->D, K = ctx->K;
  // (scratch that lives with the context: this runs between two evaluations of the optimiser, with the device idle)
  std::vector<double>&mu = ctx->t2m_mu, &sg = ctx->t2m_sg, &lm = ctx->t2m_lm, &w = ctx->t2m_w, &eta = ctx->t2m_eta;
  mu = ctx->mu;
  sg = ctx->sigma;
  lm = ctx->lambd;
  w = ctx->w;
  eta = ctx->eta;
  const int st = theta_to_arrays(D, K, theta, n_theta, optimize_mask, mu.data(), sg.data(), lm.data(),
                                 w.data(), eta.data());
  if (st == -1)
    return vbmc_fail(ctx, VBMC_E_ARG, "theta length %d does not match D=%d K=%d mask=%d", n_theta, D, K,
                     optimize_mask);
  if (st == -2) return vbmc_fail(ctx, VBMC_E_NONFINITE, "theta has a non-finite entry");
  int rc = set_mixture_host(ctx, D, K, mu.data(), sg.data(), lm.data(), w.data(), eta.data(), false);
  if (rc) return rc;
  if (mu_KxD) memcpy(mu_KxD, mu.data(), sizeof(double) * D * K);
  if (sigma_K) memcpy(sigma_K, sg.data(), sizeof(double) * K);
  if (lambd_D) memcpy(lambd_D, lm.data(), sizeof(double) * D);
  if (w_K) memcpy(w_K, w.data(), sizeof(double) * K);
  if (eta_K) memcpy(eta_K, eta.data(), sizeof(double) * K);
  return VBMC_OK;
}

static int upload_eps(vbmc_ctx* ctx, int K, int64_t n_half, int D, const double* eps_half, int64_t row_begin,
                      int64_t row_count) {
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  size_t n = (size_t)K * (size_t)row_count * (size_t)D;
  int rc = ensure_dev(ctx, &ctx->d_eps, &ctx->d_eps_cap, n ? n : 1);
  if (rc) return rc;
  if (row_begin == 0 && row_count == n_half && n > 0) {  // the whole job: one contiguous block
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_eps, eps_half, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
  } else {
    for (int j = 0; j < K && row_count > 0; ++j) {
      const double* src = eps_half + ((size_t)j * (size_t)n_half + (size_t)row_begin) * D;
      double* dst = ctx->d_eps + (size_t)j * (size_t)row_count * D;
      HIP_TRY(ctx, hipMemcpyAsync(dst, src, sizeof(double) * (size_t)row_count * D,
                                  hipMemcpyHostToDevice, ctx->stream));
    }
  }
  HIP_TRY(ctx, stream_wait(ctx));
  ctx->eps_K = K;
  ctx->eps_D = D;
  ctx->eps_rows = row_count;
  ctx->eps_row_begin = row_begin;
  ctx->eps_n_half = n_half;
  return VBMC_OK;
}

int vbmc_set_eps(vbmc_ctx* ctx, int K, int64_t n_half, int D, const double* eps_half,
                 int64_t row_begin, int64_t row_count) {
  if (!ctx || !eps_half || K < 1 || D < 1 || n_half < 0) return VBMC_E_ARG;
  if (row_begin < 0 || row_count < 0 || row_begin + row_count > n_half)
    return vbmc_fail(ctx, VBMC_E_ARG, "set_eps: rows [%lld,+%lld) outside [0,%lld)",
                     (long long)row_begin, (long long)row_count, (long long)n_half);
  NEED_DEVICE(ctx);
  return upload_eps(ctx, K, n_half, D, eps_half, row_begin, row_count);
}

int vbmc_set_eps_numpy(vbmc_ctx* ctx, uint32_t* key, int* pos, int* has_gauss, double* gauss, int K,
                       int64_t n_half, int D, int64_t row_begin, int64_t row_count, int n_threads) {
  if (!ctx || !key || !pos || !has_gauss || !gauss || K < 1 || D < 1 || n_half < 0) return VBMC_E_ARG;
  if (row_begin < 0 || row_count < 0 || row_begin + row_count > n_half)
    return vbmc_fail(ctx, VBMC_E_ARG, "set_eps_numpy: rows [%lld,+%lld) outside [0,%lld)",
                     (long long)row_begin, (long long)row_count, (long long)n_half);
  NEED_DEVICE(ctx);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  // every rank draws the whole job's values (the reference's stream has one order) and ships its rows
  const size_t n_all = (size_t)K * (size_t)n_half * (size_t)D;
  {
    const bool whole = row_begin == 0 && row_count == n_half && n_all > 0;
    if (whole) {
      const int erc = ensure_dev(ctx, &ctx->d_eps, &ctx->d_eps_cap, n_all);
      if (erc) return erc;
    }
  if (whole && ctx->opt_randn_dev && n_all >= ((size_t)1 << 16) && ctx->world == 1) {
      // the stream generated where it is consumed (device_randn.hip): no host cores, no PCIe
      const int drc = randn_device(ctx, key, pos, has_gauss, gauss, ctx->d_eps, (int64_t)n_all);
      if (drc != VBMC_W_NOT_FUSED) {
        if (drc) return drc;
        ctx->eps_K = K;
        ctx->eps_D = D;
        ctx->eps_rows = row_count;
        ctx->eps_row_begin = row_begin;
        ctx->eps_n_half = n_half;
        return VBMC_OK;
      }
    }
  }
  if (ctx->h_eps_cap < n_all || !ctx->h_eps) {
    if (ctx->h_eps) {
      HIP_TRY(ctx, stream_wait(ctx));
      HIP_TRY(ctx, hipHostFree(ctx->h_eps));
      ctx->h_eps = nullptr;
      ctx->h_eps_cap = 0;
    }
    const size_t want = n_all + n_all / 8 + 64;
    HIP_TRY(ctx, hipHostMalloc((void**)&ctx->h_eps, want * sizeof(double), hipHostMallocDefault));
    ctx->h_eps_cap = want;
  }
  // this context holds all rows (one GPU): the finished prefix of the draws goes up while the rest is
  // still being generated; a row shard is uploaded afterwards, component by component
  struct Up {
    vbmc_ctx* ctx;
    int64_t sent;
    hipError_t err;
  } up{ctx, 0, hipSuccess};
  const bool whole = row_begin == 0 && row_count == n_half && n_all > 0;
  if (whole) {
    const int erc = ensure_dev(ctx, &ctx->d_eps, &ctx->d_eps_cap, n_all);
    if (erc) return erc;
  }
  auto progress = [](void* user, int64_t m) {
    Up* u = (Up*)user;
    if (m <= u->sent || u->err != hipSuccess) return;
    u->err = hipMemcpyAsync(u->ctx->d_eps + u->sent, u->ctx->h_eps + u->sent, sizeof(double) * (size_t)(m - u->sent),
                            hipMemcpyHostToDevice, u->ctx->stream);
    u->sent = m;
  };
  const int rc = randn_with_progress(key, pos, has_gauss, gauss, ctx->h_eps, (int64_t)n_all, n_threads,
                                     whole ? +progress : nullptr, &up);
  if (rc) return vbmc_fail(ctx, rc, rc == VBMC_E_NOMEM ? "set_eps_numpy: out of host memory" : "set_eps_numpy: generator state rejected");
  if (!whole) return upload_eps(ctx, K, n_half, D, ctx->h_eps, row_begin, row_count);
  progress(&up, (int64_t)n_all);  // whatever the generator did not report
  HIP_TRY(ctx, up.err);
  HIP_TRY(ctx, stream_wait(ctx));
  ctx->eps_K = K;
  ctx->eps_D = D;
  ctx->eps_rows = row_count;
  ctx->eps_row_begin = row_begin;
  ctx->eps_n_half = n_half;
  return VBMC_OK;
}

}  // extern "C"
