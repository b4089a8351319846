// Multi-GPU: one process per GPU, ONE collective per evaluation -- an
// ncclAllReduce(sum, float64) of the raw entropy accumulator over xGMI
// (SURVEY.md 8e).  The reference has no communication layer; this is new.
// The message is <= 1+D*K+2K+D doubles (611 at config 3/4, 2221 at config 5):
// latency-bound, so no bucketing and no overlap games -- it is enqueued on the
// ctx stream right behind the reduce kernels and in front of the D2H copy.
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: librccl is bound at run time, see below

#include <cstdlib>
#include <cstring>

#include "common.h"

// librccl.so is NOT a link-time dependency.  It is dlopen'ed on first use of the
// communicator: single-GPU users never load it (it is ~570 MB and drags in rocm_smi /
// rocprofiler-register), and a process that also imports another ROCm stack later
// (PyTorch wheels bundle their own) keeps a clean shutdown.
namespace {
struct RcclApi {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclCommUserRank) CommUserRank = nullptr;
};
RcclApi g_rccl;

int rccl_load(vbmc_ctx* ctx) {
  if (g_rccl.handle) return 0;
  const char* names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
  void* h = nullptr;
  for (const char* n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (h) break;
  }
  if (!h) return vbmc_fail(ctx, VBMC_E_RCCL, "cannot load librccl.so.1: %s", dlerror());
  RcclApi a;
  a.handle = h;
  a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
  a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
  a.AllReduce = (decltype(a.AllReduce))dlsym(h, "ncclAllReduce");
  a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
  a.CommCount = (decltype(a.CommCount))dlsym(h, "ncclCommCount");
  a.CommUserRank = (decltype(a.CommUserRank))dlsym(h, "ncclCommUserRank");
  if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce || !a.GetErrorString)
    return vbmc_fail(ctx, VBMC_E_RCCL, "librccl.so.1 lacks an expected symbol");
  g_rccl = a;
  return 0;
}
}  // namespace

#define NCCL_TRY(ctx, call)                                                                 \
  do {                                                                                      \
    ncclResult_t r_ = (call);                                                               \
    if (r_ != ncclSuccess)                                                                  \
      return vbmc_fail((ctx), VBMC_E_RCCL, "%s failed: %s", #call, g_rccl.GetErrorString(r_)); \
  } while (0)

int comm_allreduce_sum(vbmc_ctx* ctx, double* d_buf, int n) {
  // world == 1 normally skips the collective; VBMC_FORCE_COLLECTIVE=1 keeps it (a 1-rank
  // all-reduce) so the RCCL call path can be exercised on a single-GPU box
  static const bool force = [] {
    const char* e = getenv("VBMC_FORCE_COLLECTIVE");
    return e && e[0] == '1';
  }();
  if (!ctx->comm || (ctx->world <= 1 && !force)) return 0;
  NCCL_TRY(ctx, g_rccl.AllReduce(d_buf, d_buf, (size_t)n, ncclDouble, ncclSum, (ncclComm_t)ctx->comm,
                                 ctx->stream));
  return 0;
}

extern "C" {

int vbmc_comm_unique_id(uint8_t id_out[128]) {
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
  int rc = rccl_load(nullptr);
  if (rc) return rc;
  ncclUniqueId id;
  ncclResult_t r = g_rccl.GetUniqueId(&id);
  if (r != ncclSuccess)
    return vbmc_fail(nullptr, VBMC_E_RCCL, "ncclGetUniqueId failed: %s", g_rccl.GetErrorString(r));
  memcpy(id_out, &id, 128);
  return VBMC_OK;
}

int vbmc_comm_init(vbmc_ctx* ctx, const uint8_t id[128], int rank, int world) {
  if (!ctx || !id || world < 1 || rank < 0 || rank >= world) return VBMC_E_ARG;
  NEED_DEVICE(ctx);
  if (ctx->comm) return vbmc_fail(ctx, VBMC_E_ARG, "comm_init: communicator already initialised");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = rccl_load(ctx);
  if (rc) return rc;
  ncclUniqueId uid;
  memcpy(&uid, id, 128);
  ncclComm_t c;
  NCCL_TRY(ctx, g_rccl.CommInitRank(&c, world, uid, rank));
  ctx->comm = (ncclComm*)c;
  ctx->rank = rank;
  ctx->world = world;
  return VBMC_OK;
}

int vbmc_comm_destroy(vbmc_ctx* ctx) {
  if (!ctx) return VBMC_E_ARG;
  if (ctx->comm) {
    if (g_rccl.CommDestroy) (void)g_rccl.CommDestroy((ncclComm_t)ctx->comm);
    ctx->comm = nullptr;
    ctx->world = 1;
    ctx->rank = 0;
  }
  return VBMC_OK;
}

int vbmc_comm_info(vbmc_ctx* ctx, int* rank_out, int* world_out) {
  if (!ctx) return VBMC_E_ARG;
  int r = 0, w = 1;
  if (ctx->comm) {
    // what RCCL itself reports for this communicator, not what the caller passed to vbmc_comm_init
    if (!g_rccl.CommCount || !g_rccl.CommUserRank)
      return vbmc_fail(ctx, VBMC_E_RCCL, "librccl.so.1 lacks ncclCommCount / ncclCommUserRank");
    NCCL_TRY(ctx, g_rccl.CommCount((ncclComm_t)ctx->comm, &w));
    NCCL_TRY(ctx, g_rccl.CommUserRank((ncclComm_t)ctx->comm, &r));
  }
  if (rank_out) *rank_out = r;
  if (world_out) *world_out = w;
  return VBMC_OK;
}

int vbmc_comm_allreduce_max(vbmc_ctx* ctx, double* value_inout) {
  if (!ctx || !value_inout) return VBMC_E_ARG;
  if (!ctx->comm || ctx->world <= 1) return VBMC_OK;
  NEED_DEVICE(ctx);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = ensure_dev(ctx, &ctx->d_out, &ctx->d_out_cap, 8);
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(ctx->d_out, value_inout, sizeof(double), hipMemcpyHostToDevice,
                              ctx->stream));
  NCCL_TRY(ctx, g_rccl.AllReduce(ctx->d_out, ctx->d_out, 1, ncclDouble, ncclMax,
                                 (ncclComm_t)ctx->comm, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(value_inout, ctx->d_out, sizeof(double), hipMemcpyDeviceToHost,
                              ctx->stream));
  HIP_TRY(ctx, stream_wait(ctx));
  return VBMC_OK;
}

int vbmc_comm_barrier(vbmc_ctx* ctx) {
  double v = 0.0;
  return vbmc_comm_allreduce_max(ctx, &v);
}

}  // extern "C"
