// C-ABI implementation of include/ddm_hip.h: host-side runtime (contexts, plans, level schedules,
// HIP graphs, the CG driver) around the kernels in kernels.hpp.  gfx950 only, no fallback path.
#include "../../include/ddm_hip.h"
#include "host_vec.hpp"
#include "kernels.hpp"
#include "trsv_pipe.hpp"
#include "trsv_box.hpp"
#include "sparse_chol_host.hpp"
#include "sn_chol.hpp"
#include "synth_host.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h> // types and enums only: the library is opened with dlopen when ddm_ctx_set_rccl is called

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace ddm;

// ---------------------------------------------------------------------------------------------
struct TimerEntry {
  double ms = 0.0;
  int64_t count = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending; // recorded, not yet resolved (no sync in the hot loop)
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;    // recycled event pairs
};

struct ddm_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;
  int rank = 0, nranks = 1;
  ddm_alltoall_fn a2a = nullptr;
  ddm_allreduce_fn allreduce = nullptr;
  void *user = nullptr;
  // in-library exchange over RCCL (xGMI): ddm_ctx_set_rccl
  struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
  } nccl;
  ncclComm_t rccl_comm = nullptr;
  // collectives of the iteration, counted as a run over several ranks issues them (one count = one RCCL launch: an all-reduce or a
  // grouped send/receive); ddm_ctx_comm_counts
  int64_t n_allreduce = 0, n_allreduce_doubles = 0, n_halo_groups = 0;
  // a scalar waiting to ride on the next coarse-defect all-reduce (ddm_cg_steps: the squared defect norm of the previous iteration)
  double *piggy = nullptr;
  bool rccl = false, rccl_self = false; // rccl_self: route the self segment through RCCL too (single-GPU self test)
  // side stream of the additive combination: the coarse level's restrict / solve / prolong run beside the latency-bound local solve
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  double *partial = nullptr; // RED_MAX_BLOCKS doubles
  double *scal = nullptr;    // 16 device scalars
  int num_cu = 256;           // compute units of the device: persistent kernels launch at most this many workgroups
  bool timing = false;
  std::map<std::string, TimerEntry> timers;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipEvent_t ev_fence = nullptr; // ddm_ctx_fence
};

static std::mutex g_err_mutex;
static thread_local std::string t_last_error;
static int fail(ddm_ctx *ctx, int code, const char *fmt, ...)
{
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  t_last_error = buf;
  if (ctx) { // (setup phases run independent host work on helper threads that may fail at the same time)
    std::lock_guard<std::mutex> lock(g_err_mutex);
    ctx->err = buf;
  }
  return code;
}
// message of the last fail() on the CALLING thread (helper threads report their own failure, not whatever another thread wrote last)
static std::string last_error_of_this_thread() { return t_last_error; }
#define HIPCHECK(ctx, call)                                                                                   \
  do {                                                                                                        \
    hipError_t e_ = (call);                                                                                   \
    if (e_ != hipSuccess) return fail(ctx, DDM_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)
#define DDMCHECK(call)            \
  do {                            \
    int rc_ = (call);             \
    if (rc_ != DDM_OK) return rc_; \
  } while (0)

// single-launch triangular solves need every workgroup resident: one workgroup per CU at most
static inline int persistent_grid(const ddm_ctx *ctx) { return std::max(8, std::min(256, ctx->num_cu) / 8 * 8); }

static inline int grid_for(int64_t n, int per_block = WG, int cap = 2048)
{
  int64_t g = (n + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

// Transfers of a BACKGROUND setup thread (the builder of the single-launch engines' schedules runs beside the caller's next setup
// steps): a synchronous hipMemcpy / hipMemset goes through the legacy default stream, and when the caller's stream is that stream and
// is being captured into a graph at that moment (the GenEO block solves capture theirs) the capture is invalidated ("operation failed
// due to a previous error during capture").  The thread therefore moves its data on a non-blocking stream of its own.
static thread_local hipStream_t t_transfer_stream = nullptr;
struct BackgroundTransfers {
  BackgroundTransfers() { (void)hipStreamCreateWithFlags(&t_transfer_stream, hipStreamNonBlocking); }
  ~BackgroundTransfers()
  {
    if (t_transfer_stream) (void)hipStreamDestroy(t_transfer_stream);
    t_transfer_stream = nullptr;
  }
};
template <class T>
static int upload(ddm_ctx *ctx, const T *host, int64_t n, T **dev)
{
  *dev = nullptr;
  if (n <= 0) {
    HIPCHECK(ctx, hipMalloc((void **)dev, sizeof(T)));
    return DDM_OK;
  }
  HIPCHECK(ctx, hipMalloc((void **)dev, sizeof(T) * (size_t)n));
  if (t_transfer_stream) { // background setup thread: its own non-blocking stream (see BackgroundTransfers)
    HIPCHECK(ctx, hipMemcpyAsync(*dev, host, sizeof(T) * (size_t)n, hipMemcpyHostToDevice, t_transfer_stream));
    HIPCHECK(ctx, hipStreamSynchronize(t_transfer_stream));
  } else
    HIPCHECK(ctx, hipMemcpy(*dev, host, sizeof(T) * (size_t)n, hipMemcpyHostToDevice));
  return DDM_OK;
}
// hipMemset that a background setup thread may call (same reason)
static hipError_t dev_memset(void *p, int v, size_t bytes)
{
  if (!t_transfer_stream) return hipMemset(p, v, bytes);
  hipError_t e = hipMemsetAsync(p, v, bytes, t_transfer_stream);
  return e != hipSuccess ? e : hipStreamSynchronize(t_transfer_stream);
}

// HIP-event timer on the context's stream.  Nothing synchronises while timing is on: the event
// pairs are resolved (hipEventElapsedTime) when the totals are read, after the stream has drained.
struct ScopedTimer {
  ddm_ctx *ctx;
  TimerEntry *t = nullptr;
  std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
  ScopedTimer(ddm_ctx *c, const char *n) : ctx(c)
  {
    if (!ctx->timing) return;
    t = &ctx->timers[n];
    if (!t->pool.empty()) {
      ev = t->pool.back();
      t->pool.pop_back();
    } else {
      (void)hipEventCreate(&ev.first);
      (void)hipEventCreate(&ev.second);
    }
    (void)hipEventRecord(ev.first, ctx->stream);
  }
  ~ScopedTimer()
  {
    if (!t) return;
    (void)hipEventRecord(ev.second, ctx->stream);
    t->pending.push_back(ev);
  }
};
static void resolve_timers(ddm_ctx *ctx)
{
  (void)hipStreamSynchronize(ctx->stream);
  for (auto &kv : ctx->timers) {
    for (auto &ev : kv.second.pending) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) {
        kv.second.ms += ms;
        kv.second.count += 1;
      }
      kv.second.pool.push_back(ev);
    }
    kv.second.pending.clear();
  }
}

// ---- context ---------------------------------------------------------------------------------
extern "C" int ddm_ctx_create(int device, void *hip_stream, ddm_ctx **out)
{
  if (!out) return DDM_EINVAL;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return DDM_EHIP; // no CPU fallback
  if (device < 0 || device >= ndev) return DDM_EINVAL;
  ddm_ctx *ctx = new ddm_ctx;
  ctx->device = device;
  if (hipSetDevice(device) != hipSuccess) {
    delete ctx;
    return DDM_EHIP;
  }
  if (hip_stream) ctx->stream = (hipStream_t)hip_stream;
  else {
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
      delete ctx;
      return DDM_EHIP;
    }
    ctx->own_stream = true;
  }
  if (hipMalloc((void **)&ctx->partial, sizeof(double) * RED_MAX_BLOCKS) != hipSuccess ||
      hipMalloc((void **)&ctx->scal, sizeof(double) * 16) != hipSuccess ||
      hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) {
    delete ctx;
    return DDM_EHIP;
  }
  (void)hipMemset(ctx->scal, 0, sizeof(double) * 16);
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ctx->num_cu = prop.multiProcessorCount;
  }
  *out = ctx;
  return DDM_OK;
}

extern "C" void ddm_ctx_destroy(ddm_ctx *ctx)
{
  if (ctx && ctx->side) {
    (void)hipStreamSynchronize(ctx->side);
    (void)hipStreamDestroy(ctx->side);
    (void)hipEventDestroy(ctx->ev_fork);
    (void)hipEventDestroy(ctx->ev_join);
    ctx->side = nullptr;
  }
  if (ctx && ctx->rccl_comm && ctx->nccl.CommDestroy) {
    (void)hipStreamSynchronize(ctx->stream);
    (void)ctx->nccl.CommDestroy(ctx->rccl_comm);
    ctx->rccl_comm = nullptr;
  }
  if (!ctx) return;
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipFree(ctx->partial);
  (void)hipFree(ctx->scal);
  if (ctx->ev_fence) (void)hipEventDestroy(ctx->ev_fence);
  (void)hipEventDestroy(ctx->ev0);
  (void)hipEventDestroy(ctx->ev1);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}
extern "C" const char *ddm_last_error(const ddm_ctx *ctx)
{
  if (!ctx) return t_last_error.empty() ? "no context" : t_last_error.c_str();   // context-free entry points: the calling thread's last failure
  static thread_local std::string copy; // (a stable pointer for the caller; ctx->err may be rewritten by a helper thread)
  std::lock_guard<std::mutex> lock(g_err_mutex);
  copy = ctx->err;
  return copy.c_str();
}
extern "C" int ddm_ctx_sync(ddm_ctx *ctx)
{
  HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
  return DDM_OK;
}
// host waits for the work enqueued on the context's stream SO FAR (an event, not a drain of the stream: work another thread or a
// later call enqueues meanwhile is not waited for, the side stream is left alone) -- what an exchange callback needs before it
// hands the packed buffer to a host-driven transport (MPI)
extern "C" int ddm_ctx_fence(ddm_ctx *ctx)
{
  if (!ctx) return DDM_EINVAL;
  if (!ctx->ev_fence) HIPCHECK(ctx, hipEventCreateWithFlags(&ctx->ev_fence, hipEventDisableTiming));
  HIPCHECK(ctx, hipEventRecord(ctx->ev_fence, ctx->stream));
  HIPCHECK(ctx, hipEventSynchronize(ctx->ev_fence));
  return DDM_OK;
}
extern "C" void *ddm_ctx_stream(ddm_ctx *ctx) { return (void *)ctx->stream; }
extern "C" int ddm_ctx_set_comm(ddm_ctx *ctx, int rank, int nranks, ddm_alltoall_fn a2a, ddm_allreduce_fn allreduce, void *user)
{
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail(ctx, DDM_EINVAL, "bad rank %d of %d", rank, nranks);
  if (nranks > 1 && (!a2a || !allreduce)) return fail(ctx, DDM_EINVAL, "multi-rank context needs both callbacks");
  ctx->rank = rank;
  ctx->nranks = nranks;
  ctx->a2a = a2a;
  ctx->allreduce = allreduce;
  ctx->user = user;
  return DDM_OK;
}
// ---- in-library exchange: RCCL over xGMI ------------------------------------------------------------
static void *rccl_open()
{
  for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
    if (void *h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) return h; // an already loaded copy (e.g. the host program's) is reused
  return nullptr;
}
extern "C" int ddm_rccl_unique_id(void *id128)
{
  if (!id128) return DDM_EINVAL;
  void *h = rccl_open();
  if (!h) return DDM_ECOMM;
  auto get = (ncclResult_t(*)(ncclUniqueId *))dlsym(h, "ncclGetUniqueId");
  ncclUniqueId id;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  if (!get || get(&id) != ncclSuccess) return DDM_ECOMM;
  std::memcpy(id128, &id, 128);
  return DDM_OK;
}
extern "C" int ddm_ctx_set_rccl(ddm_ctx *ctx, int rank, int nranks, const void *id128, int self_test)
{
  if (!ctx || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(ctx, DDM_EINVAL, "ddm_ctx_set_rccl: bad rank %d of %d", rank, nranks);
  if (ctx->rccl_comm) return fail(ctx, DDM_EINVAL, "ddm_ctx_set_rccl: the context already has a communicator");
  auto &N = ctx->nccl;
  N.lib = rccl_open();
  if (!N.lib) return fail(ctx, DDM_ECOMM, "librccl.so.1 cannot be loaded: %s", dlerror());
  N.CommInitRank = (decltype(N.CommInitRank))dlsym(N.lib, "ncclCommInitRank");
  N.CommDestroy = (decltype(N.CommDestroy))dlsym(N.lib, "ncclCommDestroy");
  N.GroupStart = (decltype(N.GroupStart))dlsym(N.lib, "ncclGroupStart");
  N.GroupEnd = (decltype(N.GroupEnd))dlsym(N.lib, "ncclGroupEnd");
  N.Send = (decltype(N.Send))dlsym(N.lib, "ncclSend");
  N.Recv = (decltype(N.Recv))dlsym(N.lib, "ncclRecv");
  N.AllReduce = (decltype(N.AllReduce))dlsym(N.lib, "ncclAllReduce");
  N.GetErrorString = (decltype(N.GetErrorString))dlsym(N.lib, "ncclGetErrorString");
  N.CommCount = (decltype(N.CommCount))dlsym(N.lib, "ncclCommCount");
  if (!N.CommInitRank || !N.CommDestroy || !N.GroupStart || !N.GroupEnd || !N.Send || !N.Recv || !N.AllReduce)
    return fail(ctx, DDM_ECOMM, "librccl lacks a required entry point");
  HIPCHECK(ctx, hipSetDevice(ctx->device));
  ncclUniqueId id;
  std::memcpy(&id, id128, 128);
  const ncclResult_t r = N.CommInitRank(&ctx->rccl_comm, nranks, id, rank);
  if (r != ncclSuccess) {
    ctx->rccl_comm = nullptr;
    return fail(ctx, DDM_ECOMM, "ncclCommInitRank failed: %s", N.GetErrorString ? N.GetErrorString(r) : "?");
  }
  ctx->rank = rank;
  ctx->nranks = nranks;
  ctx->rccl = true;
  ctx->rccl_self = self_test != 0;
  ctx->a2a = nullptr;
  ctx->allreduce = nullptr;
  return DDM_OK;
}
extern "C" int ddm_ctx_rccl_size(ddm_ctx *ctx, int *count)
{
  if (!ctx || !count) return DDM_EINVAL;
  *count = 0; // no in-library communicator
  if (!ctx->rccl_comm) return DDM_OK;
  if (!ctx->nccl.CommCount) return fail(ctx, DDM_ECOMM, "librccl lacks ncclCommCount");
  const ncclResult_t r = ctx->nccl.CommCount(ctx->rccl_comm, count);
  if (r != ncclSuccess) return fail(ctx, DDM_ECOMM, "ncclCommCount failed: %s", ctx->nccl.GetErrorString ? ctx->nccl.GetErrorString(r) : "?");
  return DDM_OK;
}
#define NCCLCHECK(ctx, call)                                                                                                   \
  do {                                                                                                                         \
    const ncclResult_t r_ = (call);                                                                                            \
    if (r_ != ncclSuccess) return fail(ctx, DDM_ECOMM, "%s failed: %s", #call, ctx->nccl.GetErrorString ? ctx->nccl.GetErrorString(r_) : "?"); \
  } while (0)
// in-place sum over all ranks of n doubles at a device pointer, enqueued on the context's stream
static int ctx_allreduce(ddm_ctx *ctx, double *buf, int64_t n, const char *what)
{
  ctx->n_allreduce += 1;
  ctx->n_allreduce_doubles += n;
  if (ctx->rccl) {
    if (ctx->nranks > 1 || ctx->rccl_self) NCCLCHECK(ctx, ctx->nccl.AllReduce(buf, buf, (size_t)n, ncclDouble, ncclSum, ctx->rccl_comm, ctx->stream));
    return DDM_OK;
  }
  if (ctx->nranks > 1)
    if (ctx->allreduce(ctx->user, buf, n) != 0) return fail(ctx, DDM_ECOMM, "allreduce callback failed (%s)", what);
  return DDM_OK;
}

// the coarse defect (K doubles at d0, room for K + 1) summed over the ranks; a scalar waiting in ctx->piggy rides along as element K
// (one RCCL launch instead of two) and is written back
__global__ void k_copy_scalar(const double *__restrict__ src, double *__restrict__ dst) { *dst = *src; }
static int coarse_allreduce(ddm_ctx *ctx, double *d0, int64_t K)
{
  double *rider = ctx->piggy;
  ctx->piggy = nullptr;
  if (!rider) return ctx_allreduce(ctx, d0, K, "coarse defect");
  hipLaunchKernelGGL(k_copy_scalar, dim3(1), dim3(1), 0, ctx->stream, (const double *)rider, d0 + K);
  DDMCHECK(ctx_allreduce(ctx, d0, K + 1, "coarse defect + deferred defect norm"));
  hipLaunchKernelGGL(k_copy_scalar, dim3(1), dim3(1), 0, ctx->stream, (const double *)(d0 + K), rider);
  return DDM_OK;
}
extern "C" int ddm_ctx_comm_counts(ddm_ctx *ctx, int64_t *counts)
{
  if (!ctx || !counts) return DDM_EINVAL;
  counts[0] = ctx->n_allreduce;
  counts[1] = ctx->n_allreduce_doubles;
  counts[2] = ctx->n_halo_groups;
  return DDM_OK;
}

extern "C" int ddm_malloc(ddm_ctx *ctx, int64_t bytes, void **dptr)
{
  HIPCHECK(ctx, hipMalloc(dptr, (size_t)std::max<int64_t>(bytes, 8)));
  return DDM_OK;
}
extern "C" int ddm_free(ddm_ctx *ctx, void *dptr)
{
  HIPCHECK(ctx, hipFree(dptr));
  return DDM_OK;
}
extern "C" int ddm_memset_zero(ddm_ctx *ctx, void *dptr, int64_t bytes)
{
  if (!dptr || bytes < 0) return fail(ctx, DDM_EINVAL, "ddm_memset_zero: bad arguments");
  HIPCHECK(ctx, hipMemsetAsync(dptr, 0, (size_t)bytes, ctx->stream));
  return DDM_OK;
}
extern "C" int ddm_memcpy_h2d(ddm_ctx *ctx, void *dst, const void *src, int64_t bytes)
{
  HIPCHECK(ctx, hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyHostToDevice, ctx->stream));
  HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
  return DDM_OK;
}
extern "C" int ddm_memcpy_d2h(ddm_ctx *ctx, void *dst, const void *src, int64_t bytes)
{
  HIPCHECK(ctx, hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
  return DDM_OK;
}
extern "C" int ddm_timing_enable(ddm_ctx *ctx, int on)
{
  ctx->timing = on != 0;
  return DDM_OK;
}
extern "C" int ddm_timing_get(ddm_ctx *ctx, const char *name, double *total_ms, int64_t *count)
{
  resolve_timers(ctx);
  auto it = ctx->timers.find(name);
  if (it == ctx->timers.end()) {
    if (total_ms) *total_ms = 0.0;
    if (count) *count = 0;
    return DDM_OK;
  }
  if (total_ms) *total_ms = it->second.ms;
  if (count) *count = it->second.count;
  return DDM_OK;
}
extern "C" int ddm_timing_reset(ddm_ctx *ctx)
{
  resolve_timers(ctx);
  for (auto &kv : ctx->timers) {
    kv.second.ms = 0.0;
    kv.second.count = 0;
  }
  return DDM_OK;
}

// ---- CSR ---------------------------------------------------------------------------------------
// Worker threads of the host-side setup phases (factorisations, schedules, assembly): the cores of the machine, but never more than
// 16 per process -- a node runs one process per GPU, and several of these pools are alive at the same time (DDM_HOST_THREADS overrides).
static unsigned host_threads()
{
  static const unsigned n = []() {
    if (const char *e = std::getenv("DDM_HOST_THREADS")) return (unsigned)std::max(1, std::atoi(e));
    return std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
  }();
  return n;
}

// dst = src with `threads` memcpy workers (fresh pages: the copy is page-fault bound on one thread)
template <class T>
static void hvec_copy(hvec<T> &dst, const T *src, size_t n)
{
  dst.resize(n);
  const size_t nth = std::min<size_t>(host_threads(), std::max<size_t>(1, n >> 22));
  if (nth <= 1) {
    if (n) std::memcpy(dst.data(), src, sizeof(T) * n);
    return;
  }
  std::vector<std::thread> th;
  for (size_t t = 0; t < nth; ++t)
    th.emplace_back([&, t]() {
      const size_t a = n * t / nth, b = n * (t + 1) / nth;
      std::memcpy(dst.data() + a, src + a, sizeof(T) * (b - a));
    });
  for (auto &t : th) t.join();
}

struct ddm_csr {
  int64_t nrows = 0, ncols = 0, nnz = 0;
  hvec<int64_t> h_rp; // host copies are kept for the ILU(0) factorisation / analysis
  hvec<int32_t> h_ci;
  hvec<double> h_va;
  int64_t *rp = nullptr;
  int32_t *ci = nullptr;
  double *va = nullptr;
  int32_t *blk_row = nullptr;
  int nblk = 0;
  bool borrowed_pattern = false; // rp / ci / blk_row belong to another ddm_csr (values-only companion on the same pattern)
  bool host_only = false;        // created by ddm_csr_create_host: no device arrays
  int32_t *row_order = nullptr;  // cache-blocked processing order of the rows for the block products (csr_row_order_tiled), or null
  std::thread uploader;          // device copies still in flight (csr_adopt): csr_wait_upload joins it
  int upload_rc = 0;
  std::string upload_err;
};

static std::vector<int32_t> csr_row_blocks(int64_t nrows, const int64_t *rowptr);
static int csr_create_impl(ddm_ctx *ctx, int64_t nrows, int64_t ncols, const int64_t *rowptr, const int32_t *col, const double *val, bool host_only, ddm_csr **out)
{
  if (!ctx || !out || nrows < 0 || !rowptr) return fail(ctx, DDM_EINVAL, "ddm_csr_create: bad arguments");
  if (nrows >= (int64_t)1 << 31 || ncols >= (int64_t)1 << 31) return fail(ctx, DDM_EINVAL, "matrix dimension exceeds int32 columns");
  const int64_t nnz = rowptr[nrows];
  for (int64_t i = 0; i < nrows; ++i)
    if (rowptr[i + 1] < rowptr[i]) return fail(ctx, DDM_EINVAL, "row pointers not monotone at row %lld", (long long)i);
  for (int64_t k = 0; k < nnz; ++k)
    if (col[k] < 0 || col[k] >= ncols) return fail(ctx, DDM_EINVAL, "column index out of range at entry %lld", (long long)k);
  ddm_csr *A = new ddm_csr;
  A->nrows = nrows;
  A->ncols = ncols;
  A->nnz = nnz;
  hvec_copy(A->h_rp, rowptr, (size_t)nrows + 1);
  hvec_copy(A->h_ci, col, (size_t)nnz);
  hvec_copy(A->h_va, val, (size_t)nnz);
  const std::vector<int32_t> blk = csr_row_blocks(nrows, rowptr);
  A->nblk = (int)blk.size() - 1;
  if (host_only) { // analysis / assembly input only (the GenEO pencil is built from the host arrays): no device copy
    A->host_only = true;
    A->nblk = 0;
    *out = A;
    return DDM_OK;
  }
  int rc = upload(ctx, rowptr, nrows + 1, &A->rp);
  if (!rc) rc = upload(ctx, col, nnz, &A->ci);
  if (!rc) rc = upload(ctx, val, nnz, &A->va);
  if (!rc) rc = upload(ctx, blk.data(), (int64_t)blk.size(), &A->blk_row);
  if (rc) {
    ddm_csr_destroy(A);
    return rc;
  }
  *out = A;
  return DDM_OK;
}
extern "C" int ddm_csr_create(ddm_ctx *ctx, int64_t nrows, int64_t ncols, const int64_t *rowptr, const int32_t *col, const double *val, ddm_csr **out)
{
  return csr_create_impl(ctx, nrows, ncols, rowptr, col, val, false, out);
}
// the same object WITHOUT device arrays: valid as A_neu / B_neu of ddm_geneo_basis (the pencil is assembled from the host arrays) and
// of the other coarse-space builders' host inputs; every entry point that would touch the device arrays returns DDM_EINVAL
extern "C" int ddm_csr_create_host(ddm_ctx *ctx, int64_t nrows, int64_t ncols, const int64_t *rowptr, const int32_t *col, const double *val, ddm_csr **out)
{
  return csr_create_impl(ctx, nrows, ncols, rowptr, col, val, true, out);
}
extern "C" void ddm_csr_destroy(ddm_csr *A)
{
  if (!A) return;
  if (A->uploader.joinable()) A->uploader.join();
  (void)hipFree(A->row_order);
  if (!A->borrowed_pattern) {
    (void)hipFree(A->rp);
    (void)hipFree(A->ci);
    (void)hipFree(A->blk_row);
  }
  (void)hipFree(A->va);
  delete A;
}
// row-block schedule of the CSR-stream kernel: <= SPMV_NNZ non-zeros and <= WG rows per block, a row longer than SPMV_NNZ gets a
// block of its own
static std::vector<int32_t> csr_row_blocks(int64_t nrows, const int64_t *rowptr)
{
  std::vector<int32_t> blk;
  blk.push_back(0);
  int64_t r = 0;
  while (r < nrows) {
    int64_t r1 = r;
    const int64_t z0 = rowptr[r];
    while (r1 < nrows && r1 - r < WG && rowptr[r1 + 1] - z0 <= SPMV_NNZ) ++r1;
    if (r1 == r) r1 = r + 1; // long row
    blk.push_back((int32_t)r1);
    r = r1;
  }
  return blk;
}
// Library-internal constructors for matrices the library assembled itself (GenEO pencil): the host arrays are MOVED in (no copy, no
// validation pass), and the device copies are made by a helper thread while the caller goes on with host work on the host arrays
// (factorisation, analysis).  Everything that touches the device arrays calls csr_wait_upload first.
static int csr_wait_upload(ddm_ctx *ctx, const ddm_csr *A)
{
  ddm_csr *M = const_cast<ddm_csr *>(A);
  if (M->uploader.joinable()) M->uploader.join();
  if (M->upload_rc) return fail(ctx, M->upload_rc, "%s", M->upload_err.c_str());
  return DDM_OK;
}
// Cache-blocked processing order of the rows of a block-diagonal matrix whose blocks come from a STRUCTURED grid in lexicographic
// numbering (possibly followed by irregularly numbered rows, e.g. an overlap shell): the strides s2 (one grid line) and s3 (one grid
// plane) are read off the column offsets that most rows share; rows are then visited brick by brick (16 x 4 x 4 points, bricks in
// lexicographic order), rows that fit no brick keep their place at the end.  Purely a performance hint -- any permutation is valid.
// Returns false (order untouched) when no such structure is found.
static bool csr_row_order_tiled(int64_t nblocks, const int64_t *block_ptr, const int64_t *rp, const int32_t *ci, std::vector<int32_t> &order)
{
  const int64_t n = block_ptr[nblocks];
  order.resize((size_t)n);
  std::vector<uint8_t> seen((size_t)n, 0);
  int64_t out = 0;
  bool any = false;
  for (int64_t b = 0; b < nblocks; ++b) {
    const int64_t r0 = block_ptr[b], r1 = block_ptr[b + 1], nb = r1 - r0;
    int64_t s2 = 0, s3 = 0;
    if (nb >= 4096) { // positive column offsets shared by most of a sample of rows from the first half of the block
      std::map<int64_t, int> hist;
      const int64_t sample = 2048, start = r0 + nb / 4;
      for (int64_t i = start; i < start + sample; ++i)
        for (int64_t k = rp[i]; k < rp[i + 1]; ++k)
          if (ci[k] > i) hist[ci[k] - i]++;
      std::vector<int64_t> P;
      for (auto &kv : hist)
        if (kv.second > sample / 2) P.push_back(kv.first);
      auto has = [&](int64_t o) { return std::binary_search(P.begin(), P.end(), o); };
      // 5- / 7-point stencils share the offsets {1, s2, s3}; 9- / 27-point ones {1, s2 - 1, s2, s2 + 1, s3 - s2 - 1, ..., s3 + s2 + 1}
      if (P.size() >= 2 && P[0] == 1) {
        const int64_t a = P[1];
        if (has(a + 1) && has(a + 2)) s2 = a + 1;
        else if (!has(a + 1)) s2 = a;
        if (s2 > 1) {
          auto it = std::upper_bound(P.begin(), P.end(), s2 + 1);
          if (it == P.end()) s3 = ((nb + s2 - 1) / s2) * s2; // two-dimensional: one plane
          else {
            const int64_t c = *it;
            if (has(c + 1) && has(c + 2)) s3 = has(c + s2 + 1) ? c + s2 + 1 : 0;
            else if (!has(c + 1)) s3 = c;
          }
        }
      }
      if (s2 < 4 || s3 < 2 * s2) s2 = s3 = 0;
    }
    if (!s2) {
      for (int64_t r = r0; r < r1; ++r) order[(size_t)out++] = (int32_t)r;
      continue;
    }
    any = true;
    const int64_t ny = s3 / s2, nz = (nb + s3 - 1) / s3;
    constexpr int64_t TX = 16, TY = 4, TZ = 4;
    for (int64_t z0 = 0; z0 < nz; z0 += TZ)
      for (int64_t y0 = 0; y0 < ny; y0 += TY)
        for (int64_t x0 = 0; x0 < s2; x0 += TX)
          for (int64_t z = z0; z < std::min(z0 + TZ, nz); ++z)
            for (int64_t y = y0; y < std::min(y0 + TY, ny); ++y)
              for (int64_t x = x0; x < std::min(x0 + TX, s2); ++x) {
                const int64_t r = x + y * s2 + z * s3;
                if (r < nb && !seen[(size_t)(r0 + r)]) {
                  seen[(size_t)(r0 + r)] = 1;
                  order[(size_t)out++] = (int32_t)(r0 + r);
                }
              }
    for (int64_t r = r0; r < r1; ++r) // (planes with s3 % s2 leftovers)
      if (!seen[(size_t)r]) order[(size_t)out++] = (int32_t)r;
  }
  return any && out == n;
}
// host-only entry for the CPU tests: order_out[n]; returns 1 when a grid structure was found (else order_out is the identity)
extern "C" int ddm_csr_row_order_tiled_host(int64_t nblocks, const int64_t *block_ptr, const int64_t *rowptr, const int32_t *col, int32_t *order_out)
{
  if (nblocks < 1 || !block_ptr || !rowptr || !col || !order_out || block_ptr[0] != 0) return DDM_EINVAL;
  std::vector<int32_t> order;
  const bool found = csr_row_order_tiled(nblocks, block_ptr, rowptr, col, order);
  std::memcpy(order_out, order.data(), sizeof(int32_t) * order.size());
  return found ? 1 : 0;
}
static ddm_csr *csr_adopt(ddm_ctx *ctx, int64_t n, hvec<int64_t> &&rp, hvec<int32_t> &&ci, hvec<double> &&va, hvec<double> &&companion_values, ddm_csr **companion,
                          int64_t nblocks = 0, const int64_t *block_ptr = nullptr /* diagonal blocks: builds the cache-blocked row order of the block products */)
{
  ddm_csr *A = new ddm_csr, *C = new ddm_csr;
  A->nrows = A->ncols = C->nrows = C->ncols = n;
  A->nnz = C->nnz = rp[(size_t)n];
  A->h_rp = std::move(rp);
  A->h_ci = std::move(ci);
  A->h_va = std::move(va);
  C->borrowed_pattern = true;
  *companion = C;
  const int device = ctx->device;
  auto cv = std::make_shared<hvec<double>>(std::move(companion_values));
  std::vector<int64_t> bp(block_ptr ? block_ptr : nullptr, block_ptr ? block_ptr + nblocks + 1 : nullptr);
  A->uploader = std::thread([A, C, cv, device, bp]() {
    auto up = [&](const void *src, size_t bytes, void **dst) {
      if (A->upload_rc) return;
      hipError_t e = hipMalloc(dst, std::max<size_t>(bytes, 8));
      if (e == hipSuccess && bytes) e = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
      if (e != hipSuccess) {
        A->upload_rc = DDM_EHIP;
        A->upload_err = std::string("matrix upload failed: ") + hipGetErrorString(e);
      }
    };
    (void)hipSetDevice(device);
    const std::vector<int32_t> blk = csr_row_blocks(A->nrows, A->h_rp.data());
    A->nblk = (int)blk.size() - 1;
    up(A->h_rp.data(), sizeof(int64_t) * A->h_rp.size(), (void **)&A->rp);
    up(A->h_ci.data(), sizeof(int32_t) * A->h_ci.size(), (void **)&A->ci);
    up(A->h_va.data(), sizeof(double) * A->h_va.size(), (void **)&A->va);
    up(blk.data(), sizeof(int32_t) * blk.size(), (void **)&A->blk_row);
    up(cv->data(), sizeof(double) * cv->size(), (void **)&C->va);
    if (bp.size() >= 2 && !std::getenv("DDM_SPMM_NATURAL_ORDER")) {
      std::vector<int32_t> order;
      if (csr_row_order_tiled((int64_t)bp.size() - 1, bp.data(), A->h_rp.data(), A->h_ci.data(), order)) up(order.data(), sizeof(int32_t) * order.size(), (void **)&A->row_order);
    }
    C->rp = A->rp;
    C->ci = A->ci;
    C->blk_row = A->blk_row;
    C->nblk = A->nblk;
  });
  return A;
}
extern "C" int64_t ddm_csr_rows(const ddm_csr *A) { return A->nrows; }
extern "C" int64_t ddm_csr_nnz(const ddm_csr *A) { return A->nnz; }

static int csr_mv_impl(ddm_ctx *ctx, const ddm_csr *A, double alpha, const double *x, double *y, bool acc)
{
  if (A->host_only) return fail(ctx, DDM_EINVAL, "the matrix was created without device arrays (ddm_csr_create_host)");
  if (A->nblk == 0) return DDM_OK;
  if (acc)
    hipLaunchKernelGGL(k_spmv_stream<true>, dim3(A->nblk), dim3(WG), 0, ctx->stream, A->rp, A->ci, A->va, A->blk_row, A->nblk, x, y, alpha);
  else
    hipLaunchKernelGGL(k_spmv_stream<false>, dim3(A->nblk), dim3(WG), 0, ctx->stream, A->rp, A->ci, A->va, A->blk_row, A->nblk, x, y, alpha);
  HIPCHECK(ctx, hipGetLastError());
  return DDM_OK;
}
extern "C" int ddm_csr_mv(ddm_ctx *ctx, const ddm_csr *A, const double *x, double *y)
{
  if (x == y) return fail(ctx, DDM_EINVAL, "ddm_csr_mv: x and y alias");
  return csr_mv_impl(ctx, A, 1.0, x, y, false);
}
extern "C" int ddm_csr_usmv(ddm_ctx *ctx, const ddm_csr *A, double alpha, const double *x, double *y)
{
  if (x == y) return fail(ctx, DDM_EINVAL, "ddm_csr_usmv: x and y alias");
  return csr_mv_impl(ctx, A, alpha, x, y, true);
}

// Y = A X, row-major n x nrhs block vectors with leading dimensions ldx / ldy (MatOp::perform_op on a block; spectra.hh:100-105)
static int csr_mm_ld(ddm_ctx *ctx, const ddm_csr *A, int nrhs, const double *X, int64_t ldx, double *Y, int64_t ldy)
{
  if (!A || !X || !Y || X == Y || nrhs < 1 || ldx < nrhs || ldy < nrhs) return fail(ctx, DDM_EINVAL, "ddm_csr_mm: bad arguments");
  if (A->host_only) return fail(ctx, DDM_EINVAL, "the matrix was created without device arrays (ddm_csr_create_host)");
  const int64_t threads = A->nrows * (int64_t)nrhs;
  if (threads == 0) return DDM_OK;
  if (nrhs % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ((uintptr_t)X & 31) == 0 && ((uintptr_t)Y & 31) == 0) {
    hipLaunchKernelGGL(k_spmm_rowmajor4<false>, dim3((unsigned)((threads / 4 + WG - 1) / WG)), dim3(WG), 0, ctx->stream, A->nrows, nrhs / 4, A->rp, A->ci, A->va,
                       (const double *)nullptr, X, ldx, Y, (double *)nullptr, ldy);
    HIPCHECK(ctx, hipGetLastError());
    return DDM_OK;
  }
  hipLaunchKernelGGL(k_spmm_rowmajor, dim3((unsigned)((threads + WG - 1) / WG)), dim3(WG), 0, ctx->stream, A->nrows, nrhs, A->rp, A->ci,
                     A->va, X, ldx, Y, ldy);
  HIPCHECK(ctx, hipGetLastError());
  return DDM_OK;
}
// Y1 = A1 X, Y2 = A2 X for two matrices on ONE pattern (same rp / ci arrays in value; checked by size only: internal use)
static int csr_mm2_ld(ddm_ctx *ctx, const ddm_csr *A1, const ddm_csr *A2, int nrhs, const double *X, int64_t ldx, double *Y1, double *Y2, int64_t ldy)
{
  const bool fast = A1->nrows == A2->nrows && A1->nnz == A2->nnz && nrhs % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ((uintptr_t)X & 31) == 0 && ((uintptr_t)Y1 & 31) == 0 &&
                    ((uintptr_t)Y2 & 31) == 0 && X != Y1 && X != Y2;
  if (!fast) {
    DDMCHECK(csr_mm_ld(ctx, A1, nrhs, X, ldx, Y1, ldy));
    return csr_mm_ld(ctx, A2, nrhs, X, ldx, Y2, ldy);
  }
  if (A1->host_only || A2->host_only) return fail(ctx, DDM_EINVAL, "the matrix was created without device arrays (ddm_csr_create_host)");
  const int64_t threads = A1->nrows * (int64_t)(nrhs / 4);
  if (threads == 0) return DDM_OK;
  if (A1->row_order && nrhs / 4 <= 8) { // cache-blocked row order: 64 rows per workgroup
    const int nq = nrhs / 4;
    hipLaunchKernelGGL(k_spmm_rowmajor4_tiled<true>, dim3((unsigned)((A1->nrows + 63) / 64)), dim3(64 * nq), 0, ctx->stream, A1->nrows, nq, A1->row_order, A1->rp, A1->ci, A1->va,
                       (const double *)A2->va, X, ldx, Y1, Y2, ldy);
    HIPCHECK(ctx, hipGetLastError());
    return DDM_OK;
  }
  hipLaunchKernelGGL(k_spmm_rowmajor4<true>, dim3((unsigned)((threads + WG - 1) / WG)), dim3(WG), 0, ctx->stream, A1->nrows, nrhs / 4, A1->rp, A1->ci, A1->va,
                     (const double *)A2->va, X, ldx, Y1, Y2, ldy);
  HIPCHECK(ctx, hipGetLastError());
  return DDM_OK;
}
extern "C" int ddm_csr_mm(ddm_ctx *ctx, const ddm_csr *A, int nrhs, const double *X, double *Y) { return csr_mm_ld(ctx, A, nrhs, X, nrhs, Y, nrhs); }

// ---- ILU(0) -----------------------------------------------------------------------------------
// Host factorisation: dune-istl blockILU0Decomposition semantics (IKJ in the pattern, multipliers
// in L, inverse pivots on the diagonal), natural row order; independent diagonal blocks
// (subdomains) are factorised by separate threads.
static int ilu0_factor_block(const int64_t *rp, const int32_t *ci, double *lu, int64_t *diag, int64_t r0, int64_t r1)
{
  for (int64_t i = r0; i < r1; ++i) {
    diag[i] = -1;
    for (int64_t k = rp[i]; k < rp[i + 1]; ++k) {
      if (ci[k] < r0 || ci[k] >= r1) return -2; // entry outside the diagonal block
      if (k > rp[i] && ci[k] <= ci[k - 1]) return -3; // unsorted row
      if (ci[k] == i) diag[i] = k;
    }
    if (diag[i] < 0) return -1;
  }
  for (int64_t i = r0; i < r1; ++i) {
    for (int64_t kk = rp[i]; kk < diag[i]; ++kk) {
      const int64_t k = ci[kk];
      lu[kk] *= lu[diag[k]];
      const double lik = lu[kk];
      int64_t pi = kk + 1;
      for (int64_t pk = diag[k] + 1; pk < rp[k + 1]; ++pk) {
        const int32_t j = ci[pk];
        while (pi < rp[i + 1] && ci[pi] < j) ++pi;
        if (pi == rp[i + 1]) break;
        if (ci[pi] == j) lu[pi] -= lik * lu[pk];
      }
    }
    if (lu[diag[i]] == 0.0) return -1;
    lu[diag[i]] = 1.0 / lu[diag[i]];
  }
  return 0;
}

struct TriSchedule { // one triangular factor, level by level in sliced ELL
  int64_t nlev = 0;
  std::vector<LevelDesc> desc;        // per level
  int32_t *rows = nullptr;            // [n] rows sorted by level
  int32_t *cols = nullptr;            // sliced ELL columns
  double *vals = nullptr;             // sliced ELL values
  double *dinv = nullptr;             // upper only: inverse pivots in level order
  float *vals_f32 = nullptr, *dinv_f32 = nullptr; // single-precision copies for the preconditioner sweeps (made on first use)
  LevelDesc *d_desc = nullptr;        // device copy (for the small-level kernel)
  struct Launch {                     // execution plan
    int first, count;                 // levels [first, first+count)
    bool small;                       // one workgroup loops over the levels
  };
  std::vector<Launch> plan;
  int64_t ell_entries = 0;
};

struct TriCsr { // one triangular factor of the sparse direct solver: rows in level order, CSR entries (kernels.hpp: CsrLevel)
  int64_t nlev = 0;
  std::vector<CsrLevel> desc;
  int64_t nrows = 0, entries = 0; // transformed rows (real + virtual unknowns of the supernodes), stored entries
  int32_t *rows = nullptr;        // destination unknown of a row
  int32_t *rhs = nullptr;         // index of its right-hand side (lower: in d, upper: in x) or -1 (none)
  int64_t *lrp = nullptr;
  int32_t *cols = nullptr;
  double *vals = nullptr;
  double *dinv = nullptr; // upper only
  CsrLevel *d_desc = nullptr;
  struct Launch {
    int first, count;
    bool fused;
  };
  std::vector<Launch> plan;
  // block-wise variant (rows ordered by (block, level)): one workgroup per block runs the block's whole solve
  int nblocks = 0;
  int32_t *blk_lev_ptr = nullptr;
};

struct ddm_ilu0 {
  int64_t n = 0, nnz = 0;
  int mode = 8;                 // 8 = pipe (default; falls back to 4 when not applicable), 4 = xcd2 (XCD-local + loader waves), 0 = one launch per level
  // xcd2 engine (mode 4): per-block (subdomain) level schedules, built on first use
  std::vector<int64_t> h_diag, h_block_ptr;
  const ddm_csr *A = nullptr;
  bool xcd_built = false;
  int ngroups = 0;
  GroupDesc *xg = nullptr;
  LevelDesc *xdesc = nullptr;
  int64_t *xflag_off = nullptr;
  int32_t *xrows = nullptr, *xcols = nullptr;
  double *xvals = nullptr, *xdinv = nullptr;
  unsigned *xflags = nullptr;
  XcdState *xstate = nullptr;
  double *xdperm = nullptr;     // right-hand side permuted into level order (loader engine)
  int64_t *xlpos = nullptr;     // positions of the L parts (only those need the permuted right-hand side)
  int64_t xnrows = 0;
  // pipe engine (mode 8): chains x tasks, see trsv_pipe_host.hpp
  int pipe_state = 0;           // 0 not built, 1 built, -1 not applicable
  // the pipe schedule is built in the background (its own host threads + uploads; 2.6 s at 216^3, nothing of it is needed before the
  // first single-vector solve): every reader of the pipe state joins first (ilu0_join)
  std::thread pipe_builder;
  int pipe_builder_rc = DDM_OK;
  std::string pipe_builder_err;
  // box engine (mode 32; trsv_box_host.hpp): structured leading box of every block + a nested factor for the rows behind it
  struct BoxEngine *box = nullptr;
  bool allow_box = true;        // (false for the nested shell factor and unless DDM_TRSV_MODE=box asks for the engine)
  pipe::Group *p_groups = nullptr;
  pipe::Task *p_tasks = nullptr;
  unsigned char *p_stream = nullptr;
  int32_t *p_koff = nullptr, *p_posU = nullptr, *p_rowU = nullptr; // p_rowU: natural row of every U position (-1: padding)
  double *p_ypos = nullptr, *p_xpos = nullptr;
  unsigned long long *p_progress = nullptr;
  unsigned *p_queue = nullptr;
  int64_t p_nposL = 0, p_nposU = 0;
  int p_spread = 0;             // placement-independent mode (set when a subdomain has more work per level than one XCD's workgroups take)
  int p_grid = 0;
  pipe::Stats p_stats;
  int64_t p_stream_bytes = 0;
  unsigned *err = nullptr;
  // sparse direct factor (ddm_chol_create): the factor lives in a fill-reducing order; d / x are permuted around the solve
  ddm_csr *own_pattern = nullptr; // host-only CSR pattern of L + D + L^T in the permuted order (owned)
  int32_t *perm = nullptr;        // device: perm[new] = old
  double *pd = nullptr, *px = nullptr; // permuted right-hand side / solution (n doubles each)
  double *pD = nullptr, *pX = nullptr; // the same for row-major blocks (n x pm_nrhs)
  int pm_nrhs = 0;
  int direct = 0;
  double direct_flops = 0.0;
  sn::Factor *sn = nullptr;       // supernodal factor computed ON THE DEVICE (sn_chol.hpp); solves run on its panels, in place in pd / pD
  // iterative refinement of the device engine (dune/ddm/eigensolvers/umfpack.hh:42-129; UMFPACK refines inside its own solve too):
  // the number of steps is fixed when the factor is created, from the backward error of a probe solve (sn_direct_create), so that
  // the solves stay captured HIP graphs; the matrix is kept as device copies of its three arrays
  int refine_steps = 0;
  double refine_omega[5] = {0, 0, 0, 0, 0}; // backward error of the probe after 0, 1, .. steps
  int64_t *ref_rp = nullptr;
  int32_t *ref_ci = nullptr;
  double *ref_va = nullptr, *pr = nullptr; // pr: residual block (n x pr_cols)
  int pr_cols = 0;
  int64_t nvirt = 0; // virtual unknowns of the supernodal transformation: the permuted work vectors hold n + nvirt entries
  hvec<double> h_lu; // factor values in the pattern of A
  TriSchedule L, U;
  TriCsr Lc, Uc;   // direct factors use these instead of L / U (global levels: multi-RHS solves, one launch per level)
  TriCsr Lb, Ub;   // the same factors ordered by (block, level): single right-hand side, one workgroup per block
  // HIP graph cache of the multi-RHS solve for one (D, X, nrhs, ld) combination
  hipGraphExec_t mgraph = nullptr;
  const double *mg_D = nullptr;
  double *mg_X = nullptr;
  int mg_nrhs = 0;
  int64_t mg_ldd = 0, mg_ldx = 0;
  bool mg_f32 = false;   // the cached graph runs the single-precision sweeps
  float *xf = nullptr;   // their n x nrhs work block
  int xf_nrhs = 0;
  // HIP graph cache of the whole solve for one (d, x) pointer pair
  hipGraphExec_t graph = nullptr;
  const double *g_d = nullptr;
  double *g_x = nullptr;
  const double *g_scale = nullptr, *g_add = nullptr; // epilogue operands the captured graph was built with
};

static constexpr int SMALL_LEVEL_ROWS = 2048;
static constexpr int SMALL_LEVELS_PER_LAUNCH = 256;

// Builds the level schedule of the lower (upper=false) or upper factor.
static int build_schedule(ddm_ctx *ctx, const ddm_csr *A, const hvec<double> &lu, const std::vector<int64_t> &diag,
                          bool upper, TriSchedule &S)
{
  const int64_t n = A->nrows;
  const int64_t *rp = A->h_rp.data();
  const int32_t *ci = A->h_ci.data();
  std::vector<int32_t> level(n, 0);
  int32_t maxlev = -1;
  if (!upper) {
    for (int64_t i = 0; i < n; ++i) {
      int32_t l = 0;
      for (int64_t k = rp[i]; k < diag[i]; ++k) l = std::max(l, level[ci[k]] + 1);
      level[i] = l;
      maxlev = std::max(maxlev, l);
    }
  } else {
    for (int64_t i = n - 1; i >= 0; --i) {
      int32_t l = 0;
      for (int64_t k = diag[i] + 1; k < rp[i + 1]; ++k) l = std::max(l, level[ci[k]] + 1);
      level[i] = l;
      maxlev = std::max(maxlev, l);
    }
  }
  const int64_t nlev = (int64_t)maxlev + 1;
  S.nlev = nlev;
  std::vector<int64_t> lptr(nlev + 1, 0);
  for (int64_t i = 0; i < n; ++i) lptr[level[i] + 1]++;
  for (int64_t l = 0; l < nlev; ++l) lptr[l + 1] += lptr[l];
  std::vector<int32_t> rows(n);
  {
    std::vector<int64_t> pos(lptr.begin(), lptr.end() - 1);
    for (int64_t i = 0; i < n; ++i) rows[pos[level[i]]++] = (int32_t)i; // ascending row inside a level
  }
  S.desc.resize(nlev);
  int64_t ent = 0;
  for (int64_t l = 0; l < nlev; ++l) {
    const int64_t m = lptr[l + 1] - lptr[l];
    int w = 0;
    for (int64_t r = lptr[l]; r < lptr[l + 1]; ++r) {
      const int64_t i = rows[r];
      const int cnt = upper ? (int)(rp[i + 1] - diag[i] - 1) : (int)(diag[i] - rp[i]);
      w = std::max(w, cnt);
    }
    S.desc[l] = LevelDesc{(int32_t)m, (int32_t)w, lptr[l], ent};
    ent += m * (int64_t)w;
  }
  S.ell_entries = ent;
  hvec<int32_t> cols((size_t)std::max<int64_t>(ent, 1));
  hvec<double> vals((size_t)std::max<int64_t>(ent, 1));
  hvec<double> dinv;
  if (upper) dinv.resize(n);
  // the sliced-ELL fill (strided writes, 1.8 GB per triangle at 216^3) on several threads: levels are independent, each thread takes a
  // run of consecutive levels with about the same number of entries (the two triangles are built at the same time: half the cores each)
  const int nfill = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)std::max(1u, host_threads() / 2), nlev, ent / (1 << 20) + 1}));
  std::vector<int64_t> cut((size_t)nfill + 1, nlev);
  cut[0] = 0;
  for (int t = 1, l = 0; t < nfill; ++t) {
    while (l < nlev && S.desc[l].ent_off < ent * t / nfill) ++l;
    cut[(size_t)t] = l;
  }
  auto fill = [&](int64_t l0, int64_t l1) {
  for (int64_t l = l0; l < l1; ++l) {
    const LevelDesc &D = S.desc[l];
    for (int64_t r = 0; r < D.m; ++r) {
      const int64_t i = rows[D.row_off + r];
      const int64_t k0 = upper ? diag[i] + 1 : rp[i];
      const int64_t k1 = upper ? rp[i + 1] : diag[i];
      int k = 0;
      for (int64_t p = k0; p < k1; ++p, ++k) {
        cols[D.ent_off + (int64_t)k * D.m + r] = ci[p];
        vals[D.ent_off + (int64_t)k * D.m + r] = lu[p];
      }
      for (; k < D.w; ++k) { // padding: a dependency that is already resolved, value 0
        cols[D.ent_off + (int64_t)k * D.m + r] = ci[k0];
        vals[D.ent_off + (int64_t)k * D.m + r] = 0.0;
      }
      if (upper) dinv[D.row_off + r] = lu[diag[i]];
    }
  }
  };
  if (nfill <= 1) fill(0, nlev);
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < nfill; ++t) th.emplace_back(fill, cut[(size_t)t], cut[(size_t)t + 1]);
    for (auto &t : th) t.join();
  }
  // launch plan: runs of small levels share one single-workgroup launch
  int l = 0;
  while (l < nlev) {
    if (S.desc[l].m <= SMALL_LEVEL_ROWS) {
      int c = 0;
      while (l + c < nlev && c < SMALL_LEVELS_PER_LAUNCH && S.desc[l + c].m <= SMALL_LEVEL_ROWS) ++c;
      S.plan.push_back({l, c, true});
      l += c;
    } else {
      S.plan.push_back({l, 1, false});
      l += 1;
    }
  }
  DDMCHECK(upload(ctx, rows.data(), n, &S.rows));
  DDMCHECK(upload(ctx, cols.data(), ent, &S.cols));
  DDMCHECK(upload(ctx, vals.data(), ent, &S.vals));
  if (upper) DDMCHECK(upload(ctx, dinv.data(), n, &S.dinv));
  DDMCHECK(upload(ctx, S.desc.data(), nlev, &S.d_desc));
  return DDM_OK;
}

// Supernodes of a direct factor: maximal runs J = [j0, j1) of consecutive eliminated indices whose diagonal block L[J, J] is a
// dense triangle (row i of J holds all columns j0 .. i-1; by the symmetric pattern of the factor U[J, J] is dense as well) -- the
// separators of the nested dissection.  Solving through such a block row by row costs |J| dependency levels; with the diagonal
// blocks INVERTED once on the host (dense triangular inverses, |J|^3 / 3 flops) it costs two:
//   t_J = rhs_J - F[J, outside J] x      (|J| independent rows; results in virtual unknowns n + q)
//   x_J = T_J^-1 t_J                     (|J| independent rows of the inverted block)
// which is how sparse triangular solves are usually made parallel on GPUs.  The inverse has as many entries as the triangle it
// replaces.  min_size: smaller runs stay row by row.
struct Supernodes {
  std::vector<int64_t> j0, j1;
  std::vector<int32_t> sn_of;   // supernode of a row or -1
  std::vector<int32_t> virt_of; // virtual unknown (>= n) of a supernode row
  int64_t nvirt = 0;
  std::vector<std::vector<double>> Linv, Uinv; // inverted diagonal blocks (dense s x s, row-major), filled by invert_supernodes
};
// T^-1 of the unit lower / M^-1 of the upper (pivots on the diagonal) diagonal block of every supernode; row-oriented substitution
// (row i of the inverse is a combination of the finished rows: contiguous updates), supernodes in parallel on the host threads
static void invert_supernodes(const hvec<double> &lu, const std::vector<int64_t> &diag, Supernodes &SN)
{
  const size_t ns = SN.j0.size();
  SN.Linv.assign(ns, {});
  SN.Uinv.assign(ns, {});
  std::vector<size_t> order(ns);
  for (size_t q = 0; q < ns; ++q) order[q] = q;
  std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return SN.j1[a] - SN.j0[a] > SN.j1[b] - SN.j0[b]; }); // largest first
  const unsigned hw = host_threads();
  const int nth = (int)std::min<size_t>(hw, std::max<size_t>(ns, 1));
  std::atomic<size_t> next{0};
  std::vector<std::thread> th;
  for (int t = 0; t < nth; ++t)
    th.emplace_back([&]() {
      for (;;) {
        const size_t w = next.fetch_add(1);
        if (w >= ns) break;
        const size_t id = order[w];
        const int64_t j0 = SN.j0[id], j1 = SN.j1[id], sz = j1 - j0;
        std::vector<double> &Li = SN.Linv[id], &Ui = SN.Uinv[id];
        Li.assign((size_t)(sz * sz), 0.0);
        Ui.assign((size_t)(sz * sz), 0.0);
        for (int64_t i = 0; i < sz; ++i) { // Linv[i, :] = e_i - sum_{k < i} L[i, k] Linv[k, :]
          double *ri = Li.data() + i * sz;
          ri[i] = 1.0;
          const int64_t gi = j0 + i;
          for (int64_t k = 0; k < i; ++k) {
            const double l = lu[diag[gi] - (i - k)];
            if (l == 0.0) continue;
            const double *rk = Li.data() + k * sz;
            for (int64_t c = 0; c <= k; ++c) ri[c] -= l * rk[c];
          }
        }
        for (int64_t i = sz - 1; i >= 0; --i) { // Uinv[i, :] = dinv_i (e_i - sum_{k > i} U[i, k] Uinv[k, :])
          double *ri = Ui.data() + i * sz;
          ri[i] = 1.0;
          const int64_t gi = j0 + i;
          for (int64_t k = i + 1; k < sz; ++k) {
            const double u = lu[diag[gi] + (k - i)];
            if (u == 0.0) continue;
            const double *rk = Ui.data() + k * sz;
            for (int64_t c = k; c < sz; ++c) ri[c] -= u * rk[c];
          }
          const double dv = lu[diag[gi]]; // stored inverse pivot
          for (int64_t c = i; c < sz; ++c) ri[c] *= dv;
        }
      }
    });
  for (auto &t : th) t.join();
}
static Supernodes detect_supernodes(const ddm_csr *A, const std::vector<int64_t> &diag, int min_size)
{
  const int64_t n = A->nrows;
  const int64_t *rp = A->h_rp.data();
  const int32_t *ci = A->h_ci.data();
  Supernodes SN;
  SN.sn_of.assign((size_t)n, -1);
  SN.virt_of.assign((size_t)n, -1);
  int64_t j0 = 0;
  while (j0 < n) {
    int64_t j1 = j0 + 1;
    while (j1 < n) {
      const int64_t w = j1 - j0;
      if (diag[j1] - rp[j1] < w || ci[diag[j1] - w] != j0) break;                 // row j1 holds columns j0 .. j1-1
      if (rp[j0 + 1] - diag[j0] - 1 < w || ci[diag[j0] + w] != j1) break;         // row j0 holds column j1 (upper part)
      ++j1;
    }
    bool ok = j1 - j0 >= min_size;
    for (int64_t i = j0; ok && i < j1; ++i) // every row of the run holds i+1 .. j1-1 right behind its diagonal
      ok = (rp[i + 1] - diag[i] - 1 >= j1 - 1 - i) && (i == j1 - 1 || ci[diag[i] + (j1 - 1 - i)] == j1 - 1);
    if (ok) {
      const int32_t id = (int32_t)SN.j0.size();
      SN.j0.push_back(j0);
      SN.j1.push_back(j1);
      for (int64_t i = j0; i < j1; ++i) {
        SN.sn_of[(size_t)i] = id;
        SN.virt_of[(size_t)i] = (int32_t)(n + SN.nvirt++);
      }
    }
    j0 = ok ? j1 : j0 + 1;
  }
  return SN;
}

// block_ptr != nullptr: rows ordered by (block, level), levels numbered per block (blk_lev_ptr), for k_trsv_csr_blocks
static int build_csr_schedule(ddm_ctx *ctx, const ddm_csr *A, const hvec<double> &lu, const std::vector<int64_t> &diag, bool upper, TriCsr &S,
                              const Supernodes &SN, int64_t nblocks = 0, const int64_t *block_ptr = nullptr)
{
  const int64_t n = A->nrows;
  const int64_t *rp = A->h_rp.data();
  const int32_t *ci = A->h_ci.data();
  const int64_t nunk = n + SN.nvirt;
  // ---- transformed rows: dst <- (rhs >= 0 ? rhsvec[rhs] : 0) - sum val * x[col], times dinv ----
  std::vector<int64_t> rptr(1, 0);
  std::vector<int32_t> rcol, rdst, rrhs, rown; // rown: original row the transformed row belongs to (for the block id)
  std::vector<double> rval, rdinv;
  rdst.reserve((size_t)nunk);
  std::vector<int32_t> level((size_t)nunk, 0);
  auto finish_row = [&](int32_t dst, int32_t rhs, double dv, int32_t owner) {
    int32_t l = 0;
    for (int64_t k = rptr.back(); k < (int64_t)rcol.size(); ++k) l = std::max(l, level[(size_t)rcol[(size_t)k]] + 1);
    level[(size_t)dst] = l;
    rptr.push_back((int64_t)rcol.size());
    rdst.push_back(dst);
    rrhs.push_back(rhs);
    rdinv.push_back(dv);
    rown.push_back(owner);
  };
  auto do_supernode = [&](int32_t id) {
    const int64_t j0 = SN.j0[(size_t)id], j1 = SN.j1[(size_t)id], s = j1 - j0;
    const std::vector<double> &Ti = upper ? SN.Uinv[(size_t)id] : SN.Linv[(size_t)id];
    if (!upper) {
      for (int64_t i = j0; i < j1; ++i) { // phase 1: t_i = d_i - F[i, < j0] x
        rcol.insert(rcol.end(), ci + rp[i], ci + (diag[i] - (i - j0)));
        rval.insert(rval.end(), lu.begin() + rp[i], lu.begin() + (diag[i] - (i - j0)));
        finish_row(SN.virt_of[(size_t)i], (int32_t)i, 1.0, (int32_t)i);
      }
      for (int64_t i = j0; i < j1; ++i) { // phase 2: x_i = sum_{c <= i} Tinv[i, c] t_c
        for (int64_t c = j0; c <= i; ++c) {
          rcol.push_back(SN.virt_of[(size_t)c]);
          rval.push_back(-Ti[(size_t)((i - j0) * s + (c - j0))]);
        }
        finish_row((int32_t)i, -1, 1.0, (int32_t)i);
      }
    } else {
      for (int64_t i = j1 - 1; i >= j0; --i) { // phase 1: t_i = y_i - F[i, >= j1] x   (y_i is read from x[i])
        rcol.insert(rcol.end(), ci + (diag[i] + (j1 - i)), ci + rp[i + 1]);
        rval.insert(rval.end(), lu.begin() + (diag[i] + (j1 - i)), lu.begin() + rp[i + 1]);
        finish_row(SN.virt_of[(size_t)i], (int32_t)i, 1.0, (int32_t)i);
      }
      for (int64_t i = j1 - 1; i >= j0; --i) { // phase 2: x_i = sum_{c >= i} Minv[i, c] t_c
        for (int64_t c = i; c < j1; ++c) {
          rcol.push_back(SN.virt_of[(size_t)c]);
          rval.push_back(-Ti[(size_t)((i - j0) * s + (c - j0))]);
        }
        finish_row((int32_t)i, -1, 1.0, (int32_t)i);
      }
    }
  };
  if (!upper) {
    for (int64_t i = 0; i < n; ++i) {
      const int32_t id = SN.sn_of[(size_t)i];
      if (id >= 0) {
        if (i == SN.j0[(size_t)id]) do_supernode(id);
        continue;
      }
      rcol.insert(rcol.end(), ci + rp[i], ci + diag[i]);
      rval.insert(rval.end(), lu.begin() + rp[i], lu.begin() + diag[i]);
      finish_row((int32_t)i, (int32_t)i, 1.0, (int32_t)i);
    }
  } else {
    for (int64_t i = n - 1; i >= 0; --i) {
      const int32_t id = SN.sn_of[(size_t)i];
      if (id >= 0) {
        if (i == SN.j1[(size_t)id] - 1) do_supernode(id);
        continue;
      }
      rcol.insert(rcol.end(), ci + diag[i] + 1, ci + rp[i + 1]);
      rval.insert(rval.end(), lu.begin() + diag[i] + 1, lu.begin() + rp[i + 1]);
      finish_row((int32_t)i, (int32_t)i, lu[diag[i]], (int32_t)i);
    }
  }
  const int64_t nr = (int64_t)rdst.size();
  // ---- levels (per block when block_ptr is given) ----
  std::vector<int32_t> rlev((size_t)nr);
  int32_t maxlev = -1;
  for (int64_t q = 0; q < nr; ++q) {
    rlev[(size_t)q] = level[(size_t)rdst[(size_t)q]];
    maxlev = std::max(maxlev, rlev[(size_t)q]);
  }
  int64_t nlev = (int64_t)maxlev + 1;
  std::vector<int32_t> blp;
  if (block_ptr) {
    std::vector<int32_t> blk_of((size_t)n);
    for (int64_t b = 0; b < nblocks; ++b)
      for (int64_t i = block_ptr[b]; i < block_ptr[b + 1]; ++i) blk_of[(size_t)i] = (int32_t)b;
    std::vector<int32_t> mx((size_t)nblocks, -1);
    for (int64_t q = 0; q < nr; ++q) mx[(size_t)blk_of[(size_t)rown[(size_t)q]]] = std::max(mx[(size_t)blk_of[(size_t)rown[(size_t)q]]], rlev[(size_t)q]);
    blp.assign(1, 0);
    for (int64_t b = 0; b < nblocks; ++b) blp.push_back(blp.back() + mx[(size_t)b] + 1);
    for (int64_t q = 0; q < nr; ++q) rlev[(size_t)q] += blp[(size_t)blk_of[(size_t)rown[(size_t)q]]];
    nlev = blp.back();
    S.nblocks = (int)nblocks;
  }
  S.nlev = nlev;
  std::vector<int64_t> lptr((size_t)nlev + 1, 0);
  for (int64_t q = 0; q < nr; ++q) lptr[(size_t)rlev[(size_t)q] + 1]++;
  for (int64_t l = 0; l < nlev; ++l) lptr[(size_t)l + 1] += lptr[(size_t)l];
  std::vector<int64_t> order((size_t)nr);
  {
    std::vector<int64_t> pos(lptr.begin(), lptr.end() - 1);
    for (int64_t q = 0; q < nr; ++q) order[(size_t)pos[(size_t)rlev[(size_t)q]]++] = q; // stable inside a level
  }
  std::vector<int32_t> rows((size_t)nr), rhs((size_t)nr), cols((size_t)std::max<int64_t>((int64_t)rcol.size(), 1));
  std::vector<int64_t> lrp((size_t)nr + 1, 0);
  std::vector<double> vals((size_t)std::max<int64_t>((int64_t)rval.size(), 1)), dinv((size_t)nr);
  for (int64_t t = 0; t < nr; ++t) {
    const int64_t q = order[(size_t)t];
    rows[(size_t)t] = rdst[(size_t)q];
    rhs[(size_t)t] = rrhs[(size_t)q];
    dinv[(size_t)t] = rdinv[(size_t)q];
    const int64_t len = rptr[(size_t)q + 1] - rptr[(size_t)q];
    lrp[(size_t)t + 1] = lrp[(size_t)t] + len;
    std::copy(rcol.begin() + rptr[(size_t)q], rcol.begin() + rptr[(size_t)q + 1], cols.begin() + lrp[(size_t)t]);
    std::copy(rval.begin() + rptr[(size_t)q], rval.begin() + rptr[(size_t)q + 1], vals.begin() + lrp[(size_t)t]);
  }
  S.desc.resize((size_t)nlev);
  for (int64_t l = 0; l < nlev; ++l) {
    const int64_t m = lptr[(size_t)l + 1] - lptr[(size_t)l];
    const int64_t ent = lrp[(size_t)lptr[(size_t)l + 1]] - lrp[(size_t)lptr[(size_t)l]];
    int Sl = 1; // lanes per row: about a quarter of the average row length
    while (Sl < 64 && 4 * Sl * m < ent) Sl <<= 1;
    S.desc[(size_t)l] = CsrLevel{(int32_t)m, Sl, lptr[(size_t)l]};
  }
  int l = 0;
  while (l < nlev) { // runs of levels whose rows x lanes fit a few rounds of one workgroup are fused
    auto small = [&](int q) { return (int64_t)S.desc[(size_t)q].m * S.desc[(size_t)q].S <= 4 * TRSV_SMALL_WG; };
    if (small(l)) {
      int c = 0;
      while (l + c < nlev && c < 4096 && small(l + c)) ++c;
      S.plan.push_back({l, c, true});
      l += c;
    } else {
      S.plan.push_back({l, 1, false});
      l += 1;
    }
  }
  S.nrows = nr;
  S.entries = (int64_t)rcol.size();
  DDMCHECK(upload(ctx, rows.data(), nr, &S.rows));
  DDMCHECK(upload(ctx, rhs.data(), nr, &S.rhs));
  DDMCHECK(upload(ctx, lrp.data(), nr + 1, &S.lrp));
  DDMCHECK(upload(ctx, cols.data(), (int64_t)rcol.size(), &S.cols));
  DDMCHECK(upload(ctx, vals.data(), (int64_t)rval.size(), &S.vals));
  if (upper) DDMCHECK(upload(ctx, dinv.data(), nr, &S.dinv));
  DDMCHECK(upload(ctx, S.desc.data(), nlev, &S.d_desc));
  if (block_ptr) DDMCHECK(upload(ctx, blp.data(), (int64_t)blp.size(), &S.blk_lev_ptr));
  return DDM_OK;
}
static void free_csr_schedule(TriCsr &S)
{
  (void)hipFree(S.blk_lev_ptr);
  (void)hipFree(S.rhs);
  (void)hipFree(S.rows);
  (void)hipFree(S.lrp);
  (void)hipFree(S.cols);
  (void)hipFree(S.vals);
  (void)hipFree(S.dinv);
  (void)hipFree(S.d_desc);
}
static int enqueue_tri_csr(ddm_ctx *ctx, const TriCsr &S, bool upper, const double *d, double *x)
{
  if (S.nblocks > 0) { // one workgroup per independent block
    if (upper)
      hipLaunchKernelGGL(k_trsv_csr_blocks<true>, dim3(S.nblocks), dim3(TRSV_SMALL_WG), 0, ctx->stream, S.blk_lev_ptr, S.d_desc, S.rows, S.rhs, S.lrp, S.cols, S.vals, S.dinv, d, x);
    else
      hipLaunchKernelGGL(k_trsv_csr_blocks<false>, dim3(S.nblocks), dim3(TRSV_SMALL_WG), 0, ctx->stream, S.blk_lev_ptr, S.d_desc, S.rows, S.rhs, S.lrp, S.cols, S.vals, S.dinv, d, x);
    HIPCHECK(ctx, hipGetLastError());
    return DDM_OK;
  }
  for (const auto &p : S.plan) {
    if (p.fused) {
      if (upper)
        hipLaunchKernelGGL(k_trsv_csr_fused<true>, dim3(1), dim3(TRSV_SMALL_WG), 0, ctx->stream, p.count, S.d_desc + p.first, S.rows, S.rhs, S.lrp, S.cols, S.vals, S.dinv, d, x);
      else
        hipLaunchKernelGGL(k_trsv_csr_fused<false>, dim3(1), dim3(TRSV_SMALL_WG), 0, ctx->stream, p.count, S.d_desc + p.first, S.rows, S.rhs, S.lrp, S.cols, S.vals, S.dinv, d, x);
    } else {
      const CsrLevel &L = S.desc[p.first];
      const int gpb = WG / L.S;
      const int grid = (int)std::min<int64_t>(((int64_t)L.m + gpb - 1) / gpb, 8192);
      if (upper)
        hipLaunchKernelGGL(k_trsv_csr_level<true>, dim3(grid), dim3(WG), 0, ctx->stream, L, S.rows, S.rhs, S.lrp, S.cols, S.vals, S.dinv, d, x);
      else
        hipLaunchKernelGGL(k_trsv_csr_level<false>, dim3(grid), dim3(WG), 0, ctx->stream, L, S.rows, S.rhs, S.lrp, S.cols, S.vals, S.dinv, d, x);
    }
  }
  HIPCHECK(ctx, hipGetLastError());
  return DDM_OK;
}
static void enqueue_multi_levels_csr(ddm_ctx *ctx, const TriCsr &S, bool upper, int nrhs, const double *D, int64_t ldd, double *X, int64_t ldx)
{
  int smax = 1;
  while (2 * smax * nrhs <= WG && smax < 64) smax <<= 1;
  for (int64_t l = 0; l < S.nlev; ++l) {
    const CsrLevel &L = S.desc[l];
    if (L.m == 0) continue;
    const int Sm = std::min(L.S, smax);
    const int rpb = WG / (Sm * nrhs);
    const unsigned grid = (unsigned)((L.m + rpb - 1) / rpb);
    if (upper)
      hipLaunchKernelGGL(k_trsv_csr_level_multi<true>, dim3(grid), dim3(WG), 0, ctx->stream, L, Sm, nrhs, S.rows, S.rhs, S.lrp, S.cols, S.vals, S.dinv, D, ldd, X, ldx);
    else
      hipLaunchKernelGGL(k_trsv_csr_level_multi<false>, dim3(grid), dim3(WG), 0, ctx->stream, L, Sm, nrhs, S.rows, S.rhs, S.lrp, S.cols, S.vals, S.dinv, D, ldd, X, ldx);
  }
}
static void free_schedule(TriSchedule &S)
{
  (void)hipFree(S.rows);
  (void)hipFree(S.cols);
  (void)hipFree(S.vals);
  (void)hipFree(S.dinv);
  (void)hipFree(S.vals_f32);
  (void)hipFree(S.dinv_f32);
  (void)hipFree(S.d_desc);
}

struct BoxEngine {
  int nblocks = 0;
  int64_t nshell = 0, nprod = 0;
  box::Block *blocks = nullptr;
  box::StepTab *steps = nullptr;
  double *stream = nullptr;
  unsigned long long *einfo = nullptr;
  double *E = nullptr, *ext_val = nullptr;
  int32_t *ext_col = nullptr;
  double *xs = nullptr;
  unsigned long long *prog = nullptr;
  unsigned *queue = nullptr;
  unsigned long long *dbg = nullptr;    // DDM_BOX_CHECK: pinned host words of the kernels' address check
  int64_t n = 0, stream_len = 0, xs_len = 0, prog_len = 0, einfo_len = 0;
  // shell system
  int64_t *srp = nullptr;
  int32_t *sci = nullptr, *srow = nullptr;
  double *sva = nullptr, *ds = nullptr, *xsol = nullptr;
  ddm_csr *shell_csr = nullptr;
  ddm_ilu0 *shell = nullptr;
  int grid = 0;
  box::Stats stats;
};
static int build_pipe_schedule(ddm_ctx *ctx, ddm_ilu0 *F);
static int build_box_engine(ddm_ctx *ctx, ddm_ilu0 *F);
// waits for the background part of the setup; its failure is reported by every call that needs the result
static int ilu0_join(ddm_ctx *ctx, ddm_ilu0 *F)
{
  if (F->pipe_builder.joinable()) F->pipe_builder.join();
  if (F->pipe_builder_rc) return fail(ctx, F->pipe_builder_rc, "%s", F->pipe_builder_err.c_str());
  return DDM_OK;
}
// diagnostic: the stamps of the box engine's last solve (DDM_BOX_CHECK=1 at creation): out[2][128][4] = per sweep and plane of block 0
// {start, end (100 MHz clock), polls of the previous plane's progress word, XCC}; zeros without the switch
extern "C" int ddm_ilu0_box_check(const ddm_ilu0 *F, unsigned long long *out1024)
{
  if (!F || !out1024) return DDM_EINVAL;
  for (int k = 0; k < 1024; ++k) out1024[k] = (F->box && F->box->dbg) ? F->box->dbg[k] : 0ull;
  return DDM_OK;
}
extern "C" int ddm_ilu0_wait(ddm_ctx *ctx, ddm_ilu0 *F) { return F ? ilu0_join(ctx, F) : fail(ctx, DDM_EINVAL, "ddm_ilu0_wait: bad arguments"); }
// Level schedules, engine selection and the pipe schedule for factor values F->h_lu stored in the pattern of A.
static int ilu0_build_engines(ddm_ctx *ctx, ddm_ilu0 *F, const ddm_csr *A, const std::vector<int64_t> &diag, int64_t nblocks, const int64_t *block_ptr,
                              bool multi_rhs_only = false)
{
  int rc = DDM_OK;
  if (F->direct) {
    int min_sn = 8;
    if (const char *e = std::getenv("DDM_DIRECT_SUPERNODE_MIN")) min_sn = std::max(2, std::atoi(e)); // a huge value switches the transformation off
    Supernodes SN = detect_supernodes(A, diag, min_sn);
    invert_supernodes(F->h_lu, diag, SN);
    F->nvirt = SN.nvirt;
    rc = build_csr_schedule(ctx, A, F->h_lu, diag, false, F->Lc, SN);
    if (!rc) rc = build_csr_schedule(ctx, A, F->h_lu, diag, true, F->Uc, SN);
    // Few, large levels (the supernodal transformation worked): one grid-wide launch per level.  Thousands of small levels and
    // enough independent blocks: one workgroup per block walks its levels with workgroup barriers instead.
    if (!rc && nblocks >= 4 && F->Lc.nlev + F->Uc.nlev > 600) {
      rc = build_csr_schedule(ctx, A, F->h_lu, diag, false, F->Lb, SN, nblocks, block_ptr);
      if (!rc) rc = build_csr_schedule(ctx, A, F->h_lu, diag, true, F->Ub, SN, nblocks, block_ptr);
    }
    if (std::getenv("DDM_PIPE_VERBOSE"))
      std::fprintf(stderr, "[ddm] direct factor: %lld rows, %lld stored entries; %zu supernodes (>= %d rows) with %lld rows; levels L/U %lld/%lld (transformed rows %lld)\n",
                   (long long)F->n, (long long)F->nnz, SN.j0.size(), min_sn, (long long)SN.nvirt, (long long)F->Lc.nlev, (long long)F->Uc.nlev, (long long)F->Lc.nrows);
    F->L.nlev = F->Lc.nlev;
    F->U.nlev = F->Uc.nlev;
  }
  // the box engine (structured blocks, trsv_box.hpp) is opt-in: bit-exact, but at the benchmark's size still slower than pipe (4.4 against
  // 3.25 ms per solve: DESIGN.md section 3d says what bounds it and what is missing)
  bool want_box = false;
  if (const char *m = std::getenv("DDM_TRSV_MODE")) {
    F->mode = !std::strcmp(m, "levels") ? 0 : (!std::strcmp(m, "xcd2") ? 4 : 8);
    want_box = !std::strcmp(m, "box");
  }
  if (!want_box) F->allow_box = false;
  if (multi_rhs_only) F->mode = 0; // only ddm_ilu0_solve_multi will be called (level kernels): no pipe schedule, no tile stream
  F->A = A;
  F->h_diag = diag;
  F->h_block_ptr.assign(block_ptr, block_ptr + nblocks + 1);
  bool pipe_started = false;
  if (!F->direct) { // the two triangles of the level schedules on two host threads (each is a single pass over the factor with scattered
                    // writes: 1.3 s at 216^3), the pipe schedule (its own thread pool) beside them
    int rcU = DDM_OK;
    std::thread tu([&]() {
      (void)hipSetDevice(ctx->device);
      BackgroundTransfers own_stream;   // (this create may itself run on a background thread: the box engine's nested factor)
      rcU = build_schedule(ctx, A, F->h_lu, diag, true, F->U);
    });
    if (F->mode == 8 && F->n > 0) {
      pipe_started = true;
      F->pipe_builder = std::thread([ctx, F]() {
        (void)hipSetDevice(ctx->device);
        BackgroundTransfers own_stream;
        int rcb = DDM_OK;
        if (F->allow_box) rcb = build_box_engine(ctx, F);   // structured blocks: mode 32 (declined: F->box stays null, pipe takes the matrix)
        F->pipe_builder_rc = rcb ? rcb : (F->box ? DDM_OK : build_pipe_schedule(ctx, F)); // (not applicable: pipe_state < 0, see ddm_ilu0_solve)
        if (F->pipe_builder_rc) F->pipe_builder_err = last_error_of_this_thread();
      });
    }
    rc = build_schedule(ctx, A, F->h_lu, diag, false, F->L);
    tu.join();
    if (!rc) rc = rcU;
    static const bool background = !std::getenv("DDM_PIPE_ASYNC") || std::atoi(std::getenv("DDM_PIPE_ASYNC")) != 0;
    if (!background || rc) {     // DDM_PIPE_ASYNC=0: the whole setup inside the create call, as before round 4
      const int rcj = ilu0_join(ctx, F);
      if (!rc) rc = rcj;
    }
  }
  // status word of the single-launch engines in pinned, device-mapped HOST memory: a wave that gives up waiting writes its code
  // straight into it, so the host can look at it without synchronising the stream (ilu0_peek_status: every apply checks the
  // applies before it -- fail fast instead of returning stale results until somebody calls ddm_ilu0_status)
  if (!rc) {
    if (hipHostMalloc((void **)&F->err, 128, hipHostMallocMapped) != hipSuccess) rc = fail(ctx, DDM_EHIP, "ILU(0): allocation failed");
    else std::memset(F->err, 0, 128);
  }
  if (!rc && F->mode == 8 && F->n > 0 && !pipe_started) { // part of the setup, not of the first solve
    if (F->allow_box) rc = build_box_engine(ctx, F);
    if (!rc && !F->box) rc = build_pipe_schedule(ctx, F);
  }
  return rc;
}
static int ilu0_create_impl(ddm_ctx *ctx, const ddm_csr *A, int64_t nblocks, const int64_t *block_ptr, bool multi_rhs_only, ddm_ilu0 **out);
extern "C" int ddm_ilu0_create(ddm_ctx *ctx, const ddm_csr *A, int64_t nblocks, const int64_t *block_ptr, ddm_ilu0 **out)
{
  return ilu0_create_impl(ctx, A, nblocks, block_ptr, false, out);
}
static int ilu0_create_impl(ddm_ctx *ctx, const ddm_csr *A, int64_t nblocks, const int64_t *block_ptr, bool multi_rhs_only, ddm_ilu0 **out)
{
  if (!ctx || !A || !out || nblocks < 1 || !block_ptr) return fail(ctx, DDM_EINVAL, "ddm_ilu0_create: bad arguments");
  if (A->nrows != A->ncols) return fail(ctx, DDM_EINVAL, "ILU(0) needs a square matrix");
  if (block_ptr[0] != 0 || block_ptr[nblocks] != A->nrows) return fail(ctx, DDM_EINVAL, "block_ptr does not cover the matrix");
  const auto t_begin = std::chrono::steady_clock::now();
  auto since = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(); };
  ddm_ilu0 *F = new ddm_ilu0;
  F->n = A->nrows;
  F->nnz = A->nnz;
  hvec_copy(F->h_lu, A->h_va.data(), A->h_va.size());
  std::vector<int64_t> diag(A->nrows);
  std::vector<int> rcs(nblocks, 0);
  {
    const unsigned hw = host_threads();
    const int nthreads = (int)std::min<int64_t>(nblocks, hw);
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t)
      th.emplace_back([&, t]() {
        for (int64_t b = t; b < nblocks; b += nthreads)
          rcs[b] = ilu0_factor_block(A->h_rp.data(), A->h_ci.data(), F->h_lu.data(), diag.data(), block_ptr[b], block_ptr[b + 1]);
      });
    for (auto &t : th) t.join();
  }
  for (int64_t b = 0; b < nblocks; ++b)
    if (rcs[b]) {
      const int rc = rcs[b];
      delete F;
      if (rc == -2) return fail(ctx, DDM_EINVAL, "ILU(0): block %lld has entries outside its diagonal block", (long long)b);
      if (rc == -3) return fail(ctx, DDM_EINVAL, "ILU(0): rows must have sorted column indices");
      return fail(ctx, DDM_ENUMERIC, "ILU(0): missing or zero pivot in block %lld", (long long)b);
    }
  const double t_factor = since();
  const int rc = ilu0_build_engines(ctx, F, A, diag, nblocks, block_ptr, multi_rhs_only);
  if (rc) {
    ddm_ilu0_destroy(F);
    return rc;
  }
  if (std::getenv("DDM_PIPE_VERBOSE"))
    std::fprintf(stderr, "[ddm] ILU(0) setup: %lld rows, factorisation (host, one thread per block) %.2f s, level schedules%s %.2f s\n", (long long)F->n, t_factor,
                 multi_rhs_only ? "" : (F->pipe_builder.joinable() ? " (single-launch engine: being built in the background)" : " + single-launch engine"), since() - t_factor);
  *out = F;
  return DDM_OK;
}

// ---- sparse direct local solver (host Cholesky, device triangular solves) ----------------------------------------
struct CholResult {
  std::vector<int32_t> perm; // perm[new] = old (rank-local indices; blocks stay contiguous)
  hvec<int64_t> rp;
  std::vector<int64_t> diag;
  hvec<int32_t> ci;
  hvec<double> lu;
  double flops = 0.0;
  int64_t nnzL = 0;
  std::string error;
};
// rc: DDM_OK, DDM_ENOTIMPL (more than max_flops: nothing was factorised), DDM_ENUMERIC (not positive definite), DDM_EINVAL
// general = true: L U without pivoting on the pattern of A + A^T (matrices with a positive definite symmetric part)
static int chol_build(int64_t n, const int64_t *rp, const int32_t *ci, const double *va, int64_t nblocks, const int64_t *block_ptr, double max_flops,
                      bool numeric, CholResult &R, bool general = false)
{
  if (n < 0 || !rp || !ci || nblocks < 1 || !block_ptr || block_ptr[0] != 0 || block_ptr[nblocks] != n) {
    R.error = "bad arguments";
    return DDM_EINVAL;
  }
  std::vector<chol::BlockFactor> BF((size_t)nblocks);
  std::vector<chol::PermutedLower> PL((size_t)nblocks);
  std::vector<chol::PermutedLowerLU> PU((size_t)(general ? nblocks : 0));
  std::vector<std::vector<double>> UX((size_t)(general ? nblocks : 0));
  const unsigned hw = host_threads();
  const int nthreads = (int)std::min<int64_t>(nblocks, hw);
  auto parallel = [&](auto fn) {
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t)
      th.emplace_back([&, t]() {
        for (int64_t b = t; b < nblocks; b += nthreads) fn(b);
      });
    for (auto &t : th) t.join();
  };
  std::vector<int> bad((size_t)nblocks, 0);
  parallel([&](int64_t b) {
    const int64_t r0 = block_ptr[b], r1 = block_ptr[b + 1];
    for (int64_t i = r0; i < r1 && !bad[(size_t)b]; ++i)
      for (int64_t k = rp[i]; k < rp[i + 1]; ++k)
        if (ci[k] < r0 || ci[k] >= r1) bad[(size_t)b] = 1;
    if (bad[(size_t)b]) return;
    chol::Graph G = chol::block_graph(rp, ci, r0, r1);
    BF[(size_t)b].perm = chol::nested_dissection(G);
    if (general) {
      PU[(size_t)b] = chol::permute_lower_lu(rp, ci, va, r0, r1, BF[(size_t)b].perm);
      chol::analyze(PU[(size_t)b].lo, (int32_t)(r1 - r0), BF[(size_t)b]);
    } else {
      PL[(size_t)b] = chol::permute_lower(rp, ci, va, r0, r1, BF[(size_t)b].perm);
      chol::analyze(PL[(size_t)b], (int32_t)(r1 - r0), BF[(size_t)b]);
    }
  });
  for (int64_t b = 0; b < nblocks; ++b)
    if (bad[(size_t)b]) {
      R.error = "block " + std::to_string(b) + " has entries outside its diagonal block";
      return DDM_EINVAL;
    }
  R.flops = 0.0;
  R.nnzL = 0;
  for (auto &f : BF) {
    R.flops += (general ? 2.0 : 1.0) * f.flops;
    R.nnzL += f.nnzL;
  }
  R.perm.resize((size_t)n);
  for (int64_t b = 0; b < nblocks; ++b)
    for (int32_t k = 0; k < BF[(size_t)b].n; ++k) R.perm[(size_t)(block_ptr[b] + k)] = (int32_t)(block_ptr[b] + BF[(size_t)b].perm[(size_t)k]);
  if (max_flops > 0.0 && R.flops > max_flops) {
    R.error = "sparse direct factorisation needs " + std::to_string(R.flops) + " flops (limit " + std::to_string(max_flops) + ")";
    return DDM_ENOTIMPL;
  }
  if (!numeric) return DDM_OK;
  if (!va) {
    R.error = "bad arguments";
    return DDM_EINVAL;
  }
  parallel([&](int64_t b) {
    if (general) {
      if (!chol::factorize_lu(PU[(size_t)b], BF[(size_t)b], UX[(size_t)b])) bad[(size_t)b] = 1;
      PU[(size_t)b] = chol::PermutedLowerLU();
    } else {
      if (!chol::factorize(PL[(size_t)b], BF[(size_t)b])) bad[(size_t)b] = 1;
      PL[(size_t)b] = chol::PermutedLower(); // release
    }
  });
  for (int64_t b = 0; b < nblocks; ++b)
    if (bad[(size_t)b]) {
      R.error = "block " + std::to_string(b) + ": " + BF[(size_t)b].error;
      return DDM_ENUMERIC;
    }
  R.rp.assign(1, 0);
  R.rp.reserve((size_t)n + 1);
  R.diag.reserve((size_t)n);
  for (int64_t b = 0; b < nblocks; ++b) {
    if (general) {
      chol::append_rows_lu(BF[(size_t)b], UX[(size_t)b], block_ptr[b], R.rp, R.ci, R.lu, R.diag);
      std::vector<double>().swap(UX[(size_t)b]);
    } else
      chol::append_rows(BF[(size_t)b], block_ptr[b], R.rp, R.ci, R.lu, R.diag);
    BF[(size_t)b] = chol::BlockFactor();
  }
  return DDM_OK;
}

struct ddm_chol_host {
  CholResult R;
};
extern "C" int ddm_chol_host_create(int64_t n, const int64_t *rp, const int32_t *ci, const double *va, int64_t nblocks, const int64_t *block_ptr,
                                    ddm_chol_host **out)
{
  return ddm_direct_host_create(n, rp, ci, va, nblocks, block_ptr, 0, out);
}
extern "C" int ddm_direct_host_create(int64_t n, const int64_t *rp, const int32_t *ci, const double *va, int64_t nblocks, const int64_t *block_ptr,
                                      int general, ddm_chol_host **out)
{
  if (!out) return DDM_EINVAL;
  ddm_chol_host *H = new ddm_chol_host;
  const int rc = chol_build(n, rp, ci, va, nblocks, block_ptr, 0.0, va != nullptr, H->R, general != 0);
  if (rc) {
    delete H;
    return rc;
  }
  *out = H;
  return DDM_OK;
}
extern "C" void ddm_chol_host_destroy(ddm_chol_host *H) { delete H; }
extern "C" int64_t ddm_chol_host_nnz(const ddm_chol_host *H) { return H ? (int64_t)H->R.ci.size() : 0; }
extern "C" int64_t ddm_chol_host_nnz_factor(const ddm_chol_host *H) { return H ? H->R.nnzL : 0; }
extern "C" double ddm_chol_host_flops(const ddm_chol_host *H) { return H ? H->R.flops : 0.0; }
extern "C" int ddm_chol_host_get(const ddm_chol_host *H, int32_t *perm, int64_t *rp, int32_t *ci, double *lu)
{
  if (!H) return DDM_EINVAL;
  if (perm) std::copy(H->R.perm.begin(), H->R.perm.end(), perm);
  if (rp) std::copy(H->R.rp.begin(), H->R.rp.end(), rp);
  if (ci) std::copy(H->R.ci.begin(), H->R.ci.end(), ci);
  if (lu) std::copy(H->R.lu.begin(), H->R.lu.end(), lu);
  return DDM_OK;
}

// Supernodal Cholesky on the device.  Returns DDM_OK / an error code, or 1 when the factorisation is too small to be worth it and
// force == false (the caller then takes the host path).
// multiply-adds of a supernodal factorisation of all blocks, estimated from the first separator of the LARGEST block alone (host only,
// one thread, ~1 s per 10^6 rows); 0 when that block has entries outside its diagonal block
static double sn_probe_largest_block(const int64_t *rp, const int32_t *ci, int64_t nblocks, const int64_t *block_ptr, bool lu)
{
  int64_t bl = 0;
  for (int64_t b = 1; b < nblocks; ++b)
    if (block_ptr[b + 1] - block_ptr[b] > block_ptr[bl + 1] - block_ptr[bl]) bl = b;
  const int64_t r0 = block_ptr[bl], r1 = block_ptr[bl + 1];
  for (int64_t i = r0; i < r1; ++i)
    for (int64_t k = rp[i]; k < rp[i + 1]; ++k)
      if (ci[k] < r0 || ci[k] >= r1) return 0.0;
  return (lu ? 2.0 : 1.0) * sn::estimate_flops(chol::block_graph(rp, ci, r0, r1)) * (double)nblocks;
}
static int ilu0_solve_epilogue(ddm_ctx *ctx, ddm_ilu0 *F, const double *d, double *x, const double *scale, const double *add);
// Fixes the number of iterative-refinement steps of a device factor (ddm_ilu0::refine_steps) from a probe solve with a pseudo-random
// right-hand side: the loop of dune/ddm/eigensolvers/umfpack.hh:42-129 -- backward error omega = ||b - A x|| / (||A||_inf ||x|| + ||b||)
// (here: the larger of that and 1e-2 x the componentwise backward error UMFPACK's own solve refines by), stop below 1e-14, stop when a step does not halve it, at most 3 steps -- run ONCE here instead of in every solve, so that the
// solves stay captured graphs.  DDM_DIRECT_REFINE = off | <steps> overrides.  Returns the last backward error in *omega_out.
static int sn_probe_refinement(ddm_ctx *ctx, ddm_ilu0 *F, const ddm_csr *A, double *omega_out)
{
  const int64_t n = F->n;
  *omega_out = 0.0;
  int forced = -1, max_steps = 3;
  if (const char *e = std::getenv("DDM_DIRECT_REFINE")) {
    if (!std::strcmp(e, "off")) return DDM_OK;
    forced = std::max(0, std::min(4, std::atoi(e)));
  }
  if (n == 0) return DDM_OK;
  const int64_t *rp = A->h_rp.data();
  const int32_t *ci = A->h_ci.data();
  const double *va = A->h_va.data();
  const unsigned nth = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  auto par_rows = [&](const std::function<void(int64_t, int64_t, unsigned)> &f) {
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nth; ++t) th.emplace_back([&, t]() { f(n * t / nth, n * (t + 1) / nth, t); });
    for (auto &t : th) t.join();
  };
  std::vector<double> part(nth, 0.0);
  par_rows([&](int64_t r0, int64_t r1, unsigned t) {
    double m = 0.0;
    for (int64_t i = r0; i < r1; ++i) {
      double a = 0.0;
      for (int64_t k = rp[i]; k < rp[i + 1]; ++k) a += std::fabs(va[k]);
      m = std::max(m, a);
    }
    part[t] = m;
  });
  double anorm = 0.0;
  for (double v : part) anorm = std::max(anorm, v);
  std::vector<double> b((size_t)n), x((size_t)n);
  uint64_t lcg = 0x9E3779B97F4A7C15ull;
  double bn2 = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
    b[(size_t)i] = (double)(int64_t)(lcg >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0;
    bn2 += b[(size_t)i] * b[(size_t)i];
  }
  double *db = nullptr, *dx = nullptr;
  HIPCHECK(ctx, hipMalloc((void **)&db, sizeof(double) * (size_t)n));
  if (hipMalloc((void **)&dx, sizeof(double) * (size_t)n) != hipSuccess) {
    (void)hipFree(db);
    return fail(ctx, DDM_EHIP, "sparse direct solver: allocation failed");
  }
  int rc = ddm_memcpy_h2d(ctx, db, b.data(), sizeof(double) * (size_t)n);
  auto omega_now = [&](double &om) -> int {
    int r = ddm_memcpy_d2h(ctx, x.data(), dx, sizeof(double) * (size_t)n); // (synchronises the stream)
    if (r) return r;
    std::vector<double> pr(nth, 0.0), px(nth, 0.0), pc(nth, 0.0);
    par_rows([&](int64_t r0, int64_t r1, unsigned t) {
      double sr = 0.0, sx = 0.0, wc = 0.0;
      for (int64_t i = r0; i < r1; ++i) {
        double res = b[(size_t)i], den = std::fabs(b[(size_t)i]);
        for (int64_t k = rp[i]; k < rp[i + 1]; ++k) {
          res -= va[k] * x[(size_t)ci[k]];
          den += std::fabs(va[k] * x[(size_t)ci[k]]);
        }
        sr += res * res;
        sx += x[(size_t)i] * x[(size_t)i];
        if (den > 0.0) wc = std::max(wc, std::fabs(res) / den);
      }
      pr[t] = sr;
      px[t] = sx;
      pc[t] = wc;
    });
    double sr = 0.0, sx = 0.0, wc = 0.0;
    for (unsigned t = 0; t < nth; ++t) sr += pr[t], sx += px[t], wc = std::max(wc, pc[t]);
    // normwise backward error of umfpack.hh:66-74, and the componentwise one UMFPACK's own solve refines by (max_i |r_i| / (|A||x| + |b|)_i),
    // weighted so that ONE threshold (1e-14) means: normwise below 1e-14 and componentwise below 1e-12
    om = std::max(std::sqrt(sr) / (anorm * std::sqrt(sx) + std::sqrt(bn2)), 1e-2 * wc);
    return DDM_OK;
  };
  auto solve_with = [&](int steps) -> int {
    if (steps > 0 && !F->ref_rp) { // device copies of the matrix for the residuals
      HIPCHECK(ctx, hipMalloc((void **)&F->ref_rp, sizeof(int64_t) * (size_t)(n + 1)));
      HIPCHECK(ctx, hipMalloc((void **)&F->ref_ci, sizeof(int32_t) * (size_t)std::max<int64_t>(A->nnz, 1)));
      HIPCHECK(ctx, hipMalloc((void **)&F->ref_va, sizeof(double) * (size_t)std::max<int64_t>(A->nnz, 1)));
      HIPCHECK(ctx, hipMalloc((void **)&F->pr, sizeof(double) * (size_t)n));
      F->pr_cols = 1;
      HIPCHECK(ctx, hipMemcpyAsync(F->ref_rp, A->rp, sizeof(int64_t) * (size_t)(n + 1), hipMemcpyDeviceToDevice, ctx->stream));
      HIPCHECK(ctx, hipMemcpyAsync(F->ref_ci, A->ci, sizeof(int32_t) * (size_t)A->nnz, hipMemcpyDeviceToDevice, ctx->stream));
      HIPCHECK(ctx, hipMemcpyAsync(F->ref_va, A->va, sizeof(double) * (size_t)A->nnz, hipMemcpyDeviceToDevice, ctx->stream));
    }
    F->refine_steps = steps;
    if (F->graph) {
      (void)hipGraphExecDestroy(F->graph);
      F->graph = nullptr;
    }
    return ilu0_solve_epilogue(ctx, F, db, dx, nullptr, nullptr);
  };
  double om = 0.0, om_prev = 0.0;
  int steps = 0;
  if (!rc) rc = solve_with(0);
  if (!rc) rc = omega_now(om);
  F->refine_omega[0] = om;
  while (!rc && steps < (forced >= 0 ? forced : max_steps)) {
    if (forced < 0) {
      if (om < 1e-14 || !(om == om)) break;            // converged (or NaN: refinement cannot help)
      if (steps > 0 && om > om_prev / 2.0) break;      // the last step did not halve the backward error
    }
    om_prev = om;
    rc = solve_with(steps + 1);
    if (!rc) rc = omega_now(om);
    ++steps;
    F->refine_omega[std::min(steps, 4)] = om;
  }
  F->refine_steps = steps;
  if (F->graph) { // (bound to the probe vectors)
    (void)hipGraphExecDestroy(F->graph);
    F->graph = nullptr;
  }
  if (steps == 0) {
    (void)hipFree(F->ref_rp);
    (void)hipFree(F->ref_ci);
    (void)hipFree(F->ref_va);
    (void)hipFree(F->pr);
    F->ref_rp = nullptr;
    F->ref_ci = nullptr;
    F->ref_va = F->pr = nullptr;
    F->pr_cols = 0;
  }
  (void)hipFree(db);
  (void)hipFree(dx);
  *omega_out = om;
  return rc;
}
static int sn_direct_create(ddm_ctx *ctx, const ddm_csr *A, int64_t nblocks, const int64_t *block_ptr, double max_flops, bool force, bool lu, bool setup_use, ddm_ilu0 **out)
{
  const int64_t n = A->nrows;
  if (block_ptr[0] != 0 || block_ptr[nblocks] != n) return fail(ctx, DDM_EINVAL, "block_ptr does not cover the matrix");
  const int64_t *rp = A->h_rp.data();
  const int32_t *ci = A->h_ci.data();
  std::vector<sn::BlockSym> BS((size_t)nblocks);
  std::vector<int> bad((size_t)nblocks, 0);
  std::vector<double> quick((size_t)nblocks, 0.0);
  if (max_flops > 0.0 && nblocks > 1) {
    // the largest block first, alone: when its first separator already says "a factor of four beyond the limit" the other blocks are
    // not looked at (the callers run other host work beside this analysis: one busy thread instead of one per block)
    const double q = sn_probe_largest_block(rp, ci, nblocks, block_ptr, lu);
    if (q > 4.0 * max_flops)
      return fail(ctx, DDM_ENOTIMPL, "sparse direct solver: the factorisation needs about %.1g flops (estimate from the first separator of the largest block; limit %.3g)", q,
                  max_flops);
  }
  {
    const unsigned hw = host_threads();
    const int nthreads = (int)std::min<int64_t>(nblocks, hw);
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t)
      th.emplace_back([&, t]() {
        for (int64_t b = t; b < nblocks; b += nthreads) {
          const int64_t r0 = block_ptr[b], r1 = block_ptr[b + 1];
          for (int64_t i = r0; i < r1 && !bad[(size_t)b]; ++i)
            for (int64_t k = rp[i]; k < rp[i + 1]; ++k)
              if (ci[k] < r0 || ci[k] >= r1) bad[(size_t)b] = 1;
          if (bad[(size_t)b]) continue;
          const chol::Graph G = chol::block_graph(rp, ci, r0, r1);
          if (max_flops > 0.0) { // early decline from the first separator alone: a factor of four beyond the limit is not worth the full ordering
            quick[(size_t)b] = (lu ? 2.0 : 1.0) * sn::estimate_flops(G);
            if (quick[(size_t)b] * (double)nblocks > 4.0 * max_flops) continue;
          }
          BS[(size_t)b] = sn::analyse(G);
        }
      });
    for (auto &t : th) t.join();
  }
  for (int64_t b = 0; b < nblocks; ++b)
    if (bad[(size_t)b]) return fail(ctx, DDM_EINVAL, "sparse direct solver: block %lld has entries outside its diagonal block", (long long)b);
  if (max_flops > 0.0) {
    double q = 0.0;
    for (double v : quick) q = std::max(q, v);
    if (q * (double)nblocks > 4.0 * max_flops)
      return fail(ctx, DDM_ENOTIMPL, "sparse direct solver: the factorisation needs about %.1g flops (estimate from the first separator; limit %.3g)", q * (double)nblocks, max_flops);
  }
  double flops = 0.0;
  int64_t entries = 0;
  for (auto &S : BS) {
    flops += (lu ? 2.0 : 1.0) * S.flops;
    entries += (lu ? 2 : 1) * S.entries; // (L U: the U^T blocks; slightly over-counted by the diagonal blocks)
  }
  double min_flops = setup_use ? 1e10 : 2e10; // (see direct_create_impl)
  if (const char *e = std::getenv("DDM_DIRECT_DEVICE_MIN_FLOPS")) min_flops = std::atof(e);
  if (!force && flops < min_flops) return 1;
  if (max_flops > 0.0 && flops > max_flops)
    return fail(ctx, DDM_ENOTIMPL, "sparse direct solver: the factorisation needs %.3g flops (limit %.3g)", flops, max_flops);
  DDMCHECK(csr_wait_upload(ctx, A)); // (matrices the library assembled itself are uploaded by a helper thread: csr_adopt)
  if (A->host_only) return fail(ctx, DDM_EINVAL, "the matrix was created without device arrays (ddm_csr_create_host)");
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && (double)entries * 8.0 > 0.85 * (double)free_b)
    return fail(ctx, DDM_ENOTIMPL, "sparse direct solver: the factor needs %.1f GB, %.1f GB of device memory are free", entries * 8e-9, free_b * 1e-9);
  const auto t0 = std::chrono::steady_clock::now();
  sn::Factor *S = new sn::Factor;
  if (!sn::build(*S, n, nblocks, block_ptr, BS, lu)) {
    delete S;
    return fail(ctx, DDM_EHIP, "sparse direct solver (device): allocation of %.1f GB failed", entries * 8e-9);
  }
  unsigned badsn = 0, perturbed = 0;
  double amax = 0.0;
  if (lu)
    for (double v : A->h_va) amax = std::max(amax, std::fabs(v));
  const hipError_t he = sn::factorize(*S, ctx->stream, A->rp, A->ci, A->va, &badsn, 1.4901161193847656e-08 * amax, &perturbed);
  if (he != hipSuccess) {
    delete S;
    return fail(ctx, DDM_EHIP, "sparse direct solver (device): %s", hipGetErrorString(he));
  }
  if (badsn) {
    delete S;
    if (lu && !force) return 1; // (not forced: the host engine takes the matrix)
    return fail(ctx, DDM_ENUMERIC, lu ? "sparse direct solver: vanishing pivot column inside the diagonal block of supernode %u (matrix singular?)"
                                     : "sparse direct solver: matrix is not positive definite (supernode %u)", badsn - 1);
  }
  if (std::getenv("DDM_PIPE_VERBOSE"))
    std::fprintf(stderr, "[ddm] device supernodal %s: %lld rows, %d supernodes, %d levels, %.2f GB of panels, %.3g flops, numeric factorisation %.3f s (%.2f TFLOP/s)\n",
                 lu ? "L U" : "Cholesky", (long long)n, S->nsn, S->nlev, (S->entries + S->uentries) * 8e-9, (lu ? 2.0 : 1.0) * S->flops, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(),
                 (lu ? 4e-12 : 2e-12) * S->flops / std::max(1e-9, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()));
  ddm_ilu0 *F = new ddm_ilu0;
  F->n = n;
  F->nnz = S->entries + S->uentries;
  F->direct = 1;
  F->direct_flops = (lu ? 2.0 : 1.0) * S->flops;
  F->sn = S;
  F->mode = 0;
  F->L.nlev = F->U.nlev = S->nlev;
  int rc = DDM_OK;
  if (hipHostMalloc((void **)&F->err, 128, hipHostMallocMapped) != hipSuccess) rc = fail(ctx, DDM_EHIP, "sparse direct solver: allocation failed");
  else std::memset(F->err, 0, 128);
  if (!rc && (hipMalloc((void **)&F->pd, sizeof(double) * (size_t)std::max<int64_t>(n, 1)) != hipSuccess ||
              hipMalloc((void **)&F->px, sizeof(double) * (size_t)std::max<int64_t>(n, 1)) != hipSuccess))
    rc = fail(ctx, DDM_EHIP, "sparse direct solver: allocation failed");
  double omega = 0.0;
  if (!rc) rc = sn_probe_refinement(ctx, F, A, &omega);
  if (rc) {
    ddm_ilu0_destroy(F);
    return rc;
  }
  if (std::getenv("DDM_PIPE_VERBOSE"))
    std::fprintf(stderr, "[ddm] device supernodal %s: single-vector solve: levels 0..%d by launches, %d top levels by the persistent kernel (%d forward phases, grid %d, %s)\n",
                 lu ? "L U" : "Cholesky", S->ltop - 1, S->ntop, S->top.nph, S->top_grid, S->top_spread ? "one group over all XCDs" : "block b on XCD b % 8");
  if (std::getenv("DDM_PIPE_VERBOSE") && S->chains_ready)
    std::fprintf(stderr, "[ddm] device supernodal %s: the top levels as %d dense chains on %d chain levels (longest %d links; inverse triangles %.1f MB, external blocks %.1f MB%s), grid %d\n",
                 lu ? "L U" : "Cholesky", S->ch.nchain(), S->ch.nclev, S->ch.max_links, S->ch.wtot * 8e-6, S->ch.etot * 8e-6, lu ? ", twice for L U" : "", S->chain_grid);
  if (std::getenv("DDM_PIPE_VERBOSE"))
    std::fprintf(stderr, "[ddm] device supernodal %s: %d refinement step(s) per solve, backward error of the probe %.2e -> %.2e%s\n", lu ? "L U" : "Cholesky", F->refine_steps,
                 F->refine_omega[0], omega, perturbed ? " (vanishing pivot columns replaced)" : "");
  if (!(omega <= 1e-9)) { // element growth beyond what pivoting inside the supernodes and three refinement steps repair
    ddm_ilu0_destroy(F);
    if (!force) return 1;
    return fail(ctx, DDM_ENUMERIC, "sparse direct solver (device): backward error %.2e after iterative refinement (the matrix needs pivoting across supernodes)", omega);
  }
  *out = F;
  return DDM_OK;
}
// host part of the device engine alone (ordering + supernodal symbolic analysis; no device needed): used by the CPU tests
struct ddm_sn_host {
  std::vector<sn::BlockSym> BS;
  std::vector<int64_t> block_ptr;
};
extern "C" int ddm_sn_host_create(int64_t n, const int64_t *rp, const int32_t *ci, int64_t nblocks, const int64_t *block_ptr, ddm_sn_host **out)
{
  if (!out || !rp || !ci || nblocks < 1 || !block_ptr || block_ptr[0] != 0 || block_ptr[nblocks] != n) return DDM_EINVAL;
  ddm_sn_host *H = new ddm_sn_host;
  H->block_ptr.assign(block_ptr, block_ptr + nblocks + 1);
  for (int64_t b = 0; b < nblocks; ++b) H->BS.push_back(sn::analyse(chol::block_graph(rp, ci, block_ptr[b], block_ptr[b + 1])));
  *out = H;
  return DDM_OK;
}
extern "C" void ddm_sn_host_destroy(ddm_sn_host *H) { delete H; }
// sizes[4] = {supernodes, entries of `rows`, panel entries, levels}; flops = multiply-adds of the factorisation
extern "C" int ddm_sn_host_sizes(const ddm_sn_host *H, int64_t block, int64_t *sizes, double *flops)
{
  if (!H || block < 0 || block >= (int64_t)H->BS.size() || !sizes) return DDM_EINVAL;
  const sn::BlockSym &S = H->BS[(size_t)block];
  int32_t nlev = 0;
  for (int32_t l : S.level) nlev = std::max(nlev, l + 1);
  sizes[0] = (int64_t)S.first.size() - 1;
  sizes[1] = (int64_t)S.rows.size();
  sizes[2] = S.entries;
  sizes[3] = nlev;
  if (flops) *flops = S.flops;
  return DDM_OK;
}
// perm[n_b] (perm[new] = old, block-local), first[nsn + 1], rptr[nsn + 1], rows[...], parent[nsn], level[nsn] of one block
extern "C" int ddm_sn_host_get(const ddm_sn_host *H, int64_t block, int32_t *perm, int32_t *first, int64_t *rptr, int32_t *rows, int32_t *parent, int32_t *level)
{
  if (!H || block < 0 || block >= (int64_t)H->BS.size()) return DDM_EINVAL;
  const sn::BlockSym &S = H->BS[(size_t)block];
  if (perm) std::copy(S.perm.begin(), S.perm.end(), perm);
  if (first) std::copy(S.first.begin(), S.first.end(), first);
  if (rptr) std::copy(S.rptr.begin(), S.rptr.end(), rptr);
  if (rows) std::copy(S.rows.begin(), S.rows.end(), rows);
  if (parent) std::copy(S.parent.begin(), S.parent.end(), parent);
  if (level) std::copy(S.level.begin(), S.level.end(), level);
  return DDM_OK;
}
extern "C" int ddm_chol_create(ddm_ctx *ctx, const ddm_csr *A, int64_t nblocks, const int64_t *block_ptr, double max_flops, ddm_ilu0 **out)
{
  return ddm_direct_create(ctx, A, nblocks, block_ptr, 0, max_flops, out);
}
static int direct_create_impl(ddm_ctx *ctx, const ddm_csr *A, int64_t nblocks, const int64_t *block_ptr, int general, double max_flops, bool setup_use, ddm_ilu0 **out);
extern "C" int ddm_direct_create(ddm_ctx *ctx, const ddm_csr *A, int64_t nblocks, const int64_t *block_ptr, int general, double max_flops, ddm_ilu0 **out)
{
  return direct_create_impl(ctx, A, nblocks, block_ptr, general, max_flops, false, out);
}
// setup_use: the factor serves a handful of block solves during a setup phase (GenEO preconditioner, harmonic extensions) -- the
// device engine pays from ~1e10 multiply-adds.  As the local solver of a Krylov loop the host engine's CSR level solves are the
// faster single-vector solves (measured on configs[4]: 1.31 against 1.75 ms), but its factorisation costs ~1 s per 1e10
// multiply-adds against ~0.05 s on the device: from 2e10 the device engine wins the time to solution of any solve shorter than
// several thousand iterations, so that is the default there (DDM_DIRECT_DEVICE_MIN_FLOPS / DDM_DIRECT_ENGINE override).
static int direct_create_impl(ddm_ctx *ctx, const ddm_csr *A, int64_t nblocks, const int64_t *block_ptr, int general, double max_flops, bool setup_use, ddm_ilu0 **out)
{
  if (!ctx || !A || !out || nblocks < 1 || !block_ptr) return fail(ctx, DDM_EINVAL, "ddm_direct_create: bad arguments");
  if (A->nrows != A->ncols) return fail(ctx, DDM_EINVAL, "the sparse direct solver needs a square matrix");
  // Engine: "device" = supernodal factorisation and solves on the GPU (sn_chol.hpp; symmetric positive definite input), "host" = the
  // up-looking host factorisation with CSR level solves on the device.  Default: the device engine when the matrix is symmetric
  // and the factorisation is worth it (DDM_DIRECT_DEVICE_MIN_FLOPS; defaults in direct_create_impl); DDM_DIRECT_ENGINE overrides.
  {
    const char *eng = std::getenv("DDM_DIRECT_ENGINE");
    if (!eng || std::strcmp(eng, "host") != 0) {
      const int rcs = sn_direct_create(ctx, A, nblocks, block_ptr, max_flops, eng && !std::strcmp(eng, "device"), general != 0, setup_use, out);
      if (rcs != 1) return rcs; // 1 = not taken (too small for the device engine): fall through to the host path
    }
  }
  CholResult R;
  const int rc0 = chol_build(A->nrows, A->h_rp.data(), A->h_ci.data(), A->h_va.data(), nblocks, block_ptr, max_flops, true, R, general != 0);
  if (rc0) return fail(ctx, rc0, "sparse direct solver: %s", R.error.c_str());
  ddm_ilu0 *F = new ddm_ilu0;
  F->n = A->nrows;
  F->nnz = (int64_t)R.ci.size();
  F->direct = 1;
  F->direct_flops = R.flops;
  F->h_lu = std::move(R.lu);
  ddm_csr *P = new ddm_csr; // host-only pattern of the factor (the schedule builders read h_rp / h_ci)
  P->nrows = P->ncols = A->nrows;
  P->nnz = F->nnz;
  P->h_rp = std::move(R.rp);
  P->h_ci = std::move(R.ci);
  F->own_pattern = P;
  // a direct factor has few, wide rows per level and thousands of levels: the level kernels (runs of small levels fused into one
  // workgroup that splits wide rows over lanes) take it; the pipe / xcd2 engines are built for the narrow rows of ILU(0)
  int rc = ilu0_build_engines(ctx, F, P, R.diag, nblocks, block_ptr, /*multi_rhs_only (= level kernels)=*/true);
  if (!rc) rc = upload(ctx, R.perm.data(), A->nrows, &F->perm);
  if (!rc && (hipMalloc((void **)&F->pd, sizeof(double) * (size_t)std::max<int64_t>(F->n, 1)) != hipSuccess ||
              hipMalloc((void **)&F->px, sizeof(double) * (size_t)std::max<int64_t>(F->n + F->nvirt, 1)) != hipSuccess))
    rc = fail(ctx, DDM_EHIP, "ddm_chol_create: allocation failed");
  if (rc) {
    ddm_ilu0_destroy(F);
    return rc;
  }
  *out = F;
  return DDM_OK;
}
extern "C" int ddm_ilu0_is_direct(const ddm_ilu0 *F) { return F ? F->direct : 0; }
extern "C" int ddm_ilu0_refinement(const ddm_ilu0 *F, double *omega)
{
  if (!F) return 0;
  if (omega)
    for (int k = 0; k < 5; ++k) omega[k] = F->refine_omega[k];
  return F->refine_steps;
}
extern "C" int64_t ddm_ilu0_nnz(const ddm_ilu0 *F) { return F ? F->nnz : 0; }
extern "C" void ddm_ilu0_destroy(ddm_ilu0 *F)
{
  if (!F) return;
  if (F->pipe_builder.joinable()) F->pipe_builder.join();
  if (F->graph) (void)hipGraphExecDestroy(F->graph);
  if (F->mgraph) (void)hipGraphExecDestroy(F->mgraph);
  (void)hipFree(F->ref_rp);
  (void)hipFree(F->ref_ci);
  (void)hipFree(F->ref_va);
  (void)hipFree(F->pr);
  (void)hipFree(F->xf);
  if (F->err) (void)hipHostFree(F->err);
  delete F->sn;
  (void)hipFree(F->perm);
  (void)hipFree(F->pd);
  (void)hipFree(F->px);
  (void)hipFree(F->pD);
  (void)hipFree(F->pX);
  delete F->own_pattern; // host-only pattern: no device arrays
  (void)hipFree(F->xg);
  (void)hipFree(F->xdesc);
  (void)hipFree(F->xflag_off);
  (void)hipFree(F->xrows);
  (void)hipFree(F->xcols);
  (void)hipFree(F->xvals);
  (void)hipFree(F->xdinv);
  (void)hipFree(F->xflags);
  (void)hipFree(F->xstate);
  (void)hipFree(F->xdperm);
  (void)hipFree(F->xlpos);
  if (BoxEngine *X = F->box) {
    ddm_ilu0_destroy(X->shell);
    ddm_csr_destroy(X->shell_csr);
    (void)hipFree(X->blocks); (void)hipFree(X->steps); (void)hipFree(X->stream); (void)hipFree(X->einfo); (void)hipFree(X->E);
    (void)hipFree(X->ext_val); (void)hipFree(X->ext_col); (void)hipFree(X->xs); (void)hipFree(X->prog); (void)hipFree(X->queue);
    if (X->dbg) (void)hipHostFree(X->dbg);
    (void)hipFree(X->srp); (void)hipFree(X->sci); (void)hipFree(X->srow); (void)hipFree(X->sva); (void)hipFree(X->ds); (void)hipFree(X->xsol);
    delete X;
  }
  (void)hipFree(F->p_groups);
  (void)hipFree(F->p_tasks);
  (void)hipFree(F->p_stream);
  (void)hipFree(F->p_koff);
  (void)hipFree(F->p_posU);
  (void)hipFree(F->p_rowU);
  (void)hipFree(F->p_ypos);
  (void)hipFree(F->p_xpos);
  (void)hipFree(F->p_progress);
  (void)hipFree(F->p_queue);
  free_schedule(F->L);
  free_schedule(F->U);
  free_csr_schedule(F->Lc);
  free_csr_schedule(F->Uc);
  free_csr_schedule(F->Lb);
  free_csr_schedule(F->Ub);
  delete F;
}
// 0 = ok, 1 = a wave of the persistent kernel gave up waiting (results invalid); synchronous
extern "C" int ddm_ilu0_status(ddm_ctx *ctx, const ddm_ilu0 *F, int *status)
{
  if (!F || !status) return fail(ctx, DDM_EINVAL, "ddm_ilu0_status: bad arguments");
  HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
  *status = (int)*(volatile unsigned *)F->err;
  return DDM_OK;
}
// the same word WITHOUT synchronising: what the solves that have finished so far reported (0 = nothing wrong yet)
static inline unsigned ilu0_peek_status(const ddm_ilu0 *F) { return (F && F->err) ? *(volatile unsigned *)F->err : 0u; }
extern "C" int ddm_ilu0_peek_status(const ddm_ilu0 *F) { return (int)ilu0_peek_status(F); }
extern "C" int64_t ddm_ilu0_num_levels(const ddm_ilu0 *F, int upper) { return upper ? F->U.nlev : F->L.nlev; }
// engine the next ddm_ilu0_solve uses: 8 = pipe, 4 = xcd2 (also when pipe declined the matrix), 0 = one launch per level
extern "C" int ddm_ilu0_engine(const ddm_ilu0 *F)
{
  if (!F) return -1;
  if (F->sn) return 16; // device supernodal factor (sn_chol.hpp)
  if (F->pipe_builder.joinable()) const_cast<ddm_ilu0 *>(F)->pipe_builder.join();   // (the answer depends on what the builder found)
  if (F->box) return 32;
  return (F->mode == 8 && F->pipe_state < 0) ? 4 : F->mode;
}
extern "C" int ddm_ilu0_get_factors_host(ddm_ctx *ctx, const ddm_ilu0 *F, double *lu_host)
{
  if (!F || !lu_host) return fail(ctx, DDM_EINVAL, "bad arguments");
  if (F->sn) return fail(ctx, DDM_ENOTIMPL, "ddm_ilu0_get_factors_host: the device supernodal factor has no CSR form");
  std::memcpy(lu_host, F->h_lu.data(), sizeof(double) * (size_t)F->nnz);
  return DDM_OK;
}

// Per-block level schedules of the XCD-local engine: for every diagonal block its L levels then its U
// levels, rows level-sorted, entries in sliced ELL; everything concatenated into one set of arrays.
static int build_xcd_schedule(ddm_ctx *ctx, ddm_ilu0 *F)
{
  const ddm_csr *A = F->A;
  const int64_t *rp = A->h_rp.data();
  const int32_t *ci = A->h_ci.data();
  const hvec<double> &lu = F->h_lu;
  const std::vector<int64_t> &diag = F->h_diag;
  const int nb = (int)F->h_block_ptr.size() - 1;
  std::vector<GroupDesc> groups(nb);
  std::vector<LevelDesc> desc;
  std::vector<int64_t> flag_off(nb);
  std::vector<int32_t> rows, cols;
  std::vector<double> vals, dinv;
  rows.reserve(2 * (size_t)A->nrows);
  dinv.reserve(2 * (size_t)A->nrows);
  cols.reserve((size_t)A->nnz);
  vals.reserve((size_t)A->nnz);
  std::vector<int32_t> level(A->nrows);
  int64_t nflag = 0;
  for (int b = 0; b < nb; ++b) {
    const int64_t r0 = F->h_block_ptr[b], r1 = F->h_block_ptr[b + 1];
    groups[b].lev_off = (int64_t)desc.size();
    flag_off[b] = nflag;
    for (int pass = 0; pass < 2; ++pass) {
      const bool upper = pass == 1;
      int32_t maxlev = -1;
      if (!upper)
        for (int64_t i = r0; i < r1; ++i) {
          int32_t l = 0;
          for (int64_t k = rp[i]; k < diag[i]; ++k) l = std::max(l, level[ci[k]] + 1);
          level[i] = l;
          maxlev = std::max(maxlev, l);
        }
      else
        for (int64_t i = r1 - 1; i >= r0; --i) {
          int32_t l = 0;
          for (int64_t k = diag[i] + 1; k < rp[i + 1]; ++k) l = std::max(l, level[ci[k]] + 1);
          level[i] = l;
          maxlev = std::max(maxlev, l);
        }
      const int64_t nlev = (int64_t)maxlev + 1;
      (upper ? groups[b].nlevU : groups[b].nlevL) = (int32_t)nlev;
      std::vector<int64_t> lptr(nlev + 1, 0);
      for (int64_t i = r0; i < r1; ++i) lptr[level[i] + 1]++;
      for (int64_t l = 0; l < nlev; ++l) lptr[l + 1] += lptr[l];
      const int64_t base = (int64_t)rows.size();
      rows.resize(base + (r1 - r0));
      dinv.resize(base + (r1 - r0), 0.0);
      {
        std::vector<int64_t> pos(lptr.begin(), lptr.end() - 1);
        for (int64_t i = r0; i < r1; ++i) rows[base + pos[level[i]]++] = (int32_t)i;
      }
      for (int64_t l = 0; l < nlev; ++l) {
        const int64_t m = lptr[l + 1] - lptr[l];
        int w = 0;
        for (int64_t r = 0; r < m; ++r) {
          const int64_t i = rows[base + lptr[l] + r];
          w = std::max(w, upper ? (int)(rp[i + 1] - diag[i] - 1) : (int)(diag[i] - rp[i]));
        }
        const int64_t ent = (int64_t)cols.size();
        desc.push_back(LevelDesc{(int32_t)m, (int32_t)w, base + lptr[l], ent});
        cols.resize(ent + m * (int64_t)w);
        vals.resize(ent + m * (int64_t)w);
        for (int64_t r = 0; r < m; ++r) {
          const int64_t i = rows[base + lptr[l] + r];
          const int64_t k0 = upper ? diag[i] + 1 : rp[i], k1 = upper ? rp[i + 1] : diag[i];
          int k = 0;
          for (int64_t p = k0; p < k1; ++p, ++k) {
            cols[ent + (int64_t)k * m + r] = ci[p];
            vals[ent + (int64_t)k * m + r] = lu[p];
          }
          for (; k < w; ++k) {
            cols[ent + (int64_t)k * m + r] = ci[k0];
            vals[ent + (int64_t)k * m + r] = 0.0;
          }
          if (upper) dinv[base + lptr[l] + r] = lu[diag[i]];
        }
      }
    }
    nflag += (int64_t)(groups[b].nlevL + groups[b].nlevU) * TRSV_X_MAXW;
  }
  F->ngroups = nb;
  DDMCHECK(upload(ctx, groups.data(), (int64_t)groups.size(), &F->xg));
  DDMCHECK(upload(ctx, desc.data(), (int64_t)desc.size(), &F->xdesc));
  DDMCHECK(upload(ctx, flag_off.data(), (int64_t)flag_off.size(), &F->xflag_off));
  DDMCHECK(upload(ctx, rows.data(), (int64_t)rows.size(), &F->xrows));
  DDMCHECK(upload(ctx, cols.data(), (int64_t)cols.size(), &F->xcols));
  DDMCHECK(upload(ctx, vals.data(), (int64_t)vals.size(), &F->xvals));
  DDMCHECK(upload(ctx, dinv.data(), (int64_t)dinv.size(), &F->xdinv));
  HIPCHECK(ctx, hipMalloc((void **)&F->xflags, sizeof(unsigned) * (size_t)std::max<int64_t>(nflag, 1)));
  HIPCHECK(ctx, dev_memset(F->xflags, 0, sizeof(unsigned) * (size_t)std::max<int64_t>(nflag, 1)));
  HIPCHECK(ctx, hipMalloc((void **)&F->xstate, sizeof(XcdState)));
  HIPCHECK(ctx, dev_memset(F->xstate, 0, sizeof(XcdState)));
  F->xnrows = (int64_t)rows.size();
  {
    std::vector<int64_t> lpos;
    lpos.reserve((size_t)A->nrows);
    int64_t base = 0;
    for (int b = 0; b < nb; ++b) {
      const int64_t nbk = F->h_block_ptr[b + 1] - F->h_block_ptr[b];
      for (int64_t p = 0; p < nbk; ++p) lpos.push_back(base + p);
      base += 2 * nbk;
    }
    DDMCHECK(upload(ctx, lpos.data(), (int64_t)lpos.size(), &F->xlpos));
  }
  HIPCHECK(ctx, hipMalloc((void **)&F->xdperm, sizeof(double) * (size_t)std::max<int64_t>(F->xnrows, 1)));
  HIPCHECK(ctx, hipFuncSetAttribute((const void *)k_trsv_xcd2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(TrsvLds)));
  F->xcd_built = true;
  return DDM_OK;
}

// Chain/task schedule of the pipe engine (mode 8); falls back to the loader engine (mode 4) when the builder
// reports that the matrix does not fit the tile format.
static int build_pipe_schedule(ddm_ctx *ctx, ddm_ilu0 *F)
{
  const ddm_csr *A = F->A;
  pipe::Options opt;
  if (const char *e = std::getenv("DDM_PIPE_DELTA")) opt.delta = std::atoi(e);
  if (const char *e = std::getenv("DDM_PIPE_SPAN")) opt.max_span = std::atoi(e);
  if (const char *e = std::getenv("DDM_PIPE_REUSE")) opt.vote = std::atoi(e);
  int spread_env = -1;
  if (const char *e = std::getenv("DDM_PIPE_SPREAD")) spread_env = std::atoi(e);
  pipe::Schedule S;
  const int nb = (int)F->h_block_ptr.size() - 1;
  if (!pipe::build(A->nrows, A->h_rp.data(), A->h_ci.data(), F->h_lu.data(), F->h_diag.data(), nb, F->h_block_ptr.data(), opt, S)) {
    F->pipe_state = -1;
    if (std::getenv("DDM_PIPE_VERBOSE")) std::fprintf(stderr, "[ddm] pipe engine not applicable: %s\n", S.error.c_str());
    return DDM_OK;
  }
  F->ngroups = nb;
  F->p_stats = S.stats;
  // one XCD hosts 64 workgroups (2 per CU): a subdomain whose sweeps are wider than ~48 wavefronts per level is spread over
  // all XCDs (write-through hand-overs); measured at 216^3: 1 subdomain 6.9 vs 9.2 ms, 2 subdomains 7.6 vs 8.3 ms
  F->p_spread = spread_env >= 0 ? spread_env : (nb < 8 && S.stats.max_rows_per_level > 48.0 * 64.0 ? 1 : 0);
  F->p_stream_bytes = (int64_t)S.stream.size();
  F->p_nposL = S.nposL;
  F->p_nposU = S.nposU;
  DDMCHECK(upload(ctx, S.groups.data(), (int64_t)S.groups.size(), &F->p_groups));
  DDMCHECK(upload(ctx, S.tasks.data(), (int64_t)S.tasks.size(), &F->p_tasks));
  DDMCHECK(upload(ctx, S.stream.data(), (int64_t)S.stream.size(), &F->p_stream));
  DDMCHECK(upload(ctx, S.koff.data(), (int64_t)S.koff.size(), &F->p_koff));
  DDMCHECK(upload(ctx, S.posU.data(), (int64_t)S.posU.size(), &F->p_posU));
  {
    std::vector<int32_t> rowU((size_t)std::max<int64_t>(S.nposU, 1), -1);
    for (size_t i = 0; i < S.posU.size(); ++i) rowU[(size_t)S.posU[i]] = (int32_t)i;
    DDMCHECK(upload(ctx, rowU.data(), (int64_t)rowU.size(), &F->p_rowU));
  }
  HIPCHECK(ctx, hipMalloc((void **)&F->p_ypos, sizeof(double) * (size_t)std::max<int64_t>(S.nposL, 1)));
  HIPCHECK(ctx, hipMalloc((void **)&F->p_xpos, sizeof(double) * (size_t)std::max<int64_t>(S.nposU, 1)));
  HIPCHECK(ctx, dev_memset(F->p_ypos, 0, sizeof(double) * (size_t)std::max<int64_t>(S.nposL, 1)));
  HIPCHECK(ctx, dev_memset(F->p_xpos, 0, sizeof(double) * (size_t)std::max<int64_t>(S.nposU, 1)));
  const size_t pbytes = sizeof(unsigned long long) * 16 * std::max<size_t>(S.tasks.size(), 1);
  HIPCHECK(ctx, hipMalloc((void **)&F->p_progress, pbytes));
  HIPCHECK(ctx, dev_memset(F->p_progress, 0, pbytes));
  HIPCHECK(ctx, hipMalloc((void **)&F->p_queue, sizeof(unsigned) * 32 * 4 * (size_t)nb));
  HIPCHECK(ctx, dev_memset(F->p_queue, 0, sizeof(unsigned) * 32 * 4 * (size_t)nb));
  if (!F->xstate) {
    HIPCHECK(ctx, hipMalloc((void **)&F->xstate, sizeof(XcdState)));
    HIPCHECK(ctx, dev_memset(F->xstate, 0, sizeof(XcdState)));
  }
  HIPCHECK(ctx, hipFuncSetAttribute((const void *)k_trsv_pipe<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PIPE_LDS_BYTES));
  HIPCHECK(ctx, hipFuncSetAttribute((const void *)k_trsv_pipe<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PIPE_LDS_BYTES));
  int per_cu = 0;
  HIPCHECK(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_trsv_pipe<false>, 64 * (PIPE_NC + PIPE_NL), PIPE_LDS_BYTES));
  per_cu = std::max(1, std::min(per_cu, 2));
  if (const char *e = std::getenv("DDM_PIPE_WG_PER_CU")) per_cu = std::max(1, std::min(per_cu, std::atoi(e)));
  F->p_grid = per_cu * (ctx->num_cu / 8 * 8);
  if (std::getenv("DDM_PIPE_VERBOSE")) {
    const pipe::Stats &st = S.stats;
    std::fprintf(stderr,
                 "[ddm] pipe schedule: %lld rows, tasks %lld+%lld, steps %lld+%lld (lane occupancy %.3f / %.3f), entries %lld: local %.3f self-global %.3f remote %.3f, "
                 "stream %.1f MB (%.2fx of 12 B/entry), max producers %lld, max steps %lld, regrouped %lld, levels <= %lld, rows/level <= %.0f, spread %d, grid %d\n",
                 (long long)st.rows, (long long)st.ntasks[0], (long long)st.ntasks[1], (long long)st.nsteps[0], (long long)st.nsteps[1],
                 (double)st.rows / (64.0 * std::max<int64_t>(st.nsteps[0], 1)), (double)st.rows / (64.0 * std::max<int64_t>(st.nsteps[1], 1)), (long long)st.entries,
                 (double)st.entries_local / std::max<int64_t>(st.entries, 1), (double)st.entries_self_global / std::max<int64_t>(st.entries, 1),
                 (double)st.entries_remote / std::max<int64_t>(st.entries, 1), S.stream.size() / 1e6, S.stream.size() / (12.0 * std::max<int64_t>(st.entries, 1)),
                 (long long)st.max_prod, (long long)st.max_steps, (long long)st.regrouped, (long long)st.max_levels, st.max_rows_per_level, F->p_spread, F->p_grid);
  }
  F->pipe_state = 1;
  return DDM_OK;
}

static unsigned perm_grid(ddm_ctx *ctx, int64_t npos) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>((npos + PERM_TILE - 1) / PERM_TILE, (int64_t)ctx->num_cu * 16)); }
static void enqueue_pipe(ddm_ctx *ctx, ddm_ilu0 *F, const double *d, double *x, unsigned long long *stamps, const double *scale = nullptr,
                         const double *add = nullptr)
{
  PipeParams P;
  P.ngroups = F->ngroups;
  P.groups = F->p_groups;
  P.tasks = F->p_tasks;
  P.stream = F->p_stream;
  P.koff = F->p_koff;
  P.d = d;
  P.ypos = F->p_ypos;
  P.xpos = F->p_xpos;
  P.progress = F->p_progress;
  P.queue = F->p_queue;
  P.st = F->xstate;
  P.err = F->err;
  P.stamps = stamps;
  P.spread = F->p_spread;
  hipLaunchKernelGGL(k_pipe_prologue, dim3(1), dim3(64), 0, ctx->stream, F->xstate, F->p_queue, F->ngroups * 4);
  if (stamps) hipLaunchKernelGGL((k_trsv_pipe<true>), dim3(F->p_grid), dim3(64 * (PIPE_NC + PIPE_NL)), PIPE_LDS_BYTES, ctx->stream, P);
  else hipLaunchKernelGGL((k_trsv_pipe<false>), dim3(F->p_grid), dim3(64 * (PIPE_NC + PIPE_NL)), PIPE_LDS_BYTES, ctx->stream, P);
  hipLaunchKernelGGL(k_pipe_permute_out, dim3(perm_grid(ctx, F->p_nposU)), dim3(PERM_WG), 0, ctx->stream, F->p_nposU, F->p_rowU, (const double *)F->p_xpos, x, scale, add);
}

// ---- box engine (trsv_box_host.hpp / trsv_box.hpp) ----
static int ilu0_build_engines(ddm_ctx *ctx, ddm_ilu0 *F, const ddm_csr *A, const std::vector<int64_t> &diag, int64_t nblocks, const int64_t *block_ptr, bool multi_rhs_only);
static int build_box_engine(ddm_ctx *ctx, ddm_ilu0 *F)
{
  const ddm_csr *A = F->A;
  const int nb = (int)F->h_block_ptr.size() - 1;
  const auto t0 = std::chrono::steady_clock::now();
  box::Schedule S;
  if (!box::build(A->nrows, A->h_rp.data(), A->h_ci.data(), F->h_lu.data(), F->h_diag.data(), nb, F->h_block_ptr.data(), S)) {
    if (std::getenv("DDM_PIPE_VERBOSE")) std::fprintf(stderr, "[ddm] box engine not applicable: %s\n", S.error.c_str());
    return DDM_OK;
  }
  BoxEngine *X = new BoxEngine;
  X->nblocks = nb;
  X->nshell = (int64_t)S.srow.size();
  X->nprod = (int64_t)S.ext_val.size();
  X->stats = S.stats;
  auto bail = [&](int rc) {
    F->box = X;           // (ddm_ilu0_destroy frees what was allocated)
    return rc;
  };
  int rc = upload(ctx, S.blocks.data(), (int64_t)S.blocks.size(), &X->blocks);
  if (!rc) rc = upload(ctx, S.steps.data(), (int64_t)S.steps.size(), &X->steps);
  if (!rc) rc = upload(ctx, S.stream.data(), (int64_t)S.stream.size(), &X->stream);
  if (!rc) rc = upload(ctx, (const unsigned long long *)S.einfo.data(), (int64_t)S.einfo.size(), &X->einfo);
  if (!rc) rc = upload(ctx, S.ext_val.data(), X->nprod, &X->ext_val);
  if (!rc) rc = upload(ctx, S.ext_col.data(), X->nprod, &X->ext_col);
  if (!rc) rc = upload(ctx, S.srp.data(), (int64_t)S.srp.size(), &X->srp);
  if (!rc) rc = upload(ctx, S.sci.data(), (int64_t)S.sci.size(), &X->sci);
  if (!rc) rc = upload(ctx, S.sva.data(), (int64_t)S.sva.size(), &X->sva);
  if (!rc) rc = upload(ctx, S.srow.data(), X->nshell, &X->srow);
  if (rc) return bail(rc);
  auto zalloc = [&](void **p, size_t bytes) {
    bytes = std::max<size_t>(bytes, 8);
    if (hipMalloc(p, bytes) != hipSuccess) return fail(ctx, DDM_EHIP, "box engine: allocation failed");
    if (dev_memset(*p, 0, bytes) != hipSuccess) return fail(ctx, DDM_EHIP, "box engine: memset failed");
    return DDM_OK;
  };
  rc = zalloc((void **)&X->E, sizeof(double) * (size_t)X->nprod);
  if (!rc) rc = zalloc((void **)&X->xs, sizeof(double) * (size_t)S.xs_len);
  if (!rc) rc = zalloc((void **)&X->prog, sizeof(unsigned long long) * (size_t)S.prog_len);
  if (!rc) rc = zalloc((void **)&X->queue, sizeof(unsigned) * 32 * 2 * (size_t)nb);
  if (!rc) rc = zalloc((void **)&X->ds, sizeof(double) * (size_t)X->nshell);
  if (!rc) rc = zalloc((void **)&X->xsol, sizeof(double) * (size_t)X->nshell);
  if (!rc && !F->xstate) rc = zalloc((void **)&F->xstate, sizeof(XcdState));
  if (rc) return bail(rc);
  if (X->nshell > 0) { // the rows behind the boxes: a factor object of their own with the general engines
    rc = csr_create_impl(ctx, X->nshell, X->nshell, S.frp.data(), S.fci.data(), S.fva.data(), /*host_only=*/true, &X->shell_csr);
    if (rc) return bail(rc);
    ddm_ilu0 *G = new ddm_ilu0;
    X->shell = G;
    G->n = X->nshell;
    G->nnz = (int64_t)S.fci.size();
    G->allow_box = false;
    hvec_copy(G->h_lu, S.fva.data(), S.fva.size());
    rc = ilu0_build_engines(ctx, G, X->shell_csr, S.fdiag, nb, S.fblock_ptr.data(), /*level kernels only=*/std::getenv("DDM_BOX_SHELL_LEVELS") != nullptr);
    if (rc) return bail(rc);
  }
  X->n = A->nrows;
  X->stream_len = (int64_t)S.stream.size();
  X->xs_len = S.xs_len;
  X->prog_len = S.prog_len;
  X->einfo_len = (int64_t)S.einfo.size();
  if (std::getenv("DDM_BOX_CHECK")) {
    if (hipHostMalloc((void **)&X->dbg, 8192, hipHostMallocMapped) != hipSuccess) return bail(fail(ctx, DDM_EHIP, "box engine: allocation failed"));
    std::memset(X->dbg, 0, 8192);
  }
  X->grid = 2 * (ctx->num_cu / 8 * 8);
  if (const char *e = std::getenv("DDM_BOX_GRID")) X->grid = std::max(8, std::atoi(e) / 8 * 8);
  F->box = X;
  if (std::getenv("DDM_PIPE_VERBOSE")) {
    const box::Block &B0 = S.blocks[0];
    std::fprintf(stderr, "[ddm] box engine: %d blocks, box rows %lld (block 0: %d x %d x %d, %d steps per plane), rows behind the boxes %lld (nested factor: %lld entries), "
                 "streams %.1f MB (%.2f B per factor entry of the boxes), shell products %lld, grid %d, built in %.2f s\n",
                 nb, (long long)S.stats.box_rows, B0.nx, B0.ny, B0.nz, B0.nsteps, (long long)X->nshell, (long long)S.fci.size(), S.stats.stream_bytes / 1e6,
                 (double)S.stats.stream_bytes / (27.0 * std::max<int64_t>(S.stats.box_rows, 1)), (long long)X->nprod, X->grid,
                 std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  }
  return DDM_OK;
}
static int enqueue_tri(ddm_ctx *ctx, const TriSchedule &S, bool upper, const double *d, double *x);
static int build_xcd_schedule(ddm_ctx *ctx, ddm_ilu0 *F);
// joins the background builders and settles which engine a factor uses (the single-vector solve and the box engine's nested factor)
static int ilu0_prepare_engine(ddm_ctx *ctx, ddm_ilu0 *F)
{
  DDMCHECK(ilu0_join(ctx, F));
  if (F->box) {
    if (F->box->shell) DDMCHECK(ilu0_prepare_engine(ctx, F->box->shell));
    return DDM_OK;
  }
  if (F->mode == 8 && F->pipe_state == 0) DDMCHECK(build_pipe_schedule(ctx, F));
  if (F->mode == 8 && F->pipe_state < 0) F->mode = 4; // not applicable: the loader engine takes any matrix
  if (F->mode == 4 && !F->xcd_built) DDMCHECK(build_xcd_schedule(ctx, F));
  return DDM_OK;
}
static int enqueue_box(ddm_ctx *ctx, ddm_ilu0 *F, const double *d, double *x, const double *scale, const double *add)
{
  BoxEngine *X = F->box;
  BoxParams P;
  P.nblocks = X->nblocks;
  P.blocks = X->blocks;
  P.steps = X->steps;
  P.stream = X->stream;
  P.einfo = X->einfo;
  P.E = X->E;
  P.xs = X->xs;
  P.prog = X->prog;
  P.queue = X->queue;
  P.st = F->xstate;
  P.err = F->err;
  P.spread = 0;
  if (const char *e = std::getenv("DDM_BOX_SPREAD")) P.spread = std::atoi(e);
  P.dbg = X->dbg;
  P.n = X->n;
  P.stream_len = X->stream_len;
  P.xs_len = X->xs_len;
  P.prog_len = X->prog_len;
  P.einfo_len = X->einfo_len;
  P.e_len = X->nprod;
  // forward sweep of the boxes: y into x
  P.rhs = d;
  P.out = x;
  P.scale = P.add = nullptr;
  int dbg = 0;   // diagnostic: DDM_BOX_DEBUG bit mask switches phases off (1 forward boxes, 2 nested solve, 4 products, 8 backward boxes, 16 shell rhs / out)
  if (const char *e = std::getenv("DDM_BOX_DEBUG")) dbg = std::atoi(e);
  hipLaunchKernelGGL(k_pipe_prologue, dim3(1), dim3(64), 0, ctx->stream, F->xstate, X->queue, X->nblocks * 2);
  hipLaunchKernelGGL(k_box_fill, dim3(grid_for(X->xs_len, WG, 4096)), dim3(WG), 0, ctx->stream, X->xs_len, (unsigned long long *)X->xs);   // "not written yet"
  if (!(dbg & 1)) hipLaunchKernelGGL((k_box_sweep<false>), dim3(X->grid), dim3(BOX_WG), 0, ctx->stream, P);
  if (X->nshell > 0 && !(dbg & 2)) {
    ddm_ilu0 *G = X->shell;
    if (!(dbg & 16))
      hipLaunchKernelGGL(k_box_shell_rhs, dim3(grid_for(X->nshell)), dim3(WG), 0, ctx->stream, X->nshell, (const int64_t *)X->srp, (const int32_t *)X->sci, (const double *)X->sva,
                         (const int32_t *)X->srow, d, (const double *)x, X->ds);
    if (G->mode == 8) enqueue_pipe(ctx, G, X->ds, X->xsol, nullptr);
    else if (G->mode == 4) {
      hipLaunchKernelGGL(k_trsv_xcd_prologue, dim3(1), dim3(64), 0, ctx->stream, G->xstate);
      hipLaunchKernelGGL(k_w_permute_in, dim3(grid_for(G->n)), dim3(WG), 0, ctx->stream, G->n, G->xlpos, G->xrows, (const double *)X->ds, G->xdperm);
      hipLaunchKernelGGL(k_trsv_xcd2, dim3(persistent_grid(ctx)), dim3(64 * (1 + TRSV_L_LOADERS)), sizeof(TrsvLds), ctx->stream, G->ngroups, G->xg, G->xdesc, G->xflag_off,
                         G->xrows, G->xcols, G->xvals, G->xdinv, G->xdperm, X->xsol, G->xflags, G->xstate, F->err, (unsigned long long *)nullptr);
    } else {
      DDMCHECK(enqueue_tri(ctx, G->L, false, X->ds, X->xsol));
      DDMCHECK(enqueue_tri(ctx, G->U, true, X->ds, X->xsol));
    }
  }
  // products of the box rows' shell entries, then the backward sweep of the boxes (with the level's tail) and the shell rows of x
  if (!(dbg & 4))
    hipLaunchKernelGGL(k_box_products, dim3(grid_for(X->nprod)), dim3(WG), 0, ctx->stream, X->nprod, (const double *)X->ext_val, (const int32_t *)X->ext_col, (const double *)X->xsol, X->E);
  P.rhs = x;
  P.scale = scale;
  P.add = add;
  hipLaunchKernelGGL(k_pipe_prologue, dim3(1), dim3(64), 0, ctx->stream, F->xstate, X->queue, X->nblocks * 2);
  hipLaunchKernelGGL(k_box_fill, dim3(grid_for(X->xs_len, WG, 4096)), dim3(WG), 0, ctx->stream, X->xs_len, (unsigned long long *)X->xs);
  if (!(dbg & 8)) hipLaunchKernelGGL((k_box_sweep<true>), dim3(X->grid), dim3(BOX_WG), 0, ctx->stream, P);
  if (X->nshell > 0 && !(dbg & 16))
    hipLaunchKernelGGL(k_box_shell_out, dim3(grid_for(X->nshell)), dim3(WG), 0, ctx->stream, X->nshell, (const int32_t *)X->srow, (const double *)X->xsol, x, scale, add);
  HIPCHECK(ctx, hipGetLastError());
  return DDM_OK;
}

static int enqueue_tri(ddm_ctx *ctx, const TriSchedule &S, bool upper, const double *d, double *x)
{
  for (const auto &p : S.plan) {
    if (p.small) {
      if (upper)
        hipLaunchKernelGGL(k_trsv_small_levels<true>, dim3(1), dim3(TRSV_SMALL_WG), 0, ctx->stream, p.count, S.d_desc + p.first,
                           S.rows, S.cols, S.vals, S.dinv, d, x);
      else
        hipLaunchKernelGGL(k_trsv_small_levels<false>, dim3(1), dim3(TRSV_SMALL_WG), 0, ctx->stream, p.count, S.d_desc + p.first,
                           S.rows, S.cols, S.vals, S.dinv, d, x);
    } else {
      const LevelDesc &D = S.desc[p.first];
      const int grid = (D.m + WG - 1) / WG;
      if (upper)
        hipLaunchKernelGGL(k_trsv_upper_level, dim3(grid), dim3(WG), 0, ctx->stream, D.m, D.w, S.rows + D.row_off, S.cols + D.ent_off,
                           S.vals + D.ent_off, S.dinv + D.row_off, x);
      else
        hipLaunchKernelGGL(k_trsv_lower_level, dim3(grid), dim3(WG), 0, ctx->stream, D.m, D.w, S.rows + D.row_off, S.cols + D.ent_off,
                           S.vals + D.ent_off, d, x);
    }
  }
  HIPCHECK(ctx, hipGetLastError());
  return DDM_OK;
}

// Diagnostic (not part of the product path): one solve with the loader engine and in-kernel cycle stamps of one
// compute wave.  out[0..5] = cycles waiting for the LDS tile, for the level flags, for the x gathers, for the
// store drain + flag; work items; total cycles (s_memtime ticks, 100 MHz constant clock on gfx9).
extern "C" int ddm_ilu0_debug_stamps(ddm_ctx *ctx, ddm_ilu0 *F, const double *d, double *x, unsigned long long *out_host)
{
  unsigned long long *st = nullptr;
  HIPCHECK(ctx, hipMalloc((void **)&st, 64));
  HIPCHECK(ctx, hipMemset(st, 0, 64));
  DDMCHECK(ilu0_join(ctx, F));
  if (!F->xcd_built) DDMCHECK(build_xcd_schedule(ctx, F));
  hipLaunchKernelGGL(k_trsv_xcd_prologue, dim3(1), dim3(64), 0, ctx->stream, F->xstate);
  hipLaunchKernelGGL(k_w_permute_in, dim3(grid_for(F->n)), dim3(WG), 0, ctx->stream, F->n, F->xlpos, F->xrows, d, F->xdperm);
  hipLaunchKernelGGL(k_trsv_xcd2, dim3(persistent_grid(ctx)), dim3(64 * (1 + TRSV_L_LOADERS)), sizeof(TrsvLds), ctx->stream, F->ngroups, F->xg, F->xdesc, F->xflag_off, F->xrows,
                     F->xcols, F->xvals, F->xdinv, F->xdperm, x, F->xflags, F->xstate, F->err, st);
  int rc = ddm_memcpy_d2h(ctx, out_host, st, 48);
  (void)hipFree(st);
  return rc;
}

// Diagnostic (not part of the product path): one solve with the stamped build of the pipe kernel.  Per task 8 words
// (layout: trsv_pipe.hpp, STAMP) followed by nothing; returns the number of tasks in *ntasks.  out_host may be null
// to query the size.  Also reports group / sweep of every task in meta_host[2 * ntasks] when given.
extern "C" int ddm_ilu0_pipe_trace(ddm_ctx *ctx, ddm_ilu0 *F, const double *d, double *x, unsigned long long *out_host, int32_t *meta_host,
                                   int64_t capacity_tasks, int64_t *ntasks)
{
  if (!F || !ntasks) return fail(ctx, DDM_EINVAL, "ddm_ilu0_pipe_trace: bad arguments");
  DDMCHECK(ilu0_join(ctx, F));
  if (F->pipe_state == 0) DDMCHECK(build_pipe_schedule(ctx, F));
  if (F->pipe_state < 0) return fail(ctx, DDM_EINVAL, "pipe engine not applicable to this matrix");
  const int64_t nt = F->p_stats.ntasks[0] + F->p_stats.ntasks[1];
  *ntasks = nt;
  if (!out_host) return DDM_OK;
  if (capacity_tasks < nt || !d || !x || d == x) return fail(ctx, DDM_EINVAL, "ddm_ilu0_pipe_trace: bad arguments");
  unsigned long long *st = nullptr;
  HIPCHECK(ctx, hipMalloc((void **)&st, sizeof(unsigned long long) * PIPE_STAMP_WORDS * (size_t)(nt + 1)));
  HIPCHECK(ctx, hipMemsetAsync(st, 0, sizeof(unsigned long long) * PIPE_STAMP_WORDS * (size_t)(nt + 1), ctx->stream));
  enqueue_pipe(ctx, F, d, x, st);
  int rc = ddm_memcpy_d2h(ctx, out_host, st, (int64_t)sizeof(unsigned long long) * PIPE_STAMP_WORDS * nt);
  if (!rc && meta_host) {
    std::vector<pipe::Task> tasks((size_t)nt);
    rc = ddm_memcpy_d2h(ctx, tasks.data(), F->p_tasks, (int64_t)sizeof(pipe::Task) * nt);
    for (int64_t t = 0; t < nt && !rc; ++t) {
      meta_host[2 * t] = tasks[(size_t)t].group;
      meta_host[2 * t + 1] = tasks[(size_t)t].sweep;
    }
  }
  (void)hipFree(st);
  return rc;
}

// x = (LU)^-1 d, then optionally x *= scale and x += add (the tail of the Schwarz level: partition of unity of the restricted
// variant and the coarse correction); the pipe engine folds both into its output permutation, the others append the two kernels.
static int ilu0_solve_epilogue(ddm_ctx *ctx, ddm_ilu0 *F, const double *d, double *x, const double *scale, const double *add)
{
  if (F && F->n == 0) return DDM_OK;
  if (!F || !d || !x || d == x) return fail(ctx, DDM_EINVAL, "ddm_ilu0_solve: bad arguments (d and x must not alias)");
  if (F->graph && F->g_d == d && F->g_x == x && F->g_scale == scale && F->g_add == add) {
    HIPCHECK(ctx, hipGraphLaunch(F->graph, ctx->stream));
    return DDM_OK;
  }
  // (re)capture the ~2*nlev launches into a graph bound to this (d, x) pair
  if (F->graph) {
    (void)hipGraphExecDestroy(F->graph);
    F->graph = nullptr;
  }
  if (F->sn && !sn::reserve(*F->sn, 1)) return fail(ctx, DDM_EHIP, "sparse direct solver: allocation failed");
  DDMCHECK(ilu0_prepare_engine(ctx, F));
  hipGraph_t g = nullptr;
  HIPCHECK(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
  int rc = DDM_OK;
  const double *d_user = d;
  double *x_user = x;
  bool epilogue_done = false;
  if (F->sn) { // supernodal device factor: gather into the permuted work vector, solve in place on the panels, scatter
    hipLaunchKernelGGL(k_perm_gather, dim3(grid_for(F->n)), dim3(WG), 0, ctx->stream, F->n, 1, F->sn->d_perm, d_user, (int64_t)1, F->pd);
    sn::solve(*F->sn, ctx->stream, 1, F->pd, 1, F->px, F->err); // (a time-out of the persistent top kernel lands in the factor's status word)
    hipLaunchKernelGGL(k_perm_scatter, dim3(grid_for(F->n)), dim3(WG), 0, ctx->stream, F->n, 1, F->sn->d_perm, (const double *)F->pd, x_user, (int64_t)1);
    for (int it = 0; it < F->refine_steps; ++it) { // x += A^-1 (d - A x)
      hipLaunchKernelGGL(k_residual_rowmajor, dim3((unsigned)((F->n + WG - 1) / WG)), dim3(WG), 0, ctx->stream, F->n, 1, (const int64_t *)F->ref_rp, (const int32_t *)F->ref_ci, (const double *)F->ref_va,
                         (const double *)x_user, (int64_t)1, d_user, (int64_t)1, F->pr, (int64_t)1);
      hipLaunchKernelGGL(k_perm_gather, dim3(grid_for(F->n)), dim3(WG), 0, ctx->stream, F->n, 1, F->sn->d_perm, (const double *)F->pr, (int64_t)1, F->pd);
      sn::solve(*F->sn, ctx->stream, 1, F->pd, 1, F->px, F->err);
      hipLaunchKernelGGL(k_perm_scatter_add, dim3(grid_for(F->n)), dim3(WG), 0, ctx->stream, F->n, 1, F->sn->d_perm, (const double *)F->pd, x_user, (int64_t)1);
    }
  } else {
  if (F->perm) { // sparse direct factor: solve in the fill-reducing order
    hipLaunchKernelGGL(k_perm_gather, dim3(grid_for(F->n)), dim3(WG), 0, ctx->stream, F->n, 1, F->perm, d_user, (int64_t)1, F->pd);
    d = F->pd;
    x = F->px;
  }
  if (F->box) {
    epilogue_done = !F->perm;
    rc = enqueue_box(ctx, F, d, x, epilogue_done ? scale : nullptr, epilogue_done ? add : nullptr);
  } else if (F->mode == 8) {
    epilogue_done = !F->perm;
    enqueue_pipe(ctx, F, d, x, nullptr, epilogue_done ? scale : nullptr, epilogue_done ? add : nullptr);
  } else if (F->mode == 4) {
    hipLaunchKernelGGL(k_trsv_xcd_prologue, dim3(1), dim3(64), 0, ctx->stream, F->xstate);
    hipLaunchKernelGGL(k_w_permute_in, dim3(grid_for(F->n)), dim3(WG), 0, ctx->stream, F->n, F->xlpos, F->xrows, d, F->xdperm);
    hipLaunchKernelGGL(k_trsv_xcd2, dim3(persistent_grid(ctx)), dim3(64 * (1 + TRSV_L_LOADERS)), sizeof(TrsvLds), ctx->stream, F->ngroups, F->xg, F->xdesc, F->xflag_off,
                       F->xrows, F->xcols, F->xvals, F->xdinv, F->xdperm, x, F->xflags, F->xstate, F->err, (unsigned long long *)nullptr);
  } else if (F->direct) {
    rc = enqueue_tri_csr(ctx, F->Lb.nblocks ? F->Lb : F->Lc, false, d, x);
    if (!rc) rc = enqueue_tri_csr(ctx, F->Ub.nblocks ? F->Ub : F->Uc, true, d, x);
  } else {
    rc = enqueue_tri(ctx, F->L, false, d, x);
    if (!rc) rc = enqueue_tri(ctx, F->U, true, d, x);
  }
  if (F->perm) hipLaunchKernelGGL(k_perm_scatter, dim3(grid_for(F->n)), dim3(WG), 0, ctx->stream, F->n, 1, F->perm, (const double *)F->px, x_user, (int64_t)1);
  }
  d = d_user;
  x = x_user;
  if (!epilogue_done) {
    if (scale) hipLaunchKernelGGL(k_scale, dim3(grid_for(F->n)), dim3(WG), 0, ctx->stream, F->n, scale, x);
    if (add) hipLaunchKernelGGL(k_axpy, dim3(grid_for(F->n)), dim3(WG), 0, ctx->stream, F->n, 1.0, add, x);
  }
  hipError_t e = hipStreamEndCapture(ctx->stream, &g);
  if (rc) return rc;
  if (e != hipSuccess) return fail(ctx, DDM_EHIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
  e = hipGraphInstantiate(&F->graph, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) {
    F->graph = nullptr;
    return fail(ctx, DDM_EHIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
  }
  F->g_d = d;
  F->g_x = x;
  F->g_scale = scale;
  F->g_add = add;
  HIPCHECK(ctx, hipGraphLaunch(F->graph, ctx->stream));
  return DDM_OK;
}

extern "C" int ddm_ilu0_solve(ddm_ctx *ctx, ddm_ilu0 *F, const double *d, double *x) { return ilu0_solve_epilogue(ctx, F, d, x, nullptr, nullptr); }

// Multi-RHS solve X = (LU)^-1 D for row-major n x nrhs block vectors with leading dimensions ldd / ldx (GenEO setup path).
// One launch per level (wide levels of direct factors: one workgroup per row); the launches of one (D, X, nrhs) combination are
// captured into a HIP graph on first use and replayed afterwards (the block eigensolver calls with the same buffers every iteration).
static void enqueue_multi_levels(ddm_ctx *ctx, ddm_ilu0 *F, int nrhs, const double *D, int64_t ldd, double *X, int64_t ldx)
{
  for (int pass = 0; pass < 2; ++pass) {
    const TriSchedule &S = pass ? F->U : F->L;
    for (int64_t l = 0; l < S.nlev; ++l) {
      const LevelDesc &L = S.desc[l];
      if (L.m == 0) continue;
      const bool wide = L.w >= 96 && nrhs <= WG;
      const bool quad = !wide && nrhs % 4 == 0 && ldd % 4 == 0 && ldx % 4 == 0 && ((uintptr_t)D & 31) == 0 && ((uintptr_t)X & 31) == 0;
      const int64_t threads = (int64_t)L.m * (quad ? nrhs / 4 : nrhs);
      const unsigned grid = wide ? (unsigned)L.m : (unsigned)((threads + WG - 1) / WG);
      if (quad) {
        if (pass)
          hipLaunchKernelGGL(k_trsv_level_multi4<true>, dim3(grid), dim3(WG), 0, ctx->stream, L.m, L.w, nrhs / 4, S.rows + L.row_off, S.cols + L.ent_off, S.vals + L.ent_off,
                             S.dinv + L.row_off, D, ldd, X, ldx);
        else
          hipLaunchKernelGGL(k_trsv_level_multi4<false>, dim3(grid), dim3(WG), 0, ctx->stream, L.m, L.w, nrhs / 4, S.rows + L.row_off, S.cols + L.ent_off, S.vals + L.ent_off,
                             (const double *)nullptr, D, ldd, X, ldx);
      } else if (pass) {
        if (wide)
          hipLaunchKernelGGL(k_trsv_level_multi_wide<true>, dim3(grid), dim3(WG), 0, ctx->stream, L.m, L.w, nrhs, S.rows + L.row_off, S.cols + L.ent_off,
                             S.vals + L.ent_off, S.dinv + L.row_off, D, ldd, X, ldx);
        else
          hipLaunchKernelGGL(k_trsv_level_multi<true>, dim3(grid), dim3(WG), 0, ctx->stream, L.m, L.w, nrhs, S.rows + L.row_off, S.cols + L.ent_off,
                             S.vals + L.ent_off, S.dinv + L.row_off, D, ldd, X, ldx);
      } else {
        if (wide)
          hipLaunchKernelGGL(k_trsv_level_multi_wide<false>, dim3(grid), dim3(WG), 0, ctx->stream, L.m, L.w, nrhs, S.rows + L.row_off, S.cols + L.ent_off,
                             S.vals + L.ent_off, (const double *)nullptr, D, ldd, X, ldx);
        else
          hipLaunchKernelGGL(k_trsv_level_multi<false>, dim3(grid), dim3(WG), 0, ctx->stream, L.m, L.w, nrhs, S.rows + L.row_off, S.cols + L.ent_off,
                             S.vals + L.ent_off, (const double *)nullptr, D, ldd, X, ldx);
      }
    }
  }
}
// single-precision preconditioner sweeps of an ILU(0) factor (kernels.hpp: k_trsv_level_multi4_f32); D, X double
// columns [c0, c0 + nc) of the block on `stream` (nc % 4 == 0): the columns are independent, so two halves can run as two chains
static void enqueue_multi_levels_f32(ddm_ilu0 *F, hipStream_t stream, int nrhs, int c0, int nc, const double *D, int64_t ldd, double *X, int64_t ldx)
{
  for (int pass = 0; pass < 2; ++pass) {
    const TriSchedule &S = pass ? F->U : F->L;
    for (int64_t l = 0; l < S.nlev; ++l) {
      const LevelDesc &L = S.desc[l];
      if (L.m == 0) continue;
      const unsigned grid = (unsigned)(((int64_t)L.m * (nc / 4) + WG - 1) / WG);
      if (pass)
        hipLaunchKernelGGL(k_trsv_level_multi4_f32<true>, dim3(grid), dim3(WG), 0, stream, L.m, L.w, nc / 4, S.rows + L.row_off, S.cols + L.ent_off, S.vals_f32 + L.ent_off,
                           S.dinv_f32 + L.row_off, D + c0, ldd, F->xf + c0, (int64_t)nrhs, X + c0, ldx);
      else
        hipLaunchKernelGGL(k_trsv_level_multi4_f32<false>, dim3(grid), dim3(WG), 0, stream, L.m, L.w, nc / 4, S.rows + L.row_off, S.cols + L.ent_off, S.vals_f32 + L.ent_off,
                           (const float *)nullptr, D + c0, ldd, F->xf + c0, (int64_t)nrhs, X + c0, ldx);
    }
  }
}
static int ilu0_solve_multi_ld(ddm_ctx *ctx, ddm_ilu0 *F, int nrhs, const double *D, int64_t ldd, double *X, int64_t ldx, bool f32 = false)
{
  if (!F || !D || !X || D == X || nrhs < 1 || ldd < nrhs || ldx < nrhs) return fail(ctx, DDM_EINVAL, "ddm_ilu0_solve_multi: bad arguments");
  if (F->n == 0) return DDM_OK;
  // single precision only for plain ILU(0) factors on aligned blocks of a multiple of 4 columns without wide levels
  f32 = f32 && !F->sn && !F->perm && nrhs % 4 == 0 && ldd % 4 == 0 && ldx % 4 == 0 && ((uintptr_t)D & 31) == 0 && ((uintptr_t)X & 31) == 0;
  if (f32)
    for (const TriSchedule *S : {&F->L, &F->U})
      for (const LevelDesc &L : S->desc) f32 = f32 && L.w < 96;
  if (F->mgraph && F->mg_D == D && F->mg_X == X && F->mg_nrhs == nrhs && F->mg_ldd == ldd && F->mg_ldx == ldx && F->mg_f32 == f32) {
    HIPCHECK(ctx, hipGraphLaunch(F->mgraph, ctx->stream));
    return DDM_OK;
  }
  if (f32) {
    for (TriSchedule *S : {&F->L, &F->U}) {
      if (!S->vals_f32 && S->ell_entries > 0) {
        HIPCHECK(ctx, hipMalloc((void **)&S->vals_f32, sizeof(float) * (size_t)S->ell_entries));
        hipLaunchKernelGGL(k_to_float, dim3((unsigned)((S->ell_entries + 255) / 256)), dim3(256), 0, ctx->stream, S->ell_entries, (const double *)S->vals, S->vals_f32);
      }
      if (S == &F->U && !S->dinv_f32) {
        HIPCHECK(ctx, hipMalloc((void **)&S->dinv_f32, sizeof(float) * (size_t)std::max<int64_t>(F->n, 1)));
        hipLaunchKernelGGL(k_to_float, dim3((unsigned)((F->n + 255) / 256)), dim3(256), 0, ctx->stream, F->n, (const double *)S->dinv, S->dinv_f32);
      }
    }
    if (F->xf_nrhs < nrhs) {
      (void)hipFree(F->xf);
      F->xf = nullptr;
      F->xf_nrhs = 0;
      HIPCHECK(ctx, hipMalloc((void **)&F->xf, sizeof(float) * (size_t)F->n * (size_t)nrhs));
      F->xf_nrhs = nrhs;
    }
    HIPCHECK(ctx, hipGetLastError());
  }
  if (F->mgraph) {
    (void)hipGraphExecDestroy(F->mgraph);
    F->mgraph = nullptr;
  }
  if (F->sn) {
    const int w = std::min(nrhs, 48); // the panel kernels take up to 48 columns: wider blocks are solved in column panels
    const double *partial_before = F->sn->d_partial, *contrib_before = F->sn->d_contrib;
    if (!sn::reserve(*F->sn, w)) return fail(ctx, DDM_EHIP, "sparse direct solver: allocation failed");
    if ((F->sn->d_partial != partial_before || F->sn->d_contrib != contrib_before) && F->graph) { // the single-vector graph's nodes hold the old scratch pointers
      (void)hipGraphExecDestroy(F->graph);
      F->graph = nullptr;
    }
    if (F->pm_nrhs < w) {
      (void)hipFree(F->pD);
      F->pD = nullptr;
      F->pm_nrhs = 0;
      HIPCHECK(ctx, hipMalloc((void **)&F->pD, sizeof(double) * (size_t)F->n * (size_t)w));
      F->pm_nrhs = w;
    }
    if (F->refine_steps > 0 && F->pr_cols < w) {
      (void)hipFree(F->pr);
      F->pr = nullptr;
      F->pr_cols = 0;
      HIPCHECK(ctx, hipMalloc((void **)&F->pr, sizeof(double) * (size_t)F->n * (size_t)w));
      F->pr_cols = w;
      if (F->graph) { // (the single-vector graph holds the old residual buffer)
        (void)hipGraphExecDestroy(F->graph);
        F->graph = nullptr;
      }
    }
  }
  if (F->perm && F->pm_nrhs < nrhs) {
    (void)hipFree(F->pD);
    (void)hipFree(F->pX);
    F->pD = F->pX = nullptr;
    F->pm_nrhs = 0;
    HIPCHECK(ctx, hipMalloc((void **)&F->pD, sizeof(double) * (size_t)F->n * (size_t)nrhs));
    HIPCHECK(ctx, hipMalloc((void **)&F->pX, sizeof(double) * (size_t)(F->n + F->nvirt) * (size_t)nrhs));
    F->pm_nrhs = nrhs;
  }
  hipGraph_t g = nullptr;
  HIPCHECK(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
  if (F->sn) {
    for (int c0 = 0; c0 < nrhs; c0 += 48) {
      const int w = std::min(48, nrhs - c0);
      hipLaunchKernelGGL(k_perm_gather, dim3(grid_for(F->n * w)), dim3(WG), 0, ctx->stream, F->n, w, F->sn->d_perm, D + c0, ldd, F->pD);
      sn::solve(*F->sn, ctx->stream, w, F->pD, w);
      hipLaunchKernelGGL(k_perm_scatter, dim3(grid_for(F->n * w)), dim3(WG), 0, ctx->stream, F->n, w, F->sn->d_perm, (const double *)F->pD, X + c0, ldx);
      for (int it = 0; it < F->refine_steps; ++it) { // X += A^-1 (D - A X), panel by panel
        hipLaunchKernelGGL(k_residual_rowmajor, dim3((unsigned)((F->n * (int64_t)w + WG - 1) / WG)), dim3(WG), 0, ctx->stream, F->n, w, (const int64_t *)F->ref_rp, (const int32_t *)F->ref_ci,
                           (const double *)F->ref_va, (const double *)(X + c0), ldx, D + c0, ldd, F->pr, (int64_t)w);
        hipLaunchKernelGGL(k_perm_gather, dim3(grid_for(F->n * w)), dim3(WG), 0, ctx->stream, F->n, w, F->sn->d_perm, (const double *)F->pr, (int64_t)w, F->pD);
        sn::solve(*F->sn, ctx->stream, w, F->pD, w);
        hipLaunchKernelGGL(k_perm_scatter_add, dim3(grid_for(F->n * w)), dim3(WG), 0, ctx->stream, F->n, w, F->sn->d_perm, (const double *)F->pD, X + c0, ldx);
      }
    }
  } else if (F->perm) { // sparse direct factor: solve in the fill-reducing order on packed work blocks
    hipLaunchKernelGGL(k_perm_gather, dim3(grid_for(F->n * nrhs)), dim3(WG), 0, ctx->stream, F->n, nrhs, F->perm, D, ldd, F->pD);
    enqueue_multi_levels_csr(ctx, F->Lc, false, nrhs, F->pD, nrhs, F->pX, nrhs);
    enqueue_multi_levels_csr(ctx, F->Uc, true, nrhs, F->pD, nrhs, F->pX, nrhs);
    hipLaunchKernelGGL(k_perm_scatter, dim3(grid_for(F->n * nrhs)), dim3(WG), 0, ctx->stream, F->n, nrhs, F->perm, (const double *)F->pX, X, ldx);
  } else if (f32) {
    // (Splitting the columns into two halves that run as two parallel chains of the captured graph -- a second stream joining the
    //  capture -- was measured and is slower: 6.8 against 5.6 s for the 109 block iterations of the headline GenEO run; every level
    //  kernel is latency-bound, so two half-width kernels cost two full ones and the chains do not overlap enough to pay for that.)
    enqueue_multi_levels_f32(F, ctx->stream, nrhs, 0, nrhs, D, ldd, X, ldx);
  } else {
    enqueue_multi_levels(ctx, F, nrhs, D, ldd, X, ldx);
  }
  hipError_t e = hipStreamEndCapture(ctx->stream, &g);
  if (e != hipSuccess) return fail(ctx, DDM_EHIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
  e = hipGraphInstantiate(&F->mgraph, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) {
    F->mgraph = nullptr;
    return fail(ctx, DDM_EHIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
  }
  F->mg_D = D;
  F->mg_X = X;
  F->mg_nrhs = nrhs;
  F->mg_ldd = ldd;
  F->mg_ldx = ldx;
  F->mg_f32 = f32;
  HIPCHECK(ctx, hipGraphLaunch(F->mgraph, ctx->stream));
  return DDM_OK;
}
extern "C" int ddm_ilu0_solve_multi(ddm_ctx *ctx, ddm_ilu0 *F, int nrhs, const double *D, double *X) { return ilu0_solve_multi_ld(ctx, F, nrhs, D, nrhs, X, nrhs); }
// the same solve with SINGLE-PRECISION sweeps (factor entries and work block in float, D read and X written in double): preconditioner
// grade -- what the GenEO block iteration applies.  Falls back to the double sweeps when nrhs is not a multiple of 4 or F is a sparse
// direct factor.
extern "C" int ddm_ilu0_solve_multi_f32(ddm_ctx *ctx, ddm_ilu0 *F, int nrhs, const double *D, double *X) { return ilu0_solve_multi_ld(ctx, F, nrhs, D, nrhs, X, nrhs, true); }

// ---- halo --------------------------------------------------------------------------------------
struct ddm_halo {
  int tag = 0, mode = 0;
  int64_t nsend = 0, nrecv = 0, ndst = 0, self_off_send = 0, self_off_recv = 0, self_count = 0;
  int64_t *send_idx = nullptr, *dst_idx = nullptr, *dst_ptr = nullptr, *src_pos = nullptr;
  double *sendbuf = nullptr, *recvbuf = nullptr;
  bool remote = false; // any traffic to/from other ranks
  std::vector<int64_t> send_counts, recv_counts; // per peer (the layout of sendbuf / recvbuf)
};

extern "C" int ddm_halo_create(ddm_ctx *ctx, int tag, int mode, int64_t nsend, const int64_t *send_idx,
                               const int64_t *send_counts, const int64_t *recv_counts, int64_t ndst, const int64_t *dst_idx,
                               const int64_t *dst_ptr, const int64_t *src_pos, ddm_halo **out)
{
  if (!ctx || !out || (mode != 0 && mode != 1)) return fail(ctx, DDM_EINVAL, "ddm_halo_create: bad arguments");
  ddm_halo *H = new ddm_halo;
  H->tag = tag;
  H->mode = mode;
  H->nsend = nsend;
  H->ndst = ndst;
  H->send_counts.assign(send_counts, send_counts + ctx->nranks);
  H->recv_counts.assign(recv_counts, recv_counts + ctx->nranks);
  int64_t ssum = 0, rsum = 0;
  for (int r = 0; r < ctx->nranks; ++r) {
    if (r == ctx->rank) {
      H->self_off_send = ssum;
      H->self_off_recv = rsum;
      H->self_count = send_counts[r];
      if (send_counts[r] != recv_counts[r]) {
        delete H;
        return fail(ctx, DDM_EINVAL, "halo: self send/recv counts differ");
      }
    } else if (send_counts[r] || recv_counts[r])
      H->remote = true;
    ssum += send_counts[r];
    rsum += recv_counts[r];
  }
  if (ssum != nsend) {
    delete H;
    return fail(ctx, DDM_EINVAL, "halo: send_counts do not sum to nsend");
  }
  H->nrecv = rsum;
  const int64_t nsrc = ndst > 0 ? dst_ptr[ndst] : 0;
  for (int64_t k = 0; k < nsrc; ++k)
    if (src_pos[k] < 0 || src_pos[k] >= rsum) {
      delete H;
      return fail(ctx, DDM_EINVAL, "halo: src_pos out of range");
    }
  int rc = upload(ctx, send_idx, nsend, &H->send_idx);
  if (!rc) rc = upload(ctx, dst_idx, ndst, &H->dst_idx);
  if (!rc) rc = upload(ctx, dst_ptr, ndst + 1, &H->dst_ptr);
  if (!rc) rc = upload(ctx, src_pos, nsrc, &H->src_pos);
  if (!rc && hipMalloc((void **)&H->sendbuf, sizeof(double) * (size_t)std::max<int64_t>(nsend, 1)) != hipSuccess) rc = DDM_EHIP;
  if (!rc && hipMalloc((void **)&H->recvbuf, sizeof(double) * (size_t)std::max<int64_t>(rsum, 1)) != hipSuccess) rc = DDM_EHIP;
  if (rc) {
    ddm_halo_destroy(H);
    return fail(ctx, rc, "halo: device allocation failed");
  }
  *out = H;
  return DDM_OK;
}
extern "C" void ddm_halo_destroy(ddm_halo *H)
{
  if (!H) return;
  (void)hipFree(H->send_idx);
  (void)hipFree(H->dst_idx);
  (void)hipFree(H->dst_ptr);
  (void)hipFree(H->src_pos);
  (void)hipFree(H->sendbuf);
  (void)hipFree(H->recvbuf);
  delete H;
}
extern "C" double *ddm_halo_sendbuf(ddm_halo *H) { return H->sendbuf; }
extern "C" double *ddm_halo_recvbuf(ddm_halo *H) { return H->recvbuf; }

static int halo_exchange_impl(ddm_ctx *ctx, ddm_halo *H, const double *src, double *v);
extern "C" int ddm_halo_exchange(ddm_ctx *ctx, ddm_halo *H, double *v) { return halo_exchange_impl(ctx, H, v, v); }
extern "C" int ddm_halo_exchange_to(ddm_ctx *ctx, ddm_halo *H, const double *src, double *dst)
{
  if (!src || !dst) return fail(ctx, DDM_EINVAL, "ddm_halo_exchange_to: bad arguments");
  return halo_exchange_impl(ctx, H, src, dst);
}
static int halo_exchange_impl(ddm_ctx *ctx, ddm_halo *H, const double *src, double *v)
{
  if (!H) return DDM_OK;
  if (H->nsend == 0 && H->ndst == 0 && !H->remote) return DDM_OK;
  if (H->nsend > 0) hipLaunchKernelGGL(k_pack, dim3(grid_for(H->nsend)), dim3(WG), 0, ctx->stream, H->nsend, H->send_idx, src, H->sendbuf);
  const double *rbuf = H->recvbuf;
  ctx->n_halo_groups += 1;
  if (ctx->rccl && (ctx->nranks > 1 || ctx->rccl_self)) {
    // one grouped point-to-point exchange on the context's stream (xGMI links are point-to-point: every peer pair is its own
    // transfer); the self segment stays a device copy unless the single-GPU self test routes it through RCCL as well
    if (H->self_count > 0 && !ctx->rccl_self)
      HIPCHECK(ctx, hipMemcpyAsync(H->recvbuf + H->self_off_recv, H->sendbuf + H->self_off_send, sizeof(double) * (size_t)H->self_count, hipMemcpyDeviceToDevice, ctx->stream));
    NCCLCHECK(ctx, ctx->nccl.GroupStart());
    int64_t so = 0, ro = 0;
    for (int r = 0; r < ctx->nranks; ++r) {
      const bool self = r == ctx->rank;
      if ((!self || ctx->rccl_self) && H->recv_counts[r] > 0) NCCLCHECK(ctx, ctx->nccl.Recv(H->recvbuf + ro, (size_t)H->recv_counts[r], ncclDouble, r, ctx->rccl_comm, ctx->stream));
      if ((!self || ctx->rccl_self) && H->send_counts[r] > 0) NCCLCHECK(ctx, ctx->nccl.Send(H->sendbuf + so, (size_t)H->send_counts[r], ncclDouble, r, ctx->rccl_comm, ctx->stream));
      so += H->send_counts[r];
      ro += H->recv_counts[r];
    }
    NCCLCHECK(ctx, ctx->nccl.GroupEnd());
  } else if (ctx->nranks > 1) {
    if (!ctx->a2a) return fail(ctx, DDM_ECOMM, "multi-rank context without an exchange (ddm_ctx_set_rccl / ddm_ctx_set_comm)");
    if (ctx->a2a(ctx->user, H->tag, H->sendbuf, H->recvbuf) != 0) return fail(ctx, DDM_ECOMM, "alltoall callback failed (tag %d)", H->tag);
  } else {
    rbuf = H->sendbuf; // single rank: the self segment is the whole buffer
  }
  if (H->ndst > 0) {
    if (H->mode == 1)
      hipLaunchKernelGGL(k_unpack<true>, dim3(grid_for(H->ndst)), dim3(WG), 0, ctx->stream, H->ndst, H->dst_idx, H->dst_ptr, H->src_pos, rbuf, v);
    else
      hipLaunchKernelGGL(k_unpack<false>, dim3(grid_for(H->ndst)), dim3(WG), 0, ctx->stream, H->ndst, H->dst_idx, H->dst_ptr, H->src_pos, rbuf, v);
  }
  HIPCHECK(ctx, hipGetLastError());
  return DDM_OK;
}

// ---- reductions --------------------------------------------------------------------------------
// result (device scalar) = sum over ranks of sum_i [mask_i] x_i y_i
static int dot_device(ddm_ctx *ctx, int64_t n, const uint8_t *mask, const double *x, const double *y, double *result_dev)
{
  const int nb = grid_for(n, WG * 4, RED_MAX_BLOCKS);
  if (mask)
    hipLaunchKernelGGL(k_dot_partial<true>, dim3(nb), dim3(WG), 0, ctx->stream, n, mask, x, y, ctx->partial);
  else
    hipLaunchKernelGGL(k_dot_partial<false>, dim3(nb), dim3(WG), 0, ctx->stream, n, mask, x, y, ctx->partial);
  hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(WG), 0, ctx->stream, nb, ctx->partial, result_dev);
  HIPCHECK(ctx, hipGetLastError());
  DDMCHECK(ctx_allreduce(ctx, result_dev, 1, "scalar product"));
  return DDM_OK;
}

// ---- NonOverlappingOperator --------------------------------------------------------------------
struct ddm_op {
  const ddm_csr *A = nullptr;
  ddm_halo *halo = nullptr;
  uint8_t *owner = nullptr;
  int64_t n = 0;
  double *tmp = nullptr;
};
extern "C" int ddm_op_create(ddm_ctx *ctx, const ddm_csr *A, ddm_halo *novlp_add, const uint8_t *owner_mask_host, ddm_op **out)
{
  if (!ctx || !A || !out || !owner_mask_host) return fail(ctx, DDM_EINVAL, "ddm_op_create: bad arguments");
  if (A->nrows != A->ncols) return fail(ctx, DDM_EINVAL, "operator matrix must be square");
  if (novlp_add && novlp_add->mode != 1) return fail(ctx, DDM_EINVAL, "operator halo must be an 'add' halo");
  ddm_op *op = new ddm_op;
  op->A = A;
  op->halo = novlp_add;
  op->n = A->nrows;
  int rc = upload(ctx, owner_mask_host, op->n, &op->owner);
  if (!rc && hipMalloc((void **)&op->tmp, sizeof(double) * (size_t)std::max<int64_t>(op->n, 1)) != hipSuccess) rc = DDM_EHIP;
  if (rc) {
    ddm_op_destroy(op);
    return fail(ctx, rc, "ddm_op_create: allocation failed");
  }
  *out = op;
  return DDM_OK;
}
extern "C" void ddm_op_destroy(ddm_op *op)
{
  if (!op) return;
  (void)hipFree(op->owner);
  (void)hipFree(op->tmp);
  delete op;
}
extern "C" int ddm_op_apply(ddm_ctx *ctx, ddm_op *op, const double *x, double *y)
{
  ScopedTimer t(ctx, "Operator/apply");
  DDMCHECK(ddm_csr_mv(ctx, op->A, x, y));           // A->mv(x, y)
  return ddm_halo_exchange(ctx, op->halo, y);       // comm->addOwnerCopyToOwnerCopy(y, y)
}
extern "C" int ddm_op_applyscaleadd(ddm_ctx *ctx, ddm_op *op, double alpha, const double *x, double *y)
{
  ScopedTimer t(ctx, "Operator/applyscaleadd");
  // y1 = y; y = 0; usmv; halo; y += y1   (only alpha*A*x is communicated, y is already consistent)
  DDMCHECK(ddm_csr_mv(ctx, op->A, x, op->tmp));
  DDMCHECK(ddm_halo_exchange(ctx, op->halo, op->tmp));
  hipLaunchKernelGGL(k_axpy, dim3(grid_for(op->n)), dim3(WG), 0, ctx->stream, op->n, alpha, op->tmp, y);
  HIPCHECK(ctx, hipGetLastError());
  return DDM_OK;
}
extern "C" int ddm_dot(ddm_ctx *ctx, ddm_op *op, const double *x, const double *y, double *result_host)
{
  DDMCHECK(dot_device(ctx, op->n, op->owner, x, y, ctx->scal + 8));
  return ddm_memcpy_d2h(ctx, result_host, ctx->scal + 8, sizeof(double));
}
extern "C" int ddm_norm(ddm_ctx *ctx, ddm_op *op, const double *x, double *result_host)
{
  DDMCHECK(ddm_dot(ctx, op, x, x, result_host));
  *result_host = std::sqrt(*result_host);
  return DDM_OK;
}

// ---- SchwarzPreconditioner ---------------------------------------------------------------------
struct ddm_schwarz {
  int64_t n = 0, n_novlp = 0;
  int type = 1;
  ddm_ilu0 *solver = nullptr;
  int32_t *ext_map = nullptr;
  double *pou = nullptr;
  double *d_ovlp = nullptr, *x_ovlp = nullptr;
  ddm_halo *copy = nullptr, *add = nullptr;
};
extern "C" int ddm_schwarz_create(ddm_ctx *ctx, const ddm_csr *A_dir, int64_t nblocks, const int64_t *block_ptr, int64_t n_novlp,
                                  const int32_t *ext_map_host, const double *pou_host, int type, ddm_halo *ovlp_copy,
                                  ddm_halo *ovlp_add, ddm_schwarz **out)
{
  return ddm_schwarz_create_ex(ctx, A_dir, nblocks, block_ptr, n_novlp, ext_map_host, pou_host, type, "ilu0", ovlp_copy, ovlp_add, out);
}
// subdomain_solver: the `type` key of the [schwarz.subdomain_solver] sub-tree (schwarz.hh:85-92): "ilu0" (dune-istl's SeqILU,
// n = 0) or one of "cholmod" / "ldl" / "spqr"-less synonyms "direct", "cholesky" for the sparse direct solver of this library
// (SPD matrices; "umfpack" is accepted for symmetric positive definite input only).
extern "C" int ddm_schwarz_create_ex(ddm_ctx *ctx, const ddm_csr *A_dir, int64_t nblocks, const int64_t *block_ptr, int64_t n_novlp,
                                     const int32_t *ext_map_host, const double *pou_host, int type, const char *subdomain_solver,
                                     ddm_halo *ovlp_copy, ddm_halo *ovlp_add, ddm_schwarz **out)
{
  if (!ctx || !A_dir || !out || !ext_map_host) return fail(ctx, DDM_EINVAL, "ddm_schwarz_create: bad arguments");
  const std::string st = subdomain_solver ? subdomain_solver : "ilu0";
  const bool direct = st == "cholmod" || st == "direct" || st == "cholesky" || st == "umfpack" || st == "ldl";
  if (!direct && st != "ilu0" && st != "ilu") return fail(ctx, DDM_ENOTIMPL, "Unknown subdomain solver type '%s'", st.c_str()); // solver factory lookup (:85-92)
  bool general = st == "umfpack";
  if (st == "direct") { // pick the factorisation by looking at the values: symmetric -> Cholesky
    general = false;
    const int64_t nn = A_dir->nrows;
    for (int64_t i = 0; i < nn && !general; ++i)
      for (int64_t k = A_dir->h_rp[i]; k < A_dir->h_rp[i + 1] && !general; ++k) {
        const int64_t j = A_dir->h_ci[k];
        if (j <= i) continue;
        const auto b = A_dir->h_ci.begin() + A_dir->h_rp[j], e = A_dir->h_ci.begin() + A_dir->h_rp[j + 1];
        const auto it = std::lower_bound(b, e, (int32_t)i);
        const double vt = (it != e && *it == i) ? A_dir->h_va[(size_t)(it - A_dir->h_ci.begin())] : 0.0;
        if (std::fabs(vt - A_dir->h_va[k]) > 1e-12 * (std::fabs(vt) + std::fabs(A_dir->h_va[k]))) general = true;
      }
  }
  if (type != 0 && type != 1) return fail(ctx, DDM_ENOTIMPL, "Unknown Schwarz type %d", type); // schwarz.hh:83
  if (ovlp_copy && ovlp_copy->mode != 0) return fail(ctx, DDM_EINVAL, "ovlp_copy must be a 'copy' halo");
  if (ovlp_add && ovlp_add->mode != 1) return fail(ctx, DDM_EINVAL, "ovlp_add must be an 'add' halo");
  const int64_t n = A_dir->nrows;
  for (int64_t i = 0; i < n; ++i)
    if (ext_map_host[i] >= n_novlp) return fail(ctx, DDM_EINVAL, "ext_map entry out of range"); // size checks, schwarz.hh:186-193
  ddm_schwarz *S = new ddm_schwarz;
  S->n = n;
  S->n_novlp = n_novlp;
  S->type = type;
  S->copy = ovlp_copy;
  S->add = ovlp_add;
  int rc = direct ? ddm_direct_create(ctx, A_dir, nblocks, block_ptr, general ? 1 : 0, 0.0, &S->solver)
                  : ddm_ilu0_create(ctx, A_dir, nblocks, block_ptr, &S->solver); // factorisation happens in the ctor (:92)
  if (!rc) rc = upload(ctx, ext_map_host, n, &S->ext_map);
  if (!rc && pou_host) rc = upload(ctx, pou_host, n, &S->pou);
  if (!rc && hipMalloc((void **)&S->d_ovlp, sizeof(double) * (size_t)std::max<int64_t>(n, 1)) != hipSuccess) rc = fail(ctx, DDM_EHIP, "alloc");
  if (!rc && hipMalloc((void **)&S->x_ovlp, sizeof(double) * (size_t)std::max<int64_t>(n, 1)) != hipSuccess) rc = fail(ctx, DDM_EHIP, "alloc");
  if (rc) {
    ddm_schwarz_destroy(S);
    return rc;
  }
  *out = S;
  return DDM_OK;
}
extern "C" void ddm_schwarz_destroy(ddm_schwarz *S)
{
  if (!S) return;
  ddm_ilu0_destroy(S->solver);
  (void)hipFree(S->ext_map);
  (void)hipFree(S->pou);
  (void)hipFree(S->d_ovlp);
  (void)hipFree(S->x_ovlp);
  delete S;
}
extern "C" int64_t ddm_schwarz_num_levels(const ddm_schwarz *S, int upper) { return ddm_ilu0_num_levels(S->solver, upper); }
extern "C" int64_t ddm_schwarz_factor_nnz(const ddm_schwarz *S) { return (S && S->solver) ? S->solver->nnz : 0; } // stored entries of L + U (+ diagonal)
extern "C" int ddm_schwarz_engine(const ddm_schwarz *S) { return S ? ddm_ilu0_engine(S->solver) : -1; }
// Synchronous.  DDM_OK, or DDM_ENUMERIC when a single-launch local solve gave up waiting (its results are invalid: the
// GPU is shared with another process, or the grid was not co-resident) -- the reference's apply has no error return
// (schwarz.hh:131 discards the InverseOperatorResult), so the adaptors poll this in post() and the Krylov drivers at the end.
extern "C" ddm_ilu0 *ddm_schwarz_local_solver(ddm_schwarz *S) { return S ? S->solver : nullptr; } // borrowed (owned by S)
extern "C" int ddm_schwarz_status(ddm_ctx *ctx, const ddm_schwarz *S)
{
  if (!S) return fail(ctx, DDM_EINVAL, "ddm_schwarz_status: bad arguments");
  int st = 0;
  DDMCHECK(ddm_ilu0_status(ctx, S->solver, &st));
  if (st) return fail(ctx, DDM_ENUMERIC, "local triangular solve timed out waiting for a dependency (code %d): results are invalid", st);
  return DDM_OK;
}
// x (= or +=) R~^T [D] A_dir^-1 R~ d
static int schwarz_apply_impl(ddm_ctx *ctx, ddm_schwarz *S, double *x, const double *d, bool acc)
{
  if (const unsigned e = ilu0_peek_status(S->solver)) // fail fast: an earlier local solve gave up (no stream synchronisation here)
    return fail(ctx, DDM_ENUMERIC, "an earlier local triangular solve timed out waiting for a dependency (code %u): results since then are invalid", e);
  {
    ScopedTimer t(ctx, "Schwarz/get defect");
    hipLaunchKernelGGL(k_extend, dim3(grid_for(S->n)), dim3(WG), 0, ctx->stream, S->n, S->ext_map, d, S->d_ovlp); // :121-122
    DDMCHECK(ddm_halo_exchange(ctx, S->copy, S->d_ovlp));                                                          // :125
  }
  {
    ScopedTimer t(ctx, "Schwarz/local solve");
    DDMCHECK(ddm_ilu0_solve(ctx, S->solver, S->d_ovlp, S->x_ovlp)); // :131-133
  }
  {
    ScopedTimer t(ctx, "Schwarz/add solution");
    if (S->type == 1 && S->pou)
      hipLaunchKernelGGL(k_scale, dim3(grid_for(S->n)), dim3(WG), 0, ctx->stream, S->n, S->pou, S->x_ovlp); // :139-141
    DDMCHECK(ddm_halo_exchange(ctx, S->add, S->x_ovlp));                                                     // :138/:142
    if (acc)
      hipLaunchKernelGGL((k_restrict<true, false>), dim3(grid_for(S->n)), dim3(WG), 0, ctx->stream, S->n, S->ext_map, S->x_ovlp, (const double *)nullptr, x);
    else
      hipLaunchKernelGGL((k_restrict<false, false>), dim3(grid_for(S->n)), dim3(WG), 0, ctx->stream, S->n, S->ext_map, S->x_ovlp, (const double *)nullptr, x); // :146
    HIPCHECK(ctx, hipGetLastError());
  }
  return DDM_OK;
}
extern "C" int ddm_schwarz_apply(ddm_ctx *ctx, ddm_schwarz *S, double *x, const double *d)
{
  ScopedTimer t(ctx, "Schwarz/apply");
  return schwarz_apply_impl(ctx, S, x, d, false);
}

// ---- GalerkinPreconditioner --------------------------------------------------------------------
struct ddm_galerkin {
  int64_t n = 0, n_novlp = 0, nsub = 0, kmax = 0, K = 0, ld = 0;
  int32_t *ext_map = nullptr;
  double *basis = nullptr;       // kmax x ld
  int64_t *coarse_index = nullptr;
  double *a0inv = nullptr;
  RowChunk *chunks = nullptr;
  int32_t *sub_chunk_ptr = nullptr;
  int nchunk = 0;
  double *partial = nullptr, *d0 = nullptr, *x0 = nullptr;
  double *d_ovlp = nullptr, *x_ovlp = nullptr;
  ddm_halo *copy = nullptr, *add = nullptr;
};
static constexpr int64_t COARSE_CHUNK_ROWS = 8192;

extern "C" int ddm_galerkin_create(ddm_ctx *ctx, int64_t n, int64_t n_novlp, const int32_t *ext_map_host, int64_t nsub,
                                   const int64_t *sub_ptr, int64_t kmax, const double *basis_host, const int64_t *coarse_index,
                                   int64_t K, const double *a0inv_host, ddm_halo *ovlp_copy, ddm_halo *ovlp_add,
                                   ddm_galerkin **out)
{
  if (!ctx || !out || !ext_map_host || !sub_ptr || !basis_host || !coarse_index || !a0inv_host)
    return fail(ctx, DDM_EINVAL, "ddm_galerkin_create: bad arguments");
  if (kmax < 1) return fail(ctx, DDM_EINVAL, "Must at least pass one template vector"); // galerkin_preconditioner.hh:129
  if (kmax > COARSE_KMAX) return fail(ctx, DDM_ENOTIMPL, "more than %d basis vectors per subdomain are not supported", COARSE_KMAX);
  if (sub_ptr[0] != 0 || sub_ptr[nsub] != n) return fail(ctx, DDM_EINVAL, "Template vectors must match size of matrix"); // :131
  for (int64_t t = 0; t < nsub * kmax; ++t)
    if (coarse_index[t] >= K) return fail(ctx, DDM_EINVAL, "coarse_index out of range");
  ddm_galerkin *G = new ddm_galerkin;
  G->n = n;
  G->n_novlp = n_novlp;
  G->nsub = nsub;
  G->kmax = kmax;
  G->K = K;
  G->ld = (n + 63) / 64 * 64;
  G->copy = ovlp_copy;
  G->add = ovlp_add;
  std::vector<RowChunk> chunks;
  std::vector<int32_t> scp(nsub + 1, 0);
  for (int64_t s = 0; s < nsub; ++s) {
    for (int64_t r = sub_ptr[s]; r < sub_ptr[s + 1]; r += COARSE_CHUNK_ROWS)
      chunks.push_back(RowChunk{r, std::min(r + COARSE_CHUNK_ROWS, sub_ptr[s + 1]), (int32_t)s, 0});
    scp[s + 1] = (int32_t)chunks.size();
  }
  G->nchunk = (int)chunks.size();
  int rc = upload(ctx, ext_map_host, n, &G->ext_map);
  if (!rc) rc = upload(ctx, coarse_index, nsub * kmax, &G->coarse_index);
  if (!rc) rc = upload(ctx, a0inv_host, K * K, &G->a0inv);
  if (!rc) rc = upload(ctx, chunks.data(), (int64_t)chunks.size(), &G->chunks);
  if (!rc) rc = upload(ctx, scp.data(), nsub + 1, &G->sub_chunk_ptr);
  auto dalloc = [&](double **p, int64_t cnt) {
    if (!rc && hipMalloc((void **)p, sizeof(double) * (size_t)std::max<int64_t>(cnt, 1)) != hipSuccess) rc = fail(ctx, DDM_EHIP, "galerkin: allocation failed");
  };
  dalloc(&G->basis, kmax * G->ld);
  dalloc(&G->partial, (int64_t)G->nchunk * kmax);
  dalloc(&G->d0, K + 1); // (+ 1: a scalar may ride on the all-reduce, coarse_allreduce)
  dalloc(&G->x0, K);
  dalloc(&G->d_ovlp, n);
  dalloc(&G->x_ovlp, n);
  if (!rc && hipMemset(G->basis, 0, sizeof(double) * (size_t)(kmax * G->ld)) != hipSuccess) rc = DDM_EHIP;
  if (!rc && hipMemcpy2D(G->basis, sizeof(double) * (size_t)G->ld, basis_host, sizeof(double) * (size_t)n, sizeof(double) * (size_t)n,
                         (size_t)kmax, hipMemcpyHostToDevice) != hipSuccess)
    rc = fail(ctx, DDM_EHIP, "galerkin: basis upload failed");
  if (rc) {
    ddm_galerkin_destroy(G);
    return rc;
  }
  *out = G;
  return DDM_OK;
}
extern "C" void ddm_galerkin_destroy(ddm_galerkin *G)
{
  if (!G) return;
  (void)hipFree(G->ext_map);
  (void)hipFree(G->basis);
  (void)hipFree(G->coarse_index);
  (void)hipFree(G->a0inv);
  (void)hipFree(G->chunks);
  (void)hipFree(G->sub_chunk_ptr);
  (void)hipFree(G->partial);
  (void)hipFree(G->d0);
  (void)hipFree(G->x0);
  (void)hipFree(G->d_ovlp);
  (void)hipFree(G->x_ovlp);
  delete G;
}
// d_ovlp_ready: the overlapping defect (extended + owner values copied to all holders) if the caller already has it -- in the
// additive combination both levels start from the same defect (schwarz.hh:121-125 and galerkin_preconditioner.hh:159-162)
static int galerkin_apply_impl(ddm_ctx *ctx, ddm_galerkin *G, double *x, const double *d, bool acc, const double *d_ovlp_ready = nullptr)
{
  ScopedTimer t(ctx, "GalerkinPrec/apply");
  const double *dov = d_ovlp_ready;
  if (!dov) {
    hipLaunchKernelGGL(k_extend, dim3(grid_for(G->n)), dim3(WG), 0, ctx->stream, G->n, G->ext_map, d, G->d_ovlp); // :159
    DDMCHECK(ddm_halo_exchange(ctx, G->copy, G->d_ovlp));                                                         // :162
    dov = G->d_ovlp;
  }
  hipLaunchKernelGGL(k_coarse_restrict_partial, dim3(G->nchunk), dim3(WG), 0, ctx->stream, (int)G->kmax, G->ld, G->basis, dov,
                     G->chunks, G->partial, G->nchunk); // :165-167
  hipLaunchKernelGGL(k_coarse_restrict_final, dim3(1), dim3(WG), 0, ctx->stream, (int)G->nsub, (int)G->kmax, G->sub_chunk_ptr, G->partial,
                     G->coarse_index, G->K, G->d0);
  HIPCHECK(ctx, hipGetLastError());
  DDMCHECK(coarse_allreduce(ctx, G->d0, G->K)); // replaces MPI_Gatherv (:170-171): every rank obtains the full coarse defect
  hipLaunchKernelGGL(k_dense_mv, dim3((unsigned)((G->K + 3) / 4)), dim3(WG), 0, ctx->stream, G->K, G->a0inv, G->d0, G->x0); // :174-179 (replicated)
  hipLaunchKernelGGL(k_coarse_prolong, dim3(G->nchunk), dim3(WG), 0, ctx->stream, (int)G->kmax, G->ld, G->basis, G->x0, G->coarse_index,
                     G->chunks, G->x_ovlp, G->nchunk);       // :186-188
  DDMCHECK(ddm_halo_exchange(ctx, G->add, G->x_ovlp)); // :190
  if (acc)
    hipLaunchKernelGGL((k_restrict<true, false>), dim3(grid_for(G->n)), dim3(WG), 0, ctx->stream, G->n, G->ext_map, G->x_ovlp, (const double *)nullptr, x);
  else
    hipLaunchKernelGGL((k_restrict<false, false>), dim3(grid_for(G->n)), dim3(WG), 0, ctx->stream, G->n, G->ext_map, G->x_ovlp, (const double *)nullptr, x); // :193
  HIPCHECK(ctx, hipGetLastError());
  return DDM_OK;
}
extern "C" int ddm_galerkin_apply(ddm_ctx *ctx, ddm_galerkin *G, double *x, const double *d)
{
  return galerkin_apply_impl(ctx, G, x, d, false);
}

extern "C" int ddm_galerkin_products(ddm_ctx *ctx, const ddm_csr *A_dir, int64_t nleft, const double *left, int64_t nright,
                                     const double *right, int64_t row0, int64_t row1, double *out_host)
{
  // out[j*nleft + i] = <left_i, A_dir right_j> over rows [row0,row1)   (column-major nleft x nright,
  // the slab layout of galerkin_preconditioner.hh:294 / helpers.hh:252)
  if (!A_dir || !left || !right || !out_host || nleft < 1 || nleft > COARSE_KMAX || nright < 1 || row0 < 0 || row1 > A_dir->nrows || row0 > row1)
    return fail(ctx, DDM_EINVAL, "ddm_galerkin_products: bad arguments");
  if (A_dir->host_only) return fail(ctx, DDM_EINVAL, "the matrix was created without device arrays (ddm_csr_create_host)");
  const int64_t n = A_dir->nrows;
  double *y = nullptr, *partial = nullptr, *outd = nullptr;
  RowChunk *chunks = nullptr;
  std::vector<RowChunk> hc;
  for (int64_t r = row0; r < row1; r += COARSE_CHUNK_ROWS) hc.push_back(RowChunk{r, std::min(r + COARSE_CHUNK_ROWS, row1), 0, 0});
  const int nchunk = (int)hc.size();
  HIPCHECK(ctx, hipMalloc((void **)&y, sizeof(double) * (size_t)n));
  HIPCHECK(ctx, hipMalloc((void **)&partial, sizeof(double) * (size_t)std::max<int64_t>((int64_t)nchunk * nleft, 1)));
  HIPCHECK(ctx, hipMalloc((void **)&outd, sizeof(double) * (size_t)(nleft * nright)));
  int rc = upload(ctx, hc.data(), (int64_t)hc.size(), &chunks);
  std::vector<int32_t> scp = {0, nchunk};
  std::vector<int64_t> cidx(nleft);
  int32_t *d_scp = nullptr;
  int64_t *d_cidx = nullptr;
  if (!rc) rc = upload(ctx, scp.data(), 2, &d_scp);
  for (int64_t j = 0; j < nright && !rc; ++j) {
    // y[row0:row1) = (A_dir right_j)[row0:row1): only the rows the products below read (a whole-matrix product per vector and call
    // was 1 s of the headline setup: 1 280 passes over 3.5 GB); same row sums in the same order as ddm_csr_mv
    if (row1 > row0)
      hipLaunchKernelGGL(k_spmm_rowmajor, dim3((unsigned)((row1 - row0 + WG - 1) / WG)), dim3(WG), 0, ctx->stream, row1 - row0, 1, A_dir->rp + row0, A_dir->ci, A_dir->va,
                         right + j * n, (int64_t)1, y + row0, (int64_t)1);
    if (hipGetLastError() != hipSuccess) rc = fail(ctx, DDM_EHIP, "ddm_galerkin_products: kernel launch failed");
    for (int64_t i = 0; i < nleft; ++i) cidx[i] = i;
    if (!d_cidx) rc = upload(ctx, cidx.data(), nleft, &d_cidx);
    if (rc) break;
    if (nchunk > 0)
      hipLaunchKernelGGL(k_coarse_restrict_partial, dim3(nchunk), dim3(WG), 0, ctx->stream, (int)nleft, n, left, y, chunks, partial, nchunk);
    hipLaunchKernelGGL(k_coarse_restrict_final, dim3(1), dim3(WG), 0, ctx->stream, 1, (int)nleft, d_scp, partial, d_cidx, nleft, outd + j * nleft);
  }
  if (!rc) rc = ddm_memcpy_d2h(ctx, out_host, outd, sizeof(double) * (size_t)(nleft * nright));
  (void)hipFree(y);
  (void)hipFree(partial);
  (void)hipFree(outd);
  (void)hipFree(chunks);
  (void)hipFree(d_scp);
  (void)hipFree(d_cidx);
  return rc;
}

// ---- CombinedPreconditioner --------------------------------------------------------------------
struct ddm_combined {
  int mode = 0;
  ddm_op *op = nullptr;
  ddm_schwarz *schwarz = nullptr;
  ddm_galerkin *galerkin = nullptr;
  double *dnext = nullptr;
  int64_t n = 0;
  bool fused = false;   // additive mode: the levels' overlapping results are summed before ONE halo add (combined_apply_fused)
  bool overlap = false; // ... and the coarse chain runs on a side stream beside the local solve (measured slower: off by default)
};
extern "C" int ddm_combined_create(ddm_ctx *ctx, int mode, ddm_op *op, ddm_schwarz *schwarz, ddm_galerkin *galerkin, ddm_combined **out)
{
  if (!ctx || !out || !schwarz) return fail(ctx, DDM_EINVAL, "ERROR: No preconditioners added yet"); // combined_preconditioner.hh:77
  if (mode != 0 && mode != 1) return fail(ctx, DDM_ENOTIMPL, "Unknown apply mode in CombinedPreconditioner, use either additive or multiplicative"); // :68
  if (mode == 1 && galerkin && !op) return fail(ctx, DDM_EINVAL, "ERROR: ApplyMode is multiplicative but operator A is not provided. Set with `set_op`"); // :146
  ddm_combined *C = new ddm_combined;
  C->mode = mode;
  C->op = op;
  C->schwarz = schwarz;
  C->galerkin = galerkin;
  C->n = schwarz->n_novlp;
  if (mode == 0 && galerkin) {
    const char *f = std::getenv("DDM_FUSE_LEVELS");    // "0": the two levels one after the other (two halo adds: the reference's order of sums)
    const char *e = std::getenv("DDM_OVERLAP_COARSE"); // "1": coarse chain on a side stream
    C->fused = !(f && f[0] == '0') && galerkin->copy == schwarz->copy && galerkin->add == schwarz->add && galerkin->n == schwarz->n && galerkin->n_novlp == schwarz->n_novlp;
    C->overlap = C->fused && e && e[0] == '1' && (ctx->nranks == 1 || ctx->rccl);
  }
  if (hipMalloc((void **)&C->dnext, sizeof(double) * (size_t)std::max<int64_t>(C->n, 1)) != hipSuccess) {
    delete C;
    return fail(ctx, DDM_EHIP, "combined: allocation failed");
  }
  *out = C;
  return DDM_OK;
}
extern "C" int ddm_combined_status(ddm_ctx *ctx, const ddm_combined *C)
{
  if (!C) return fail(ctx, DDM_EINVAL, "ddm_combined_status: bad arguments");
  return C->schwarz ? ddm_schwarz_status(ctx, C->schwarz) : DDM_OK;
}
extern "C" void ddm_combined_destroy(ddm_combined *C)
{
  if (!C) return;
  (void)hipFree(C->dnext);
  delete C;
}
// Additive combination, fused: both levels start from the same extended defect and add over the same interface, so their
// overlapping results are summed BEFORE the exchange (linearity of addOwnerCopyToAll; schwarz.hh:138-146 +
// galerkin_preconditioner.hh:190-193 + combined_preconditioner.hh:136-142) -- one extend, one copy-halo, one halo add and one restrict
// instead of two each; the result differs from the two-pass order by rounding only (measured: 5.54 -> 5.31 ms per iteration at 216^3).
//   extend + copy-halo -> local solve -> (POU scale) -> R d -> all-reduce -> A0^-1 -> R^T x0 -> x_s += x_c -> halo add -> restrict
// two_streams (DDM_OVERLAP_COARSE=1; needs the in-library exchange or a single rank): the coarse chain runs on a side stream BESIDE the
// local solve -- the local solves are latency-bound and leave 85 % of the HBM bandwidth idle, the coarse level is bandwidth-bound.
// Measured at 216^3 it LOSES: the local solve slows from 3.39 to 4.34 ms (its dependent L2 / HBM round trips queue behind the
// basis stream), the coarse chain from 0.87 to 2.2 ms, 5.58 ms per iteration against 5.31 -- off by default.
static int combined_apply_fused(ddm_ctx *ctx, ddm_combined *C, double *x, const double *d, bool two_streams)
{
  ddm_schwarz *S = C->schwarz;
  ddm_galerkin *G = C->galerkin;
  if (two_streams && !ctx->side) {
    HIPCHECK(ctx, hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
    HIPCHECK(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    HIPCHECK(ctx, hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
  }
  {
    ScopedTimer t(ctx, "Schwarz/get defect");
    hipLaunchKernelGGL(k_extend, dim3(grid_for(S->n)), dim3(WG), 0, ctx->stream, S->n, S->ext_map, d, S->d_ovlp);
    DDMCHECK(ddm_halo_exchange(ctx, S->copy, S->d_ovlp));
  }
  auto coarse_chain = [&](int grid) -> int {
    ScopedTimer t(ctx, "GalerkinPrec/apply");
    hipLaunchKernelGGL(k_coarse_restrict_partial, dim3(grid), dim3(WG), 0, ctx->stream, (int)G->kmax, G->ld, G->basis, (const double *)S->d_ovlp, G->chunks, G->partial, G->nchunk);
    hipLaunchKernelGGL(k_coarse_restrict_final, dim3(1), dim3(WG), 0, ctx->stream, (int)G->nsub, (int)G->kmax, G->sub_chunk_ptr, G->partial, G->coarse_index, G->K, G->d0);
    DDMCHECK(coarse_allreduce(ctx, G->d0, G->K));
    hipLaunchKernelGGL(k_dense_mv, dim3((unsigned)((G->K + 3) / 4)), dim3(WG), 0, ctx->stream, G->K, G->a0inv, G->d0, G->x0);
    hipLaunchKernelGGL(k_coarse_prolong, dim3(grid), dim3(WG), 0, ctx->stream, (int)G->kmax, G->ld, G->basis, G->x0, G->coarse_index, G->chunks, G->x_ovlp, G->nchunk);
    return DDM_OK;
  };
  if (two_streams) {
    // inter-rank operations stay totally ordered: copy-halo (main) -> all-reduce (side) -> [join] -> halo add (main)
    HIPCHECK(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
    hipStream_t main = ctx->stream;
    HIPCHECK(ctx, hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
    ctx->stream = ctx->side; // the coarse chain is enqueued on the side stream (kernels, RCCL all-reduce, timer)
    // a small grid: the chain only has to finish within the (latency-bound, ~3 ms) local solve, and a full-rate basis stream would
    // queue in front of the pipe kernel's dependent L2 / HBM round trips (DDM_OVERLAP_GRID: workgroups, default 64)
    static const int side_grid = std::getenv("DDM_OVERLAP_GRID") ? std::max(1, std::atoi(std::getenv("DDM_OVERLAP_GRID"))) : 64;
    const int rc = coarse_chain(std::min(G->nchunk, side_grid));
    const hipError_t e = hipEventRecord(ctx->ev_join, ctx->side);
    ctx->stream = main;
    if (rc) return rc;
    if (e != hipSuccess) return fail(ctx, DDM_EHIP, "hipEventRecord failed: %s", hipGetErrorString(e));
  }
  const double *pou = S->type == 1 ? S->pou : nullptr;
  // one stream: the coarse chain runs first, so that the local solve's last kernel can also apply "x *= pou; x += x_coarse"
  if (!two_streams) DDMCHECK(coarse_chain(G->nchunk));
  {
    ScopedTimer t(ctx, "Schwarz/local solve");
    DDMCHECK(ilu0_solve_epilogue(ctx, S->solver, S->d_ovlp, S->x_ovlp, two_streams ? nullptr : pou, two_streams ? nullptr : (const double *)G->x_ovlp));
  }
  {
    ScopedTimer t(ctx, "Schwarz/add solution");
    if (two_streams) {
      if (pou) hipLaunchKernelGGL(k_scale, dim3(grid_for(S->n)), dim3(WG), 0, ctx->stream, S->n, pou, S->x_ovlp);
      HIPCHECK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
      hipLaunchKernelGGL(k_axpy, dim3(grid_for(S->n)), dim3(WG), 0, ctx->stream, S->n, 1.0, (const double *)G->x_ovlp, S->x_ovlp);
    }
    DDMCHECK(ddm_halo_exchange(ctx, S->add, S->x_ovlp));
    hipLaunchKernelGGL((k_restrict<false, false>), dim3(grid_for(S->n)), dim3(WG), 0, ctx->stream, S->n, S->ext_map, S->x_ovlp, (const double *)nullptr, x);
    HIPCHECK(ctx, hipGetLastError());
  }
  return DDM_OK;
}

extern "C" int ddm_combined_apply(ddm_ctx *ctx, ddm_combined *C, double *x, const double *d)
{
  ScopedTimer t(ctx, "CombinedPreconditioner/apply");
  if (const unsigned e = C->schwarz ? ilu0_peek_status(C->schwarz->solver) : 0u) // fail fast, no synchronisation (see ddm_ilu0_status)
    return fail(ctx, DDM_ENUMERIC, "an earlier local triangular solve timed out waiting for a dependency (code %u): results since then are invalid", e);
  if (C->mode == 0 && C->galerkin && C->fused) return combined_apply_fused(ctx, C, x, d, C->overlap);
  // x = 0; precs[0]->apply(x, d)  (:133-134)  -- the restrict kernel overwrites every entry of x
  DDMCHECK(schwarz_apply_impl(ctx, C->schwarz, x, d, false));
  if (!C->galerkin) return DDM_OK;
  if (C->mode == 0) { // additive: xnext = P1 d; x += xnext (:136-142) -- fused into the restrict of the coarse level
    // both levels extend the same defect over the same interface: the Schwarz level's copy is reused (the local solves read it only)
    static const bool no_share = std::getenv("DDM_NO_SHARED_DEFECT") != nullptr; // diagnostic switch
    const bool share = !no_share && C->galerkin->copy == C->schwarz->copy && C->galerkin->n == C->schwarz->n && C->galerkin->n_novlp == C->schwarz->n_novlp;
    return galerkin_apply_impl(ctx, C->galerkin, x, d, true, share ? C->schwarz->d_ovlp : nullptr);
  }
  // multiplicative: dnext = d - A x; x += P1 dnext (:149-158)
  HIPCHECK(ctx, hipMemcpyAsync(C->dnext, d, sizeof(double) * (size_t)C->n, hipMemcpyDeviceToDevice, ctx->stream));
  DDMCHECK(ddm_op_applyscaleadd(ctx, C->op, -1.0, x, C->dnext));
  return galerkin_apply_impl(ctx, C->galerkin, x, C->dnext, true);
}

// ---- CG ----------------------------------------------------------------------------------------
// dune-istl CGSolver::apply, split so that a caller can time an exact number of
// iterations: begin = "b -= A x; def0 = ||b||", one step = "prec.apply; rho; [beta; p = beta p + q];
// q = A p; alpha; lambda; x += lambda p; b -= lambda q; def = ||b||".
struct ddm_cg {
  ddm_op *op = nullptr;
  ddm_combined *prec = nullptr;
  double *x = nullptr, *b = nullptr, *p = nullptr, *q = nullptr;
  int64_t n = 0;
  int it = 0;
  double def0 = 0.0;
};
extern "C" int ddm_cg_begin(ddm_ctx *ctx, ddm_op *op, ddm_combined *prec, double *x, double *b, ddm_cg **out)
{
  if (!ctx || !op || !prec || !x || !b || !out) return fail(ctx, DDM_EINVAL, "ddm_cg_begin: bad arguments");
  ddm_cg *S = new ddm_cg;
  S->op = op;
  S->prec = prec;
  S->x = x;
  S->b = b;
  S->n = op->n;
  if (hipMalloc((void **)&S->p, sizeof(double) * (size_t)std::max<int64_t>(S->n, 1)) != hipSuccess ||
      hipMalloc((void **)&S->q, sizeof(double) * (size_t)std::max<int64_t>(S->n, 1)) != hipSuccess) {
    (void)hipFree(S->p);
    delete S;
    return fail(ctx, DDM_EHIP, "ddm_cg_begin: allocation failed");
  }
  int rc = ddm_op_applyscaleadd(ctx, op, -1.0, x, b); // prec.pre(x,b); b -= A x
  double bb = 0.0;
  if (!rc) rc = dot_device(ctx, S->n, op->owner, b, b, ctx->scal + 5);
  if (!rc) rc = ddm_memcpy_d2h(ctx, &bb, ctx->scal + 5, sizeof(double));
  if (rc) {
    (void)hipFree(S->p);
    (void)hipFree(S->q);
    delete S;
    return rc;
  }
  S->def0 = std::sqrt(bb);
  *out = S;
  return DDM_OK;
}
extern "C" void ddm_cg_end(ddm_ctx *ctx, ddm_cg *S)
{
  if (!S) return;
  if (ctx) (void)hipStreamSynchronize(ctx->stream);
  (void)hipFree(S->p);
  (void)hipFree(S->q);
  delete S;
}
extern "C" double ddm_cg_def0(const ddm_cg *S) { return S->def0; }
// Enqueues k iterations without synchronising; the squared defect of the last one is left in
// device scalar 5 (read it with ddm_cg_defect).
extern "C" int ddm_cg_steps(ddm_ctx *ctx, ddm_cg *S, int k)
{
  double *scal = ctx->scal;
  const int G = grid_for(S->n);
  for (int i = 0; i < k; ++i) {
    const bool first = S->it == 0;
    DDMCHECK(ddm_combined_apply(ctx, S->prec, first ? S->p : S->q, S->b));                 // q = M^-1 b  (p on the first step)
    DDMCHECK(dot_device(ctx, S->n, S->op->owner, first ? S->p : S->q, S->b, scal + (first ? 0 : 3))); // rho = <q, b>
    if (!first) {
      hipLaunchKernelGGL(k_cg_beta, dim3(1), dim3(1), 0, ctx->stream, scal);                 // beta = rho / rholast; rholast = rho
      hipLaunchKernelGGL(k_cg_direction, dim3(G), dim3(WG), 0, ctx->stream, S->n, scal, S->q, S->p); // p = beta p + q
    }
    DDMCHECK(ddm_op_apply(ctx, S->op, S->p, S->q));                                          // q = A p
    DDMCHECK(dot_device(ctx, S->n, S->op->owner, S->p, S->q, scal + 1));                     // alpha = <p, q>
    hipLaunchKernelGGL(k_cg_lambda, dim3(1), dim3(1), 0, ctx->stream, scal);                 // lambda = rholast / alpha
    { // x += lambda p; b -= lambda q; def^2 = <b, b> (partial sums in the same kernel)
      const int nb = grid_for(S->n, WG * 4, RED_MAX_BLOCKS);
      if (S->op->owner)
        hipLaunchKernelGGL(k_cg_update_norm<true>, dim3(nb), dim3(WG), 0, ctx->stream, S->n, scal, S->op->owner, S->p, S->q, S->x, S->b, ctx->partial);
      else
        hipLaunchKernelGGL(k_cg_update_norm<false>, dim3(nb), dim3(WG), 0, ctx->stream, S->n, scal, S->op->owner, S->p, S->q, S->x, S->b, ctx->partial);
      hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(WG), 0, ctx->stream, nb, ctx->partial, scal + 5);
      // The rank-local sum is complete; its all-reduce rides on the coarse-defect all-reduce of the NEXT iteration's preconditioner
      // (one RCCL launch saved per iteration) unless this is the chunk's last iteration -- whoever reads the defect (ddm_cg_defect)
      // needs it now -- or there is no coarse level to ride on.
      if (i + 1 < k && S->prec->galerkin) ctx->piggy = scal + 5;
      else DDMCHECK(ctx_allreduce(ctx, scal + 5, 1, "scalar product"));
    }
    S->it += 1;
  }
  if (ctx->piggy) { // (cannot happen: the last iteration of a chunk reduces its own norm)
    ctx->piggy = nullptr;
    DDMCHECK(ctx_allreduce(ctx, scal + 5, 1, "scalar product"));
  }
  HIPCHECK(ctx, hipGetLastError());
  return DDM_OK;
}
extern "C" int ddm_cg_defect(ddm_ctx *ctx, ddm_cg *S, double *def_host) // synchronous
{
  double bb = 0.0;
  DDMCHECK(ddm_memcpy_d2h(ctx, &bb, ctx->scal + 5, sizeof(double)));
  *def_host = std::sqrt(bb);
  (void)S;
  return DDM_OK;
}

extern "C" int ddm_cg_solve(ddm_ctx *ctx, ddm_op *op, ddm_combined *prec, double *x, double *b, double reduction, int maxit,
                            int fixed_iterations, double *hist_host, ddm_solve_result *res)
{
  if (!res) return fail(ctx, DDM_EINVAL, "ddm_cg_solve: bad arguments");
  ddm_cg *S = nullptr;
  DDMCHECK(ddm_cg_begin(ctx, op, prec, x, b, &S));
  const double def0 = S->def0;
  res->def0 = def0;
  res->iterations = 0;
  res->converged = 0;
  res->reduction = 1.0;
  res->elapsed_s = 0.0;
  if (hist_host) hist_host[0] = def0;
  if (!(def0 == def0)) {
    ddm_cg_end(ctx, S);
    return fail(ctx, DDM_ENUMERIC, "initial defect is NaN");
  }
  if (def0 < 1e-30) {
    res->converged = 1;
    ddm_cg_end(ctx, S);
    return DDM_OK;
  }
  (void)hipStreamSynchronize(ctx->stream);
  const auto t0 = std::chrono::steady_clock::now();
  int rc = DDM_OK;
  double deff = def0;
  if (fixed_iterations > 0 && !hist_host) {
    rc = ddm_cg_steps(ctx, S, fixed_iterations);
    if (!rc) rc = ddm_cg_defect(ctx, S, &deff);
    res->iterations = fixed_iterations;
  } else {
    const int iters = fixed_iterations > 0 ? fixed_iterations : maxit;
    for (int i = 1; i <= iters && !rc; ++i) {
      rc = ddm_cg_steps(ctx, S, 1);
      if (!rc) rc = ddm_cg_defect(ctx, S, &deff); // the Krylov loop tests the defect every iteration
      if (rc) break;
      res->iterations = i;
      if (hist_host) hist_host[i] = deff;
      if (!(deff == deff)) {
        rc = fail(ctx, DDM_ENUMERIC, "defect is NaN in iteration %d", i);
        break;
      }
      if (fixed_iterations <= 0 && (deff < def0 * reduction || deff < 1e-30)) {
        res->converged = 1;
        break;
      }
    }
  }
  (void)hipStreamSynchronize(ctx->stream);
  res->elapsed_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  res->reduction = deff / def0;
  if (!rc && prec->schwarz) {
    int st = 0;
    rc = ddm_ilu0_status(ctx, prec->schwarz->solver, &st);
    if (!rc && st) rc = fail(ctx, DDM_ENUMERIC, "persistent triangular solve timed out waiting for a level (results invalid)");
  }
  ddm_cg_end(ctx, S);
  return rc;
}

// ---- restarted GMRES -----------------------------------------------------------------------------
// dune-istl RestartedGMResSolver::apply (DUNE 2.10 solvers.hh; not in the snapshot, restated from the
// published implementation): left preconditioning, modified Gram-Schmidt, Givens rotations; the
// monitored quantity is the norm of the PRECONDITIONED defect.  Selected by [solver] type =
// restartedgmressolver in examples/poisson.ini:12-17 (restart = 100) and the default of
// dune/ddm/twolevel_schwarz.hh:121-130 (restart = 30).  Krylov basis, dots and updates stay on the
// device; per iteration the i+2 Hessenberg entries are read back for the rotations on the host.
static void gmres_generate_rotation(double dx, double dy, double &cs, double &sn)
{
  const double ndx = std::fabs(dx), ndy = std::fabs(dy);
  if (ndy < 1e-15) {
    cs = 1.0;
    sn = 0.0;
  } else if (ndx < 1e-15) {
    cs = 0.0;
    sn = 1.0;
  } else if (ndy > ndx) {
    const double t = ndx / ndy;
    cs = 1.0 / std::sqrt(1.0 + t * t);
    sn = cs;
    cs *= t;
    sn *= dx / ndx;
    sn *= dy / ndy;
  } else {
    const double t = ndy / ndx;
    cs = 1.0 / std::sqrt(1.0 + t * t);
    sn = cs;
    sn *= dy / dx;
  }
}
static void gmres_apply_rotation(double &dx, double &dy, double cs, double sn)
{
  const double t = cs * dx + sn * dy;
  dy = -sn * dx + cs * dy;
  dx = t;
}

extern "C" int ddm_gmres_solve(ddm_ctx *ctx, ddm_op *op, ddm_combined *prec, double *x, double *b, double reduction, int maxit,
                               int restart, double *hist_host, ddm_solve_result *res)
{
  if (!ctx || !op || !prec || !x || !b || !res || restart < 1) return fail(ctx, DDM_EINVAL, "ddm_gmres_solve: bad arguments");
  const int64_t n = op->n;
  const int m = restart;
  const int G = grid_for(n);
  double *V = nullptr, *w = nullptr, *hdev = nullptr;
  HIPCHECK(ctx, hipMalloc((void **)&V, sizeof(double) * (size_t)std::max<int64_t>(n, 1) * (size_t)(m + 1)));
  HIPCHECK(ctx, hipMalloc((void **)&w, sizeof(double) * (size_t)std::max<int64_t>(n, 1)));
  HIPCHECK(ctx, hipMalloc((void **)&hdev, sizeof(double) * (size_t)(m + 2)));
  auto v = [&](int k) { return V + (size_t)k * (size_t)n; };
  auto cleanup = [&](int rc) {
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(V);
    (void)hipFree(w);
    (void)hipFree(hdev);
    return rc;
  };
  std::vector<double> s(m + 1), cs(m), sn(m), hcol(m + 2), y(m);
  std::vector<std::vector<double>> H(m + 1, std::vector<double>(m, 0.0));
  int rc = ddm_op_applyscaleadd(ctx, op, -1.0, x, b); // b -= A x
  if (!rc) rc = ddm_combined_apply(ctx, prec, v(0), b); // v0 = M^-1 b
  double nn = 0.0;
  if (!rc) rc = dot_device(ctx, n, op->owner, v(0), v(0), hdev);
  if (!rc) rc = ddm_memcpy_d2h(ctx, &nn, hdev, sizeof(double));
  if (rc) return cleanup(rc);
  double norm = std::sqrt(nn);
  const double def0 = norm;
  res->def0 = def0;
  res->iterations = 0;
  res->converged = 0;
  res->reduction = 1.0;
  res->elapsed_s = 0.0;
  if (hist_host) hist_host[0] = def0;
  if (!(def0 == def0)) return cleanup(fail(ctx, DDM_ENUMERIC, "initial defect is NaN"));
  if (def0 < 1e-30) {
    res->converged = 1;
    return cleanup(DDM_OK);
  }
  const auto t0 = std::chrono::steady_clock::now();
  int j = 0;
  bool conv = false;
  while (j < maxit && !conv && !rc) {
    hipLaunchKernelGGL(k_scal, dim3(G), dim3(WG), 0, ctx->stream, n, 1.0 / norm, v(0));
    std::fill(s.begin(), s.end(), 0.0);
    s[0] = norm;
    int i = 0;
    for (; i < m && j < maxit && !conv; ++i, ++j) {
      rc = ddm_op_apply(ctx, op, v(i), v(i + 1));                 // v[i+1] = A v[i] (temporary)
      if (!rc) rc = ddm_combined_apply(ctx, prec, w, v(i + 1));   // w = M^-1 A v[i]
      for (int k = 0; k <= i && !rc; ++k) {                       // modified Gram-Schmidt
        rc = dot_device(ctx, n, op->owner, v(k), w, hdev + k);
        hipLaunchKernelGGL(k_axpy_negdev, dim3(G), dim3(WG), 0, ctx->stream, n, hdev + k, v(k), w);
      }
      if (!rc) rc = dot_device(ctx, n, op->owner, w, w, hdev + i + 1);
      if (!rc) rc = ddm_memcpy_d2h(ctx, hcol.data(), hdev, sizeof(double) * (size_t)(i + 2));
      if (rc) break;
      for (int k = 0; k <= i; ++k) H[k][i] = hcol[k];
      H[i + 1][i] = std::sqrt(hcol[i + 1]);
      if (std::fabs(H[i + 1][i]) < 1e-80) {
        rc = fail(ctx, DDM_ENUMERIC, "breakdown in GMRes - |w| == 0.0 after %d iterations", j);
        break;
      }
      HIPCHECK(ctx, hipMemcpyAsync(v(i + 1), w, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
      hipLaunchKernelGGL(k_scal, dim3(G), dim3(WG), 0, ctx->stream, n, 1.0 / H[i + 1][i], v(i + 1));
      for (int k = 0; k < i; ++k) gmres_apply_rotation(H[k][i], H[k + 1][i], cs[k], sn[k]);
      gmres_generate_rotation(H[i][i], H[i + 1][i], cs[i], sn[i]);
      gmres_apply_rotation(H[i][i], H[i + 1][i], cs[i], sn[i]);
      gmres_apply_rotation(s[i], s[i + 1], cs[i], sn[i]);
      norm = std::fabs(s[i + 1]);
      res->iterations = j + 1;
      if (hist_host) hist_host[j + 1] = norm;
      if (!(norm == norm)) {
        rc = fail(ctx, DDM_ENUMERIC, "defect is NaN in iteration %d", j + 1);
        break;
      }
      if (norm < def0 * reduction || norm < 1e-30) conv = true;
    }
    if (rc) break;
    // update(w, i, H, s, v): solve the triangular system, w = sum_k y_k v[k]; x += w
    for (int a = i - 1; a >= 0; --a) {
      double t = s[a];
      for (int c = a + 1; c < i; ++c) t -= H[a][c] * y[c];
      y[a] = t / H[a][a];
    }
    HIPCHECK(ctx, hipMemsetAsync(w, 0, sizeof(double) * (size_t)n, ctx->stream));
    for (int a = 0; a < i; ++a) hipLaunchKernelGGL(k_axpy, dim3(G), dim3(WG), 0, ctx->stream, n, y[a], v(a), w);
    hipLaunchKernelGGL(k_axpy, dim3(G), dim3(WG), 0, ctx->stream, n, 1.0, w, x);
    if (!conv && j < maxit) { // restart: b -= A w; v0 = M^-1 b
      rc = ddm_op_applyscaleadd(ctx, op, -1.0, w, b);
      if (!rc) rc = ddm_combined_apply(ctx, prec, v(0), b);
      if (!rc) rc = dot_device(ctx, n, op->owner, v(0), v(0), hdev);
      if (!rc) rc = ddm_memcpy_d2h(ctx, &nn, hdev, sizeof(double));
      norm = std::sqrt(nn);
    }
  }
  if (!rc && hipGetLastError() != hipSuccess) rc = fail(ctx, DDM_EHIP, "kernel launch failed in GMRES");
  (void)hipStreamSynchronize(ctx->stream);
  res->elapsed_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  res->converged = conv ? 1 : 0;
  res->reduction = norm / def0;
  if (!rc && prec->schwarz) {
    int st = 0;
    rc = ddm_ilu0_status(ctx, prec->schwarz->solver, &st);
    if (!rc && st) rc = fail(ctx, DDM_ENUMERIC, "persistent triangular solve timed out waiting for a level (results invalid)");
  }
  return cleanup(rc);
}

// ---- BiCGSTAB ------------------------------------------------------------------------------------
// dune-istl BiCGSTABSolver::apply ([solver] type = bicgstabsolver; DUNE 2.10 solvers.hh, not in the snapshot -- restated in
// oracle/apply_oracle.py::bicgstab_solve): right-preconditioned, two half steps per iteration, the defect norm is tested after each
// half step (hist_host receives both: up to 2 maxit + 1 entries); result.iterations = ceil of the half-step counter, as dune-istl reports.
extern "C" int ddm_bicgstab_solve(ddm_ctx *ctx, ddm_op *op, ddm_combined *prec, double *x, double *b, double reduction, int maxit, double *hist_host,
                                  int32_t *nhist, ddm_solve_result *res)
{
  if (!ctx || !op || !prec || !x || !b || !res) return fail(ctx, DDM_EINVAL, "ddm_bicgstab_solve: bad arguments");
  const int64_t n = op->n;
  const int G = grid_for(n);
  const size_t bytes = sizeof(double) * (size_t)std::max<int64_t>(n, 1);
  double *buf[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; // rt, p, v, y, t
  auto cleanup = [&](int rc) {
    (void)hipStreamSynchronize(ctx->stream);
    for (double *q : buf) (void)hipFree(q);
    return rc;
  };
  for (auto &q : buf)
    if (hipMalloc((void **)&q, bytes) != hipSuccess) return cleanup(fail(ctx, DDM_EHIP, "ddm_bicgstab_solve: allocation failed"));
  double *rt = buf[0], *p = buf[1], *v = buf[2], *y = buf[3], *t = buf[4], *r = b;
  const double EPS = 1e-80;
  const bool verbose = std::getenv("DDM_KRYLOV_VERBOSE") != nullptr;
  int rc = ddm_op_applyscaleadd(ctx, op, -1.0, x, r); // r = b - A x (b is overwritten by the defect, as in dune-istl)
  if (rc) return cleanup(rc);
  HIPCHECK(ctx, hipMemcpyAsync(rt, r, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  double norm = 0.0;
  if ((rc = ddm_norm(ctx, op, r, &norm))) return cleanup(rc);
  const double def0 = norm;
  res->def0 = def0;
  res->iterations = 0;
  res->converged = 0;
  res->reduction = 1.0;
  res->elapsed_s = 0.0;
  int nh = 0;
  if (hist_host) hist_host[nh] = def0;
  ++nh;
  if (!(def0 == def0)) return cleanup(fail(ctx, DDM_ENUMERIC, "initial defect is NaN"));
  if (def0 < 1e-30) {
    res->converged = 1;
    if (nhist) *nhist = nh;
    return cleanup(DDM_OK);
  }
  HIPCHECK(ctx, hipMemsetAsync(p, 0, bytes, ctx->stream));
  HIPCHECK(ctx, hipMemsetAsync(v, 0, bytes, ctx->stream));
  double rho = 1.0, alpha = 1.0, omega = 1.0, rho_new = 0.0, h = 0.0;
  const auto t0 = std::chrono::steady_clock::now();
  double it = 0.5;
  bool conv = false;
  auto record = [&](double nrm) {
    if (hist_host) hist_host[nh] = nrm;
    ++nh;
    res->reduction = nrm / def0;
    return nrm <= def0 * reduction;
  };
  for (; it < maxit && !rc; it += 0.5) {
    if ((rc = ddm_dot(ctx, op, rt, r, &rho_new))) break;
    if (verbose) std::fprintf(stderr, "[ddm bicgstab] it %.1f rho_new %.17g rho %.17g alpha %.17g omega %.17g norm %.17g\n", it, rho_new, rho, alpha, omega, norm);
    if (std::fabs(rho) <= EPS) { rc = fail(ctx, DDM_ENUMERIC, "breakdown in BiCGSTAB - rho %g <= EPSILON after %g iterations", rho, it); break; }
    if (std::fabs(omega) <= EPS) { rc = fail(ctx, DDM_ENUMERIC, "breakdown in BiCGSTAB - omega %g <= EPSILON after %g iterations", omega, it); break; }
    if (it < 1) {
      HIPCHECK(ctx, hipMemcpyAsync(p, r, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    } else {
      const double beta = (rho_new / rho) * (alpha / omega);
      hipLaunchKernelGGL(k_axpy, dim3(G), dim3(WG), 0, ctx->stream, n, -omega, (const double *)v, p); // p = r + beta (p - omega v)
      hipLaunchKernelGGL(k_scal, dim3(G), dim3(WG), 0, ctx->stream, n, beta, p);
      hipLaunchKernelGGL(k_axpy, dim3(G), dim3(WG), 0, ctx->stream, n, 1.0, (const double *)r, p);
    }
    if ((rc = ddm_combined_apply(ctx, prec, y, p))) break;  // y = W^-1 p
    if ((rc = ddm_op_apply(ctx, op, y, v))) break;           // v = A y
    if ((rc = ddm_dot(ctx, op, rt, v, &h))) break;
    if (std::fabs(h) < EPS) { rc = fail(ctx, DDM_ENUMERIC, "abs(h) < EPSILON in BiCGSTAB - abort"); break; }
    alpha = rho_new / h;
    hipLaunchKernelGGL(k_axpy, dim3(G), dim3(WG), 0, ctx->stream, n, alpha, (const double *)y, x);
    hipLaunchKernelGGL(k_axpy, dim3(G), dim3(WG), 0, ctx->stream, n, -alpha, (const double *)v, r);
    if ((rc = ddm_norm(ctx, op, r, &norm))) break;
    if (record(norm)) { conv = true; break; }
    it += 0.5;
    if ((rc = ddm_combined_apply(ctx, prec, y, r))) break;  // y = W^-1 r
    if ((rc = ddm_op_apply(ctx, op, y, t))) break;           // t = A y
    double tt = 0.0, tr = 0.0;
    if ((rc = ddm_dot(ctx, op, t, t, &tt))) break;
    if ((rc = ddm_dot(ctx, op, t, r, &tr))) break;
    omega = tr / tt;
    hipLaunchKernelGGL(k_axpy, dim3(G), dim3(WG), 0, ctx->stream, n, omega, (const double *)y, x);
    hipLaunchKernelGGL(k_axpy, dim3(G), dim3(WG), 0, ctx->stream, n, -omega, (const double *)t, r);
    rho = rho_new;
    if ((rc = ddm_norm(ctx, op, r, &norm))) break;
    if (record(norm)) { conv = true; break; }
  }
  if (rc) return cleanup(rc);
  (void)hipStreamSynchronize(ctx->stream);
  res->elapsed_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  res->iterations = (int32_t)std::ceil(std::min(it, (double)maxit));
  res->converged = conv ? 1 : 0;
  if (nhist) *nhist = nh;
  int st = 0;
  if (prec->schwarz && !ddm_ilu0_status(ctx, prec->schwarz->solver, &st) && st) return cleanup(fail(ctx, DDM_ENUMERIC, "local triangular solve timed out (code %d)", st));
  return cleanup(DDM_OK);
}

#include "geneo.hpp"

// ---- dense host helpers exposed for the CPU tests (host logic of the GenEO Rayleigh-Ritz step) -------------------------
extern "C" int ddm_dense_sym_eig_host(int n, double *V, double *w) { return dense::sym_eig(n, V, w) ? DDM_OK : DDM_ENUMERIC; }
extern "C" int ddm_dense_rayleigh_ritz_host(int p, const double *gA, const double *gC, int keep, double tau, double *mu, double *Y)
{
  return dense::rayleigh_ritz(p, gA, gC, keep, tau, mu, Y);
}

// ---- input synthesis on the host (bench.py / tests: the matrices PDELab's assembler hands to the reference) ----------------------
extern "C" int ddm_synth_q1_matrix(int dim, const int64_t *bshape, const double *ke, const int64_t *eshape, const int64_t *eoff, const double *K,
                                   const uint8_t *inset, const int64_t *loc_of_box, int64_t n, const int64_t *box_index, const uint8_t *dmask,
                                   const double *diag, int64_t *indptr, int32_t *indices, double *data, int nthreads)
{
    if ((dim != 2 && dim != 3) || !bshape || !ke || !eshape || !eoff || !K || !indptr || n < 0 || (indices && !data))
        return fail(nullptr, DDM_EINVAL, "ddm_synth_q1_matrix: bad arguments");
    int64_t nbox = 1;
    for (int d = 0; d < dim; ++d) {
        if (bshape[d] < 1 || eshape[d] < 0 || eoff[d] < 0) return fail(nullptr, DDM_EINVAL, "ddm_synth_q1_matrix: bad box");
        nbox *= bshape[d];
    }
    if (nbox >= INT32_MAX || (!box_index && n != nbox)) return fail(nullptr, DDM_EINVAL, "ddm_synth_q1_matrix: box too large or row count does not match the box");
    synth::Q1Args A{dim, bshape, ke, eshape, eoff, K, inset, loc_of_box, n, box_index, dmask, diag};
    synth::q1_rows(A, indptr, indices, data, nthreads);
    return DDM_OK;
}
