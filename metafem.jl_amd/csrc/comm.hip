// Multi-GPU support (new: the reference is single-GPU): one process per GPU.
//   * slab decomposition along i: a rank owns a contiguous range of node planes, local vectors are
//     [owned, field-major | ghost planes: (field 0 lo, field 0 hi, field 1 lo, ...)];
//   * SpMV halo: one block of `plane_len` doubles per neighbour and field (planes are contiguous, nothing is packed on the
//     RCCL path).  The exchange is split into begin / end so that it runs on a second stream beside the SpMV of the rows
//     that reference no ghost column (krylov_kernels.h: mfem_spmv_halo);
//   * Krylov scalars: one all-reduce(sum, f64) of <= 8 fused device scalars per reduction group -- the scalars never
//     visit the host on the RCCL path.
// Two backends behind the same three operations (all-reduce, neighbour exchange, reverse exchange):
//   0  RCCL over xGMI: ncclSend/ncclRecv inside one group on the halo stream, ncclAllReduce on the context stream;
//   1  host-staged: device -> pinned host -> caller-supplied callbacks (gloo / MPI / anything) -> device.  It exists so that
//      the very same solver code paths can be executed with several ranks where RCCL cannot run (ranks sharing one GPU:
//      RCCL rejects duplicate devices) and for hosts whose MPI is not GPU-aware; it is not the fast path.
#include <rccl/rccl.h>

#include <limits>

#include "krylov.h"

struct mfem_comm_s {
  int backend;
  ncclComm_t comm;
  mfem_comm_host_ops host;
  int rank, world;
  int64_t n_owned_nodes;
  int failed;
  hipStream_t halo_stream;      // RCCL: the exchange runs here, fenced against the context stream by the two events
  hipEvent_t ev_ready, ev_done;
  double* h_stage;              // host-staged: pinned [send_lo | send_hi | recv_lo | recv_hi], each h_block doubles
  size_t h_block;
  double* d_stage;              // device staging of the reverse exchange [from_lo | from_hi], each d_block doubles
  size_t d_block;
  double* pending_x;            // begin() issued for this vector, end() not yet
  // Optional timing of the communication the solver's stream is EXPOSED to (mfem_prof_comm_enable / _read; bench.py prints it per rank): kind 0 = the
  // wait for the halo exchange (RCCL: event pair on the context stream around the wait for the halo stream -- zero-length when the exchange
  // finished beside the interior rows; host callbacks: host clock around staging + callback), kind 1 = one all-reduce of a reduction group (event
  // pair around ncclAllReduce: includes waiting for the slowest rank; host callbacks: host clock).
  int prof_on;
  hipEvent_t* pev;              // [2 * COMM_PROF_PAIRS]
  unsigned char* pkind;         // [COMM_PROF_PAIRS]
  int pused;
  double p_ms[2];
  int64_t p_n[2];
};
#define COMM_PROF_PAIRS 2048

static int comm_prof_flush(mfem_comm_s* c) {
  if (!c->pused) return MFEM_OK;
  MFEM_CHECK_HIP(hipEventSynchronize(c->pev[2 * c->pused - 1]));
  for (int k = 0; k < c->pused; ++k) {
    float ms = 0.f;
    MFEM_CHECK_HIP(hipEventElapsedTime(&ms, c->pev[2 * k], c->pev[2 * k + 1]));
    c->p_ms[c->pkind[k]] += ms;
    c->p_n[c->pkind[k]] += 1;
  }
  c->pused = 0;
  return MFEM_OK;
}
// event bracket on the context stream (RCCL backend); k = -1: not timed
static int comm_prof_begin(mfem_context_s* ctx, mfem_comm_s* c, int kind, int* k) {
  *k = -1;
  if (!c->prof_on || !c->pev) return MFEM_OK;
  if (c->pused == COMM_PROF_PAIRS) {
    int rc = comm_prof_flush(c);
    if (rc) return rc;
  }
  *k = c->pused;
  c->pkind[*k] = (unsigned char)kind;
  MFEM_CHECK_HIP(hipEventRecord(c->pev[2 * *k], ctx->stream));
  return MFEM_OK;
}
static int comm_prof_end(mfem_context_s* ctx, mfem_comm_s* c, int k) {
  if (k < 0) return MFEM_OK;
  MFEM_CHECK_HIP(hipEventRecord(c->pev[2 * k + 1], ctx->stream));
  c->pused = k + 1;
  return MFEM_OK;
}
#include <chrono>
static double comm_now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct CommHostTimer {  // host-callback backend: wall time of the staged operation (its stream synchronisations included)
  mfem_comm_s* c;
  int kind;
  int count;  // 1: this bracket completes one operation (the begin half of an exchange adds time only)
  double t0;
  CommHostTimer(mfem_comm_s* c_, int kind_, int count_ = 1) : c(c_), kind(kind_), count(count_), t0(c_->prof_on ? comm_now_ms() : 0.0) {}
  ~CommHostTimer() {
    if (c->prof_on) {
      c->p_ms[kind] += comm_now_ms() - t0;
      c->p_n[kind] += count;
    }
  }
};

extern "C" int mfem_prof_comm_enable(mfem_context ctx, int on) try {
  MFEM_REQUIRE(ctx, "null ctx");
  mfem_comm_s* c = ctx->comm;
  if (!c) return MFEM_OK;  // nothing to time on one rank
  if (on && c->backend == 0 && !c->pev) {
    mfem_host_alloc_probe();
    hipEvent_t* ev = new hipEvent_t[2 * COMM_PROF_PAIRS]();
    unsigned char* kd = new unsigned char[COMM_PROF_PAIRS]();
    for (int i = 0; i < 2 * COMM_PROF_PAIRS; ++i) {
      const hipError_t e = hipEventCreate(&ev[i]);
      if (e != hipSuccess) {  // nothing half-made stays behind: the timers of the all-reduce / halo path would record on null events
        for (int j = 0; j < i; ++j) (void)hipEventDestroy(ev[j]);
        delete[] ev;
        delete[] kd;
        mfem_set_error("mfem_prof_comm_enable: hipEventCreate -> %s", hipGetErrorString(e));
        return MFEM_ERR_HIP;
      }
    }
    c->pev = ev;
    c->pkind = kd;
  }
  if (!on && c->backend == 0) {
    int rc = comm_prof_flush(c);
    if (rc) return rc;
  }
  c->prof_on = on ? 1 : 0;
  return MFEM_OK;
} MFEM_API_CATCH("mfem_prof_comm_enable")

extern "C" int mfem_prof_comm_read(mfem_context ctx, double* halo_wait_ms, int64_t* halo_waits, double* allreduce_ms, int64_t* allreduces,
                                   int reset) try {
  MFEM_REQUIRE(ctx && halo_wait_ms && halo_waits && allreduce_ms && allreduces, "null argument");
  *halo_wait_ms = *allreduce_ms = 0.0;
  *halo_waits = *allreduces = 0;
  mfem_comm_s* c = ctx->comm;
  if (!c) return MFEM_OK;
  if (c->backend == 0) {
    int rc = comm_prof_flush(c);
    if (rc) return rc;
  }
  *halo_wait_ms = c->p_ms[0]; *halo_waits = c->p_n[0];
  *allreduce_ms = c->p_ms[1]; *allreduces = c->p_n[1];
  if (reset) {
    c->p_ms[0] = c->p_ms[1] = 0.0;
    c->p_n[0] = c->p_n[1] = 0;
  }
  return MFEM_OK;
} MFEM_API_CATCH("mfem_prof_comm_read")

#define MFEM_CHECK_NCCL(expr)                                                              \
  do {                                                                                     \
    ncclResult_t _r = (expr);                                                              \
    if (_r != ncclSuccess) {                                                               \
      mfem_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, ncclGetErrorString(_r)); \
      return MFEM_ERR_COMM;                                                                \
    }                                                                                      \
  } while (0)

// inside ncclGroupStart / ncclGroupEnd: a failing call must not leave the group open (the next collective on the
// communicator would hang); the communicator is marked failed and every later operation on it fails fast
#define MFEM_GROUP_NCCL(c, expr)                                                           \
  do {                                                                                     \
    ncclResult_t _r = (expr);                                                              \
    if (_r != ncclSuccess) {                                                               \
      mfem_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, ncclGetErrorString(_r)); \
      ncclGroupEnd();                                                                      \
      (c)->failed = 1;                                                                     \
      return MFEM_ERR_COMM;                                                                \
    }                                                                                      \
  } while (0)

#define MFEM_COMM_ALIVE(c)                                                          \
  do {                                                                              \
    if ((c)->failed) {                                                              \
      mfem_set_error("communicator is in a failed state (an earlier operation failed)"); \
      return MFEM_ERR_COMM;                                                         \
    }                                                                               \
  } while (0)

extern "C" int mfem_comm_unique_id(void* out128) try {
  MFEM_REQUIRE(out128, "null buffer");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  MFEM_CHECK_NCCL(ncclGetUniqueId(&id));
  memcpy(out128, &id, sizeof(id));
  return MFEM_OK;
} MFEM_API_CATCH("mfem_comm_unique_id")

static mfem_comm_s* comm_new(int backend, int rank, int world) {
  mfem_host_alloc_probe();
  mfem_comm_s* c = new mfem_comm_s();
  memset(c, 0, sizeof(*c));
  c->backend = backend;
  c->rank = rank;
  c->world = world;
  return c;
}

extern "C" int mfem_comm_create(mfem_context ctx, int32_t rank, int32_t world, const void* unique_id128, mfem_comm* out) try {
  MFEM_REQUIRE(ctx && unique_id128 && out, "null argument");
  MFEM_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bad rank/world");
  MFEM_CHECK_HIP(hipSetDevice(ctx->device));
  ncclUniqueId id;
  memcpy(&id, unique_id128, sizeof(id));
  mfem_comm_s* c = comm_new(0, rank, world);
  ncclResult_t r = ncclCommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) {
    mfem_set_error("ncclCommInitRank failed: %s", ncclGetErrorString(r));
    delete c;
    return MFEM_ERR_COMM;
  }
  // the exchange is a few hundred kilobytes per neighbour: give its stream priority so that its kernels are placed before
  // the (persistent, chip-filling) SpMV workgroups of the interior rows that become ready at the same moment
  int prio_lo = 0, prio_hi = 0;
  hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
  if (hipStreamCreateWithPriority(&c->halo_stream, hipStreamNonBlocking, prio_hi) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming) != hipSuccess) {
    mfem_set_error("could not create the halo stream / events");
    ncclCommDestroy(c->comm);
    delete c;
    return MFEM_ERR_HIP;
  }
  *out = c;
  return MFEM_OK;
} MFEM_API_CATCH("mfem_comm_create")

extern "C" int mfem_comm_create_host(mfem_context ctx, int32_t rank, int32_t world, const mfem_comm_host_ops* ops,
                                     mfem_comm* out) try {
  MFEM_REQUIRE(ctx && ops && out, "null argument");
  MFEM_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bad rank/world");
  MFEM_REQUIRE(ops->allreduce_sum && ops->neighbour_exchange, "both callbacks are required");
  mfem_comm_s* c = comm_new(1, rank, world);
  c->host = *ops;
  *out = c;
  return MFEM_OK;
} MFEM_API_CATCH("mfem_comm_create_host")

extern "C" int mfem_comm_destroy(mfem_comm c) try {
  if (!c) return MFEM_OK;
  if (c->backend == 0) {
    if (c->halo_stream) hipStreamSynchronize(c->halo_stream);
    ncclCommDestroy(c->comm);
    if (c->ev_ready) hipEventDestroy(c->ev_ready);
    if (c->ev_done) hipEventDestroy(c->ev_done);
    if (c->halo_stream) hipStreamDestroy(c->halo_stream);
  }
  if (c->h_stage) hipHostFree(c->h_stage);
  if (c->d_stage) hipFree(c->d_stage);
  if (c->pev) {
    for (int i = 0; i < 2 * COMM_PROF_PAIRS; ++i)
      if (c->pev[i]) hipEventDestroy(c->pev[i]);
    delete[] c->pev;
    delete[] c->pkind;
  }
  delete c;
  return MFEM_OK;
} MFEM_API_CATCH("mfem_comm_destroy")

extern "C" int mfem_context_set_comm(mfem_context ctx, mfem_comm c, int64_t n_owned_nodes, int64_t plane_len,
                                     int32_t n_fields) try {
  MFEM_REQUIRE(ctx, "null ctx");
  if (!c) {
    ctx->comm = nullptr;
    ctx->halo_plane_len = 0;
    ctx->halo_fields = 0;
    return MFEM_OK;
  }
  MFEM_REQUIRE(plane_len > 0 && n_fields >= 1 && n_owned_nodes >= plane_len, "bad halo geometry");
  c->n_owned_nodes = n_owned_nodes;
  ctx->comm = c;
  ctx->halo_plane_len = plane_len;
  ctx->halo_fields = n_fields;
  return MFEM_OK;
} MFEM_API_CATCH("mfem_context_set_comm")

int mfem_comm_world(const mfem_context_s* ctx) { return ctx->comm ? ctx->comm->world : 1; }
// Can the communication of a Krylov cycle be recorded into a hipGraph?  RCCL calls on the context stream and the fork / join of the halo stream through
// events are capturable; the host-callback transport synchronises the stream and calls the host, the exposed-communication timers read events: neither is.
bool mfem_comm_capturable(const mfem_context_s* ctx) { return ctx->comm && ctx->comm->backend == 0 && !ctx->comm->prof_on && !ctx->comm->failed; }
int mfem_comm_rank(const mfem_context_s* ctx) { return ctx->comm ? ctx->comm->rank : 0; }
int64_t mfem_comm_owned_nodes(const mfem_context_s* ctx) { return ctx->comm ? ctx->comm->n_owned_nodes : 0; }

static int stage_reserve(mfem_context_s* ctx, mfem_comm_s* c, size_t block_doubles, bool device_too) {
  if (c->backend == 1 && c->h_block < block_doubles) {
    MFEM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    if (c->h_stage) MFEM_CHECK_HIP(hipHostFree(c->h_stage));
    c->h_stage = nullptr;
    c->h_block = 0;
    MFEM_CHECK_HIP(hipHostMalloc(&c->h_stage, sizeof(double) * 4 * block_doubles));
    c->h_block = block_doubles;
  }
  if (device_too && c->d_block < block_doubles) {
    MFEM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    if (c->d_stage) MFEM_CHECK_HIP(hipFree(c->d_stage));
    c->d_stage = nullptr;
    c->d_block = 0;
    MFEM_CHECK_HIP(hipMalloc(&c->d_stage, sizeof(double) * 2 * block_doubles));
    c->d_block = block_doubles;
  }
  return MFEM_OK;
}

int mfem_comm_allreduce(mfem_context_s* ctx, double* dev, int count) {
  mfem_comm_s* c = ctx->comm;
  if (!c || count <= 0) return MFEM_OK;
  MFEM_COMM_ALIVE(c);
  if (c->backend == 0) {
    int k = -1;
    int rc = comm_prof_begin(ctx, c, 1, &k);
    if (rc) return rc;
    MFEM_CHECK_NCCL(ncclAllReduce(dev, dev, (size_t)count, ncclDouble, ncclSum, c->comm, ctx->stream));
    return comm_prof_end(ctx, c, k);
  }
  // host-staged: the scalars visit the host (one stream sync per reduction group)
  CommHostTimer timer(c, 1);
  int rc = stage_reserve(ctx, c, (size_t)(count > 64 ? count : 64), false);
  if (rc) return rc;
  MFEM_CHECK_HIP(hipMemcpyAsync(c->h_stage, dev, sizeof(double) * count, hipMemcpyDeviceToHost, ctx->stream));
  MFEM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  if (c->world > 1) {
    const int r = c->host.allreduce_sum(c->host.user, c->h_stage, count);
    if (r != 0) {
      mfem_set_error("host all-reduce callback returned %d", r);
      c->failed = 1;
      return MFEM_ERR_COMM;
    }
  }
  MFEM_CHECK_HIP(hipMemcpyAsync(dev, c->h_stage, sizeof(double) * count, hipMemcpyHostToDevice, ctx->stream));
  MFEM_CHECK_HIP(hipStreamSynchronize(ctx->stream));  // the staging buffer is reused by the next operation
  return MFEM_OK;
}

__global__ __launch_bounds__(MFEM_BLOCK) void k_fill_nan(int64_t n, double* __restrict__ x) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const double q = __longlong_as_double(0x7ff8000000000000ll);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride) x[i] = q;
}

__global__ __launch_bounds__(MFEM_BLOCK) void k_add_block(int64_t n, const double* __restrict__ src, double* __restrict__ dst) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride) dst[i] += src[i];
}

// Start filling the ghost blocks of a local vector from the neighbours' boundary planes.  Until mfem_comm_halo_end the ghost
// entries of x must not be read and its first / last plane_len owned entries per field must not be written.
int mfem_comm_halo_begin(mfem_context_s* ctx, double* x) {
  mfem_comm_s* c = ctx->comm;
  if (!c || c->world == 1) return MFEM_OK;
  MFEM_COMM_ALIVE(c);
  MFEM_REQUIRE(!c->pending_x, "a halo exchange is already in flight on this communicator");
  const int64_t PL = ctx->halo_plane_len, NO = c->n_owned_nodes;
  const int F = ctx->halo_fields;
  double* ghost = x + (int64_t)F * NO;
  const bool lo = c->rank > 0, hi = c->rank < c->world - 1;
  if (c->backend == 0) {
    MFEM_CHECK_HIP(hipEventRecord(c->ev_ready, ctx->stream));
    MFEM_CHECK_HIP(hipStreamWaitEvent(c->halo_stream, c->ev_ready, 0));
    MFEM_CHECK_NCCL(ncclGroupStart());
    for (int f = 0; f < F; ++f) {
      double* own = x + (int64_t)f * NO;
      if (lo) {
        MFEM_GROUP_NCCL(c, ncclSend(own, (size_t)PL, ncclDouble, c->rank - 1, c->comm, c->halo_stream));
        MFEM_GROUP_NCCL(c, ncclRecv(ghost + (int64_t)(2 * f + 0) * PL, (size_t)PL, ncclDouble, c->rank - 1, c->comm, c->halo_stream));
      }
      if (hi) {
        MFEM_GROUP_NCCL(c, ncclSend(own + NO - PL, (size_t)PL, ncclDouble, c->rank + 1, c->comm, c->halo_stream));
        MFEM_GROUP_NCCL(c, ncclRecv(ghost + (int64_t)(2 * f + 1) * PL, (size_t)PL, ncclDouble, c->rank + 1, c->comm, c->halo_stream));
      }
    }
    ncclResult_t r = ncclGroupEnd();
    if (r != ncclSuccess) {
      mfem_set_error("ncclGroupEnd (halo): %s", ncclGetErrorString(r));
      c->failed = 1;
      return MFEM_ERR_COMM;
    }
    MFEM_CHECK_HIP(hipEventRecord(c->ev_done, c->halo_stream));
    c->pending_x = x;
    return MFEM_OK;
  }
  CommHostTimer timer(c, 0, 0);  // (host callbacks: nothing overlaps, the whole exchange is exposed)
  const size_t blk = (size_t)F * (size_t)PL;
  int rc = stage_reserve(ctx, c, blk > 64 ? blk : 64, false);
  if (rc) return rc;
  double* s_lo = c->h_stage;
  double* s_hi = c->h_stage + c->h_block;
  double* r_lo = c->h_stage + 2 * c->h_block;
  double* r_hi = c->h_stage + 3 * c->h_block;
  for (int f = 0; f < F; ++f) {
    const double* own = x + (int64_t)f * NO;
    if (lo) MFEM_CHECK_HIP(hipMemcpyAsync(s_lo + (size_t)f * PL, own, sizeof(double) * PL, hipMemcpyDeviceToHost, ctx->stream));
    if (hi) MFEM_CHECK_HIP(hipMemcpyAsync(s_hi + (size_t)f * PL, own + NO - PL, sizeof(double) * PL, hipMemcpyDeviceToHost, ctx->stream));
  }
  if (c->host.flags & MFEM_COMM_HOST_POISON_GHOSTS) {
    // test aid: between begin and end the ghost entries are NaN, so a kernel that reads them too early is found out
    hipLaunchKernelGGL(k_fill_nan, dim3(mfem_grid_for(2 * (int64_t)blk, MFEM_BLOCK, 64)), dim3(MFEM_BLOCK), 0, ctx->stream,
                       2 * (int64_t)blk, ghost);
    MFEM_CHECK_LAUNCH();
  }
  MFEM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  const int r = c->host.neighbour_exchange(c->host.user, lo ? s_lo : nullptr, lo ? r_lo : nullptr, hi ? s_hi : nullptr,
                                           hi ? r_hi : nullptr, (int64_t)blk);
  if (r != 0) {
    mfem_set_error("host neighbour-exchange callback returned %d", r);
    c->failed = 1;
    return MFEM_ERR_COMM;
  }
  c->pending_x = x;
  return MFEM_OK;
}

int mfem_comm_halo_end(mfem_context_s* ctx) {
  mfem_comm_s* c = ctx->comm;
  if (!c || c->world == 1 || !c->pending_x) return MFEM_OK;
  double* x = c->pending_x;
  c->pending_x = nullptr;
  if (c->backend == 0) {
    int k = -1;
    int rc = comm_prof_begin(ctx, c, 0, &k);
    if (rc) return rc;
    MFEM_CHECK_HIP(hipStreamWaitEvent(ctx->stream, c->ev_done, 0));
    return comm_prof_end(ctx, c, k);
  }
  CommHostTimer timer(c, 0);
  const int64_t PL = ctx->halo_plane_len, NO = c->n_owned_nodes;
  const int F = ctx->halo_fields;
  double* ghost = x + (int64_t)F * NO;
  const bool lo = c->rank > 0, hi = c->rank < c->world - 1;
  const double* r_lo = c->h_stage + 2 * c->h_block;
  const double* r_hi = c->h_stage + 3 * c->h_block;
  for (int f = 0; f < F; ++f) {
    if (lo)
      MFEM_CHECK_HIP(hipMemcpyAsync(ghost + (int64_t)(2 * f + 0) * PL, r_lo + (size_t)f * PL, sizeof(double) * PL, hipMemcpyHostToDevice, ctx->stream));
    if (hi)
      MFEM_CHECK_HIP(hipMemcpyAsync(ghost + (int64_t)(2 * f + 1) * PL, r_hi + (size_t)f * PL, sizeof(double) * PL, hipMemcpyHostToDevice, ctx->stream));
  }
  MFEM_CHECK_HIP(hipStreamSynchronize(ctx->stream));  // the pinned staging buffer is reused by the next operation
  return MFEM_OK;
}

int mfem_comm_halo(mfem_context_s* ctx, double* x) {
  int rc = mfem_comm_halo_begin(ctx, x);
  if (rc) return rc;
  return mfem_comm_halo_end(ctx);
}

// Reverse exchange: what this rank accumulated in its ghost blocks (contributions to entries other ranks own) is sent to
// the owners and ADDED to their first / last plane_len owned entries per field.  Used for column sums of a slab matrix
// (Jacobi2_By_Colomn): the rows that hit a boundary column live on two ranks.
int mfem_comm_halo_reduce(mfem_context_s* ctx, double* x) {
  mfem_comm_s* c = ctx->comm;
  if (!c || c->world == 1) return MFEM_OK;
  MFEM_COMM_ALIVE(c);
  MFEM_REQUIRE(!c->pending_x, "a halo exchange is in flight on this communicator");
  const int64_t PL = ctx->halo_plane_len, NO = c->n_owned_nodes;
  const int F = ctx->halo_fields;
  double* ghost = x + (int64_t)F * NO;
  const bool lo = c->rank > 0, hi = c->rank < c->world - 1;
  const size_t blk = (size_t)F * (size_t)PL;
  int rc = stage_reserve(ctx, c, blk > 64 ? blk : 64, true);
  if (rc) return rc;
  double* from_lo = c->d_stage;
  double* from_hi = c->d_stage + c->d_block;
  if (c->backend == 0) {
    MFEM_CHECK_NCCL(ncclGroupStart());
    for (int f = 0; f < F; ++f) {
      if (lo) {
        MFEM_GROUP_NCCL(c, ncclSend(ghost + (int64_t)(2 * f + 0) * PL, (size_t)PL, ncclDouble, c->rank - 1, c->comm, ctx->stream));
        MFEM_GROUP_NCCL(c, ncclRecv(from_lo + (size_t)f * PL, (size_t)PL, ncclDouble, c->rank - 1, c->comm, ctx->stream));
      }
      if (hi) {
        MFEM_GROUP_NCCL(c, ncclSend(ghost + (int64_t)(2 * f + 1) * PL, (size_t)PL, ncclDouble, c->rank + 1, c->comm, ctx->stream));
        MFEM_GROUP_NCCL(c, ncclRecv(from_hi + (size_t)f * PL, (size_t)PL, ncclDouble, c->rank + 1, c->comm, ctx->stream));
      }
    }
    ncclResult_t r = ncclGroupEnd();
    if (r != ncclSuccess) {
      mfem_set_error("ncclGroupEnd (reverse halo): %s", ncclGetErrorString(r));
      c->failed = 1;
      return MFEM_ERR_COMM;
    }
  } else {
    double* s_lo = c->h_stage;
    double* s_hi = c->h_stage + c->h_block;
    double* r_lo = c->h_stage + 2 * c->h_block;
    double* r_hi = c->h_stage + 3 * c->h_block;
    for (int f = 0; f < F; ++f) {
      if (lo) MFEM_CHECK_HIP(hipMemcpyAsync(s_lo + (size_t)f * PL, ghost + (int64_t)(2 * f + 0) * PL, sizeof(double) * PL, hipMemcpyDeviceToHost, ctx->stream));
      if (hi) MFEM_CHECK_HIP(hipMemcpyAsync(s_hi + (size_t)f * PL, ghost + (int64_t)(2 * f + 1) * PL, sizeof(double) * PL, hipMemcpyDeviceToHost, ctx->stream));
    }
    MFEM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    const int r = c->host.neighbour_exchange(c->host.user, lo ? s_lo : nullptr, lo ? r_lo : nullptr, hi ? s_hi : nullptr,
                                             hi ? r_hi : nullptr, (int64_t)blk);
    if (r != 0) {
      mfem_set_error("host neighbour-exchange callback returned %d", r);
      c->failed = 1;
      return MFEM_ERR_COMM;
    }
    if (lo) MFEM_CHECK_HIP(hipMemcpyAsync(from_lo, r_lo, sizeof(double) * blk, hipMemcpyHostToDevice, ctx->stream));
    if (hi) MFEM_CHECK_HIP(hipMemcpyAsync(from_hi, r_hi, sizeof(double) * blk, hipMemcpyHostToDevice, ctx->stream));
  }
  for (int f = 0; f < F; ++f) {
    double* own = x + (int64_t)f * NO;
    const int g = mfem_grid_for(PL, MFEM_BLOCK, 256);
    if (lo) hipLaunchKernelGGL(k_add_block, dim3(g), dim3(MFEM_BLOCK), 0, ctx->stream, PL, from_lo + (size_t)f * PL, own);
    if (hi) hipLaunchKernelGGL(k_add_block, dim3(g), dim3(MFEM_BLOCK), 0, ctx->stream, PL, from_hi + (size_t)f * PL, own + NO - PL);
  }
  MFEM_CHECK_LAUNCH();
  if (c->backend == 1) MFEM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return MFEM_OK;
}

extern "C" int mfem_allreduce_sum(mfem_context ctx, double* dev_scalars, int32_t count) try {
  MFEM_REQUIRE(ctx && dev_scalars && count >= 0, "bad argument");
  MFEM_REQUIRE(ctx->comm, "no communicator attached (mfem_context_set_comm)");
  return mfem_comm_allreduce(ctx, dev_scalars, count);
} MFEM_API_CATCH("mfem_allreduce_sum")

extern "C" int mfem_halo_exchange(mfem_context ctx, double* x_local) try {
  MFEM_REQUIRE(ctx && x_local, "bad argument");
  MFEM_REQUIRE(ctx->comm, "no communicator attached (mfem_context_set_comm)");
  return mfem_comm_halo(ctx, x_local);
} MFEM_API_CATCH("mfem_halo_exchange")

extern "C" int mfem_halo_reduce(mfem_context ctx, double* x_local) try {
  MFEM_REQUIRE(ctx && x_local, "bad argument");
  MFEM_REQUIRE(ctx->comm, "no communicator attached (mfem_context_set_comm)");
  return mfem_comm_halo_reduce(ctx, x_local);
} MFEM_API_CATCH("mfem_halo_reduce")

// ---- diagnostic: the RCCL choreography of one overlapped SpMV + reduction group, on a ring ------------------------------------
// What the solver issues per iteration with the RCCL transport -- grouped ncclSend / ncclRecv on the high-priority halo stream fenced
// by two events, a kernel on the context stream beside it, the stream wait, then ncclAllReduce on the context stream with the SAME
// communicator -- but with ring neighbours (rank + 1, rank - 1 modulo world), so that a one-rank communicator exercises every call
// too (a self send / receive inside a group).  Returns MFEM_OK when every received entry and the reduced scalars are what the ring
// must deliver.  mfem_debug_* : not part of the drop-in surface.
__global__ __launch_bounds__(MFEM_BLOCK) void k_ring_fill(int64_t n, double base, double* __restrict__ s) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride) s[i] = base + (double)i;
}
__global__ __launch_bounds__(MFEM_BLOCK) void k_ring_check(int64_t n, double base, const double* __restrict__ r, int32_t* __restrict__ bad) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride)
    if (r[i] != base + (double)i) atomicAdd(bad, 1);
}

extern "C" int mfem_debug_comm_selftest(mfem_context ctx, int64_t count, int32_t rounds) try {
  MFEM_REQUIRE(ctx && count > 0 && rounds > 0, "bad argument");
  mfem_comm_s* c = ctx->comm;
  MFEM_REQUIRE(c && c->backend == 0, "needs an attached RCCL communicator (mfem_comm_create + mfem_context_set_comm)");
  MFEM_COMM_ALIVE(c);
  MFEM_REQUIRE(!c->pending_x, "a halo exchange is in flight on this communicator");
  double *snd = nullptr, *rcv = nullptr;
  MFEM_CHECK_HIP(hipMalloc(&snd, sizeof(double) * (size_t)count));
  if (hipMalloc(&rcv, sizeof(double) * (size_t)count) != hipSuccess) {
    hipFree(snd);
    mfem_set_error("hipMalloc failed");
    return MFEM_ERR_HIP;
  }
  int32_t* bad = ctx->d_flags + 10;
  double* sc = ctx->d_scalars + (MFEM_NSCALARS - 8);
  const int next = (c->rank + 1) % c->world, prev = (c->rank + c->world - 1) % c->world;
  const int g = mfem_grid_for(count, MFEM_BLOCK, 1024);
  int rc = MFEM_OK;
  auto run = [&]() -> int {
    MFEM_CHECK_HIP(hipMemsetAsync(bad, 0, sizeof(int32_t), ctx->stream));
    for (int r = 0; r < rounds; ++r) {
      hipLaunchKernelGGL(k_ring_fill, dim3(g), dim3(MFEM_BLOCK), 0, ctx->stream, count, 1e6 * c->rank + 1e3 * r, snd);
      MFEM_CHECK_LAUNCH();
      MFEM_CHECK_HIP(hipEventRecord(c->ev_ready, ctx->stream));
      MFEM_CHECK_HIP(hipStreamWaitEvent(c->halo_stream, c->ev_ready, 0));
      MFEM_CHECK_NCCL(ncclGroupStart());
      MFEM_GROUP_NCCL(c, ncclSend(snd, (size_t)count, ncclDouble, next, c->comm, c->halo_stream));
      MFEM_GROUP_NCCL(c, ncclRecv(rcv, (size_t)count, ncclDouble, prev, c->comm, c->halo_stream));
      ncclResult_t e = ncclGroupEnd();
      if (e != ncclSuccess) {
        mfem_set_error("ncclGroupEnd (ring): %s", ncclGetErrorString(e));
        c->failed = 1;
        return MFEM_ERR_COMM;
      }
      MFEM_CHECK_HIP(hipEventRecord(c->ev_done, c->halo_stream));
      // "interior rows": work on the context stream that does not depend on the exchange
      hipLaunchKernelGGL(k_ring_fill, dim3(1), dim3(MFEM_BLOCK), 0, ctx->stream, (int64_t)3, (double)(c->rank + 1), sc);
      MFEM_CHECK_LAUNCH();
      MFEM_CHECK_HIP(hipStreamWaitEvent(ctx->stream, c->ev_done, 0));
      hipLaunchKernelGGL(k_ring_check, dim3(g), dim3(MFEM_BLOCK), 0, ctx->stream, count, 1e6 * prev + 1e3 * r, rcv, bad);
      MFEM_CHECK_LAUNCH();
      MFEM_CHECK_NCCL(ncclAllReduce(sc, sc, 3, ncclDouble, ncclSum, c->comm, ctx->stream));
    }
    MFEM_CHECK_HIP(hipMemcpyAsync(ctx->h_flags + 10, bad, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    MFEM_CHECK_HIP(hipMemcpyAsync(ctx->h_scalars + (MFEM_NSCALARS - 8), sc, sizeof(double) * 3, hipMemcpyDeviceToHost, ctx->stream));
    MFEM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    MFEM_CHECK_HIP(hipStreamSynchronize(c->halo_stream));
    if (ctx->h_flags[10] != 0) {
      mfem_set_error("ring exchange delivered %d wrong entries", ctx->h_flags[10]);
      return MFEM_ERR_COMM;
    }
    const double W = (double)c->world, tri = W * (W + 1.0) / 2.0;  // sum over ranks of (rank + 1 + i), i = 0..2
    for (int i = 0; i < 3; ++i)
      if (ctx->h_scalars[MFEM_NSCALARS - 8 + i] != tri + W * i) {
        mfem_set_error("all-reduce delivered %g for scalar %d, expected %g", ctx->h_scalars[MFEM_NSCALARS - 8 + i], i, tri + W * i);
        return MFEM_ERR_COMM;
      }
    return MFEM_OK;
  };
  rc = run();
  hipStreamSynchronize(ctx->stream);
  hipStreamSynchronize(c->halo_stream);
  hipFree(snd);
  hipFree(rcv);
  return rc;
} MFEM_API_CATCH("mfem_debug_comm_selftest")
