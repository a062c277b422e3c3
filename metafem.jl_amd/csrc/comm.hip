// Multi-GPU support (new: the reference is single-GPU): one process per GPU, RCCL over xGMI.
//   * slab decomposition along i: a rank owns a contiguous range of node planes, local vectors are
//     [owned, field-major | ghost planes: (field 0 lo, field 0 hi, field 1 lo, ...)];
//   * SpMV halo: one node plane per neighbour and field, ncclSend/ncclRecv inside one group on the
//     context stream (point-to-point over a single xGMI link; planes are contiguous, no packing);
//   * Krylov scalars: one ncclAllReduce(sum, f64) of <= 8 fused device scalars per reduction group --
//     the scalars never visit the host.
#include <rccl/rccl.h>

#include "krylov.h"

struct mfem_comm_s {
  ncclComm_t comm;
  int rank, world;
  int64_t n_owned_nodes;
};

#define MFEM_CHECK_NCCL(expr)                                                              \
  do {                                                                                     \
    ncclResult_t _r = (expr);                                                              \
    if (_r != ncclSuccess) {                                                               \
      mfem_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, ncclGetErrorString(_r)); \
      return MFEM_ERR_COMM;                                                                \
    }                                                                                      \
  } while (0)

extern "C" int mfem_comm_unique_id(void* out128) {
  MFEM_REQUIRE(out128, "null buffer");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  MFEM_CHECK_NCCL(ncclGetUniqueId(&id));
  memcpy(out128, &id, sizeof(id));
  return MFEM_OK;
}

extern "C" int mfem_comm_create(mfem_context ctx, int32_t rank, int32_t world, const void* unique_id128, mfem_comm* out) {
  MFEM_REQUIRE(ctx && unique_id128 && out, "null argument");
  MFEM_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bad rank/world");
  MFEM_CHECK_HIP(hipSetDevice(ctx->device));
  ncclUniqueId id;
  memcpy(&id, unique_id128, sizeof(id));
  mfem_comm_s* c = new mfem_comm_s();
  c->rank = rank;
  c->world = world;
  c->n_owned_nodes = 0;
  ncclResult_t r = ncclCommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) {
    mfem_set_error("ncclCommInitRank failed: %s", ncclGetErrorString(r));
    delete c;
    return MFEM_ERR_COMM;
  }
  *out = c;
  return MFEM_OK;
}

extern "C" int mfem_comm_destroy(mfem_comm c) {
  if (!c) return MFEM_OK;
  ncclCommDestroy(c->comm);
  delete c;
  return MFEM_OK;
}

extern "C" int mfem_context_set_comm(mfem_context ctx, mfem_comm c, int64_t n_owned_nodes, int64_t plane_len,
                                     int32_t n_fields) {
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
}

int mfem_comm_allreduce(mfem_context_s* ctx, double* dev, int count) {
  if (!ctx->comm) return MFEM_OK;
  MFEM_CHECK_NCCL(ncclAllReduce(dev, dev, (size_t)count, ncclDouble, ncclSum, ctx->comm->comm, ctx->stream));
  return MFEM_OK;
}

// Fill the ghost planes of a local vector from the neighbours' boundary planes.
int mfem_comm_halo(mfem_context_s* ctx, double* x) {
  mfem_comm_s* c = ctx->comm;
  if (!c || c->world == 1) return MFEM_OK;
  const int64_t PL = ctx->halo_plane_len, NO = c->n_owned_nodes;
  const int F = ctx->halo_fields;
  double* ghost = x + (int64_t)F * NO;
  MFEM_CHECK_NCCL(ncclGroupStart());
  for (int f = 0; f < F; ++f) {
    double* own = x + (int64_t)f * NO;
    if (c->rank > 0) {
      MFEM_CHECK_NCCL(ncclSend(own, (size_t)PL, ncclDouble, c->rank - 1, c->comm, ctx->stream));
      MFEM_CHECK_NCCL(ncclRecv(ghost + (int64_t)(2 * f + 0) * PL, (size_t)PL, ncclDouble, c->rank - 1, c->comm, ctx->stream));
    }
    if (c->rank < c->world - 1) {
      MFEM_CHECK_NCCL(ncclSend(own + NO - PL, (size_t)PL, ncclDouble, c->rank + 1, c->comm, ctx->stream));
      MFEM_CHECK_NCCL(ncclRecv(ghost + (int64_t)(2 * f + 1) * PL, (size_t)PL, ncclDouble, c->rank + 1, c->comm, ctx->stream));
    }
  }
  MFEM_CHECK_NCCL(ncclGroupEnd());
  return MFEM_OK;
}

extern "C" int mfem_allreduce_sum(mfem_context ctx, double* dev_scalars, int32_t count) {
  MFEM_REQUIRE(ctx && dev_scalars && count >= 0, "bad argument");
  MFEM_REQUIRE(ctx->comm, "no communicator attached (mfem_context_set_comm)");
  return mfem_comm_allreduce(ctx, dev_scalars, count);
}

extern "C" int mfem_halo_exchange(mfem_context ctx, double* x_local) {
  MFEM_REQUIRE(ctx && x_local, "bad argument");
  MFEM_REQUIRE(ctx->comm, "no communicator attached (mfem_context_set_comm)");
  return mfem_comm_halo(ctx, x_local);
}
