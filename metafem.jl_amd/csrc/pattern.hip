// Sparse pattern for UNSTRUCTURED connectivity: the replacement of assemble_SparseID! + assemble_KIJ! +
// sort_CUSPARSE_COO! + generate_J_ptr (reference solver/03_GlobalAssembly.jl:77-168, misc/04_GPU_Utils.jl:87-118,
// misc/06_GPU_Dict.jl).  The reference inserts the itp^2*nel (cp_i, cp_j) keys into a chained GPU hash table
// (atomicCAS contention, 32 B per slot, table sized from the duplicate-inclusive key count), runs itp^2 lookup
// launches to fill sparse_IDs_by_el, then sorts the COO with CUSPARSE and keeps values in hash order plus a
// permutation.  Here: one radix sort of the packed 64-bit keys (rocPRIM via hipCUB), run-length unique, row
// pointers and slot ids by binary search in the sorted unique keys -- the result IS row-sorted CSR
// (K_val_ids = identity), deterministic, and the slot table addresses it directly.
#include <hipcub/hipcub.hpp>

#include "common.h"

int mfem_csr_plan(mfem_context_s* ctx, mfem_csr_s* A);

__global__ __launch_bounds__(MFEM_BLOCK) void k_make_keys(int itp, int64_t nel, const int32_t* __restrict__ cp, int base,
                                                            uint64_t* __restrict__ keys) {
  const int64_t total = (int64_t)itp * itp * nel;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t e = t / (itp * itp);
    const int p = (int)(t - e * itp * itp), a = p % itp, b = p / itp;  // [a, b, e] column-major
    const uint64_t i = (uint64_t)(uint32_t)(cp[a + (int64_t)itp * e] - base);
    const uint64_t j = (uint64_t)(uint32_t)(cp[b + (int64_t)itp * e] - base);
    keys[t] = (i << 32) | j;  // I32I32_To_UI64 (06_GPU_Dict.jl:233)
  }
}

__device__ __forceinline__ int64_t lower_bound_u64(const uint64_t* __restrict__ a, int64_t n, uint64_t key) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (a[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// rowptr1[i] = first unique key with row >= i
__global__ __launch_bounds__(MFEM_BLOCK) void k_rowptr1(int64_t ncp, const uint64_t* __restrict__ ukeys, int64_t U,
                                                          int64_t* __restrict__ rowptr1) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i <= ncp) rowptr1[i] = lower_bound_u64(ukeys, U, (uint64_t)i << 32);
}

// field-major expansion: row (f,i) = F blocks of the node's unique neighbours
__global__ __launch_bounds__(MFEM_BLOCK) void k_expand(int64_t ncp, int F, int64_t U, const int64_t* __restrict__ rowptr1,
                                                         const uint64_t* __restrict__ ukeys, int64_t* __restrict__ rowptr,
                                                         int32_t* __restrict__ col) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < ncp; i += stride) {
    const int64_t lo = rowptr1[i], len = rowptr1[i + 1] - lo;
    for (int f = 0; f < F; ++f) {
      const int64_t start = (int64_t)f * F * U + (int64_t)F * lo;
      rowptr[(int64_t)f * ncp + i] = start;
      for (int g = 0; g < F; ++g)
        for (int64_t j = 0; j < len; ++j)
          col[start + g * len + j] = (int32_t)((int64_t)g * ncp + (int64_t)(ukeys[lo + j] & 0xFFFFFFFFull));
    }
    if (i == ncp - 1) rowptr[(int64_t)F * ncp] = (int64_t)F * F * U;
  }
}

// slots[u][a,b,e] for block u = f*F + g
__global__ __launch_bounds__(MFEM_BLOCK) void k_slots(int itp, int64_t nel, int64_t ncp, int F, int64_t U,
                                                        const int32_t* __restrict__ cp, int base,
                                                        const uint64_t* __restrict__ ukeys, const int64_t* __restrict__ rowptr1,
                                                        int out_base, int32_t* __restrict__ slots) {
  const int64_t total = (int64_t)itp * itp * nel;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t e = t / (itp * itp);
    const int p = (int)(t - e * itp * itp), a = p % itp, b = p / itp;
    const uint64_t i = (uint64_t)(uint32_t)(cp[a + (int64_t)itp * e] - base);
    const uint64_t j = (uint64_t)(uint32_t)(cp[b + (int64_t)itp * e] - base);
    const int64_t pos = lower_bound_u64(ukeys, U, (i << 32) | j);
    const int64_t lo = rowptr1[i], len = rowptr1[i + 1] - lo;
    for (int f = 0; f < F; ++f)
      for (int g = 0; g < F; ++g)
        slots[(int64_t)(f * F + g) * total + t] =
            (int32_t)((int64_t)f * F * U + (int64_t)F * lo + (int64_t)g * len + (pos - lo) + out_base);
  }
}

extern "C" int mfem_pattern_build(mfem_context ctx, int32_t itp, int64_t nel, int64_t ncp, const int32_t* controlpoint_IDs,
                                  int32_t index_base, int32_t n_fields, mfem_csr* out, int32_t* sparse_IDs_by_el) try {
  MFEM_REQUIRE(ctx && controlpoint_IDs && out, "null argument");
  MFEM_REQUIRE(itp > 0 && nel > 0 && ncp > 0, "sizes must be positive");
  MFEM_REQUIRE(index_base == 0 || index_base == 1, "index_base must be 0 or 1");
  MFEM_REQUIRE(n_fields >= 1 && n_fields <= 8, "n_fields out of range");
  MFEM_REQUIRE((int64_t)n_fields * ncp < ((int64_t)1 << 31), "global DOF ids must fit int32 (FEM_Int)");
  const int64_t nkeys = (int64_t)itp * itp * nel;
  uint64_t *keys = nullptr, *sorted = nullptr, *ukeys = nullptr;
  int64_t* d_num = nullptr;
  void* tmp = nullptr;
  int rc = MFEM_OK;
  mfem_csr_s* A = nullptr;
  int64_t* rowptr1 = nullptr;
#define PB_CHECK(expr)                                                                     \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      mfem_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      rc = MFEM_ERR_HIP;                                                                   \
      goto done;                                                                           \
    }                                                                                      \
  } while (0)
  {
    PB_CHECK(hipMalloc(&keys, sizeof(uint64_t) * nkeys));
    PB_CHECK(hipMalloc(&sorted, sizeof(uint64_t) * nkeys));
    PB_CHECK(hipMalloc(&ukeys, sizeof(uint64_t) * nkeys));
    PB_CHECK(hipMalloc(&d_num, sizeof(int64_t)));
    hipLaunchKernelGGL(k_make_keys, dim3(mfem_grid_for(nkeys, MFEM_BLOCK, ctx->num_cus * 16)), dim3(MFEM_BLOCK), 0, ctx->stream,
                       itp, nel, controlpoint_IDs, index_base, keys);
    PB_CHECK(hipGetLastError());
    size_t tb = 0, tb2 = 0;
    PB_CHECK(hipcub::DeviceRadixSort::SortKeys(nullptr, tb, keys, sorted, nkeys, 0, 64, ctx->stream));
    PB_CHECK(hipcub::DeviceSelect::Unique(nullptr, tb2, sorted, ukeys, d_num, nkeys, ctx->stream));
    if (tb2 > tb) tb = tb2;
    PB_CHECK(hipMalloc(&tmp, tb > 0 ? tb : 16));
    PB_CHECK(hipcub::DeviceRadixSort::SortKeys(tmp, tb, keys, sorted, nkeys, 0, 64, ctx->stream));
    PB_CHECK(hipcub::DeviceSelect::Unique(tmp, tb, sorted, ukeys, d_num, nkeys, ctx->stream));
    int64_t U = 0;
    PB_CHECK(hipMemcpyAsync(&U, d_num, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    PB_CHECK(hipStreamSynchronize(ctx->stream));
    const int F = n_fields;
    const int64_t n = (int64_t)F * ncp, nnz = (int64_t)F * F * U;
    mfem_host_alloc_probe();
    A = new mfem_csr_s();
    memset(A, 0, sizeof(*A));
    A->ctx = ctx;
    A->n = n;
    A->nnz = nnz;
    A->rowptr_bits = 64;
    A->index_base = 0;
    PB_CHECK(hipMalloc(&A->owned_rowptr, sizeof(int64_t) * (n + 1)));
    PB_CHECK(hipMalloc(&A->owned_colidx, sizeof(int32_t) * (nnz > 0 ? nnz : 1)));
    A->rowptr = A->owned_rowptr;
    A->colidx = (const int32_t*)A->owned_colidx;
    PB_CHECK(hipMalloc(&rowptr1, sizeof(int64_t) * (ncp + 1)));
    hipLaunchKernelGGL(k_rowptr1, dim3((int)((ncp + 1 + MFEM_BLOCK - 1) / MFEM_BLOCK)), dim3(MFEM_BLOCK), 0, ctx->stream, ncp, ukeys,
                       U, rowptr1);
    hipLaunchKernelGGL(k_expand, dim3(mfem_grid_for(ncp, MFEM_BLOCK, ctx->num_cus * 16)), dim3(MFEM_BLOCK), 0, ctx->stream, ncp, F, U,
                       rowptr1, ukeys, (int64_t*)A->owned_rowptr, (int32_t*)A->owned_colidx);
    PB_CHECK(hipGetLastError());
    if (sparse_IDs_by_el) {
      MFEM_REQUIRE(nnz + index_base < ((int64_t)1 << 31), "slot ids must fit int32 (FEM_Int)");
      hipLaunchKernelGGL(k_slots, dim3(mfem_grid_for(nkeys, MFEM_BLOCK, ctx->num_cus * 16)), dim3(MFEM_BLOCK), 0, ctx->stream, itp,
                         nel, ncp, F, U, controlpoint_IDs, index_base, ukeys, rowptr1, index_base, sparse_IDs_by_el);
      PB_CHECK(hipGetLastError());
    }
    PB_CHECK(hipStreamSynchronize(ctx->stream));
    rc = mfem_csr_plan(ctx, A);
  }
done:
  if (keys) hipFree(keys);
  if (sorted) hipFree(sorted);
  if (ukeys) hipFree(ukeys);
  if (d_num) hipFree(d_num);
  if (tmp) hipFree(tmp);
  if (rowptr1) hipFree(rowptr1);
  if (rc != MFEM_OK) {
    if (A) mfem_csr_destroy(A);
    return rc;
  }
  *out = A;
  return MFEM_OK;
#undef PB_CHECK
} MFEM_API_CATCH("mfem_pattern_build")
