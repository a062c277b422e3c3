// Jacobi preconditioner kernels: reference linear_solver/02_Preconditioner.jl
//   Jacobi_By_Diagonal :122-130, Jacobi2_By_Colomn :132-139, Mat_Div_Jacobi :141-148,
//   Jacobi_By_Row :170-177.
// The reference runs one thread per row over a 27..81-entry row (stride-row, uncoalesced).  Here a
// row is scanned by an 8-lane group (row entries are contiguous, so a wave reads 8 rows x 32..64 B
// runs), and Mat_Div_Jacobi -- which needs no row structure at all -- is a flat 16-byte-per-lane
// stream over the nonzeros.
#include "blas1.h"

typedef double d2_t __attribute__((ext_vector_type(2)));
typedef int i2_t __attribute__((ext_vector_type(2)));

// mode 0: d[r] = |K_rr| (rows without a stored diagonal keep their preset value; so do rows whose stored diagonal is exactly
//         0 -- the reference would take |0| and divide by it; the solver layouts of spmv_ell.hip cannot tell a stored zero
//         from a padding slot, so every layout applies this one guarded rule)
// mode 1: d[r] = sqrt(sum_j K_rj^2)
template <typename RP>
__global__ __launch_bounds__(MFEM_BLOCK) void k_jacobi_rows(int64_t n, const RP* __restrict__ rowptr,
                                                              const int32_t* __restrict__ col,
                                                              const double* __restrict__ vals, double* __restrict__ d,
                                                              int base, int mode) {
  const int g = threadIdx.x & 7;
  const int64_t grp = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 3;
  const int64_t ngrp = ((int64_t)gridDim.x * blockDim.x) >> 3;
  for (int64_t r = grp; r < n; r += ngrp) {
    const int64_t lo = (int64_t)rowptr[r] - base, hi = (int64_t)rowptr[r + 1] - base;
    if (mode == 0) {
      for (int64_t j = lo + g; j < hi; j += 8)
        if ((int64_t)col[j] - base == r && vals[j] != 0.0) d[r] = fabs(vals[j]);  // a stored ZERO diagonal keeps the preset too
    } else {
      double s = 0.0;
      for (int64_t j = lo + g; j < hi; j += 8) s += vals[j] * vals[j];
      s += __shfl_xor(s, 1, MFEM_WAVE);
      s += __shfl_xor(s, 2, MFEM_WAVE);
      s += __shfl_xor(s, 4, MFEM_WAVE);
      if (g == 0) d[r] = sqrt(s);
    }
  }
}

__global__ __launch_bounds__(MFEM_BLOCK) void k_jacobi2_by_column(int64_t nnz, const int32_t* __restrict__ col,
                                                                    const double* __restrict__ vals,
                                                                    double* __restrict__ d, int base) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < nnz; j += stride)
    atomicAdd(&d[col[j] - base], vals[j] * vals[j]);
}

__global__ __launch_bounds__(MFEM_BLOCK) void k_sqrt_inplace(int64_t n, double* __restrict__ d) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride) d[i] = sqrt(d[i]);
}

// vals[j] /= d[col[j]] over the flat nonzero stream.
// vals[j] = src[j] / d[col[j]] (src == vals: in place; the solver scales its working copy straight from the caller's values)
__global__ __launch_bounds__(MFEM_BLOCK) void k_mat_div_jacobi(int64_t nnz, const int32_t* __restrict__ col,
                                                                 const double* src, double* vals,
                                                                 const double* __restrict__ d, int base) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const bool al = ((((uintptr_t)vals) & 15) == 0) && ((((uintptr_t)src) & 15) == 0) && ((((uintptr_t)col) & 7) == 0);
  if (al) {
    const int64_t n2 = nnz >> 1;
    d2_t* v2 = reinterpret_cast<d2_t*>(vals);
    const d2_t* s2 = reinterpret_cast<const d2_t*>(src);
    const i2_t* c2 = reinterpret_cast<const i2_t*>(col);
    for (int64_t i = tid; i < n2; i += stride) {
      d2_t v = s2[i];
      const i2_t c = c2[i];
      v.x /= d[c.x - base];
      v.y /= d[c.y - base];
      v2[i] = v;
    }
    if (tid == 0 && (nnz & 1)) vals[nnz - 1] = src[nnz - 1] / d[col[nnz - 1] - base];
  } else {
    for (int64_t j = tid; j < nnz; j += stride) vals[j] = src[j] / d[col[j] - base];
  }
}

// vals[j] /= d[row(j)]  (left Jacobi folded into the matrix; 8 lanes per row like k_jacobi_rows)
template <typename RP>
__global__ __launch_bounds__(MFEM_BLOCK) void k_mat_div_rows(int64_t n, const RP* __restrict__ rowptr, double* __restrict__ vals,
                                                               const double* __restrict__ d, int base) {
  const int g = threadIdx.x & 7;
  const int64_t grp = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 3;
  const int64_t ngrp = ((int64_t)gridDim.x * blockDim.x) >> 3;
  for (int64_t r = grp; r < n; r += ngrp) {
    const int64_t lo = (int64_t)rowptr[r] - base, hi = (int64_t)rowptr[r + 1] - base;
    const double dr = d[r];
    for (int64_t j = lo + g; j < hi; j += 8) vals[j] /= dr;
  }
}

int mfem_mat_div_rows(mfem_context_s* ctx, mfem_csr_s* A, double* vals, const double* d) {
  if (A->n == 0) return MFEM_OK;
  const int grid = mfem_grid_for(A->n * 8, MFEM_BLOCK, ctx->num_cus * 16);
  if (A->rowptr_bits == 64)
    hipLaunchKernelGGL(k_mat_div_rows<int64_t>, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, A->n,
                       (const int64_t*)A->rowptr, vals, d, A->index_base);
  else
    hipLaunchKernelGGL(k_mat_div_rows<int32_t>, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, A->n,
                       (const int32_t*)A->rowptr, vals, d, A->index_base);
  MFEM_CHECK_LAUNCH();
  return MFEM_OK;
}

// off[r] = offset of row r's diagonal entry inside the row (0xFFFF: none stored); one scan of the pattern, once per handle
template <typename RP>
__global__ __launch_bounds__(MFEM_BLOCK) void k_diag_offsets(int64_t n, const RP* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                               int base, uint16_t* __restrict__ off) {
  const int g = threadIdx.x & 7;
  const int64_t grp = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 3;
  const int64_t ngrp = ((int64_t)gridDim.x * blockDim.x) >> 3;
  for (int64_t r = grp; r < n; r += ngrp) {
    const int64_t lo = (int64_t)rowptr[r] - base, hi = (int64_t)rowptr[r + 1] - base;
    int found = 0xFFFF;
    for (int64_t j = lo + g; j < hi; j += 8)
      if ((int64_t)col[j] - base == r) found = (int)(j - lo);  // (a row lists a column once)
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) {
      const int o = __shfl_xor(found, m, MFEM_WAVE);
      found = o < found ? o : found;
    }
    if (g == 0) off[r] = (uint16_t)found;
  }
}
// d[r] = |K_rr| through the table (same guarded rule as k_jacobi_rows mode 0: no stored diagonal, or a stored zero, keeps the preset)
template <typename RP>
__global__ __launch_bounds__(MFEM_BLOCK) void k_jacobi_diag_table(int64_t n, const RP* __restrict__ rowptr, const uint16_t* __restrict__ off,
                                                                    const double* __restrict__ vals, double* __restrict__ d, int base) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n; r += stride) {
    const int o = off[r];
    if (o == 0xFFFF) continue;
    const double v = vals[(int64_t)rowptr[r] - base + o];
    if (v != 0.0) d[r] = fabs(v);
  }
}

int mfem_jacobi_diag_launch(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, double* d, int mode) {
  if (A->n == 0) return MFEM_OK;
  if (mode == 0 && A->max_row_nnz > 0 && A->max_row_nnz < 0xFFFF) {
    // |diag|: an n-sized gather through a per-pattern table of diagonal positions (built by the first call: one scan of the pattern, what
    // every call used to cost)
    if (!A->diag_off) {
      uint16_t* t = nullptr;
      MFEM_CHECK_HIP(hipMalloc(&t, sizeof(uint16_t) * (size_t)A->n));
      const int g8 = mfem_grid_for(A->n * 8, MFEM_BLOCK, ctx->num_cus * 16);
      if (A->rowptr_bits == 64)
        hipLaunchKernelGGL(k_diag_offsets<int64_t>, dim3(g8), dim3(MFEM_BLOCK), 0, ctx->stream, A->n, (const int64_t*)A->rowptr, A->colidx, A->index_base, t);
      else
        hipLaunchKernelGGL(k_diag_offsets<int32_t>, dim3(g8), dim3(MFEM_BLOCK), 0, ctx->stream, A->n, (const int32_t*)A->rowptr, A->colidx, A->index_base, t);
      if (hipGetLastError() != hipSuccess) {
        hipFree(t);
        mfem_set_error("k_diag_offsets launch failed");
        return MFEM_ERR_HIP;
      }
      A->diag_off = t;
    }
    const int g1 = mfem_grid_for(A->n, MFEM_BLOCK, ctx->num_cus * 16);
    if (A->rowptr_bits == 64)
      hipLaunchKernelGGL(k_jacobi_diag_table<int64_t>, dim3(g1), dim3(MFEM_BLOCK), 0, ctx->stream, A->n, (const int64_t*)A->rowptr, A->diag_off, vals, d, A->index_base);
    else
      hipLaunchKernelGGL(k_jacobi_diag_table<int32_t>, dim3(g1), dim3(MFEM_BLOCK), 0, ctx->stream, A->n, (const int32_t*)A->rowptr, A->diag_off, vals, d, A->index_base);
    MFEM_CHECK_LAUNCH();
    return MFEM_OK;
  }
  const int grid = mfem_grid_for(A->n * 8, MFEM_BLOCK, ctx->num_cus * 16);
  if (A->rowptr_bits == 64)
    hipLaunchKernelGGL(k_jacobi_rows<int64_t>, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, A->n,
                       (const int64_t*)A->rowptr, A->colidx, vals, d, A->index_base, mode);
  else
    hipLaunchKernelGGL(k_jacobi_rows<int32_t>, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, A->n,
                       (const int32_t*)A->rowptr, A->colidx, vals, d, A->index_base, mode);
  MFEM_CHECK_LAUNCH();
  return MFEM_OK;
}

extern "C" int mfem_jacobi_by_diagonal(mfem_context ctx, mfem_csr A, const double* vals, double* d) try {
  MFEM_REQUIRE(ctx && A, "null handle");
  MFEM_REQUIRE(A->n == 0 || (vals && d), "null array");
  return mfem_jacobi_diag_launch(ctx, A, vals, d, 0);
} MFEM_API_CATCH("mfem_jacobi_by_diagonal")

extern "C" int mfem_jacobi_by_row(mfem_context ctx, mfem_csr A, const double* vals, double* d) try {
  MFEM_REQUIRE(ctx && A, "null handle");
  MFEM_REQUIRE(A->n == 0 || (vals && d), "null array");
  return mfem_jacobi_diag_launch(ctx, A, vals, d, 1);
} MFEM_API_CATCH("mfem_jacobi_by_row")

extern "C" int mfem_jacobi2_by_column(mfem_context ctx, mfem_csr A, const double* vals, double* d) try {
  MFEM_REQUIRE(ctx && A, "null handle");
  if (A->n == 0) return MFEM_OK;
  MFEM_REQUIRE(vals && d, "null array");
  // d has mfem_csr_ncols(A) entries: a slab pattern addresses ghost columns behind the owned ones
  const int64_t ncols = A->ncols > 0 ? A->ncols : A->n;
  MFEM_CHECK_HIP(hipMemsetAsync(d, 0, sizeof(double) * ncols, ctx->stream));
  hipLaunchKernelGGL(k_jacobi2_by_column, dim3(mfem_grid_for(A->nnz, MFEM_BLOCK, ctx->num_cus * 16)), dim3(MFEM_BLOCK),
                     0, ctx->stream, A->nnz, A->colidx, vals, d, A->index_base);
  MFEM_CHECK_LAUNCH();
  const bool slab = ncols > A->n && mfem_comm_world(ctx) > 1;
  if (slab) {
    // the rows that hit a column next to a slab interface live on two ranks: the squares this rank summed into its ghost
    // columns go to the owners, who add them -- d becomes the column norm of the GLOBAL matrix, as on one GPU
    int rc = mfem_comm_halo_reduce(ctx, d);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(k_sqrt_inplace, dim3(mfem_grid_for(A->n, MFEM_BLOCK, ctx->num_cus * 8)), dim3(MFEM_BLOCK), 0,
                     ctx->stream, A->n, d);
  MFEM_CHECK_LAUNCH();
  if (slab) return mfem_comm_halo(ctx, d);  // ghost entries = the owners' d
  return MFEM_OK;
} MFEM_API_CATCH("mfem_jacobi2_by_column")

int mfem_mat_div_jacobi_from(mfem_context_s* ctx, mfem_csr_s* A, const double* src, double* vals, const double* d) {
  if (A->nnz == 0) return MFEM_OK;
  hipLaunchKernelGGL(k_mat_div_jacobi, dim3(mfem_grid_for((A->nnz + 1) / 2, MFEM_BLOCK, ctx->num_cus * 16)),
                     dim3(MFEM_BLOCK), 0, ctx->stream, A->nnz, A->colidx, src, vals, d, A->index_base);
  MFEM_CHECK_LAUNCH();
  return MFEM_OK;
}

extern "C" int mfem_mat_div_jacobi(mfem_context ctx, mfem_csr A, double* vals, const double* d) try {
  MFEM_REQUIRE(ctx && A, "null handle");
  if (A->nnz == 0) return MFEM_OK;
  MFEM_REQUIRE(vals && d, "null array");
  return mfem_mat_div_jacobi_from(ctx, A, vals, vals, d);
} MFEM_API_CATCH("mfem_mat_div_jacobi")
