// Internal launch helpers shared by the solver translation units.
#pragma once
#include "common.h"

int mfem_vec_grid(mfem_context_s* ctx, int64_t n);
struct FoldList {
  const double* src[4];
  int cnt[4];
  int m;
};
int mfem_fold_list(mfem_context_s* ctx, const FoldList& L, double* d_out, const int32_t* done_flag = nullptr);
int mfem_dot_device(mfem_context_s* ctx, int64_t n, const double* x, const double* y, double* d_out);
int mfem_spmv_launch(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, const double* x, double* y,
                     double alpha, double beta, const double* dotw, double* partials, int* n_partials,
                     const int32_t* done_flag = nullptr);
int mfem_jacobi_diag_launch(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, double* d, int mode);
int mfem_mat_div_rows(mfem_context_s* ctx, mfem_csr_s* A, double* vals, const double* d);
int mfem_mat_div_jacobi_from(mfem_context_s* ctx, mfem_csr_s* A, const double* src, double* vals, const double* d);  // vals = src / d[col]
int mfem_comm_allreduce(mfem_context_s* ctx, double* dev, int count);
int mfem_comm_halo(mfem_context_s* ctx, double* x_local);
// split form: between begin and end the ghost entries of x must not be read and its boundary planes must not be written
int mfem_comm_halo_begin(mfem_context_s* ctx, double* x_local);
int mfem_comm_halo_end(mfem_context_s* ctx);
int mfem_comm_halo_reduce(mfem_context_s* ctx, double* x_local);
int mfem_comm_world(const mfem_context_s* ctx);
int mfem_comm_rank(const mfem_context_s* ctx);
bool mfem_comm_capturable(const mfem_context_s* ctx);
int64_t mfem_comm_owned_nodes(const mfem_context_s* ctx);
// SpMV with the halo exchange of x overlapped with the rows that need no ghost entry (spmv.hip)
int mfem_spmv_halo(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, double* x, double* y, double alpha, double beta,
                   const double* dotw, double* partials, int* n_partials, const int32_t* done_flag = nullptr);

// Every thread of the workgroup returns sum(p[0..np)); deterministic order, identical in all
// workgroups.  np <= MFEM_MAX_PARTIALS.  smem >= 4 doubles.
__device__ __forceinline__ double reduce_partials_bcast(const double* __restrict__ p, int np, double* smem) {
  double acc = 0.0;
  // four loads of a thread in flight, added in the order of the plain loop (a `for ... acc += p[i]` of unknown trip count waits for every load: three to
  // eight round trips at the head of every kernel that folds its predecessor's partial sums); a term past the end adds +0.0
  const int B = blockDim.x;
  for (int i = threadIdx.x; i < np; i += 4 * B) {
    const double v0 = p[i], v1 = i + B < np ? p[i + B] : 0.0, v2 = i + 2 * B < np ? p[i + 2 * B] : 0.0, v3 = i + 3 * B < np ? p[i + 3 * B] : 0.0;
    acc += v0;
    acc += v1;
    acc += v2;
    acc += v3;
  }
  acc = wave_reduce_sum(acc);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) smem[w] = acc;
  __syncthreads();
  double r = 0.0;
  const int nw = (blockDim.x + 63) >> 6;
  for (int i = 0; i < nw; ++i) r += smem[i];
  return r;
}
