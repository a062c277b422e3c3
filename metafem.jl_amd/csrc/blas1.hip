// Level-1 kernels of the Krylov loop: replaces the GPUArrays broadcasts and the CUBLAS dot/nrm2
// behind LinearAlgebra.dot/norm in the reference solvers (e.g. linear_solver/03_BiCGstabl.jl:45-90).
// All are HBM-bound streams: 16-byte per-lane accesses, persistent grid, wave-shuffle reductions,
// deterministic two-stage sums (no FP64 atomics).
#include "krylov.h"

#include "rng.h"

typedef double d2_t __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(MFEM_BLOCK) void k_axpby(int64_t n, double a, const double* __restrict__ x, double b,
                                                        double* __restrict__ y) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t n2 = n >> 1;
  const bool al = ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0;
  if (al) {
    const d2_t* x2 = reinterpret_cast<const d2_t*>(x);
    d2_t* y2 = reinterpret_cast<d2_t*>(y);
    if (b == 0.0) {
      for (int64_t i = tid; i < n2; i += stride) y2[i] = a * x2[i];
    } else {
      for (int64_t i = tid; i < n2; i += stride) y2[i] = a * x2[i] + b * y2[i];
    }
    if (tid == 0 && (n & 1)) y[n - 1] = (b == 0.0) ? a * x[n - 1] : a * x[n - 1] + b * y[n - 1];
  } else {
    for (int64_t i = tid; i < n; i += stride) y[i] = (b == 0.0) ? a * x[i] : a * x[i] + b * y[i];
  }
}

__global__ __launch_bounds__(MFEM_BLOCK) void k_dot_partials(int64_t n, const double* __restrict__ x,
                                                               const double* __restrict__ y,
                                                               double* __restrict__ partials) {
  __shared__ double red[4];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  double acc = 0.0;
  const bool al = ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0;
  if (al) {
    const d2_t* x2 = reinterpret_cast<const d2_t*>(x);
    const d2_t* y2 = reinterpret_cast<const d2_t*>(y);
    const int64_t n2 = n >> 1;
    for (int64_t i = tid; i < n2; i += stride) {
      const d2_t a = x2[i], b = y2[i];
      acc += a.x * b.x + a.y * b.y;
    }
    if (tid == 0 && (n & 1)) acc += x[n - 1] * y[n - 1];
  } else {
    for (int64_t i = tid; i < n; i += stride) acc += x[i] * y[i];
  }
  const double b = block_reduce_sum(acc, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = b;
}

// Single workgroup: out[k] = sum(src[k][0 .. cnt[k])) for k < m -- the local scalars of one reduction group, contiguous for the
// all-reduce that follows when a communicator is attached
__global__ __launch_bounds__(MFEM_BLOCK) void k_fold_list(FoldList L, double* __restrict__ out, const int32_t* __restrict__ done_flag) {
  __shared__ double red[4];
  if (done_flag && done_flag[0]) return;
  for (int k = 0; k < L.m; ++k) {
    const double v = reduce_partials_bcast(L.src[k], L.cnt[k], red);
    if (threadIdx.x == 0) out[k] = v;
  }
}

int mfem_fold_list(mfem_context_s* ctx, const FoldList& L, double* d_out, const int32_t* done_flag) {
  hipLaunchKernelGGL(k_fold_list, dim3(1), dim3(MFEM_BLOCK), 0, ctx->stream, L, d_out, done_flag);
  MFEM_CHECK_LAUNCH();
  return MFEM_OK;
}

// Single workgroup: out[0] = sum(partials[0..np))
__global__ __launch_bounds__(MFEM_BLOCK) void k_sum_partials(const double* __restrict__ partials, int np,
                                                               double* __restrict__ out) {
  __shared__ double red[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < np; i += blockDim.x) acc += partials[i];
  const double b = block_reduce_sum(acc, red);
  if (threadIdx.x == 0) out[0] = b;
}

__global__ __launch_bounds__(MFEM_BLOCK) void k_rand(int64_t n, uint64_t seed, uint32_t stream_id,
                                                       double* __restrict__ x) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride)
    x[i] = mfem_u01(seed, stream_id, (uint64_t)i);
}

// 3 workgroups per CU: CG iteration at 256^3 1.057 ms vs 1.078 ms with 8 (fewer partial sums for the consumer kernels to
// re-reduce, longer unit-stride runs per workgroup); 1: 1.138, 2: 1.059, 4: 1.062, 5: 1.074, 16: 1.095 (tools/probe_vecgrid.py)
static std::atomic<int> g_vec_grid_mult{3};
extern "C" int mfem_debug_set_vec_grid(int workgroups_per_cu) try {
  ++mfem_debug_epoch;
  if (workgroups_per_cu > 0) g_vec_grid_mult = workgroups_per_cu;
  return MFEM_OK;
} MFEM_API_CATCH("mfem_debug_set_vec_grid")

int mfem_vec_grid(mfem_context_s* ctx, int64_t n) {
  // 16 B per lane; persistent grid of at most g_vec_grid_mult workgroups per CU
  int cap = ctx->num_cus * g_vec_grid_mult;
  if (cap > MFEM_MAX_PARTIALS) cap = MFEM_MAX_PARTIALS;
  return mfem_grid_for((n + 1) / 2, MFEM_BLOCK, cap);
}

int mfem_sum_partials(mfem_context_s* ctx, const double* partials, int np, double* d_out) {
  hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(MFEM_BLOCK), 0, ctx->stream, partials, np, d_out);
  MFEM_CHECK_LAUNCH();
  return MFEM_OK;
}

int mfem_dot_device(mfem_context_s* ctx, int64_t n, const double* x, const double* y, double* d_out) {
  const int grid = mfem_vec_grid(ctx, n);
  hipLaunchKernelGGL(k_dot_partials, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, n, x, y, ctx->d_partials);
  MFEM_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(MFEM_BLOCK), 0, ctx->stream, ctx->d_partials, grid, d_out);
  MFEM_CHECK_LAUNCH();
  return MFEM_OK;
}

extern "C" int mfem_axpby(mfem_context ctx, int64_t n, double a, const double* x, double b, double* y) try {
  MFEM_REQUIRE(ctx, "null ctx");
  MFEM_REQUIRE(n >= 0, "negative n");
  if (n == 0) return MFEM_OK;
  MFEM_REQUIRE(x && y, "null vector");
  hipLaunchKernelGGL(k_axpby, dim3(mfem_vec_grid(ctx, n)), dim3(MFEM_BLOCK), 0, ctx->stream, n, a, x, b, y);
  MFEM_CHECK_LAUNCH();
  return MFEM_OK;
} MFEM_API_CATCH("mfem_axpby")

extern "C" int mfem_dot(mfem_context ctx, int64_t n, const double* x, const double* y, double* out) try {
  MFEM_REQUIRE(ctx && out, "null argument");
  MFEM_REQUIRE(n >= 0, "negative n");
  if (n == 0) {
    *out = 0.0;
    return MFEM_OK;
  }
  MFEM_REQUIRE(x && y, "null vector");
  int rc = mfem_dot_device(ctx, n, x, y, ctx->d_scalars + MFEM_NSCALARS - 1);
  if (rc) return rc;
  MFEM_CHECK_HIP(hipMemcpyAsync(ctx->h_scalars + MFEM_NSCALARS - 1, ctx->d_scalars + MFEM_NSCALARS - 1, sizeof(double), hipMemcpyDeviceToHost,
                                ctx->stream));
  MFEM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  *out = ctx->h_scalars[MFEM_NSCALARS - 1];
  return MFEM_OK;
} MFEM_API_CATCH("mfem_dot")

extern "C" int mfem_nrm2(mfem_context ctx, int64_t n, const double* x, double* out) try {
  double d = 0.0;
  int rc = mfem_dot(ctx, n, x, x, &d);
  if (rc) return rc;
  *out = sqrt(d);
  return MFEM_OK;
} MFEM_API_CATCH("mfem_nrm2")

extern "C" int mfem_rand(mfem_context ctx, int64_t n, uint64_t seed, uint32_t stream_id, double* x) try {
  MFEM_REQUIRE(ctx, "null ctx");
  MFEM_REQUIRE(n >= 0, "negative n");
  if (n == 0) return MFEM_OK;
  MFEM_REQUIRE(x, "null vector");
  hipLaunchKernelGGL(k_rand, dim3(mfem_vec_grid(ctx, n)), dim3(MFEM_BLOCK), 0, ctx->stream, n, seed, stream_id, x);
  MFEM_CHECK_LAUNCH();
  return MFEM_OK;
} MFEM_API_CATCH("mfem_rand")
