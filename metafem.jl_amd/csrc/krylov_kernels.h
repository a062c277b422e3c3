// Building blocks shared by the BiCGStab(l) and IDR(s) drivers: vector updates whose coefficients
// live in device memory, multi-dot reductions, and the DONE-flag guard.  All vectors are padded to an
// even length and 16-byte aligned (see mfem_solve), so every kernel streams d2 (16 B / lane).
#pragma once
#include "krylov.h"
#include "rng.h"

typedef double d2_t __attribute__((ext_vector_type(2)));

// coefficient = sign * S[slot]  (slot >= 0)   or   the immediate `value` (slot < 0)
struct Coef {
  double value;
  double sign;
  int slot;
};
static inline Coef coef_imm(double v) { return Coef{v, 1.0, -1}; }
static inline Coef coef_dev(int slot, double sign = 1.0) { return Coef{0.0, sign, slot}; }
__device__ __forceinline__ double coef_get(const Coef& c, const double* __restrict__ S) {
  return c.slot >= 0 ? c.sign * S[c.slot] : c.value;
}

// y = a*x + b*y
// Streaming (nontemporal) LOADS in the vector kernels of bicgstabl_GS!, idrs! and the multi-dot (round 4): every vector is read once per kernel; with plain loads
// they sweep the L2 / Infinity Cache clean for the SpMV that runs next.  C3 (6.4 M rows, bicgstabl_GS! s = 2), A/B on one box: 9.77e9 -> 9.97e9
// DOF-updates/s, the SpMV pair itself 0.531 -> 0.515 ms.  -DKB_NT=0 builds the plain form.
#ifndef KB_NT
#define KB_NT 1
#endif
#if KB_NT
#define KB_LD(p, i) __builtin_nontemporal_load(&(p)[i])
#else
#define KB_LD(p, i) ((p)[i])
#endif
static __global__ __launch_bounds__(MFEM_BLOCK) void kk_axpby(int64_t n2, Coef a, const d2_t* __restrict__ x, Coef b,
                                                               d2_t* __restrict__ y, const double* __restrict__ S,
                                                               const int32_t* __restrict__ flags) {
  if (flags[F_DONE]) return;
  const double ca = coef_get(a, S), cb = coef_get(b, S);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n2; i += stride) y[i] = ca * x[i] + cb * y[i];
}

// z = a*x + b*y   (z may alias x or y)
static __global__ __launch_bounds__(MFEM_BLOCK) void kk_lin2(int64_t n2, Coef a, const d2_t* x, Coef b, const d2_t* y,
                                                              d2_t* z, const double* __restrict__ S,
                                                              const int32_t* __restrict__ flags) {
  if (flags[F_DONE]) return;
  const double ca = coef_get(a, S), cb = coef_get(b, S);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n2; i += stride) z[i] = ca * x[i] + cb * y[i];
}

// Two updates in one pass, in this order per element:  y1 += a1*x1 ;  y2 += a2*x2   (x1 may alias y2)
static __global__ __launch_bounds__(MFEM_BLOCK) void kk_axpy2(int64_t n2, Coef a1, const d2_t* x1, d2_t* y1, Coef a2,
                                                               const d2_t* x2, d2_t* y2,
                                                               const double* __restrict__ S,
                                                               const int32_t* __restrict__ flags) {
  if (flags[F_DONE]) return;
  const double c1 = coef_get(a1, S), c2 = coef_get(a2, S);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n2; i += stride) {
    y1[i] = y1[i] + c1 * x1[i];
    y2[i] = y2[i] + c2 * x2[i];
  }
}

#define KK_MAX_DOTS 8
struct DotList {
  const d2_t* x[KK_MAX_DOTS];
  const d2_t* y[KK_MAX_DOTS];
  int m;
};

// partials[k*G + blockIdx] = partial of x_k . y_k  for k < m   (one pass over memory per distinct vector read).
// Only the first n entries count: behind them a slab vector carries ghost entries (copies of the neighbours' values, which
// their owners sum) and every vector carries padding.
// M = number of dot products when it is a template constant (1 .. 8; round 5): all 2 M (M + 1 with SAMEY: every y_k is the same vector, as in P' g) loads of
// an index are issued before the first product -- behind the run-time test `k < L.m` each pair of loads sat in its own branch and was waited for there, one
// memory round trip after the other.  M = 0: the run-time form.  Same products, same order of the sums: bitwise the same partials.
template <int M, bool SAMEY>
static __global__ __launch_bounds__(MFEM_BLOCK) void kk_multi_dot(int64_t n, DotList L, double* __restrict__ partials,
                                                                   const int32_t* __restrict__ flags) {
  __shared__ double red[4];
  if (flags[F_DONE]) return;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t n2 = n >> 1;
  double acc[KK_MAX_DOTS];
#pragma unroll
  for (int k = 0; k < KK_MAX_DOTS; ++k) acc[k] = 0.0;
  if constexpr (M > 0) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n2; i += stride) {
      d2_t a[M], b[M];
#pragma unroll
      for (int k = 0; k < M; ++k) a[k] = KB_LD(L.x[k], i);
      if constexpr (SAMEY) {
        b[0] = KB_LD(L.y[0], i);
#pragma unroll
        for (int k = 1; k < M; ++k) b[k] = b[0];
      } else {
#pragma unroll
        for (int k = 0; k < M; ++k) b[k] = KB_LD(L.y[k], i);
      }
#pragma unroll
      for (int k = 0; k < M; ++k) acc[k] += a[k].x * b[k].x + a[k].y * b[k].y;
    }
  } else {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n2; i += stride) {
#pragma unroll
      for (int k = 0; k < KK_MAX_DOTS; ++k)
        if (k < L.m) {
          const d2_t a = KB_LD(L.x[k], i), b = KB_LD(L.y[k], i);
          acc[k] += a.x * b.x + a.y * b.y;
        }
    }
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {  // odd n: the last entry shares its 16 bytes with the first ghost / pad entry
#pragma unroll
    for (int k = 0; k < KK_MAX_DOTS; ++k)
      if (k < L.m) acc[k] += L.x[k][n2].x * L.y[k][n2].x;
  }
#pragma unroll
  for (int k = 0; k < KK_MAX_DOTS; ++k)
    if (k < L.m) {
      const double s = block_reduce_sum(acc[k], red);
      if (threadIdx.x == 0) partials[(int64_t)k * gridDim.x + blockIdx.x] = s;
    }
}
static inline void kk_multi_dot_launch(int G, hipStream_t st, int64_t n, const DotList& L, double* part, const int32_t* F) {
  bool same = true;
  for (int k = 1; k < L.m; ++k) same = same && L.y[k] == L.y[0];
#define KK_MD(M_)                                                                                                                   \
  case M_:                                                                                                                          \
    if (same) hipLaunchKernelGGL((kk_multi_dot<M_, true>), dim3(G), dim3(MFEM_BLOCK), 0, st, n, L, part, F);                        \
    else hipLaunchKernelGGL((kk_multi_dot<M_, false>), dim3(G), dim3(MFEM_BLOCK), 0, st, n, L, part, F);                            \
    break;
  switch (L.m) {
    KK_MD(1) KK_MD(2) KK_MD(3) KK_MD(4) KK_MD(5) KK_MD(6) KK_MD(7) KK_MD(8)
    default: hipLaunchKernelGGL((kk_multi_dot<0, false>), dim3(G), dim3(MFEM_BLOCK), 0, st, n, L, part, F); break;
  }
#undef KK_MD
}

// partials[k*G + blockIdx] = partial of P_(k0+k) . y for k < M, P_k = the +-1 vector of bit k of mfem_sign_word(seed, row) (rng.h): ONE stream (y) for M dot
// products.  A product with +-1 is a sign flip: y's sign bit is XORed with the complement of the word's bit.
template <int M>
static __global__ __launch_bounds__(MFEM_BLOCK) void kk_sign_dots(int64_t n, uint64_t seed, int k0, const d2_t* __restrict__ y, double* __restrict__ partials,
                                                                   const int32_t* __restrict__ flags) {
  __shared__ double red[4];
  if (flags[F_DONE]) return;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t n2 = n >> 1;
  double acc[M];
#pragma unroll
  for (int k = 0; k < M; ++k) acc[k] = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n2; i += stride) {
    const d2_t v = KB_LD(y, i);
    const uint64_t w0 = ~mfem_sign_word(seed, 2 * (uint64_t)i) >> k0, w1 = ~mfem_sign_word(seed, 2 * (uint64_t)i + 1) >> k0;
    const uint64_t b0 = (uint64_t)__double_as_longlong(v.x), b1 = (uint64_t)__double_as_longlong(v.y);
#pragma unroll
    for (int k = 0; k < M; ++k)
      acc[k] += __longlong_as_double((long long)(b0 ^ (((w0 >> k) & 1ull) << 63))) + __longlong_as_double((long long)(b1 ^ (((w1 >> k) & 1ull) << 63)));
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {  // odd n: the last entry shares its 16 bytes with the first ghost / pad entry
    const uint64_t w0 = ~mfem_sign_word(seed, (uint64_t)(n - 1)) >> k0;
    const uint64_t b0 = (uint64_t)__double_as_longlong(y[n2].x);
#pragma unroll
    for (int k = 0; k < M; ++k) acc[k] += __longlong_as_double((long long)(b0 ^ (((w0 >> k) & 1ull) << 63)));
  }
#pragma unroll
  for (int k = 0; k < M; ++k) {
    const double s = block_reduce_sum(acc[k], red);
    if (threadIdx.x == 0) partials[(int64_t)k * gridDim.x + blockIdx.x] = s;
  }
}
static inline void kk_sign_dots_launch(int G, hipStream_t st, int64_t n, uint64_t seed, int k0, int m, const double* y, double* part, const int32_t* F) {
#define KK_SD(M_) case M_: hipLaunchKernelGGL((kk_sign_dots<M_>), dim3(G), dim3(MFEM_BLOCK), 0, st, n, seed, k0, (const d2_t*)y, part, F); break;
  switch (m) { KK_SD(1) KK_SD(2) KK_SD(3) KK_SD(4) KK_SD(5) KK_SD(6) KK_SD(7) default: KK_SD(8) }
#undef KK_SD
}
// x[i] = +-1 by bit k of the sign word: the explicit form of P_k (tests, the literal orthogonalisation loop)
static __global__ __launch_bounds__(MFEM_BLOCK) void kk_sign_vector(int64_t n, uint64_t seed, int k, double* __restrict__ x) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride) x[i] = ((mfem_sign_word(seed, (uint64_t)i) >> k) & 1ull) ? 1.0 : -1.0;
}

// Single workgroup: S[out + k] = sum(partials[k*G .. (k+1)*G)) for k < m
static __global__ __launch_bounds__(MFEM_BLOCK) void kk_fold(const double* __restrict__ partials, int G, int m, int out,
                                                              double* __restrict__ S, const int32_t* __restrict__ flags) {
  __shared__ double red[4];
  if (flags[F_DONE]) return;
  for (int k = 0; k < m; ++k) {
    const double v = reduce_partials_bcast(partials + (int64_t)k * G, G, red);
    if (threadIdx.x == 0) S[out + k] = v;
    __syncthreads();
  }
}

// A reduction whose partial sums are folded by the scalar kernel that consumes it (one launch less per dot): the consumer is
// launched with a full workgroup, sums the partials in a fixed order into S[out + k] and continues on thread 0.
struct FoldArg {
  const double* part;
  int G, m, out;
};
// (round 6: the m sums run side by side -- all partial loads in flight, one pair of barriers -- instead of one after the other; every sum keeps the
//  order of reduce_partials_bcast, so the results are the same bits.  An IDR(8) step's ki_ortho folds eight: 16 -> 7 us.)
__device__ __forceinline__ void kk_fold_dev(const FoldArg& f, double* __restrict__ S) {
  __shared__ double fold_red[KK_MAX_DOTS][MFEM_BLOCK / 64];
  if (f.m > KK_MAX_DOTS) {  // (not issued by the solvers; kept correct)
    for (int k = 0; k < f.m; ++k) {
      const double v = reduce_partials_bcast(f.part + (int64_t)k * f.G, f.G, fold_red[0]);
      if (threadIdx.x == 0) S[f.out + k] = v;
      __syncthreads();
    }
    return;
  }
  double acc[KK_MAX_DOTS];
#pragma unroll
  for (int k = 0; k < KK_MAX_DOTS; ++k) acc[k] = 0.0;
  for (int i = threadIdx.x; i < f.G; i += blockDim.x) {
    double v[KK_MAX_DOTS];  // (branch-free: behind `if (k < m)` every load sits in its own block and is waited for on its own)
#pragma unroll
    for (int k = 0; k < KK_MAX_DOTS; ++k) v[k] = f.part[(int64_t)(k < f.m ? k : 0) * f.G + i];
#pragma unroll
    for (int k = 0; k < KK_MAX_DOTS; ++k) acc[k] += v[k];
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int k = 0; k < KK_MAX_DOTS; ++k) acc[k] = wave_reduce_sum(acc[k]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < KK_MAX_DOTS; ++k) fold_red[k][w] = acc[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 0; k < f.m; ++k) {
      double r = 0.0;
      for (int i = 0; i < nw; ++i) r += fold_red[k][i];
      S[f.out + k] = r;
    }
  }
  __syncthreads();
}

struct KK {
  mfem_context_s* ctx;
  int64_t nv;   // padded vector length (owned + ghost + pad): what the vector updates stream over
  int64_t n;    // owned entries: what the dot products sum over
  int G;
  double* S;
  int32_t* F;
  hipStream_t st;

  int axpby(Coef a, const double* x, Coef b, double* y) const {
    hipLaunchKernelGGL(kk_axpby, dim3(G), dim3(MFEM_BLOCK), 0, st, nv / 2, a, (const d2_t*)x, b, (d2_t*)y, S, F);
    MFEM_CHECK_LAUNCH();
    return MFEM_OK;
  }
  int lin2(Coef a, const double* x, Coef b, const double* y, double* z) const {
    hipLaunchKernelGGL(kk_lin2, dim3(G), dim3(MFEM_BLOCK), 0, st, nv / 2, a, (const d2_t*)x, b, (const d2_t*)y, (d2_t*)z, S, F);
    MFEM_CHECK_LAUNCH();
    return MFEM_OK;
  }
  int axpy2(Coef a1, const double* x1, double* y1, Coef a2, const double* x2, double* y2) const {
    hipLaunchKernelGGL(kk_axpy2, dim3(G), dim3(MFEM_BLOCK), 0, st, nv / 2, a1, (const d2_t*)x1, (d2_t*)y1, a2,
                       (const d2_t*)x2, (d2_t*)y2, S, F);
    MFEM_CHECK_LAUNCH();
    return MFEM_OK;
  }
  // S[out + k] = x_k . y_k (all-reduced over ranks when a communicator is attached)
  int dots(const DotList& L, int out) const {
    double* part = ctx->d_partials;
    kk_multi_dot_launch(G, st, n, L, part, F);
    MFEM_CHECK_LAUNCH();
    hipLaunchKernelGGL(kk_fold, dim3(1), dim3(MFEM_BLOCK), 0, st, part, G, L.m, out, S, F);
    MFEM_CHECK_LAUNCH();
    if (ctx->comm) return mfem_comm_allreduce(ctx, S + out, L.m);
    return MFEM_OK;
  }
  // partial sums only; the returned descriptor goes to the scalar kernel that uses the result (launch it with K1F).  With a
  // communicator the fold + all-reduce happen here and the descriptor is empty.
  int dots_partials(const DotList& L, int out, FoldArg* fa) const {
    double* part = ctx->d_partials;
    kk_multi_dot_launch(G, st, n, L, part, F);
    MFEM_CHECK_LAUNCH();
    *fa = FoldArg{part, G, L.m, out};
    if (ctx->comm) {
      hipLaunchKernelGGL(kk_fold, dim3(1), dim3(MFEM_BLOCK), 0, st, part, G, L.m, out, S, F);
      MFEM_CHECK_LAUNCH();
      fa->m = 0;
      return mfem_comm_allreduce(ctx, S + out, L.m);
    }
    return MFEM_OK;
  }
  // the m (<= 8) dot products P_(k0 ..)' y with the +-1 shadow vectors of `seed`: y is the only stream
  int sign_dots_partials(uint64_t seed, int k0, int m, const double* y, int out, FoldArg* fa) const {
    double* part = ctx->d_partials;
    kk_sign_dots_launch(G, st, n, seed, k0, m, y, part, F);
    MFEM_CHECK_LAUNCH();
    *fa = FoldArg{part, G, m, out};
    if (ctx->comm) {
      hipLaunchKernelGGL(kk_fold, dim3(1), dim3(MFEM_BLOCK), 0, st, part, G, m, out, S, F);
      MFEM_CHECK_LAUNCH();
      fa->m = 0;
      return mfem_comm_allreduce(ctx, S + out, m);
    }
    return MFEM_OK;
  }
  int dot1_partials(const double* x, const double* y, int out, FoldArg* fa) const {
    DotList L;
    L.m = 1;
    L.x[0] = (const d2_t*)x;
    L.y[0] = (const d2_t*)y;
    return dots_partials(L, out, fa);
  }
  int dot1(const double* x, const double* y, int out) const {
    DotList L;
    L.m = 1;
    L.x[0] = (const d2_t*)x;
    L.y[0] = (const d2_t*)y;
    return dots(L, out);
  }
  int spmv(mfem_csr_s* A, const double* vals, double* x, double* y, int* spmv_count) const {
    ++*spmv_count;  // with a communicator: the halo exchange of x runs beside the rows that need no ghost entry
    return mfem_spmv_halo(ctx, A, vals, x, y, 1.0, 0.0, nullptr, nullptr, nullptr, F);
  }
};
