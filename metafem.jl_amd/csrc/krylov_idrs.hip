// idrs!  -- bi-orthogonal IDR(s) (reference linear_solver/04_IDRs.jl:26-95, modify_Omega :1-8; default s = 4,
// the examples pass s = 8 / 20).  This is the reference's default Sv_func! (02_Preconditioner.jl:32).
//
// The reference solves the small lower-triangular system M c = f on the HOST every inner step (:47) and
// reads every dot back; here M, f, c, omega, alpha, beta live in ctx->d_scalars, the triangular solve is a
// one-thread device kernel, and the vector work of an inner step is fused:
//   V = sum c_i G_i ; Q = sum c_i U_i ; V = r - V ; U_k = Q + omega V     -> ONE kernel (ki_combine)
//   G_k -= alpha G_i ; U_k -= alpha U_i                                    -> ONE kernel (kk_axpy2)
//   x += beta U_k ; r -= beta G_k                                          -> ONE kernel (kk_axpy2)
//   M[k:s,k] = P[k:s]' G_k and the s dots f = P' r                          -> batched multi-dot passes
// The stop test after every inner step (:79) is evaluated on device into the DONE flag.
//
// Round 3: the bi-orthogonalisation loop (:62-66: for i < k: alpha = p_i' g / M_ii; g -= alpha g_i; u -= alpha u_i -- k dependent dot products and
// 2 k vector updates, 8 k vector streams, more than half of all streams of a cycle at s = 8) is MERGED: one multi-dot pass gives d = P' g for the
// untouched g; the alphas follow from d by forward substitution with the lower triangle of M that is already there (p_i' g^(i) = d_i - sum_{t<i}
// alpha_t M_it), column k of M is d_i - sum_t alpha_t M_it as well -- no second pass --, and ONE kernel applies all alphas to g and u, updates x and r
// with beta and leaves the partial sums of r'r: 35 streams per inner step instead of 46 on average (s = 8), and one reduction (all-reduce) per inner
// step instead of k + 2.  The same numbers in exact arithmetic (Collignon & van Gijzen's re-ordering); mfem_debug_set_idrs(1) runs the literal loop.
#include "krylov_kernels.h"

#define IS_MAXS MFEM_MAX_S
enum {
  I_OMEGA = S_SOLVER + 0, I_ALPHA, I_BETA, I_SP0, I_SP1, I_SP2,
  I_DOT = S_SOLVER + 8,          // batched dot scratch [KK_MAX_DOTS]
  I_F = I_DOT + KK_MAX_DOTS,     // f [IS_MAXS]
  I_C = I_F + IS_MAXS,           // c [IS_MAXS]
  I_M = I_C + IS_MAXS,           // M [IS_MAXS * IS_MAXS], M[i + IS_MAXS*j]
  I_D = I_M + IS_MAXS * IS_MAXS, // d [IS_MAXS]: P' G_k before the bi-orthogonalisation (merged form)
  I_AL = I_D + IS_MAXS           // alpha [IS_MAXS]: its coefficients
};
static_assert(I_AL + IS_MAXS <= MFEM_NSCALARS, "IDR(s) scalars do not fit the device scalar block");

struct IdArgs {
  double n_inv, tol;
  int32_t maxiter, fixed, s;
};

__global__ void ki_init(IdArgs a, double* __restrict__ S, int32_t* __restrict__ F) {
  for (int j = 0; j < a.s; ++j)
    for (int i = 0; i < a.s; ++i) S[I_M + i + IS_MAXS * j] = (i == j) ? 1.0 : 0.0;  // M = I (:38)
  for (int i = 0; i < a.s; ++i) S[I_F + i] = S[I_C + i] = 0.0;
  S[I_OMEGA] = 1.0;
  F[F_ITER] = 1;
  const bool conv = !a.fixed && sqrt(S[S_RR] * a.n_inv) <= a.tol;
  F[F_DONE] = conv ? 1 : 0;
  if (conv) F[F_ITER] = 0;
}
// f[first + t] = dots[t]
// beta_k >= 0: this was the last chunk of column k of M -> also beta = f[k] / M[k,k]   (:73)
__global__ void ki_store(FoldArg fa, int dst, int m, int beta_k, double* __restrict__ S, const int32_t* __restrict__ F) {
  if (F[F_DONE]) return;
  kk_fold_dev(fa, S);
  if (threadIdx.x != 0) return;
  for (int t = 0; t < m; ++t) S[dst + t] = S[I_DOT + t];
  if (beta_k >= 0) S[I_BETA] = S[I_F + beta_k] / S[I_M + beta_k + IS_MAXS * beta_k];
}
// alpha = dot(P[i], G[k]) / M[i,i]   (:62)
__global__ void ki_alpha(FoldArg fa, int i, double* __restrict__ S, const int32_t* __restrict__ F) {
  if (F[F_DONE]) return;
  kk_fold_dev(fa, S);
  if (threadIdx.x != 0) return;
  S[I_ALPHA] = S[I_DOT] / S[I_M + i + IS_MAXS * i];
}
// stop test after the inner step, then f[k+1:] -= beta*M[k+1:,k]; iter += 1   (:79-81); S[I_DOT] = r.r
__global__ void ki_step_end(FoldArg fa, IdArgs a, int k, double* __restrict__ S, int32_t* __restrict__ F, int f_done = 0) {
  if (F[F_DONE]) return;
  kk_fold_dev(fa, S);
  if (threadIdx.x != 0) return;
  S[S_RR] = S[I_DOT];
  const int iter = F[F_ITER];
  if ((!a.fixed && sqrt(S[I_DOT] * a.n_inv) <= a.tol) || iter >= a.maxiter) {
    F[F_DONE] = 1;
    return;
  }
  if (k >= 0 && !f_done) {
    const double beta = S[I_BETA];
    for (int i = k + 1; i < a.s; ++i) S[I_F + i] -= beta * S[I_M + i + IS_MAXS * k];
  }
  F[F_ITER] = iter + 1;
}
// omega = modify_Omega(Ar, r) (:1-8); dots: [0] Ar.Ar  [1] r.r  [2] Ar.r
__global__ void ki_omega(FoldArg fa, double* __restrict__ S, const int32_t* __restrict__ F) {
  if (F[F_DONE]) return;
  kk_fold_dev(fa, S);
  if (threadIdx.x != 0) return;
  const double angle = 0.70710678118654752440;  // sqrt(2)/2
  const double n1 = sqrt(S[I_DOT]), n2 = sqrt(S[I_DOT + 1]), d = S[I_DOT + 2];
  const double rho = fabs(d / (n1 * n2));
  const double omega = d / (n1 * n1);
  S[I_OMEGA] = (rho < angle) ? omega * angle / rho : omega;
}

struct CombineList {
  const d2_t* G[IS_MAXS];
  const d2_t* U[IS_MAXS];
  int m;  // number of terms (s - k)
};
// U_k = sum_t c[t] U[k+t] + omega * (r - sum_t c[t] G[k+t])      (:49-58)
// The coefficients c = LowerTriangular(M[k:s, k:s]) \ f[k:s] (:47; a host solve in the reference) are computed by every
// workgroup for itself from a shared-memory copy of the m x m block (m <= 32: a few hundred flops) -- no separate 1-thread kernel.
// M = L.m as a template constant (1 .. 8; round 5): the 2 M + 1 loads of an index are issued before the first product (the run-time loop waited for each
// pair before the next was requested); M = 0: the run-time form (s > 8).  Same arithmetic in the same order.
template <int M>
__global__ __launch_bounds__(MFEM_BLOCK) void ki_combine(int64_t n2, CombineList L, int k, const d2_t* __restrict__ r, d2_t* Uk,
                                                          const double* __restrict__ S, const int32_t* __restrict__ F) {
  __shared__ double c[IS_MAXS];
  __shared__ double Ms[IS_MAXS * IS_MAXS];
  __shared__ double fs[IS_MAXS];
  if (F[F_DONE]) return;
  const int m = L.m;  // = s - k
  for (int t = threadIdx.x; t < m * m; t += blockDim.x) {
    const int i = t % m, j = t / m;
    Ms[i + IS_MAXS * j] = S[I_M + (k + i) + IS_MAXS * (k + j)];
  }
  if (threadIdx.x < m) fs[threadIdx.x] = S[I_F + k + threadIdx.x];
  __syncthreads();
  if (threadIdx.x == 0)
    for (int i = 0; i < m; ++i) {
      double v = fs[i];
      for (int j = 0; j < i; ++j) v -= Ms[i + IS_MAXS * j] * c[j];
      c[i] = v / Ms[i + IS_MAXS * i];
    }
  __syncthreads();
  const double omega = S[I_OMEGA];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n2; i += stride) {
    if constexpr (M > 0) {
      d2_t g[M], u[M];
#pragma unroll
      for (int t = 0; t < M; ++t) {
        g[t] = KB_LD(L.G[t], i);
        u[t] = KB_LD(L.U[t], i);
      }
      const d2_t rv = KB_LD(r, i);
      d2_t v = c[0] * g[0];
      d2_t q = c[0] * u[0];
#pragma unroll
      for (int t = 1; t < M; ++t) {
        v += c[t] * g[t];
        q += c[t] * u[t];
      }
      v = rv - v;
      Uk[i] = q + omega * v;
    } else {
      d2_t v = c[0] * KB_LD(L.G[0], i);
      d2_t q = c[0] * KB_LD(L.U[0], i);
      for (int t = 1; t < L.m; ++t) {
        v += c[t] * KB_LD(L.G[t], i);
        q += c[t] * KB_LD(L.U[t], i);
      }
      v = KB_LD(r, i) - v;
      Uk[i] = q + omega * v;
    }
  }
}

static std::atomic<int> g_idrs_literal{0};
static std::atomic<int> g_idrs_uniform{0};   // bit 1: P = U(0,1) vectors from mfem_rand, streamed (the default until round 5)
static std::atomic<int> g_idrs_unfused{0};   // bit 2: ki_update and ki_combine as separate kernels (round 5's sequence)
extern "C" int mfem_debug_set_idrs(int bits) try {
  ++mfem_debug_epoch;
  g_idrs_literal = (bits & 1) ? 1 : 0;
  g_idrs_uniform = (bits & 2) ? 1 : 0;
  g_idrs_unfused = (bits & 4) ? 1 : 0;
  return MFEM_OK;
} MFEM_API_CATCH("mfem_debug_set_idrs")

// merged bi-orthogonalisation, scalar part: d = P' g (I_D) -> alpha[0..k) (I_AL), column k of M from row k on, beta = f_k / M_kk   (:62-73)
// f_update != 0 (fused update + combine, round 6): f[k+1:] -= beta M[k+1:, k] happens HERE -- the next step's c is formed by the kernel that applies this
// step's alphas, before ki_step_end runs (which then leaves f alone).  If the stop test of this step ends the solve, f is not used again.
__global__ void ki_ortho(FoldArg fa, int dst, int m, int k, int s, int last, int f_update, double* __restrict__ S, const int32_t* __restrict__ F) {
  // (round 6) the scalars the recurrences below walk through -- f, M, d, alpha -- are mirrored in LDS by the whole workgroup first: thread 0's chain of
  // ~40 dependent reads then costs LDS latency, not a global round trip each (the kernel took 16 us per inner step of idrs!(8); same arithmetic, same bits)
  constexpr int NL = I_AL + IS_MAXS - I_F;
  __shared__ double Lm[NL];
#define LS(i) Lm[(i) - I_F]
  if (F[F_DONE]) return;
  for (int i = threadIdx.x; i < NL; i += blockDim.x) Lm[i] = S[I_F + i];
  kk_fold_dev(fa, S);  // (its barriers also order the copy above)
  if (threadIdx.x != 0) return;
  for (int t = 0; t < m; ++t) {
    const double v = S[I_DOT + t];
    S[dst + t] = v;
    LS(dst + t) = v;
  }
  if (!last) return;  // (more chunks of d to come)
  for (int j = 0; j < k; ++j) {
    double v = LS(I_D + j);
    for (int t = 0; t < j; ++t) v -= LS(I_AL + t) * LS(I_M + j + IS_MAXS * t);
    v = v / LS(I_M + j + IS_MAXS * j);
    LS(I_AL + j) = v;
    S[I_AL + j] = v;
  }
  for (int i = k; i < s; ++i) {
    double v = LS(I_D + i);
    for (int t = 0; t < k; ++t) v -= LS(I_AL + t) * LS(I_M + i + IS_MAXS * t);
    LS(I_M + i + IS_MAXS * k) = v;
    S[I_M + i + IS_MAXS * k] = v;
  }
  const double beta = LS(I_F + k) / LS(I_M + k + IS_MAXS * k);
  S[I_BETA] = beta;
  if (f_update)
    for (int i = k + 1; i < s; ++i) S[I_F + i] = LS(I_F + i) - beta * LS(I_M + i + IS_MAXS * k);
#undef LS
}
// ... vector part: g -= sum alpha_t g_t ; u -= sum alpha_t u_t ; x += beta u ; r -= beta g ; partial sums of r'r over the owned entries
template <int M>  // (M = L.m for 0 .. 8 -- all loads of an index up front, see ki_combine --, -1: the run-time form)
__global__ __launch_bounds__(MFEM_BLOCK) void ki_update(int64_t n2, int64_t n_owned, CombineList L, d2_t* __restrict__ Gk, d2_t* __restrict__ Uk,
                                                         d2_t* __restrict__ x, d2_t* __restrict__ r, const double* __restrict__ S,
                                                         const int32_t* __restrict__ F, double* __restrict__ partials) {
  __shared__ double al[IS_MAXS];
  __shared__ double red[4];
  if (F[F_DONE]) return;
  if (threadIdx.x < L.m) al[threadIdx.x] = S[I_AL + threadIdx.x];
  __syncthreads();
  const double beta = S[I_BETA];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double acc = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n2; i += stride) {
    d2_t g = KB_LD(Gk, i), u = KB_LD(Uk, i);
    d2_t xv, rv;
    if constexpr (M >= 0) {
      d2_t gt[M > 0 ? M : 1], ut[M > 0 ? M : 1];
#pragma unroll
      for (int t = 0; t < M; ++t) {
        gt[t] = KB_LD(L.G[t], i);
        ut[t] = KB_LD(L.U[t], i);
      }
      xv = KB_LD(x, i);
      rv = KB_LD(r, i);
#pragma unroll
      for (int t = 0; t < M; ++t) {
        g -= al[t] * gt[t];
        u -= al[t] * ut[t];
      }
    } else {
      for (int t = 0; t < L.m; ++t) {
        g -= al[t] * KB_LD(L.G[t], i);
        u -= al[t] * KB_LD(L.U[t], i);
      }
      xv = KB_LD(x, i);
      rv = KB_LD(r, i);
    }
    Gk[i] = g;
    Uk[i] = u;
    x[i] = xv + beta * u;
    const d2_t rn = rv - beta * g;
    r[i] = rn;
    if (2 * i < n_owned) acc += rn.x * rn.x;
    if (2 * i + 1 < n_owned) acc += rn.y * rn.y;
  }
  const double b = block_reduce_sum(acc, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = b;
}

// ki_update(k) and ki_combine(k + 1) in ONE pass (round 6): r' = r - beta g is used where it is formed, U_(k+1) leaves with it -- one stream (r) and one
// launch less per inner step.  L lists the T = s - 1 other vector pairs: first the K = k new ones (alphas), then the s - k - 1 old ones of step k + 1's
// combination (c = LowerTriangular(M[k+1:s, k+1:s]) \ f[k+1:s], f already updated by ki_ortho).  T is a template constant for s <= 8 (every load of an
// index issued up front, on every path), -1 = the run-time form.  Same products and sums in the same order as the two kernels: bitwise the same vectors.
template <int T>
__global__ __launch_bounds__(MFEM_BLOCK) void ki_update_combine(int64_t n2, int64_t n_owned, CombineList L, int K, int kn, d2_t* __restrict__ Gk,
                                                                 d2_t* __restrict__ Uk, d2_t* __restrict__ x, d2_t* __restrict__ r, d2_t* Un,
                                                                 const double* __restrict__ S, const int32_t* __restrict__ F, double* __restrict__ partials) {
  __shared__ double al[IS_MAXS];
  __shared__ double c[IS_MAXS];
  __shared__ double Ms[IS_MAXS * IS_MAXS];
  __shared__ double fs[IS_MAXS];
  __shared__ double red[4];
  if (F[F_DONE]) return;
  const int m = L.m - K;  // terms of the next combination (= s - kn)
  if (threadIdx.x < K) al[threadIdx.x] = S[I_AL + threadIdx.x];
  for (int t = threadIdx.x; t < m * m; t += blockDim.x) {
    const int i = t % m, j = t / m;
    Ms[i + IS_MAXS * j] = S[I_M + (kn + i) + IS_MAXS * (kn + j)];
  }
  if (threadIdx.x < m) fs[threadIdx.x] = S[I_F + kn + threadIdx.x];
  __syncthreads();
  if (threadIdx.x == 0)
    for (int i = 0; i < m; ++i) {
      double v = fs[i];
      for (int j = 0; j < i; ++j) v -= Ms[i + IS_MAXS * j] * c[j];
      c[i] = v / Ms[i + IS_MAXS * i];
    }
  __syncthreads();
  const double beta = S[I_BETA], omega = S[I_OMEGA];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double acc = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n2; i += stride) {
    d2_t g = KB_LD(Gk, i), u = KB_LD(Uk, i);
    d2_t v, q;
    if constexpr (T >= 0) {
      d2_t gt[T > 0 ? T : 1], ut[T > 0 ? T : 1];
#pragma unroll
      for (int t = 0; t < T; ++t) {
        gt[t] = KB_LD(L.G[t], i);
        ut[t] = KB_LD(L.U[t], i);
      }
      const d2_t xv = KB_LD(x, i), rv = KB_LD(r, i);
#pragma unroll
      for (int t = 0; t < T; ++t)
        if (t < K) {
          g -= al[t] * gt[t];
          u -= al[t] * ut[t];
        }
      Gk[i] = g;
      Uk[i] = u;
      x[i] = xv + beta * u;
      const d2_t rn = rv - beta * g;
      r[i] = rn;
      if (2 * i < n_owned) acc += rn.x * rn.x;
      if (2 * i + 1 < n_owned) acc += rn.y * rn.y;
      if (m > 0) {
        v = 0.0;  // (0 + c0 g0 = c0 g0: the sums of ki_combine)
        q = 0.0;
#pragma unroll
        for (int t = 0; t < T; ++t)
          if (t >= K) {
            v += c[t - K] * gt[t];
            q += c[t - K] * ut[t];
          }
        v = rn - v;
        Un[i] = q + omega * v;
      }
    } else {
      for (int t = 0; t < K; ++t) {
        g -= al[t] * KB_LD(L.G[t], i);
        u -= al[t] * KB_LD(L.U[t], i);
      }
      const d2_t xv = KB_LD(x, i), rv = KB_LD(r, i);
      Gk[i] = g;
      Uk[i] = u;
      x[i] = xv + beta * u;
      const d2_t rn = rv - beta * g;
      r[i] = rn;
      if (2 * i < n_owned) acc += rn.x * rn.x;
      if (2 * i + 1 < n_owned) acc += rn.y * rn.y;
      if (m > 0) {
        v = c[0] * KB_LD(L.G[K], i);
        q = c[0] * KB_LD(L.U[K], i);
        for (int t = 1; t < m; ++t) {
          v += c[t] * KB_LD(L.G[K + t], i);
          q += c[t] * KB_LD(L.U[K + t], i);
        }
        v = rn - v;
        Un[i] = q + omega * v;
      }
    }
  }
  const double b = block_reduce_sum(acc, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = b;
}

#define RC(x)            \
  do {                   \
    int _rc = (x);       \
    if (_rc) return _rc; \
  } while (0)
#define K1F(kernel, ...)                                                               \
  do {                                                                                 \
    hipLaunchKernelGGL(kernel, dim3(1), dim3(MFEM_BLOCK), 0, ctx->stream, __VA_ARGS__); \
    MFEM_CHECK_LAUNCH();                                                               \
  } while (0)
#define K1(kernel, ...)                                                       \
  do {                                                                        \
    hipLaunchKernelGGL(kernel, dim3(1), dim3(1), 0, ctx->stream, __VA_ARGS__); \
    MFEM_CHECK_LAUNCH();                                                      \
  } while (0)

int mfem_idrs_pass(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, KrylovVecs& V, const mfem_solve_options* o,
                   int s, double tol, int64_t n_global, int* iters_out, int* spmv_out) {
  MFEM_REQUIRE(s >= 1 && s <= IS_MAXS, "idrs: 1 <= s <= 32 supported");
  double* S = ctx->d_scalars;
  int32_t* F = ctx->d_flags;
  const int64_t nv = V.nv;
  double* r = V.w[0];
  double* Ar = V.w[1];
  double** P = V.w + 4;
  double** U = V.w + 4 + s;
  double** G = V.w + 4 + 2 * s;
  KK k{ctx, nv, V.n, mfem_vec_grid(ctx, nv), S, F, ctx->stream};
  IdArgs a{1.0 / (double)n_global, tol, o->maxiter, o->fixed_iterations, s};

  RC(mfem_pass_residual(ctx, A, vals, V, r, S + S_RR, spmv_out));  // :27-29
  K1(ki_init, a, S, F);
  // P (:35: FEM_rand, unseeded in the reference): the caller's shadow vectors if given; otherwise (round 6) the +-1 vectors of the seed's sign words, which are
  // never stored -- P' g reads g only (kk_sign_dots) -- unless the literal loop wants them as vectors; mfem_debug_set_idrs(2): U(0,1) vectors, streamed
  const bool user_shadow = ctx->shadow && ctx->shadow_count >= s;
  const bool signs = !user_shadow && !g_idrs_uniform && !g_idrs_literal;
  const bool fused = !g_idrs_literal && !g_idrs_unfused;
  for (int i = 0; i < s; ++i) {
    if (user_shadow)
      MFEM_CHECK_HIP(hipMemcpyAsync(P[i], ctx->shadow + (int64_t)i * V.n, sizeof(double) * V.n, hipMemcpyDeviceToDevice, ctx->stream));
    else if (g_idrs_uniform)
      RC(mfem_rand(ctx, V.n, o->seed, (uint32_t)i, P[i]));
    else if (!signs) {
      hipLaunchKernelGGL(kk_sign_vector, dim3(k.G), dim3(MFEM_BLOCK), 0, ctx->stream, V.n, o->seed, i, P[i]);
      MFEM_CHECK_LAUNCH();
    }
    MFEM_CHECK_HIP(hipMemsetAsync(U[i], 0, sizeof(double) * nv, ctx->stream));
    MFEM_CHECK_HIP(hipMemsetAsync(G[i], 0, sizeof(double) * nv, ctx->stream));
  }
  MFEM_CHECK_HIP(hipMemsetAsync(Ar, 0, sizeof(double) * nv, ctx->stream));

  const int check = o->check_every > 0 ? o->check_every : 32;
  int since_poll = 0, host_iter = 1;
  RC(mfem_read_flags(ctx));
  uint64_t key = mfem_hash(MFEM_HASH_SEED, (int)MFEM_SOLVER_IDRS);
  key = mfem_hash(key, s); key = mfem_csr_graph_key(key, A); key = mfem_hash(key, vals); key = mfem_hash(key, V.w[0]);
  key = mfem_hash(key, V.x); key = mfem_hash(key, nv); key = mfem_hash(key, tol); key = mfem_hash(key, n_global);
  key = mfem_hash(key, o->maxiter); key = mfem_hash(key, o->fixed_iterations); key = mfem_hash(key, g_idrs_literal);
  key = mfem_hash(key, (int)signs); key = mfem_hash(key, (int)fused); key = mfem_hash(key, o->seed);
  int dummy_spmv = 0;
  // one IDR cycle = s steps in G_j + the step into G_j+1: s + 1 SpMVs, constant kernel arguments
  auto cycle = [&](int* spmv_cnt) -> int {
    // f = P' r  (:43-45)
    for (int i0 = 0; i0 < s; i0 += KK_MAX_DOTS) {
      DotList L;
      L.m = (s - i0) < KK_MAX_DOTS ? (s - i0) : KK_MAX_DOTS;
      FoldArg fa;
      if (signs) {
        RC(k.sign_dots_partials(o->seed, i0, L.m, r, I_DOT, &fa));
      } else {
        for (int t = 0; t < L.m; ++t) {
          L.x[t] = (const d2_t*)P[i0 + t];
          L.y[t] = (const d2_t*)r;
        }
        RC(k.dots_partials(L, I_DOT, &fa));
      }
      K1F(ki_store, fa, I_F + i0, L.m, -1, S, F);
    }
    for (int kk = 0; kk < s; ++kk) {
      CombineList C;
      C.m = s - kk;
      for (int t = 0; t < C.m; ++t) {
        C.G[t] = (const d2_t*)G[kk + t];
        C.U[t] = (const d2_t*)U[kk + t];
      }
      if (!fused || kk == 0) {  // (fused form: U_kk came out of the previous step's ki_update_combine)
#define KI_COMBINE(M_) case M_: hipLaunchKernelGGL(ki_combine<M_>, dim3(k.G), dim3(MFEM_BLOCK), 0, ctx->stream, nv / 2, C, kk, (const d2_t*)r, (d2_t*)U[kk], S, F); break;
      switch (C.m) {
        KI_COMBINE(1) KI_COMBINE(2) KI_COMBINE(3) KI_COMBINE(4) KI_COMBINE(5) KI_COMBINE(6) KI_COMBINE(7) KI_COMBINE(8)
        default: hipLaunchKernelGGL(ki_combine<0>, dim3(k.G), dim3(MFEM_BLOCK), 0, ctx->stream, nv / 2, C, kk, (const d2_t*)r, (d2_t*)U[kk], S, F); break;
      }
#undef KI_COMBINE
      MFEM_CHECK_LAUNCH();
      }
      RC(k.spmv(A, vals, U[kk], G[kk], spmv_cnt));  // :59
      if (g_idrs_literal) {
        for (int i = 0; i < kk; ++i) {                 // bi-orthogonalise (:62-66)
          FoldArg fa;
          RC(k.dot1_partials(P[i], G[kk], I_DOT, &fa));
          K1F(ki_alpha, fa, i, S, F);
          RC(k.axpy2(coef_dev(I_ALPHA, -1.0), G[i], G[kk], coef_dev(I_ALPHA, -1.0), U[i], U[kk]));
        }
        for (int i0 = kk; i0 < s; i0 += KK_MAX_DOTS) {  // M[k:s, k] = P[k:s]' G[k]  (:69-71)
          DotList L;
          L.m = (s - i0) < KK_MAX_DOTS ? (s - i0) : KK_MAX_DOTS;
          for (int t = 0; t < L.m; ++t) {
            L.x[t] = (const d2_t*)P[i0 + t];
            L.y[t] = (const d2_t*)G[kk];
          }
          FoldArg fa;
          RC(k.dots_partials(L, I_DOT, &fa));
          K1F(ki_store, fa, I_M + i0 + IS_MAXS * kk, L.m, (i0 + KK_MAX_DOTS >= s) ? kk : -1, S, F);
        }
        RC(k.axpy2(coef_dev(I_BETA), U[kk], V.x, coef_dev(I_BETA, -1.0), G[kk], r));  // :75-76
        FoldArg fe;
        RC(k.dot1_partials(r, r, I_DOT, &fe));
        K1F(ki_step_end, fe, a, kk, S, F);
      } else {
        // merged form (see the head of this file): d = P' G_k in one pass, the scalars, one vector kernel
        for (int i0 = 0; i0 < s; i0 += KK_MAX_DOTS) {
          DotList L;
          L.m = (s - i0) < KK_MAX_DOTS ? (s - i0) : KK_MAX_DOTS;
          FoldArg fa;
          if (signs) {
            RC(k.sign_dots_partials(o->seed, i0, L.m, G[kk], I_DOT, &fa));
          } else {
            for (int t = 0; t < L.m; ++t) {
              L.x[t] = (const d2_t*)P[i0 + t];
              L.y[t] = (const d2_t*)G[kk];
            }
            RC(k.dots_partials(L, I_DOT, &fa));
          }
          K1F(ki_ortho, fa, I_D + i0, L.m, kk, s, (i0 + KK_MAX_DOTS >= s) ? 1 : 0, fused ? 1 : 0, S, F);
        }
        double* part = ctx->d_partials;
        if (fused) {
          // this step's alphas and x / r update + the next step's combination in one pass: the s - 1 other pairs, new ones first
          CombineList Q;
          Q.m = s - 1;
          for (int t = 0; t < kk; ++t) {
            Q.G[t] = (const d2_t*)G[t];
            Q.U[t] = (const d2_t*)U[t];
          }
          for (int t = kk + 1; t < s; ++t) {
            Q.G[t - 1] = (const d2_t*)G[t];
            Q.U[t - 1] = (const d2_t*)U[t];
          }
          d2_t* Un = (d2_t*)U[kk + 1 < s ? kk + 1 : kk];  // (not written at the last step: no terms)
#define KI_UC(T_) case T_: hipLaunchKernelGGL(ki_update_combine<T_>, dim3(k.G), dim3(MFEM_BLOCK), 0, ctx->stream, nv / 2, V.n, Q, kk, kk + 1, (d2_t*)G[kk], (d2_t*)U[kk], (d2_t*)V.x, (d2_t*)r, Un, S, F, part); break;
          switch (Q.m) {
            KI_UC(0) KI_UC(1) KI_UC(2) KI_UC(3) KI_UC(4) KI_UC(5) KI_UC(6) KI_UC(7)
            default: hipLaunchKernelGGL(ki_update_combine<-1>, dim3(k.G), dim3(MFEM_BLOCK), 0, ctx->stream, nv / 2, V.n, Q, kk, kk + 1, (d2_t*)G[kk], (d2_t*)U[kk], (d2_t*)V.x, (d2_t*)r, Un, S, F, part); break;
          }
#undef KI_UC
          MFEM_CHECK_LAUNCH();
        } else {
        CombineList Q;
        Q.m = kk;
        for (int t = 0; t < kk; ++t) {
          Q.G[t] = (const d2_t*)G[t];
          Q.U[t] = (const d2_t*)U[t];
        }
#define KI_UPDATE(M_) case M_: hipLaunchKernelGGL(ki_update<M_>, dim3(k.G), dim3(MFEM_BLOCK), 0, ctx->stream, nv / 2, V.n, Q, (d2_t*)G[kk], (d2_t*)U[kk], (d2_t*)V.x, (d2_t*)r, S, F, part); break;
        switch (Q.m) {
          KI_UPDATE(0) KI_UPDATE(1) KI_UPDATE(2) KI_UPDATE(3) KI_UPDATE(4) KI_UPDATE(5) KI_UPDATE(6) KI_UPDATE(7) KI_UPDATE(8)
          default: hipLaunchKernelGGL(ki_update<-1>, dim3(k.G), dim3(MFEM_BLOCK), 0, ctx->stream, nv / 2, V.n, Q, (d2_t*)G[kk], (d2_t*)U[kk], (d2_t*)V.x, (d2_t*)r, S, F, part); break;
        }
#undef KI_UPDATE
        MFEM_CHECK_LAUNCH();
        }
        FoldArg fe{part, k.G, 1, I_DOT};
        if (ctx->comm) {  // (fold + all-reduce here, like KK::dots_partials)
          hipLaunchKernelGGL(kk_fold, dim3(1), dim3(MFEM_BLOCK), 0, ctx->stream, part, k.G, 1, I_DOT, S, F);
          MFEM_CHECK_LAUNCH();
          fe.m = 0;
          RC(mfem_comm_allreduce(ctx, S + I_DOT, 1));
        }
        K1F(ki_step_end, fe, a, kk, S, F, fused ? 1 : 0);
      }
    }
    // r in G_j+1  (:85-93)
    RC(k.spmv(A, vals, r, Ar, spmv_cnt));
    DotList L;
    L.m = 3;
    L.x[0] = (const d2_t*)Ar; L.y[0] = (const d2_t*)Ar;
    L.x[1] = (const d2_t*)r;  L.y[1] = (const d2_t*)r;
    L.x[2] = (const d2_t*)Ar; L.y[2] = (const d2_t*)r;
    FoldArg fo;
    RC(k.dots_partials(L, I_DOT, &fo));
    K1F(ki_omega, fo, S, F);
    RC(k.axpy2(coef_dev(I_OMEGA), r, V.x, coef_dev(I_OMEGA, -1.0), Ar, r));  // x += omega r ; r -= omega Ar
    FoldArg fe;
    RC(k.dot1_partials(r, r, I_DOT, &fe));
    K1F(ki_step_end, fe, a, -1, S, F);
    return MFEM_OK;
  };
  while (!ctx->h_flags[F_DONE]) {
    RC(mfem_cycle_run(ctx, key, [&]() -> int { return cycle(&dummy_spmv); }));
    *spmv_out += s + 1;
    host_iter += s + 1;
    since_poll += s + 1;
    if (since_poll >= check || host_iter >= o->maxiter) {
      RC(mfem_read_flags(ctx));
      since_poll = 0;
    }
  }
  RC(mfem_read_flags(ctx));
  *iters_out = ctx->h_flags[F_ITER];
  return MFEM_OK;
}
