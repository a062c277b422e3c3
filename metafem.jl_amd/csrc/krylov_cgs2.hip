// cgs2!  -- CGS with a second shadow vector (reference linear_solver/07_CGS.jl:54-105); this is what
// examples/incompressible_flow/lid_driven_cavity_flow/2D_Script.jl:97 selects.  Same recurrences and order as
// the reference; scalars stay on device, the two dot products of each half-step share one pass over memory and
// the vector updates of each half-step are one fused kernel.  Like the reference it recomputes the true
// residual r = b - A x every iteration (2 SpMV per iteration, :96-98).
#include "krylov_kernels.h"

enum { C_ALPHA = S_SOLVER + 0, C_ALPHABAR, C_BETA, C_BETABAR, C_RHO, C_RHOBAR, C_SIGMA, C_SIGMABAR, C_DOT = S_SOLVER + 8 };

struct C2Args {
  double n_inv, tol;
  int32_t maxiter, fixed;
};

__global__ void kc_init(C2Args a, double* __restrict__ S, int32_t* __restrict__ F) {
  for (int i = C_ALPHA; i <= C_SIGMABAR; ++i) S[i] = 1.0;  // :66
  F[F_ITER] = 1;
  const bool conv = !a.fixed && sqrt(S[S_RR] * a.n_inv) <= a.tol;
  F[F_DONE] = conv ? 1 : 0;
  if (conv) F[F_ITER] = 0;
}
// rho = r.r0, rhobar = r.s0 ; beta = 1/alphabar * rho/sigma ; betabar = 1/alpha * rhobar/sigmabar  (:75-81)
__global__ void kc_betas(FoldArg fa, double* __restrict__ S, const int32_t* __restrict__ F) {
  if (F[F_DONE]) return;
  kk_fold_dev(fa, S);
  if (threadIdx.x != 0) return;
  S[C_RHO] = S[C_DOT];
  S[C_RHOBAR] = S[C_DOT + 1];
  S[C_BETA] = 1.0 / S[C_ALPHABAR] * S[C_RHO] / S[C_SIGMA];
  S[C_BETABAR] = 1.0 / S[C_ALPHA] * S[C_RHOBAR] / S[C_SIGMABAR];
}
// sigma = c.r0, alpha = rho/sigma ; sigmabar = c.s0, alphabar = rhobar/sigmabar  (:87-92)
__global__ void kc_alphas(FoldArg fa, double* __restrict__ S, const int32_t* __restrict__ F) {
  if (F[F_DONE]) return;
  kk_fold_dev(fa, S);
  if (threadIdx.x != 0) return;
  S[C_SIGMA] = S[C_DOT];
  S[C_ALPHA] = S[C_RHO] / S[C_SIGMA];
  S[C_SIGMABAR] = S[C_DOT + 1];
  S[C_ALPHABAR] = S[C_RHOBAR] / S[C_SIGMABAR];
}
// iter += 1 ; stop if normalized_norm(r) <= tol || iter > maxiter  (:100-101); S[S_RR] = r.r
__global__ void kc_end(C2Args a, double* __restrict__ S, int32_t* __restrict__ F) {
  if (F[F_DONE]) return;
  const int iter = F[F_ITER] + 1;
  F[F_ITER] = iter;
  if ((!a.fixed && sqrt(S[S_RR] * a.n_inv) <= a.tol) || iter > a.maxiter) F[F_DONE] = 1;
}
// v = r + beta u ; t = r + betabar s ; w = t + beta (u + betabar w)   (:77,82-83)
__global__ __launch_bounds__(MFEM_BLOCK) void kc_half1(int64_t n2, const d2_t* __restrict__ r, const d2_t* __restrict__ u,
                                                        const d2_t* __restrict__ s, d2_t* __restrict__ v, d2_t* __restrict__ t,
                                                        d2_t* __restrict__ w, const double* __restrict__ S,
                                                        const int32_t* __restrict__ F) {
  if (F[F_DONE]) return;
  const double beta = S[C_BETA], betabar = S[C_BETABAR];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n2; i += stride) {
    const d2_t ri = r[i], ui = u[i];
    v[i] = ri + beta * ui;
    const d2_t ti = ri + betabar * s[i];
    t[i] = ti;
    w[i] = ti + beta * (ui + betabar * w[i]);
  }
}
// s = t - alpha c ; u = v - alphabar c ; x += alpha v + alphabar s   (:89,93,95)
__global__ __launch_bounds__(MFEM_BLOCK) void kc_half2(int64_t n2, const d2_t* __restrict__ t, const d2_t* __restrict__ c,
                                                        const d2_t* __restrict__ v, d2_t* __restrict__ s, d2_t* __restrict__ u,
                                                        d2_t* __restrict__ x, const double* __restrict__ S,
                                                        const int32_t* __restrict__ F) {
  if (F[F_DONE]) return;
  const double alpha = S[C_ALPHA], alphabar = S[C_ALPHABAR];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n2; i += stride) {
    const d2_t ci = c[i], vi = v[i];
    const d2_t si = t[i] - alpha * ci;
    s[i] = si;
    u[i] = vi - alphabar * ci;
    x[i] = x[i] + (alpha * vi + alphabar * si);
  }
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

int mfem_cgs2_pass(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, KrylovVecs& V, const mfem_solve_options* o,
                   double tol, int64_t n_global, int* iters_out, int* spmv_out) {
  double* S = ctx->d_scalars;
  int32_t* F = ctx->d_flags;
  const int64_t nv = V.nv;
  double *r = V.w[0], *r0 = V.w[1], *s0 = V.w[2], *u = V.w[3], *w = V.w[4], *s = V.w[5], *v = V.w[6], *t = V.w[7], *c = V.w[8];
  KK k{ctx, nv, V.n, mfem_vec_grid(ctx, nv), S, F, ctx->stream};
  C2Args a{1.0 / (double)n_global, tol, o->maxiter, o->fixed_iterations};
  RC(mfem_pass_residual(ctx, A, vals, V, r, S + S_RR, spmv_out));
  K1(kc_init, a, S, F);
  MFEM_CHECK_HIP(hipMemcpyAsync(r0, r, sizeof(double) * nv, hipMemcpyDeviceToDevice, ctx->stream));  // r0 = copy(r)
  if (ctx->shadow && ctx->shadow_count >= 1)
    MFEM_CHECK_HIP(hipMemcpyAsync(s0, ctx->shadow, sizeof(double) * V.n, hipMemcpyDeviceToDevice, ctx->stream));
  else
    RC(mfem_rand(ctx, V.n, o->seed, 0, s0));
  for (double* z : {u, w, s, v, t, c}) MFEM_CHECK_HIP(hipMemsetAsync(z, 0, sizeof(double) * nv, ctx->stream));
  const int check = o->check_every > 0 ? o->check_every : 32;
  int since = 0, host_iter = 1;
  RC(mfem_read_flags(ctx));
  uint64_t key = mfem_hash(MFEM_HASH_SEED, (int)MFEM_SOLVER_CGS2);
  key = mfem_csr_graph_key(key, A); key = mfem_hash(key, vals); key = mfem_hash(key, V.w[0]); key = mfem_hash(key, V.x);
  key = mfem_hash(key, V.b); key = mfem_hash(key, nv); key = mfem_hash(key, tol); key = mfem_hash(key, n_global);
  key = mfem_hash(key, o->maxiter); key = mfem_hash(key, o->fixed_iterations);
  int dummy_spmv = 0;
  int* const spmv_cnt = &dummy_spmv;
  auto step = [&]() -> int {  // one CGS2 step: 2 SpMVs + the true-residual SpMV, constant kernel arguments
    DotList L;
    L.m = 2;
    L.x[0] = (const d2_t*)r; L.y[0] = (const d2_t*)r0;
    L.x[1] = (const d2_t*)r; L.y[1] = (const d2_t*)s0;
    FoldArg fa;
    RC(k.dots_partials(L, C_DOT, &fa));
    K1F(kc_betas, fa, S, F);
    hipLaunchKernelGGL(kc_half1, dim3(k.G), dim3(MFEM_BLOCK), 0, ctx->stream, nv / 2, (const d2_t*)r, (const d2_t*)u,
                       (const d2_t*)s, (d2_t*)v, (d2_t*)t, (d2_t*)w, S, F);
    MFEM_CHECK_LAUNCH();
    RC(k.spmv(A, vals, w, c, spmv_cnt));
    L.x[0] = (const d2_t*)c; L.y[0] = (const d2_t*)r0;
    L.x[1] = (const d2_t*)c; L.y[1] = (const d2_t*)s0;
    RC(k.dots_partials(L, C_DOT, &fa));
    K1F(kc_alphas, fa, S, F);
    hipLaunchKernelGGL(kc_half2, dim3(k.G), dim3(MFEM_BLOCK), 0, ctx->stream, nv / 2, (const d2_t*)t, (const d2_t*)c,
                       (const d2_t*)v, (d2_t*)s, (d2_t*)u, (d2_t*)V.x, S, F);
    MFEM_CHECK_LAUNCH();
    // r = b - A x (:96-98).  The kernels below are not DONE-guarded, which is harmless: once DONE is set x no longer
    // changes, so they recompute the same r.
    RC(mfem_true_residual(ctx, A, vals, V.b, V.x, r, nv, S + S_RR));
    K1(kc_end, a, S, F);
    return MFEM_OK;
  };
  while (!ctx->h_flags[F_DONE]) {
    RC(mfem_cycle_run(ctx, key, step));
    *spmv_out += 2;
    ++host_iter;
    if (++since >= check || host_iter > o->maxiter) {
      RC(mfem_read_flags(ctx));
      since = 0;
    }
  }
  RC(mfem_read_flags(ctx));
  *iters_out = ctx->h_flags[F_ITER];
  return MFEM_OK;
}
