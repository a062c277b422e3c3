// bicgstabl_GS!  -- BiCGStab(l) with the Gram-Schmidt minimal-residual part
// (reference linear_solver/03_BiCGstabl.jl:18-96; `s` kwarg = l, default 2).
//
// Same recurrences, same operation order as the reference; what changes is where the scalars live.
// The reference reads every dot product back to the host (10 synced reductions per l = 2 sweep);
// here rho, alpha, beta, omega, tau, sigma, gamma', gamma, gamma'' stay in ctx->d_scalars, small
// single-thread "scalar step" kernels advance them, and independent reductions of one phase are
// batched into one pass over memory (the MR part needs sigma_j and gamma'_j together; the BiCG part
// needs one dot at a time).  The stop test (normalized_norm(R[1]) <= tol || iter >= maxiter, :94) is
// evaluated on device into the DONE flag.
#include "krylov_kernels.h"

#include "rng.h"

#define BL_MAXL 8
// scalar slots (relative to S_SOLVER)
enum {
  B_RHO0 = S_SOLVER + 0, B_OMEGA, B_ALPHA, B_BETA, B_RHO1, B_SIGMA, B_RNORM2, B_SPARE,
  B_G = S_SOLVER + 8,            // gamma   [BL_MAXL]
  B_GP = B_G + BL_MAXL,          // gamma'  [BL_MAXL]
  B_GPP = B_GP + BL_MAXL,        // gamma'' [BL_MAXL]
  B_SIG = B_GPP + BL_MAXL,       // sigma   [BL_MAXL]
  B_TAU = B_SIG + BL_MAXL,       // tau     [BL_MAXL * BL_MAXL], tau[i + BL_MAXL*j]
  B_DOT = B_TAU + BL_MAXL * BL_MAXL  // scratch for batched dots [KK_MAX_DOTS]
};

struct BlArgs {
  double n_inv, tol;
  int32_t maxiter, fixed, l;
};

__global__ void kb_init(BlArgs a, double* __restrict__ S, int32_t* __restrict__ F) {
  // after r = b - A x: S[S_RR] = r.r
  S[B_OMEGA] = 1.0;
  S[B_RHO0] = 1.0;
  S[B_ALPHA] = 0.0;
  F[F_ITER] = 1;  // "iter = 1" (:25)
  const bool conv = !a.fixed && sqrt(S[S_RR] * a.n_inv) <= a.tol;
  F[F_DONE] = conv ? 1 : 0;
  if (conv) F[F_ITER] = 0;  // "(r_norm <= tol) && return 0" (:24)
}

__global__ void kb_sweep_begin(double* __restrict__ S, const int32_t* __restrict__ F) {
  if (F[F_DONE]) return;
  S[B_RHO0] *= -S[B_OMEGA];  // :41
}
// rho1 = dot(r_shadow, R[j]) in S[B_DOT]; beta = alpha*rho1/rho0; rho0 = rho1  (:44-46)
__global__ void kb_beta(FoldArg fa, double* __restrict__ S, const int32_t* __restrict__ F) {
  if (F[F_DONE]) return;
  kk_fold_dev(fa, S);
  if (threadIdx.x != 0) return;
  const double rho1 = S[B_DOT];
  S[B_BETA] = S[B_ALPHA] * rho1 / S[B_RHO0];
  S[B_RHO0] = rho1;
}
// alpha = rho0 / dot(r_shadow, U[j+1])  (:53)
__global__ void kb_alpha(FoldArg fa, double* __restrict__ S, const int32_t* __restrict__ F) {
  if (F[F_DONE]) return;
  kk_fold_dev(fa, S);
  if (threadIdx.x != 0) return;
  S[B_ALPHA] = S[B_RHO0] / S[B_DOT];
}
// tau[i,j] = dot(R[i+1], R[j+1]) / sigma[i]  (:66)
__global__ void kb_tau(FoldArg fa, int i, int j, double* __restrict__ S, const int32_t* __restrict__ F) {
  if (F[F_DONE]) return;
  kk_fold_dev(fa, S);
  if (threadIdx.x != 0) return;
  S[B_TAU + i + BL_MAXL * j] = S[B_DOT] / S[B_SIG + i];
}
// sigma[j] = dot(R[j+1],R[j+1]) ; gamma'[j] = dot(R[1],R[j+1]) / sigma[j]  (:69-70)
__global__ void kb_sigma(FoldArg fa, int j, double* __restrict__ S, const int32_t* __restrict__ F) {
  if (F[F_DONE]) return;
  kk_fold_dev(fa, S);
  if (threadIdx.x != 0) return;
  S[B_SIG + j] = S[B_DOT];
  S[B_GP + j] = S[B_DOT + 1] / S[B_DOT];
}
// gamma, omega, gamma''  (:72-80)
__global__ void kb_gamma(int l, double* __restrict__ S, const int32_t* __restrict__ F) {
  if (F[F_DONE]) return;
  S[B_G + l - 1] = S[B_GP + l - 1];
  S[B_OMEGA] = S[B_G + l - 1];
  for (int j = l - 2; j >= 0; --j) {
    double d = 0.0;
    for (int k = j + 1; k < l; ++k) d += S[B_TAU + j + BL_MAXL * k] * S[B_G + k];
    S[B_G + j] = S[B_GP + j] - d;
  }
  for (int j = 0; j < l - 1; ++j) {
    double d = 0.0;
    for (int k = j + 1; k < l - 1; ++k) d += S[B_TAU + j + BL_MAXL * k] * S[B_G + k + 1];
    S[B_GPP + j] = S[B_G + j + 1] + d;
  }
}
// iter += l; stop test on R[1] (:93-94); S[B_DOT] = R[1].R[1]
__global__ void kb_sweep_end(FoldArg fa, BlArgs a, double* __restrict__ S, int32_t* __restrict__ F) {
  if (F[F_DONE]) return;
  kk_fold_dev(fa, S);
  if (threadIdx.x != 0) return;
  const int iter = F[F_ITER] + a.l;
  F[F_ITER] = iter;
  S[S_RR] = S[B_DOT];
  if ((!a.fixed && sqrt(S[B_DOT] * a.n_inv) <= a.tol) || iter >= a.maxiter) F[F_DONE] = 1;
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

int mfem_bicgstabl_pass(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, KrylovVecs& V,
                        const mfem_solve_options* o, int l, double tol, int64_t n_global, int* iters_out, int* spmv_out) {
  MFEM_REQUIRE(l >= 1 && l <= BL_MAXL, "bicgstabl_GS: 1 <= s <= 8 supported");
  double* S = ctx->d_scalars;
  int32_t* F = ctx->d_flags;
  const int64_t nv = V.nv;
  double** R = V.w;              // R[0..l]
  double** U = V.w + (l + 1);    // U[0..l]
  double* shadow = V.w[2 * (l + 1)];
  KK k{ctx, nv, V.n, mfem_vec_grid(ctx, nv), S, F, ctx->stream};
  BlArgs a{1.0 / (double)n_global, tol, o->maxiter, o->fixed_iterations, l};

  // r = b - A x ; Pl(r) (identity)   (:19-21)
  RC(mfem_true_residual(ctx, A, vals, V.b, V.x, R[0], nv, S + S_RR));
  ++*spmv_out;
  K1(kb_init, a, S, F);
  // r_shadow = FEM_rand (:37); U, R[2..] = FEM_buffer (zeros)
  if (ctx->shadow && ctx->shadow_count >= 1) {
    MFEM_CHECK_HIP(hipMemcpyAsync(shadow, ctx->shadow, sizeof(double) * V.n, hipMemcpyDeviceToDevice, ctx->stream));
  } else {
    RC(mfem_rand(ctx, V.n, o->seed, 0, shadow));
  }
  for (int i = 1; i <= l; ++i) MFEM_CHECK_HIP(hipMemsetAsync(R[i], 0, sizeof(double) * nv, ctx->stream));
  for (int i = 0; i <= l; ++i) MFEM_CHECK_HIP(hipMemsetAsync(U[i], 0, sizeof(double) * nv, ctx->stream));

  const int check = o->check_every > 0 ? o->check_every : 32;
  int sweeps_since_poll = 0;
  RC(mfem_read_flags(ctx));
  int host_iter = 1;
  uint64_t key = mfem_hash(MFEM_HASH_SEED, (int)MFEM_SOLVER_BICGSTABL_GS);
  key = mfem_hash(key, l); key = mfem_csr_graph_key(key, A); key = mfem_hash(key, vals); key = mfem_hash(key, V.w[0]);
  key = mfem_hash(key, V.x); key = mfem_hash(key, nv); key = mfem_hash(key, tol); key = mfem_hash(key, n_global);
  key = mfem_hash(key, o->maxiter); key = mfem_hash(key, o->fixed_iterations);
  int dummy_spmv = 0;
  int* const spmv_cnt = &dummy_spmv;
  auto sweep = [&]() -> int {  // one BiCGStab(l) sweep: 2 l SpMVs, constant kernel arguments
    K1(kb_sweep_begin, S, F);
    // ---- BiCG part (:43-62)
    for (int j = 0; j < l; ++j) {
      FoldArg fa;
      RC(k.dot1_partials(shadow, R[j], B_DOT, &fa));
      K1F(kb_beta, fa, S, F);
      for (int i = 0; i <= j; ++i) RC(k.lin2(coef_imm(1.0), R[i], coef_dev(B_BETA, -1.0), U[i], U[i]));  // U[i] = R[i] - beta U[i]
      RC(k.spmv(A, vals, U[j], U[j + 1], spmv_cnt));
      RC(k.dot1_partials(shadow, U[j + 1], B_DOT, &fa));
      K1F(kb_alpha, fa, S, F);
      for (int i = 0; i <= j; ++i) RC(k.axpby(coef_dev(B_ALPHA, -1.0), U[i + 1], coef_imm(1.0), R[i]));  // R[i] -= alpha U[i+1]
      RC(k.spmv(A, vals, R[j], R[j + 1], spmv_cnt));
      RC(k.axpby(coef_dev(B_ALPHA), U[0], coef_imm(1.0), V.x));  // x += alpha U[1]
    }
    // ---- MR part, modified Gram-Schmidt (:64-71)
    for (int j = 0; j < l; ++j) {
      for (int i = 0; i < j; ++i) {
        FoldArg ft;
        RC(k.dot1_partials(R[i + 1], R[j + 1], B_DOT, &ft));
        K1F(kb_tau, ft, i, j, S, F);
        RC(k.axpby(coef_dev(B_TAU + i + BL_MAXL * j, -1.0), R[i + 1], coef_imm(1.0), R[j + 1]));
      }
      DotList L;
      L.m = 2;
      L.x[0] = (const d2_t*)R[j + 1]; L.y[0] = (const d2_t*)R[j + 1];
      L.x[1] = (const d2_t*)R[0];     L.y[1] = (const d2_t*)R[j + 1];
      FoldArg fs;
      RC(k.dots_partials(L, B_DOT, &fs));
      K1F(kb_sigma, fs, j, S, F);
    }
    K1(kb_gamma, l, S, F);
    // ---- updates (:82-91)
    RC(k.axpby(coef_dev(B_G + 0), R[0], coef_imm(1.0), V.x));                  // x += gamma[1] R[1]
    RC(k.axpby(coef_dev(B_GP + l - 1, -1.0), R[l], coef_imm(1.0), R[0]));      // R[1] -= gamma'[l] R[l+1]
    RC(k.axpby(coef_dev(B_G + l - 1, -1.0), U[l], coef_imm(1.0), U[0]));       // U[1] -= gamma[l] U[l+1]
    for (int j = 0; j < l - 1; ++j) {
      RC(k.axpby(coef_dev(B_G + j, -1.0), U[j + 1], coef_imm(1.0), U[0]));
      RC(k.axpby(coef_dev(B_GPP + j), R[j + 1], coef_imm(1.0), V.x));
      RC(k.axpby(coef_dev(B_GP + j, -1.0), R[j + 1], coef_imm(1.0), R[0]));
    }
    FoldArg fe;
    RC(k.dot1_partials(R[0], R[0], B_DOT, &fe));
    K1F(kb_sweep_end, fe, a, S, F);
    return MFEM_OK;
  };
  while (!ctx->h_flags[F_DONE]) {
    RC(mfem_cycle_run(ctx, key, sweep));
    *spmv_out += 2 * l;
    host_iter += l;
    if (++sweeps_since_poll * l >= check || host_iter >= o->maxiter) {
      RC(mfem_read_flags(ctx));
      sweeps_since_poll = 0;
    }
  }
  RC(mfem_read_flags(ctx));
  *iters_out = ctx->h_flags[F_ITER];
  return MFEM_OK;
}
