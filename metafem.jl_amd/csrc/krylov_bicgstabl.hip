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
//
// Round 3, fused form (default; mfem_debug_set_bicgstabl(1) runs the literal sequence): the same recurrences with fewer passes over the vectors --
//   * rho1 = r~' R[j] and r~' U[j+1] are produced by the SpMVs that write R[j] / U[j+1] (fused dot of the SpMV kernels), rho1 of the next sweep and
//     R[1]' R[1] by the kernel that ends a sweep;
//   * the vector updates of one BiCG step are one launch each (U[0..j]; R[0..j] and x);
//   * the minimal-residual part never touches the vectors: ONE multi-dot pass gives the Gram matrix of R[0..l]; the modified Gram-Schmidt loop of
//     :64-71 runs on it in a scalar kernel (R'[j] = sum_k c_jk R[k]: tau, sigma, gamma' are bilinear forms of the Gram matrix -- the same numbers in
//     exact arithmetic), and the updates of :82-91 become one kernel with the combined coefficients (x, R[1], U[1], + the two sums above).
//   l = 2: 62 -> 37 vector streams per sweep (4 SpMVs), 10 -> 6 reductions.  Used for l <= 2 (the Gram-matrix form squares the condition of R[1..l]).
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
__global__ void kb_beta(FoldArg fa, double* __restrict__ S, const int32_t* __restrict__ F, int slot = B_DOT) {
  if (F[F_DONE]) return;
  kk_fold_dev(fa, S);
  if (threadIdx.x != 0) return;
  const double rho1 = S[slot];
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

// ---- fused form ------------------------------------------------------------------------------------------------------------------------
static std::atomic<int> g_bicgstabl_literal{0};
extern "C" int mfem_debug_set_bicgstabl(int literal_sequence) try {
  ++mfem_debug_epoch;
  g_bicgstabl_literal = literal_sequence ? 1 : 0;
  return MFEM_OK;
} MFEM_API_CATCH("mfem_debug_set_bicgstabl")
enum {
  B_Z = B_DOT + KK_MAX_DOTS,              // Gram matrix Z[p + (BL_MAXL + 1) * q] = R[p]' R[q], 0 <= p <= q <= l
  B_AX = B_Z + (BL_MAXL + 1) * (BL_MAXL + 1),  // x  += sum_k ax[k] R[k], k = 0..l
  B_AR = B_AX + BL_MAXL + 1,               // R[0] -= sum_k ar[k] R[k], k = 1..l
  B_AU = B_AR + BL_MAXL + 1,               // U[0] -= sum_k au[k] U[k], k = 1..l
  B_END = B_AU + BL_MAXL + 1               // [0] R[1]' R[1], [1] r~' R[1] after the sweep
};
static_assert(B_END + 2 <= MFEM_NSCALARS, "BiCGStab(l) scalars do not fit the device scalar block");

struct VecList {
  d2_t* a[BL_MAXL + 1];
  const d2_t* b[BL_MAXL + 1];
  int m;
};
// U[i] = R[i] - beta U[i], i < m   (:48-50)
__global__ __launch_bounds__(MFEM_BLOCK) void kb_ulist(int64_t n2, VecList L, const double* __restrict__ S, const int32_t* __restrict__ F) {
  if (F[F_DONE]) return;
  const double beta = S[B_BETA];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n2; i += stride)
    for (int t = 0; t < L.m; ++t) L.a[t][i] = KB_LD(L.b[t], i) - beta * KB_LD(L.a[t], i);
}
// R[i] -= alpha U[i+1], i < m ; x += alpha U[0]   (:55-61)
__global__ __launch_bounds__(MFEM_BLOCK) void kb_rlist(int64_t n2, VecList L, const d2_t* __restrict__ U0, d2_t* __restrict__ x,
                                                        const double* __restrict__ S, const int32_t* __restrict__ F) {
  if (F[F_DONE]) return;
  const double alpha = S[B_ALPHA];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n2; i += stride) {
    for (int t = 0; t < L.m; ++t) L.a[t][i] = KB_LD(L.a[t], i) - alpha * KB_LD(L.b[t], i);
    x[i] = KB_LD(x, i) + alpha * KB_LD(U0, i);
  }
}
// the MR part on the Gram matrix (:64-80) and the combined coefficients of the updates (:82-91)
struct ZMap {
  int slot[KK_MAX_DOTS];  // where dot product t of a pass belongs in S
  int m;
};
__global__ void kb_mr(FoldArg fa, ZMap zm, int last, int l, double* __restrict__ S, const int32_t* __restrict__ F) {
  if (F[F_DONE]) return;
  kk_fold_dev(fa, S);
  if (threadIdx.x != 0) return;
  for (int t = 0; t < zm.m; ++t) S[zm.slot[t]] = S[B_DOT + t];
  if (!last) return;
  const int W = BL_MAXL + 1;
  auto Z = [&](int p, int q) { return p <= q ? S[B_Z + p + W * q] : S[B_Z + q + W * p]; };
  double c[BL_MAXL + 1][BL_MAXL + 1];  // R'[j] = sum_k c[j][k] R[k], 1 <= k <= j
  for (int j = 1; j <= l; ++j) {
    for (int k = 1; k <= l; ++k) c[j][k] = k == j ? 1.0 : 0.0;
    for (int i = 1; i < j; ++i) {
      double num = 0.0;
      for (int p = 1; p <= i; ++p)
        for (int q = 1; q <= j; ++q) num += c[i][p] * c[j][q] * Z(p, q);
      const double tau = num / S[B_SIG + i - 1];
      S[B_TAU + (i - 1) + BL_MAXL * (j - 1)] = tau;
      for (int k = 1; k <= i; ++k) c[j][k] -= tau * c[i][k];
    }
    double sig = 0.0, g0 = 0.0;
    for (int p = 1; p <= j; ++p) {
      for (int q = 1; q <= j; ++q) sig += c[j][p] * c[j][q] * Z(p, q);
      g0 += c[j][p] * Z(0, p);
    }
    S[B_SIG + j - 1] = sig;
    S[B_GP + j - 1] = g0 / sig;
  }
  // gamma, omega, gamma''  (:72-80; kb_gamma)
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
  // x += gamma[1] R[1] + sum_{j<l} gamma''[j] R'[j+1] ; R[1] -= sum_j gamma'[j] R'[j+1] ; U[1] -= sum_j gamma[j] U[j+1]   (1-based as in the reference)
  S[B_AX + 0] = S[B_G + 0];
  for (int k = 1; k <= l; ++k) {
    double ax = 0.0, ar = 0.0;
    for (int j = 1; j <= l; ++j) {
      if (j <= l - 1) ax += S[B_GPP + j - 1] * c[j][k];
      ar += S[B_GP + j - 1] * c[j][k];
    }
    S[B_AX + k] = ax;
    S[B_AR + k] = ar;
    S[B_AU + k] = S[B_G + k - 1];
  }
}
struct FinalList {
  const d2_t* R[BL_MAXL + 1];
  const d2_t* U[BL_MAXL + 1];
  int l;
};
// the updates of :82-91 in one pass; partials[blockIdx] = R[1]' R[1], partials[gridDim + blockIdx] = r~' R[1] over the owned entries
__global__ __launch_bounds__(MFEM_BLOCK) void kb_final(int64_t n2, int64_t n_owned, FinalList L, d2_t* __restrict__ R0, d2_t* __restrict__ U0,
                                                        d2_t* __restrict__ x, const d2_t* __restrict__ shadow, const double* __restrict__ S,
                                                        const int32_t* __restrict__ F, double* __restrict__ partials) {
  __shared__ double red[4];
  __shared__ double ax[BL_MAXL + 1], ar[BL_MAXL + 1], au[BL_MAXL + 1];
  if (F[F_DONE]) return;
  if (threadIdx.x <= L.l) {
    ax[threadIdx.x] = S[B_AX + threadIdx.x];
    ar[threadIdx.x] = S[B_AR + threadIdx.x];
    au[threadIdx.x] = S[B_AU + threadIdx.x];
  }
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double a_rr = 0.0, a_sr = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n2; i += stride) {
    d2_t r0 = KB_LD(R0, i), u0 = KB_LD(U0, i);
    d2_t xv = KB_LD(x, i) + ax[0] * r0;
    for (int k = 1; k <= L.l; ++k) {
      const d2_t rk = KB_LD(L.R[k], i);
      xv += ax[k] * rk;
      r0 -= ar[k] * rk;
      u0 -= au[k] * KB_LD(L.U[k], i);
    }
    x[i] = xv;
    R0[i] = r0;
    U0[i] = u0;
    const d2_t sh = KB_LD(shadow, i);
    if (2 * i < n_owned) { a_rr += r0.x * r0.x; a_sr += sh.x * r0.x; }
    if (2 * i + 1 < n_owned) { a_rr += r0.y * r0.y; a_sr += sh.y * r0.y; }
  }
  const double b0 = block_reduce_sum(a_rr, red);
  const double b1 = block_reduce_sum(a_sr, red);
  if (threadIdx.x == 0) {
    partials[blockIdx.x] = b0;
    partials[gridDim.x + blockIdx.x] = b1;
  }
}
// iter += l; stop test on R[1] (:93-94); S[B_END] = R[1]' R[1], S[B_END + 1] = r~' R[1]
__global__ void kb_sweep_end2(FoldArg fa, BlArgs a, double* __restrict__ S, int32_t* __restrict__ F) {
  if (F[F_DONE]) return;
  kk_fold_dev(fa, S);
  if (threadIdx.x != 0) return;
  const int iter = F[F_ITER] + a.l;
  F[F_ITER] = iter;
  S[S_RR] = S[B_END];
  if ((!a.fixed && sqrt(S[B_END] * a.n_inv) <= a.tol) || iter >= a.maxiter) F[F_DONE] = 1;
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
  RC(mfem_pass_residual(ctx, A, vals, V, R[0], S + S_RR, spmv_out));
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
  // ---- the fused form of the same sweep (see the head of this file)
  auto reduce_to = [&](double* part, int G, int m, int out, FoldArg* fa) -> int {  // partial sums -> S[out..out+m): by the consuming scalar kernel, or --
    *fa = FoldArg{part, G, m, out};                                               // with a communicator -- folded and all-reduced here
    if (ctx->comm) {
      hipLaunchKernelGGL(kk_fold, dim3(1), dim3(MFEM_BLOCK), 0, ctx->stream, part, G, m, out, S, F);
      MFEM_CHECK_LAUNCH();
      fa->m = 0;
      return mfem_comm_allreduce(ctx, S + out, m);
    }
    return MFEM_OK;
  };
  auto spmv_dot = [&](double* xin, double* yout, const double* dotw, int out, FoldArg* fa) -> int {  // y = A x ; S[out] = dotw' y
    int np = 0;
    ++*spmv_cnt;
    RC(mfem_spmv_halo(ctx, A, vals, xin, yout, 1.0, 0.0, dotw, dotw ? ctx->d_partials : nullptr, &np, F));
    *fa = FoldArg{nullptr, 0, 0, out};
    return dotw ? reduce_to(ctx->d_partials, np, 1, out, fa) : MFEM_OK;
  };
  // first: the first sweep of a pass, r~' R[0] is not known yet (later sweeps get it from the kernel that ended the sweep before)
  auto sweep_fused = [&](bool first) -> int {
    K1(kb_sweep_begin, S, F);
    FoldArg fa{nullptr, 0, 0, B_DOT};
    for (int j = 0; j < l; ++j) {
      if (j == 0 && first) {
        RC(k.dot1_partials(shadow, R[0], B_DOT, &fa));
        K1F(kb_beta, fa, S, F, B_DOT);
      } else if (j == 0) {
        FoldArg none{nullptr, 0, 0, B_END + 1};
        K1F(kb_beta, none, S, F, B_END + 1);
      } else {
        K1F(kb_beta, fa, S, F, B_DOT);  // (r~' R[j] came with the SpMV that wrote R[j])
      }
      VecList LU;
      LU.m = j + 1;
      for (int i = 0; i <= j; ++i) { LU.a[i] = (d2_t*)U[i]; LU.b[i] = (const d2_t*)R[i]; }
      hipLaunchKernelGGL(kb_ulist, dim3(k.G), dim3(MFEM_BLOCK), 0, ctx->stream, nv / 2, LU, S, F);
      MFEM_CHECK_LAUNCH();
      RC(spmv_dot(U[j], U[j + 1], shadow, B_DOT, &fa));
      K1F(kb_alpha, fa, S, F);
      VecList LR;
      LR.m = j + 1;
      for (int i = 0; i <= j; ++i) { LR.a[i] = (d2_t*)R[i]; LR.b[i] = (const d2_t*)U[i + 1]; }
      hipLaunchKernelGGL(kb_rlist, dim3(k.G), dim3(MFEM_BLOCK), 0, ctx->stream, nv / 2, LR, (const d2_t*)U[0], (d2_t*)V.x, S, F);
      MFEM_CHECK_LAUNCH();
      RC(spmv_dot(R[j], R[j + 1], (j + 1 < l) ? shadow : nullptr, B_DOT, &fa));
    }
    // Gram matrix of R[0..l]: Z[p][q], 0 <= p <= q <= l, q >= 1, in passes of KK_MAX_DOTS dot products; the scalar kernel of the last pass runs the MR part
    {
      int npairs = 0, pp[(BL_MAXL + 1) * (BL_MAXL + 2) / 2], qq[(BL_MAXL + 1) * (BL_MAXL + 2) / 2];
      for (int q = 1; q <= l; ++q)
        for (int p2 = 0; p2 <= q; ++p2) { pp[npairs] = p2; qq[npairs] = q; ++npairs; }
      for (int i0 = 0; i0 < npairs; i0 += KK_MAX_DOTS) {
        DotList L;
        ZMap zm;
        L.m = zm.m = (npairs - i0) < KK_MAX_DOTS ? (npairs - i0) : KK_MAX_DOTS;
        for (int t = 0; t < L.m; ++t) {
          L.x[t] = (const d2_t*)R[pp[i0 + t]];
          L.y[t] = (const d2_t*)R[qq[i0 + t]];
          zm.slot[t] = B_Z + pp[i0 + t] + (BL_MAXL + 1) * qq[i0 + t];
        }
        FoldArg fz;
        RC(k.dots_partials(L, B_DOT, &fz));
        K1F(kb_mr, fz, zm, (i0 + KK_MAX_DOTS >= npairs) ? 1 : 0, l, S, F);
      }
    }
    FinalList FL;
    FL.l = l;
    for (int i = 0; i <= l; ++i) { FL.R[i] = (const d2_t*)R[i]; FL.U[i] = (const d2_t*)U[i]; }
    hipLaunchKernelGGL(kb_final, dim3(k.G), dim3(MFEM_BLOCK), 0, ctx->stream, nv / 2, V.n, FL, (d2_t*)R[0], (d2_t*)U[0], (d2_t*)V.x, (const d2_t*)shadow,
                       S, F, ctx->d_partials);
    MFEM_CHECK_LAUNCH();
    FoldArg fe;
    RC(reduce_to(ctx->d_partials, k.G, 2, B_END, &fe));
    K1F(kb_sweep_end2, fe, a, S, F);
    return MFEM_OK;
  };
  bool first_sweep = true;
  // The Gram-matrix form of the minimal-residual part squares the condition of R[1..l] (= A^k r: ever more parallel): fine for l <= 2 (the default and
  // what the elasticity examples use; 1e-11 from the literal loop after three sweeps, also at l = 4), visibly worse at l = 6 (1e-8): larger l keep the
  // literal sequence with its modified Gram-Schmidt on the vectors.
  const bool literal = g_bicgstabl_literal || l > 2;
  key = mfem_hash(key, literal ? 1 : 0);
  while (!ctx->h_flags[F_DONE]) {
    if (literal) {
      RC(mfem_cycle_run(ctx, key, sweep));
    } else if (first_sweep) {  // (its kernel sequence differs from the later sweeps': not a cached cycle)
      RC(sweep_fused(true));
      first_sweep = false;
    } else {
      RC(mfem_cycle_run(ctx, key, [&]() -> int { return sweep_fused(false); }));
    }
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
