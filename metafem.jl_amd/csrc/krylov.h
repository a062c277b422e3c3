// Shared declarations of the Krylov translation units.
#pragma once
#include "blas1.h"

#define MFEM_MAX_S 32

// device scalar slots (ctx->d_scalars)
enum {
  S_RZ0 = 0, S_RZ1 = 1, S_RR = 2, S_PAP = 3, S_TMP0 = 4, S_TMP1 = 5, S_TMP2 = 6, S_TMP3 = 7,
  S_SOLVER = 16  // solver-private block [16, 240)
};
// device flags (ctx->d_flags)
enum { F_DONE = 0, F_ITER = 1, F_AUX = 2 };

struct KrylovVecs {
  int64_t n, nv;
  double* x;
  double* b;
  double* d;           // Jacobi vector (right preconditioner) or |diag| (CG)
  const double* dinv;  // CG only
  double* w[3 * MFEM_MAX_S + 8];
  int nwork;
};

int mfem_fill(mfem_context_s* ctx, int64_t n, double v, double* x);
int mfem_sum_partials(mfem_context_s* ctx, const double* partials, int np, double* d_out);
int mfem_true_residual(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, const double* b, const double* x,
                       double* r, int64_t nv, double* d_rr);
int mfem_read_scalars(mfem_context_s* ctx, int first, int count);
int mfem_read_flags(mfem_context_s* ctx);
int mfem_bicgstabl_pass(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, KrylovVecs& V,
                        const mfem_solve_options* o, int l, double tol, int64_t n_global, int* iters_out, int* spmv_out);
int mfem_cgs2_pass(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, KrylovVecs& V, const mfem_solve_options* o,
                   double tol, int64_t n_global, int* iters_out, int* spmv_out);
int mfem_idrs_pass(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, KrylovVecs& V, const mfem_solve_options* o,
                   int s, double tol, int64_t n_global, int* iters_out, int* spmv_out);
