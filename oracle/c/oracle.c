/*
 * oracle.c -- C/OpenMP restatement of the reference hot path (TEST INFRASTRUCTURE ONLY).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product (metafem.jl_amd) never does.  It exists because the reference (jxx2/MetaFEM.jl,
 * 100 % Julia on CUDA.jl) has no CPU execution path and cannot run in this image (SURVEY.md
 * F2/F4): this file keeps the reference's ALGORITHM and DATA LAYOUT -- per-element physical
 * basis tables, one operator launch per bilinear term, atomics into the global arrays, a slot
 * table per element -- so that timing it on the host cores is a fair "reference algorithm on
 * CPU" baseline.  It is checked against the numpy oracle in tests/test_oracle_c.py.
 *
 * Restated reference code (paths under src/):
 *   update_BasicElements_3D, inv_Jac_3D, update_Basic_itgval_1_3D   mesh/unstructured_mesh/4_Update_Integrator.jl:2-33,90-154
 *   _Var_Basic, _Kval_Basic, _Res_Basic                               solver/06_FEM_Kernel.jl:1-13,28-45,65-79
 *   assemble_SparseID! (pattern + per-element slot table)             solver/03_GlobalAssembly.jl:77-140
 *   mul! (CSR SpMV), Jacobi_By_Diagonal                               misc/04_GPU_Utils.jl:131; linear_solver/02_Preconditioner.jl:122-130
 *   the added Jacobi-PCG (see oracle/solvers.py::cg)
 * Arrays are column-major like the reference; ids are 0-based.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

void orc_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* First-touch helpers: numpy allocates and fills on the main thread, which homes every page on one NUMA node.
 * Re-homing the big arrays with the same static schedule the kernels use keeps the CPU baseline from being
 * limited by a single memory controller on multi-socket hosts. */
void orc_parallel_copy(void* dst, const void* src, int64_t nbytes) {
  const int64_t n = nbytes / 8;
  int64_t* d = (int64_t*)dst;
  const int64_t* s = (const int64_t*)src;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) d[i] = s[i];
  memcpy((char*)dst + n * 8, (const char*)src + n * 8, (size_t)(nbytes - n * 8));
}
void orc_parallel_zero(void* dst, int64_t nbytes) {
  const int64_t n = nbytes / 8;
  int64_t* d = (int64_t*)dst;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) d[i] = 0;
  memset((char*)dst + n * 8, 0, (size_t)(nbytes - n * 8));
}

/* ---- geometry: integral_vals[q, a, s, e] (s = 0 value, 1..3 d/dx_s), integral_weights[q, e] ------- */
/* ref_vals[q, a, s] (s = 0 value, 1..3 d/dxi_s), itg_weight[q], coords[ncp*3] (x|y|z SoA), cp_ids[a, e]. */
void orc_update_basic_elements_3d(int itg, int itp, int64_t nel, const double* ref_vals, const double* itg_weight,
                                  const double* coords, int64_t ncp, const int64_t* cp_ids, double* integral_vals,
                                  double* integral_weights) {
#pragma omp parallel for schedule(static)
  for (int64_t e = 0; e < nel; ++e) {
    for (int q = 0; q < itg; ++q) {
      double J[3][3] = {{0}};
      for (int a = 0; a < itp; ++a) {
        const int64_t cp = cp_ids[a + (int64_t)itp * e];
        for (int i = 0; i < 3; ++i) {
          const double xi = coords[cp + (int64_t)i * ncp];
          for (int X = 0; X < 3; ++X) J[i][X] += ref_vals[q + itg * (a + itp * (1 + X))] * xi;
        }
      }
      const double det = J[0][0] * J[1][1] * J[2][2] - J[0][0] * J[1][2] * J[2][1] - J[0][1] * J[1][0] * J[2][2] +
                         J[0][1] * J[1][2] * J[2][0] + J[0][2] * J[1][0] * J[2][1] - J[0][2] * J[1][1] * J[2][0];
      double I[3][3];
      I[0][0] = (J[1][1] * J[2][2] - J[1][2] * J[2][1]) / det;
      I[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) / det;
      I[0][2] = (J[0][1] * J[1][2] - J[1][1] * J[0][2]) / det;
      I[1][0] = (J[1][2] * J[2][0] - J[2][2] * J[1][0]) / det;
      I[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) / det;
      I[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) / det;
      I[2][0] = (J[1][0] * J[2][1] - J[1][1] * J[2][0]) / det;
      I[2][1] = (J[0][1] * J[2][0] - J[2][1] * J[0][0]) / det;
      I[2][2] = (J[0][0] * J[1][1] - J[1][0] * J[0][1]) / det;
      for (int a = 0; a < itp; ++a) {
        double* out = integral_vals + q + (int64_t)itg * (a + (int64_t)itp * 4 * e);
        out[0] = ref_vals[q + itg * a];
        for (int s = 0; s < 3; ++s) {
          double v = 0.0;
          for (int m = 0; m < 3; ++m) v += ref_vals[q + itg * (a + itp * (1 + m))] * I[m][s];
          out[(int64_t)itg * itp * (1 + s)] = v;
        }
      }
      integral_weights[q + (int64_t)itg * e] = itg_weight[q] * det;
    }
  }
}

#define IV(q, a, s, h) itp_vals[(q) + (int64_t)itg * ((a) + (int64_t)itp * ((s) + (int64_t)n_sd * (h)))]

/* _Var_Basic: target[q, t] += sum_a N[q,a,sd,host_t] * x[cp[a, el_t] + shift] */
void orc_var_basic(int itg, int itp, int n_sd, const double* itp_vals, int sd, int64_t shift, const int64_t* el_g_cpIDs,
                   const double* x, double* target, const int64_t* host_ids, const int64_t* el_ids, int64_t nt) {
#pragma omp parallel for schedule(static)
  for (int64_t t = 0; t < nt; ++t) {
    const int64_t e = el_ids[t], h = host_ids[t];
    for (int a = 0; a < itp; ++a) {
      const double xv = x[el_g_cpIDs[a + (int64_t)itp * e] + shift];
      for (int q = 0; q < itg; ++q) target[q + (int64_t)itg * t] += IV(q, a, sd, h) * xv;
    }
  }
}

/* _Kval_Basic: K[slot[a,b,el_t] + shift] += sum_q N[q,a,dsd] N[q,b,bsd] vals[q,t]   (FP64 atomics) */
void orc_kval_basic(int itg, int itp, int n_sd, const double* itp_vals, int dual_sd, int base_sd, const double* vals,
                    const int64_t* sparse_ids_by_el, int64_t shift, double* K_val, const int64_t* host_ids,
                    const int64_t* el_ids, int64_t nt) {
#pragma omp parallel for schedule(static)
  for (int64_t t = 0; t < nt; ++t) {
    const int64_t e = el_ids[t], h = host_ids[t];
    for (int a = 0; a < itp; ++a)
      for (int b = 0; b < itp; ++b) {
        double sum = 0.0;
        for (int q = 0; q < itg; ++q) sum += IV(q, a, dual_sd, h) * IV(q, b, base_sd, h) * vals[q + (int64_t)itg * t];
        double* dst = K_val + sparse_ids_by_el[a + (int64_t)itp * (b + (int64_t)itp * e)] + shift;
#pragma omp atomic
        *dst += sum;
      }
  }
}

/* _Res_Basic: residue[cp[a, el_t] + shift] += sum_q N[q,a,dsd] vals[q,t] */
void orc_res_basic(int itg, int itp, int n_sd, const double* itp_vals, int dual_sd, const double* vals, int64_t shift,
                   const int64_t* el_g_cpIDs, double* residue, const int64_t* host_ids, const int64_t* el_ids,
                   int64_t nt) {
#pragma omp parallel for schedule(static)
  for (int64_t t = 0; t < nt; ++t) {
    const int64_t e = el_ids[t], h = host_ids[t];
    for (int a = 0; a < itp; ++a) {
      double sum = 0.0;
      for (int q = 0; q < itg; ++q) sum += IV(q, a, dual_sd, h) * vals[q + (int64_t)itg * t];
      double* dst = residue + el_g_cpIDs[a + (int64_t)itp * e] + shift;
#pragma omp atomic
      *dst += sum;
    }
  }
}

/* vals[q, t] = coeff * w[q, host_t] (the generated `vals = @. coeff * K_params * w[:, ids]` broadcast) */
void orc_scale_weights(int itg, double coeff, const double* w, const int64_t* host_ids, int64_t nt, double* vals) {
#pragma omp parallel for schedule(static)
  for (int64_t t = 0; t < nt; ++t)
    for (int q = 0; q < itg; ++q) vals[q + (int64_t)itg * t] = coeff * w[q + (int64_t)itg * host_ids[t]];
}

/* ---- pattern: unique (cp_i, cp_j) pairs of one field block -> row-sorted CSR + per-element slots ---- */
static int cmp_i64(const void* a, const void* b) {
  const int64_t x = *(const int64_t*)a, y = *(const int64_t*)b;
  return (x > y) - (x < y);
}

/* Pass 1 (rowptr == NULL on input colidx): returns nnz.  Caller allocates colidx[nnz] and slots[itp*itp*nel]. */
int64_t orc_pattern(int itp, int64_t nel, int64_t ncp, const int64_t* cp_ids, int64_t* rowptr /* ncp+1 */,
                    int32_t* colidx, int64_t* sparse_ids_by_el) {
  /* count with duplicates */
  int64_t* cnt = (int64_t*)calloc((size_t)ncp + 1, sizeof(int64_t));
  for (int64_t e = 0; e < nel; ++e)
    for (int a = 0; a < itp; ++a) cnt[cp_ids[a + (int64_t)itp * e] + 1] += itp;
  for (int64_t i = 0; i < ncp; ++i) cnt[i + 1] += cnt[i];
  int64_t* fill = (int64_t*)malloc((size_t)ncp * sizeof(int64_t));
  memcpy(fill, cnt, (size_t)ncp * sizeof(int64_t));
  int64_t* dup = (int64_t*)malloc((size_t)cnt[ncp] * sizeof(int64_t));
  for (int64_t e = 0; e < nel; ++e)
    for (int a = 0; a < itp; ++a) {
      const int64_t r = cp_ids[a + (int64_t)itp * e];
      for (int b = 0; b < itp; ++b) dup[fill[r]++] = cp_ids[b + (int64_t)itp * e];
    }
  rowptr[0] = 0;
#pragma omp parallel for schedule(dynamic, 1024)
  for (int64_t r = 0; r < ncp; ++r) {
    int64_t* p = dup + cnt[r];
    const int64_t m = cnt[r + 1] - cnt[r];
    qsort(p, (size_t)m, sizeof(int64_t), cmp_i64);
    int64_t u = 0;
    for (int64_t i = 0; i < m; ++i)
      if (i == 0 || p[i] != p[i - 1]) p[u++] = p[i];
    fill[r] = u; /* unique count */
  }
  for (int64_t r = 0; r < ncp; ++r) rowptr[r + 1] = rowptr[r] + fill[r];
  const int64_t nnz = rowptr[ncp];
  if (colidx) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < ncp; ++r)
      for (int64_t i = 0; i < fill[r]; ++i) colidx[rowptr[r] + i] = (int32_t)dup[cnt[r] + i];
    if (sparse_ids_by_el) {
#pragma omp parallel for schedule(static)
      for (int64_t e = 0; e < nel; ++e)
        for (int a = 0; a < itp; ++a) {
          const int64_t r = cp_ids[a + (int64_t)itp * e];
          const int32_t* row = colidx + rowptr[r];
          const int64_t len = rowptr[r + 1] - rowptr[r];
          for (int b = 0; b < itp; ++b) {
            const int32_t c = (int32_t)cp_ids[b + (int64_t)itp * e];
            int64_t lo = 0, hi = len - 1;
            while (lo < hi) {
              const int64_t mid = (lo + hi) >> 1;
              if (row[mid] < c) lo = mid + 1; else hi = mid;
            }
            sparse_ids_by_el[a + (int64_t)itp * (b + (int64_t)itp * e)] = rowptr[r] + lo;
          }
        }
    }
  }
  free(cnt); free(fill); free(dup);
  return nnz;
}

/* ---- linear algebra ------------------------------------------------------------------------------- */
void orc_spmv(int64_t n, const int64_t* rowptr, const int32_t* col, const double* vals, const double* x, double* y,
              double alpha, double beta) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < n; ++r) {
    double s = 0.0;
    for (int64_t j = rowptr[r]; j < rowptr[r + 1]; ++j) s += vals[j] * x[col[j]];
    y[r] = (beta == 0.0) ? alpha * s : alpha * s + beta * y[r];
  }
}

static double dotp(int64_t n, const double* a, const double* b) {
  double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
  for (int64_t i = 0; i < n; ++i) s += a[i] * b[i];
  return s;
}

void orc_jacobi_by_diagonal(int64_t n, const int64_t* rowptr, const int32_t* col, const double* vals, double* d) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < n; ++r) {
    d[r] = 1.0;
    for (int64_t j = rowptr[r]; j < rowptr[r + 1]; ++j)
      if (col[j] == r) d[r] = fabs(vals[j]);
  }
}

/* Jacobi-PCG, x0 = given x; stop: ||r||/sqrt(n) <= tol or iters >= maxiter (fixed != 0: exactly maxiter).
 * Returns the iteration count; *final_res = ||r||/sqrt(n) from the recurrence. */
int orc_cg_jacobi(int64_t n, const int64_t* rowptr, const int32_t* col, const double* vals, const double* b, double* x,
                  double tol, int maxiter, int fixed, double* final_res) {
  double* r = (double*)malloc(sizeof(double) * (size_t)n);
  double* p = (double*)malloc(sizeof(double) * (size_t)n);
  double* Ap = (double*)malloc(sizeof(double) * (size_t)n);
  double* dinv = (double*)malloc(sizeof(double) * (size_t)n);
  orc_jacobi_by_diagonal(n, rowptr, col, vals, dinv);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) dinv[i] = 1.0 / dinv[i];
  orc_spmv(n, rowptr, col, vals, x, r, -1.0, 0.0);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    r[i] += b[i];
    p[i] = r[i] * dinv[i];
  }
  double rz = dotp(n, r, p), rr = dotp(n, r, r);
  int it = 0;
  if (fixed || sqrt(rr / (double)n) > tol) {
    while (it < maxiter) {
      orc_spmv(n, rowptr, col, vals, p, Ap, 1.0, 0.0);
      const double alpha = rz / dotp(n, p, Ap);
      double rz_new = 0.0;
      rr = 0.0;
#pragma omp parallel for reduction(+ : rz_new, rr) schedule(static)
      for (int64_t i = 0; i < n; ++i) {
        x[i] += alpha * p[i];
        const double ri = r[i] - alpha * Ap[i];
        r[i] = ri;
        rz_new += ri * ri * dinv[i];
        rr += ri * ri;
      }
      ++it;
      if ((!fixed && sqrt(rr / (double)n) <= tol) || it >= maxiter) break;
      const double beta = rz_new / rz;
      rz = rz_new;
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i) p[i] = r[i] * dinv[i] + beta * p[i];
    }
  }
  if (final_res) *final_res = sqrt(rr / (double)n);
  free(r); free(p); free(Ap); free(dinv);
  return it;
}
