/*
 * c_abi_smoke.c -- a plain-C consumer of include/metafem_mi355x.h (no Python, no ctypes mirror of the structs).
 *
 * Compiled by gcc (-std=c99) against the header and linked to libmetafem_mi355x.so by tests/test_gpu_c_abi.py; run on the GPU box.
 * It walks the three seams on the 4 x 4 x 4 hex-8 thermal fixture (tests/golden/oracle_thermal_hex8_4x4x4.h: oracle output):
 *   make_Brick + mesh_Classical + assemble_SparseID!   mfem_brick_create / mfem_brick_pattern      pattern == the oracle's CSR
 *   K_linear_func  (solver/01_Types.jl:164)             mfem_brick_assemble_thermal                 K   vs gold_K   (1e-12)
 *   K_nonlinear_func (:165)                             mfem_brick_residual_thermal                 R0  vs gold_R0  (1e-12)
 *   fem_domain.linear_solver (:166) = iterative_Solve!  mfem_solve, bicgstabl_GS! and idrs!         T = -delta vs gold_T (1e-9)
 *   mul! (misc/04_GPU_Utils.jl:131)                     mfem_spmv_csr on a caller-made 1-based Int32 CSR (the reference's arrays)
 * Exit code 0 and a line "C_ABI_SMOKE OK" on success.
 *
 * The four HIP runtime calls it needs are declared by hand so that gcc needs no HIP headers (they come from libamdhip64, which the
 * library itself is linked against).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "metafem_mi355x.h"
#include "golden/oracle_thermal_hex8_4x4x4.h"

extern int hipMalloc(void** ptr, size_t size);
extern int hipFree(void* ptr);
extern int hipMemcpy(void* dst, const void* src, size_t size, int kind); /* 1 = host to device, 2 = device to host */
extern int hipDeviceSynchronize(void);

#define CHECK(call)                                                                              \
  do {                                                                                           \
    int rc_ = (call);                                                                            \
    if (rc_ != 0) {                                                                              \
      fprintf(stderr, "%s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, #call, rc_, mfem_last_error()); \
      return 1;                                                                                  \
    }                                                                                            \
  } while (0)

static double rel_err(const double* a, const double* b, int n) {
  double num = 0.0, den = 0.0;
  for (int i = 0; i < n; ++i) {
    if (fabs(a[i] - b[i]) > num) num = fabs(a[i] - b[i]);
    if (fabs(b[i]) > den) den = fabs(b[i]);
  }
  return num / den;
}

static void* dev_alloc(size_t bytes) {
  void* p = NULL;
  if (hipMalloc(&p, bytes) != 0) {
    fprintf(stderr, "hipMalloc(%zu) failed\n", bytes);
    exit(2);
  }
  return p;
}

int main(void) {
  if (mfem_abi_version() != MFEM_ABI_VERSION) {
    fprintf(stderr, "library ABI %d, header %d\n", mfem_abi_version(), MFEM_ABI_VERSION);
    return 1;
  }
  mfem_context ctx = NULL;
  CHECK(mfem_context_create(0, NULL, &ctx));

  /* ---- mesh + pattern */
  mfem_brick brick = NULL;
  CHECK(mfem_brick_create(ctx, gold_num[0], gold_num[1], gold_num[2], gold_size[0], gold_size[1], gold_size[2], 1, 3, &brick));
  if (mfem_brick_num_controlpoints(brick) != GOLD_N) return 1;
  mfem_csr A = NULL;
  CHECK(mfem_brick_pattern(ctx, brick, 1, &A));
  if (mfem_csr_n(A) != GOLD_N || mfem_csr_nnz(A) != GOLD_NNZ) {
    fprintf(stderr, "pattern %lld x %lld nnz, expected %d x %d\n", (long long)mfem_csr_n(A), (long long)mfem_csr_nnz(A), GOLD_N, GOLD_NNZ);
    return 1;
  }
  static int64_t rowptr[GOLD_N + 1];
  static int32_t colidx[GOLD_NNZ];
  hipMemcpy(rowptr, mfem_csr_rowptr64(A), sizeof(rowptr), 2);
  hipMemcpy(colidx, mfem_csr_colidx(A), sizeof(colidx), 2);
  for (int i = 0; i <= GOLD_N; ++i)
    if (rowptr[i] != gold_rowptr[i]) { fprintf(stderr, "rowptr[%d]\n", i); return 1; }
  for (int i = 0; i < GOLD_NNZ; ++i)
    if (colidx[i] != gold_colidx[i]) { fprintf(stderr, "colidx[%d]\n", i); return 1; }

  /* ---- S2: K_linear_func, K_nonlinear_func */
  mfem_thermal_params p;
  memset(&p, 0, sizeof(p));
  p.k = 0.6; p.h = 25.0; p.Tenv = 293.15; p.robin_faces = 0x3f;
  double *dK = dev_alloc(sizeof(double) * GOLD_NNZ), *dR = dev_alloc(sizeof(double) * GOLD_N);
  double *dx0 = dev_alloc(sizeof(double) * GOLD_N), *ds = dev_alloc(sizeof(double) * GOLD_N), *dd = dev_alloc(sizeof(double) * GOLD_N);
  static double hK[GOLD_NNZ], hR[GOLD_N], hs[GOLD_N], hd[GOLD_N], hz[GOLD_N];
  for (int i = 0; i < GOLD_N; ++i) { hs[i] = 1600.0; hz[i] = 0.0; }
  hipMemcpy(ds, hs, sizeof(hs), 1);
  hipMemcpy(dx0, hz, sizeof(hz), 1);
  CHECK(mfem_brick_assemble_thermal(ctx, brick, A, &p, dK));
  CHECK(mfem_brick_residual_thermal(ctx, brick, &p, dx0, ds, dR));
  CHECK(mfem_context_sync(ctx));
  hipMemcpy(hK, dK, sizeof(hK), 2);
  hipMemcpy(hR, dR, sizeof(hR), 2);
  const double eK = rel_err(hK, gold_K, GOLD_NNZ), eR = rel_err(hR, gold_R0, GOLD_N);
  printf("K rel err %.3e, R0 rel err %.3e\n", eK, eR);
  if (!(eK < 1e-12 && eR < 1e-12)) return 1;

  /* ---- S1: linear_solver = x -> iterative_Solve!(x; Sv_func!, maxiter, max_pass, s); update_OneStep!: x += -(delta) */
  const int methods[2] = {MFEM_SOLVER_BICGSTABL_GS, MFEM_SOLVER_IDRS};
  const int s_arg[2] = {2, 8};
  for (int m = 0; m < 2; ++m) {
    mfem_solve_options o;
    memset(&o, 0, sizeof(o));
    o.method = methods[m];
    o.precond = MFEM_PRECOND_JACOBI_RIGHT_DIAG;
    o.l_or_s = s_arg[m];
    o.maxiter = 2000;
    o.max_pass = 10;
    o.check_every = 1;
    o.converge_tol = 1e-11;
    o.seed = 0x5EED;
    mfem_solve_stats st;
    memset(&st, 0, sizeof(st));
    CHECK(mfem_solve(ctx, A, dK, dR, dd, &o, &st));
    hipMemcpy(hd, dd, sizeof(hd), 2);
    for (int i = 0; i < GOLD_N; ++i) hd[i] = -hd[i];
    const double eT = rel_err(hd, gold_T, GOLD_N);
    printf("solver %d: %d passes, %d iterations, res %.3e, T rel err %.3e\n", methods[m], st.passes, st.iterations, st.final_res, eT);
    if (!(st.converged && eT < 1e-9)) return 1;
  }

  /* ---- mul! on the reference's own array types: 1-based Int32 row pointers / columns, borrowed */
  static int32_t rp1[GOLD_N + 1], ci1[GOLD_NNZ];
  static double hx[GOLD_N], hy[GOLD_N], want[GOLD_N];
  for (int i = 0; i <= GOLD_N; ++i) rp1[i] = (int32_t)gold_rowptr[i] + 1;
  for (int i = 0; i < GOLD_NNZ; ++i) ci1[i] = gold_colidx[i] + 1;
  for (int i = 0; i < GOLD_N; ++i) hx[i] = 1.0 + 0.01 * i;
  for (int r = 0; r < GOLD_N; ++r) {
    double acc = 0.0;
    for (long long j = gold_rowptr[r]; j < gold_rowptr[r + 1]; ++j) acc += gold_K[j] * hx[gold_colidx[j]];
    want[r] = 2.0 * acc;
  }
  int32_t *drp = dev_alloc(sizeof(rp1)), *dci = dev_alloc(sizeof(ci1));
  double *dxx = dev_alloc(sizeof(hx)), *dy = dev_alloc(sizeof(hy)), *dKg = dev_alloc(sizeof(double) * GOLD_NNZ);
  hipMemcpy(drp, rp1, sizeof(rp1), 1);
  hipMemcpy(dci, ci1, sizeof(ci1), 1);
  hipMemcpy(dxx, hx, sizeof(hx), 1);
  hipMemcpy(dKg, gold_K, sizeof(double) * GOLD_NNZ, 1);
  mfem_csr B = NULL;
  CHECK(mfem_csr_create(ctx, GOLD_N, GOLD_NNZ, drp, 32, dci, 1, &B));
  CHECK(mfem_spmv_csr(ctx, B, dKg, dxx, dy, 2.0, 0.0));
  CHECK(mfem_context_sync(ctx));
  hipMemcpy(hy, dy, sizeof(hy), 2);
  const double eY = rel_err(hy, want, GOLD_N);
  printf("mul! rel err %.3e\n", eY);
  if (!(eY < 1e-14)) return 1;
  double nrm = 0.0, dt = 0.0;
  CHECK(mfem_nrm2(ctx, GOLD_N, dy, &nrm));
  CHECK(mfem_dot(ctx, GOLD_N, dy, dy, &dt));
  if (!(fabs(nrm * nrm - dt) <= 1e-12 * dt)) return 1;

  /* ---- error convention: a bad argument returns a negative status and a message, nothing aborts */
  if (mfem_spmv_csr(ctx, NULL, dKg, dxx, dy, 1.0, 0.0) != MFEM_ERR_INVALID || strlen(mfem_last_error()) == 0) return 1;

  CHECK(mfem_csr_destroy(B));
  CHECK(mfem_csr_destroy(A));
  CHECK(mfem_brick_destroy(brick));
  CHECK(mfem_context_destroy(ctx));
  hipFree(dK); hipFree(dR); hipFree(dx0); hipFree(ds); hipFree(dd); hipFree(drp); hipFree(dci); hipFree(dxx); hipFree(dy); hipFree(dKg);
  hipDeviceSynchronize();
  printf("C_ABI_SMOKE OK\n");
  return 0;
}
