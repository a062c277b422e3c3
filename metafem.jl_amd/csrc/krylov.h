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

// 1 / d to ~1 ulp for a normal, non-zero d: v_rcp_f64 (about 24 good bits) + two Newton steps -- 5 instructions where the IEEE
// division sequence takes ~14.  Used where r = z / dinv only feeds the dot products of the z-carrying recurrences.
__device__ __forceinline__ double mfem_recip_nr(double d) {
  double y = __builtin_amdgcn_rcp(d);
  double e = fma(-d, y, 1.0);
  y = fma(y, e, y);
  e = fma(-d, y, 1.0);
  return fma(y, e, y);
}

// what the fused pass 2 + residual update of the lattice tiles (spmv_lat27.hip: k_lat27_gather_cg) needs of a CG iteration: the fields of CgArgs that k_cg_update reads,
// the iteration's scalar bank, the vectors, and p . A p as the partials pass 1 left (mfem_lat27_dot_partials).
struct LatCgUpdate {
  int32_t zrec, cur;
  const double* sw;
  double smax2, gate2, n_inv;
  const double* dinv;
  double* r;
  const double* S;
  const int32_t* flags;
  double* partials2;  // [0, grid) r.z, [grid, 2 grid) r.r
  const double* pap_partials;  // p . A p: np partials (mfem_lat27_dot_partials), or np = 0 and the sum in S[S_PAP]
  int32_t np;
};
bool mfem_lat27_cg_fused(const mfem_context_s* ctx, const mfem_csr_s* A, const double* vals);
const double* mfem_lat27_dot_partials(const mfem_csr_s* A, int* np);
int mfem_lat27_gather_cg_update(mfem_context_s* ctx, mfem_csr_s* A, const LatCgUpdate& U, int grid);

struct KrylovVecs {
  int64_t n, nv;
  double* x;
  double* b;
  double* d;           // Jacobi vector (right preconditioner) or |diag| (CG)
  const double* dinv;  // CG only
  const double* cg_s = nullptr;  // scaled CG (cg_variant 4): S = sqrt|diag|, its extremes (the solver iterates on S^-1 A S^-1)
  double cg_smax = 1.0, cg_smin = 1.0;
  double* w[3 * MFEM_MAX_S + 8];
  int nwork;
  bool x_zero = false;  // x is still the zero vector a solve starts from (x0 = 0, 02_Preconditioner.jl:45): the first pass's r = b - A x is b itself
};

int mfem_fill(mfem_context_s* ctx, int64_t n, double v, double* x);
int mfem_sum_partials(mfem_context_s* ctx, const double* partials, int np, double* d_out);
int mfem_true_residual(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, const double* b, const double* x,
                       double* r, int64_t nv, double* d_rr);
// r = b - A x at the start of a pass; the SpMV is skipped while V.x_zero holds (r = b, r.r summed in the order the SpMV path sums it: the same bits).
// *spmv_out is advanced only when a product ran.
int mfem_pass_residual(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, const KrylovVecs& V, double* r, double* d_rr, int* spmv_out);
int mfem_read_scalars(mfem_context_s* ctx, int first, int count);
int mfem_read_flags(mfem_context_s* ctx);
int mfem_bicgstabl_pass(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, KrylovVecs& V,
                        const mfem_solve_options* o, int l, double tol, int64_t n_global, int* iters_out, int* spmv_out);
int mfem_cgs2_pass(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, KrylovVecs& V, const mfem_solve_options* o,
                   double tol, int64_t n_global, int* iters_out, int* spmv_out);
int mfem_idrs_pass(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, KrylovVecs& V, const mfem_solve_options* o,
                   int s, double tol, int64_t n_global, int* iters_out, int* spmv_out);

// ---- hipGraph replay of one solver cycle -------------------------------------------------------------------------
// The Krylov drivers keep every recurrence scalar on the device and guard their kernels with the DONE flag, so the kernel
// sequence of one cycle (an IDR(s) cycle of s + 1 steps, a BiCGStab(l) sweep, a CGS2 step, a CG iteration pair) has
// constant arguments and no host dependency: it is captured once and replayed.  On small systems (the reference's
// examples: 1e3 - 1e5 unknowns) a cycle is 10 - 200 kernels of a few microseconds each and the host launch rate, not
// the GPU, sets the time per iteration.
inline uint64_t mfem_hash_bytes(uint64_t h, const void* data, size_t bytes) {  // FNV-1a
  const unsigned char* p = (const unsigned char*)data;
  for (size_t i = 0; i < bytes; ++i) h = (h ^ p[i]) * 1099511628211ull;
  return h;
}
template <typename T>
inline uint64_t mfem_hash(uint64_t h, const T& v) { return mfem_hash_bytes(h, &v, sizeof(T)); }
#define MFEM_HASH_SEED 1469598103934665603ull

// Everything about the matrix a captured cycle's kernel arguments depend on.  The pattern is identified by a serial number
// given at creation (never by the address of the handle: a destroyed pattern's address and the caller's value buffer are
// routinely handed out again for the next pattern of the same size) plus the arrays and sizes the launches bake in.
inline uint64_t mfem_csr_graph_key(uint64_t key, const mfem_csr_s* A) {
  key = mfem_hash(key, A->serial); key = mfem_hash(key, A->rowptr); key = mfem_hash(key, A->colidx);
  key = mfem_hash(key, A->n); key = mfem_hash(key, A->nnz); key = mfem_hash(key, A->max_row_nnz);
  key = mfem_hash(key, A->index_base); key = mfem_hash(key, A->ell_vals);
  key = mfem_hash(key, A->ell_bound_mode + 16 * A->sym_bound + 32 * A->symp_bound); key = mfem_hash(key, A->sell_vals); key = mfem_hash(key, A->lat27_vals); key = mfem_hash(key, A->lat8_vals); key = mfem_hash(key, A->symp_vals);
  // the column scaling the lattice-tile kernels apply to x is a kernel argument too: a solve with right Jacobi (dsc = the solve's d) and one without
  // (dsc = null) on the same pattern, values and workspace must not share a captured cycle
  key = mfem_hash(key, A->lat27_dsc); key = mfem_hash(key, A->lat8_dsc);
  // ... and so are the arrays of the skew remainder a tile bind may carry (spmv_rem.hip)
  key = mfem_hash(key, A->rem_active);
  if (A->rem_active) { key = mfem_hash(key, A->rem_nrows); key = mfem_hash(key, A->rem_rows); key = mfem_hash(key, A->rem_col); key = mfem_hash(key, A->rem_val); }
  key = mfem_hash(key, mfem_debug_epoch.load());
  return key;
}

extern std::atomic<int> mfem_graph_comm_broken;
extern std::atomic<long long> mfem_graph_comm_captures;
template <class Body>
inline int mfem_cycle_run(mfem_context_s* ctx, uint64_t key, Body body) {
  if (!ctx->graph_active) return body();
  for (int i = 0; i < MFEM_GRAPH_SLOTS; ++i)
    if (ctx->graph_exec[i] && ctx->graph_key[i] == key) {
      MFEM_CHECK_HIP(hipGraphLaunch(ctx->graph_exec[i], ctx->stream));
      return MFEM_OK;
    }
  const int slot = ctx->graph_next;
  ctx->graph_next = (slot + 1) % MFEM_GRAPH_SLOTS;
  if (ctx->graph_exec[slot]) {
    hipGraphExecDestroy(ctx->graph_exec[slot]);
    ctx->graph_exec[slot] = nullptr;
  }
  MFEM_CHECK_HIP(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
  const int rc = body();
  hipGraph_t g = nullptr;
  const hipError_t e = hipStreamEndCapture(ctx->stream, &g);
  // With a communicator (MFEM_GRAPH_COMM=1) a cycle holds RCCL calls: if recording them does not work here, nothing has been launched yet -- the cycle
  // runs as direct launches, now and for the rest of the process.  (Without a communicator a failed capture is an error, as before.)
  auto comm_fallback = [&](const char* what, hipError_t err) -> int {
    (void)hipGetLastError();
    mfem_graph_comm_broken = 1;
    ctx->graph_active = 0;
    fprintf(stderr, "metafem_mi355x: %s failed with a communicator attached (%s): cycles run as direct launches from now on\n", what, hipGetErrorString(err));
    return body();
  };
  if (rc) {
    if (g) hipGraphDestroy(g);
    if (ctx->comm && e != hipSuccess) return comm_fallback("capturing a Krylov cycle", e);
    return rc;
  }
  if (e != hipSuccess || !g) {
    if (ctx->comm) return comm_fallback("hipStreamEndCapture", e);
    mfem_set_error("hipStreamEndCapture: %s", hipGetErrorString(e));
    return MFEM_ERR_HIP;
  }
  hipGraphExec_t exec = nullptr;
  const hipError_t ei = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
  hipGraphDestroy(g);
  if (ei != hipSuccess) {
    if (ctx->comm) return comm_fallback("hipGraphInstantiate", ei);
    mfem_set_error("hipGraphInstantiate: %s", hipGetErrorString(ei));
    return MFEM_ERR_HIP;
  }
  if (ctx->comm) ++mfem_graph_comm_captures;
  ctx->graph_exec[slot] = exec;
  ctx->graph_key[slot] = key;
  MFEM_CHECK_HIP(hipGraphLaunch(exec, ctx->stream));
  return MFEM_OK;
}
