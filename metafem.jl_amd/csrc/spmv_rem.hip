// A = S + N: the sparse skew remainder of the symmetric lattice-tile layouts (solver layout modes 4 and 5, spmv_lat27.hip / spmv_lat8.hip).
//
// The tiles store one triangle of the matrix (per node the diagonal block's upper entries and the entries towards the "upper" lattice neighbours)
// and mirror it, i.e. they apply S = U + U^T - D.  That is the caller's matrix only if its values are symmetric -- and the reference's OWN way of
// fixing a value makes them nonsymmetric: the Nitsche term k*Bilinear(T, n{i}*T{;i}) of examples/thermal_conduction/2D_Script.jl:58 (and the SUPG /
// boundary terms of the flow examples) couples a face node's row to the interior nodes of its element with no mirrored counterpart.  The asymmetry is
// confined to the O(n^(2/3)) rows next to those faces, so instead of refusing the whole solve (round 4: every nonsymmetric solve fell back to the
// layouts that read ALL entries, 1.5 - 1.8x slower per SpMV) the bind now keeps the tiles and carries the difference
//     N[r][c] = A[r][c] - A[c][r]     for the entries (r, c) the tiles do NOT store (the mirrored triangle), in the rows where it is non-zero
// as a small CSR of its own, applied after the tiles' gather pass:  y = S x + N x = A x  to round-off (one rounding in the difference).
//
// Which rows: the symmetry probe (mfem_sym_probe) already measures, per row, |(S x - A x)_r| / |a_rr| for a random-sign probe vector; the rows above
// the gate (4e-13) are the rows of N.  The bind accepts when they are at most n / 8 AND the probe repeated with N applied passes the same gate
// on every row -- the acceptance test is the same as for symmetric values, only the operator it is applied to is S + N.
// Everything here is generic in the pattern: the mirrored entry of (r, c) is found by searching row c of the caller's CSR for column r.
#include "blas1.h"
#include <hipcub/hipcub.hpp>

#define MFEM_REM_MAX_FRACTION 8  // a remainder of more than n / 8 rows is not taken

static std::atomic<int> g_rem_enable{1};
static std::atomic<long long> g_rem_count{0};  // SpMVs that applied a remainder (tests / bench.py: which path ran)
extern "C" long long mfem_debug_rem_spmv_count(void) { return g_rem_count; }
static std::atomic<int> g_rem_diag{0};  // bit 1 of mfem_debug_set_remainder: the diagnostic SpMV entry (mfem_spmv_solver_layout, which answers for cg!) allows it too
extern "C" int mfem_debug_set_remainder(int enable) try {
  ++mfem_debug_epoch;
  g_rem_enable = (enable & 1) ? 1 : 0;
  g_rem_diag = (enable >> 1) & 1;
  return MFEM_OK;
} MFEM_API_CATCH("mfem_debug_set_remainder")
bool mfem_rem_diag() { return g_rem_diag != 0; }
// rows / entries of the remainder the last bind on this pattern built (0 / 0: none needed or none accepted); asymmetry the probe saw before it
extern "C" int mfem_debug_remainder_info(mfem_csr A, int64_t* rows, int64_t* entries, double* asym_before) try {
  MFEM_REQUIRE(A, "null pattern");
  if (rows) *rows = A->rem_last_rows;
  if (entries) *entries = A->rem_last_ent;
  if (asym_before) *asym_before = A->rem_asym_before;
  return MFEM_OK;
} MFEM_API_CATCH("mfem_debug_remainder_info")
bool mfem_rem_enabled() { return g_rem_enable != 0; }

// (r, c) is an entry the tiles take from its mirror (c, r): column node below the row node, or the same node and a lower field.  Ghost columns
// (c >= n: slab patterns) are never mirrored -- the tiles read them from the stored triangle or from the caller's CSR values.
__device__ __forceinline__ bool rem_is_mirrored(int64_t r, int64_t c, int64_t n, int64_t N) {
  if (c >= n) return false;
  const int64_t pr = r % N, pc = c % N;
  return pc < pr || (pc == pr && c < r);
}

// rows whose probe difference is above the gate -> rows[] (any order), count in cnt[0]
__global__ __launch_bounds__(MFEM_BLOCK) void k_rem_flag(int64_t n, const double* __restrict__ y1, const double* __restrict__ y2,
                                                           const double* __restrict__ scale, double gate, int32_t* __restrict__ rows, int64_t cap,
                                                           unsigned long long* __restrict__ cnt) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double e = fabs(y1[i] - y2[i]) / scale[i];
    if (!(e <= gate)) {  // (NaN counts as above)
      const unsigned long long k = atomicAdd(cnt, 1ull);
      if ((int64_t)k < cap) rows[k] = (int32_t)i;
    }
  }
}

// a wave per flagged row.  FILL = false: len[k] = mirrored entries of the row, ptr[k] = their place (reserved from cnt[1]).
// FILL = true: col / val of every mirrored entry: val = A[r][c] - A[c][r], the mirror found by a wave-wide scan of row c.
template <typename RP, bool FILL>
__global__ __launch_bounds__(MFEM_BLOCK) void k_rem_rows(int64_t n, int64_t N, const RP* __restrict__ rowptr, const int32_t* __restrict__ col, int base,
                                                           const double* __restrict__ vals, const int32_t* __restrict__ rows, int64_t nrows,
                                                           int64_t* __restrict__ ptr, int32_t* __restrict__ len, int32_t* __restrict__ ocol,
                                                           double* __restrict__ oval, unsigned long long* __restrict__ cnt) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t k = wave; k < nrows; k += nwaves) {
    const int64_t r = rows[k];
    const int64_t lo = (int64_t)rowptr[r] - base, hi = (int64_t)rowptr[r + 1] - base;
    int64_t out = FILL ? ptr[k] : 0;
    int total = 0;
    for (int64_t j0 = lo; j0 < hi; j0 += 64) {
      const int64_t j = j0 + lane;
      const int64_t c = j < hi ? (int64_t)col[j] - base : n;
      const bool mir = j < hi && rem_is_mirrored(r, c, n, N);
      const unsigned long long mask = __ballot(mir);
      if (FILL) {
        // the entries of this round, one after the other: all 64 lanes scan row c for column r
        unsigned long long m = mask;
        while (m) {
          const int src = __ffsll((long long)m) - 1;
          m &= m - 1;
          const int64_t cc = __shfl(c, src, MFEM_WAVE);
          const double arc = __shfl(mir ? vals[j] : 0.0, src, MFEM_WAVE);
          const int64_t clo = (int64_t)rowptr[cc] - base, chi = (int64_t)rowptr[cc + 1] - base;
          double acr = 0.0;  // (no mirrored entry in the pattern: the tiles take 0 for it -- their structure check guarantees it exists)
          for (int64_t t0 = clo; t0 < chi; t0 += 64) {
            const int64_t t = t0 + lane;
            const bool hit = t < chi && (int64_t)col[t] - base == r;
            const unsigned long long hm = __ballot(hit);
            if (hm) {
              const int hl = __ffsll((long long)hm) - 1;
              acr = __shfl(hit ? vals[t] : 0.0, hl, MFEM_WAVE);
              break;
            }
          }
          if (lane == 0) {
            ocol[out] = (int32_t)cc;
            oval[out] = arc - acr;
          }
          ++out;
        }
      } else {
        total += __popcll(mask);
      }
    }
    if (!FILL && lane == 0) {
      len[k] = total;
      ptr[k] = (int64_t)atomicAdd(cnt + 1, (unsigned long long)total);
    }
  }
}

// y[r] += alpha * sum_c N[r][c] x[c] / dsc[c] for the remainder's rows, 16 lanes per row; the fused dot product's correction goes to partials[blockIdx]
__global__ __launch_bounds__(MFEM_BLOCK) void k_rem_apply(int64_t nrows, const int32_t* __restrict__ rows, const int64_t* __restrict__ ptr,
                                                            const int32_t* __restrict__ len, const int32_t* __restrict__ col,
                                                            const double* __restrict__ val, const double* __restrict__ x, const double* __restrict__ dsc,
                                                            double* __restrict__ y, double alpha, const double* __restrict__ dotw,
                                                            double* __restrict__ partials, const int32_t* __restrict__ done_flag) {
  __shared__ double red[4];
  if (done_flag && done_flag[0]) return;
  const int sub = threadIdx.x & 15;
  const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4, ngrp = ((int64_t)gridDim.x * blockDim.x) >> 4;
  double dacc = 0.0;
  for (int64_t k0 = grp - (grp % (MFEM_BLOCK / 16)); k0 < nrows; k0 += ngrp) {  // (whole workgroups iterate together: the shuffles below need full waves)
    const int64_t k = k0 + (grp % (MFEM_BLOCK / 16));
    double s = 0.0;
    if (k < nrows) {
      const int64_t p0 = ptr[k];
      const int L = len[k];
      for (int t = sub; t < L; t += 16) {
        const int64_t c = col[p0 + t];
        s += val[p0 + t] * (dsc ? x[c] / dsc[c] : x[c]);
      }
    }
    s += __shfl_xor(s, 8, MFEM_WAVE);
    s += __shfl_xor(s, 4, MFEM_WAVE);
    s += __shfl_xor(s, 2, MFEM_WAVE);
    s += __shfl_xor(s, 1, MFEM_WAVE);
    if (k < nrows && sub == 0) {
      const int64_t r = rows[k];
      const double dy = alpha * s;
      y[r] += dy;
      if (dotw) dacc += dy * dotw[r];
    }
  }
  if (partials) {
    const double bsum = block_reduce_sum(dacc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = bsum;
  }
}

void mfem_rem_clear(mfem_csr_s* A) { A->rem_active = 0; }
void mfem_rem_free(mfem_csr_s* A) {
  if (A->rem_rows) hipFree(A->rem_rows);
  if (A->rem_ptr) hipFree(A->rem_ptr);
  if (A->rem_len) hipFree(A->rem_len);
  if (A->rem_col) hipFree(A->rem_col);
  if (A->rem_val) hipFree(A->rem_val);
  if (A->rem_cnt) hipFree(A->rem_cnt);
  if (A->rem_sort_tmp) hipFree(A->rem_sort_tmp);
  A->rem_sort_tmp = nullptr;
  A->rem_sort_bytes = 0;
  A->rem_rows = nullptr; A->rem_ptr = nullptr; A->rem_len = nullptr; A->rem_col = nullptr; A->rem_val = nullptr; A->rem_cnt = nullptr;
  A->rem_cap_rows = A->rem_cap_ent = 0;
  A->rem_active = 0;
}

static int rem_apply_grid(const mfem_context_s* ctx, int64_t nrows) {
  int64_t g = (nrows + MFEM_BLOCK / 16 - 1) / (MFEM_BLOCK / 16);
  int cap = ctx->num_cus * 4;  // (its partial sums follow the gather pass's in one array of MFEM_MAX_PARTIALS: the gather takes at most half)
  if (cap > MFEM_MAX_PARTIALS / 4) cap = MFEM_MAX_PARTIALS / 4;
  if (g > cap) g = cap;
  return g < 1 ? 1 : (int)g;
}

// Builds N for `vals` from the probe's two products (y1: tiles, y2: CSR kernel; scale: the rows' weights) -- storage owned by the pattern, grown on
// demand, kept between solves.  *built = false (and no error) when the rows above the gate exceed n / 8 (MFEM_REM_MAX_FRACTION; a remainder row costs
// about 290 bytes per product against the 108 a mirrored row saves, and moves them less well): the caller refuses the tiles as before.
// n_fields: fields of the field-major row numbering (1 for the hex-27 tiles).  Synchronises the stream twice (the two counts).
int mfem_rem_build(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, int n_fields, const double* y1, const double* y2, const double* scale,
                   double gate, bool* built) {
  *built = false;
  A->rem_active = 0;
  if (!g_rem_enable || n_fields < 1 || A->n % n_fields != 0 || A->n >= ((int64_t)1 << 31)) return MFEM_OK;
  const int64_t n = A->n, N = n / n_fields;
  const int64_t cap_rows = n / MFEM_REM_MAX_FRACTION + 1;
  if (A->rem_cap_rows < cap_rows) {
    if (A->rem_rows) hipFree(A->rem_rows);
    if (A->rem_ptr) hipFree(A->rem_ptr);
    if (A->rem_len) hipFree(A->rem_len);
    A->rem_rows = nullptr; A->rem_ptr = nullptr; A->rem_len = nullptr;
    A->rem_cap_rows = 0;
    MFEM_CHECK_HIP(hipMalloc(&A->rem_rows, sizeof(int32_t) * (size_t)cap_rows));
    MFEM_CHECK_HIP(hipMalloc(&A->rem_ptr, sizeof(int64_t) * (size_t)cap_rows));
    MFEM_CHECK_HIP(hipMalloc(&A->rem_len, sizeof(int32_t) * (size_t)cap_rows));
    A->rem_cap_rows = cap_rows;
  }
  if (!A->rem_cnt) MFEM_CHECK_HIP(hipMalloc(&A->rem_cnt, 2 * sizeof(unsigned long long)));
  MFEM_CHECK_HIP(hipMemsetAsync(A->rem_cnt, 0, 2 * sizeof(unsigned long long), ctx->stream));
  hipLaunchKernelGGL(k_rem_flag, dim3(mfem_vec_grid(ctx, n)), dim3(MFEM_BLOCK), 0, ctx->stream, n, y1, y2, scale, gate, A->rem_rows, A->rem_cap_rows,
                     A->rem_cnt);
  MFEM_CHECK_LAUNCH();
  unsigned long long h[2] = {0, 0};
  MFEM_CHECK_HIP(hipMemcpyAsync(h, A->rem_cnt, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
  MFEM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  const int64_t nrows = (int64_t)h[0];
  if (nrows == 0 || nrows > n / MFEM_REM_MAX_FRACTION) return MFEM_OK;
  {
    // the flagged rows in ASCENDING order (the flag pass appends them in the order its atomics land): the application then writes y and gathers x with
    // neighbouring groups on neighbouring lines, and the layout is the same from run to run.  Sorted into rem_len's storage (same size), copied back.
    size_t tb = 0;
    int32_t* alt = A->rem_len;
    MFEM_CHECK_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tb, (const int32_t*)A->rem_rows, alt, (int)nrows, 0, 32, ctx->stream));
    if (tb > A->rem_sort_bytes) {  // kept with the pattern (a hipMalloc / hipFree and a stream synchronisation per solve before)
      if (A->rem_sort_tmp) (void)hipFree(A->rem_sort_tmp);
      A->rem_sort_tmp = nullptr;
      A->rem_sort_bytes = 0;
      MFEM_CHECK_HIP(hipMalloc(&A->rem_sort_tmp, tb));
      A->rem_sort_bytes = tb;
    }
    hipError_t es = hipcub::DeviceRadixSort::SortKeys(A->rem_sort_tmp, tb, (const int32_t*)A->rem_rows, alt, (int)nrows, 0, 32, ctx->stream);
    if (es == hipSuccess) es = hipMemcpyAsync(A->rem_rows, alt, sizeof(int32_t) * (size_t)nrows, hipMemcpyDeviceToDevice, ctx->stream);
    if (es != hipSuccess) {
      mfem_set_error("remainder: sorting the rows -> %s", hipGetErrorString(es));
      return MFEM_ERR_HIP;
    }
  }
  const int grid = mfem_grid_for(nrows * 64, MFEM_BLOCK, ctx->num_cus * 16);
#define REM_ROWS(RP, FILL_)                                                                                                                         \
  hipLaunchKernelGGL((k_rem_rows<RP, FILL_>), dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, n, N, (const RP*)A->rowptr, A->colidx, A->index_base, vals, \
                     (const int32_t*)A->rem_rows, nrows, A->rem_ptr, A->rem_len, A->rem_col, A->rem_val, A->rem_cnt)
  if (A->rowptr_bits == 64) REM_ROWS(int64_t, false); else REM_ROWS(int32_t, false);
  MFEM_CHECK_LAUNCH();
  MFEM_CHECK_HIP(hipMemcpyAsync(h + 1, A->rem_cnt + 1, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
  MFEM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  const int64_t nent = (int64_t)h[1];
  if (nent == 0) return MFEM_OK;  // (rows above the gate without a mirrored entry: nothing a remainder could repair)
  if (A->rem_cap_ent < nent) {
    if (A->rem_col) hipFree(A->rem_col);
    if (A->rem_val) hipFree(A->rem_val);
    A->rem_col = nullptr; A->rem_val = nullptr;
    A->rem_cap_ent = 0;
    const int64_t want = nent + nent / 4 + 1024;
    MFEM_CHECK_HIP(hipMalloc(&A->rem_col, sizeof(int32_t) * (size_t)want));
    MFEM_CHECK_HIP(hipMalloc(&A->rem_val, sizeof(double) * (size_t)want));
    A->rem_cap_ent = want;
  }
  if (A->rowptr_bits == 64) REM_ROWS(int64_t, true); else REM_ROWS(int32_t, true);
#undef REM_ROWS
  MFEM_CHECK_LAUNCH();
  A->rem_nrows = nrows;
  A->rem_nent = nent;
  *built = true;  // (not active yet: the caller verifies S + N against the CSR kernel first)
  return MFEM_OK;
}

// y += alpha N (x ./ dsc); the fused dot product's correction is appended to the partial sums the gather pass wrote (*n_partials advanced)
int mfem_rem_apply(mfem_context_s* ctx, mfem_csr_s* A, const double* x, const double* dsc, double* y, double alpha, const double* dotw,
                   double* partials, int* n_partials, const int32_t* done_flag) {
  if (A->rem_nrows <= 0) return MFEM_OK;
  const int grid = rem_apply_grid(ctx, A->rem_nrows);
  double* p = nullptr;
  if (dotw && partials && n_partials) {
    if (*n_partials + grid > MFEM_MAX_PARTIALS) {
      mfem_set_error("remainder SpMV: %d + %d partial sums (> %d)", *n_partials, grid, MFEM_MAX_PARTIALS);
      return MFEM_ERR_INVALID;
    }
    p = partials + *n_partials;
  }
  hipLaunchKernelGGL(k_rem_apply, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, A->rem_nrows, (const int32_t*)A->rem_rows, (const int64_t*)A->rem_ptr,
                     (const int32_t*)A->rem_len, (const int32_t*)A->rem_col, (const double*)A->rem_val, x, dsc, y, alpha, p ? dotw : nullptr, p, done_flag);
  MFEM_CHECK_LAUNCH();
  if (p) *n_partials += grid;
  if (!ctx->probe_active) ++g_rem_count;
  return MFEM_OK;
}
// bytes one application moves by design (accounting: bench.py adds them to the tiles' bytes)
int64_t mfem_rem_design_bytes(const mfem_csr_s* A) {  // (of the last accepted remainder: the query comes after the solve has unbound its layout)
  if (!A->rem_last_rows) return 0;
  return A->rem_last_ent * (8 + 4 + 8) + A->rem_last_rows * (4 + 8 + 4 + 16);  // value, column, gathered x per entry; row id, pointer, length, y read + written per row
}
