// Inspector-executor CSR SpMV: the ABI still takes plain CSR (rowptr, colidx, vals); at pattern-creation time an
// inspector pass derives, per row tile, the list of distinct 128-byte lines of x the tile touches and a 16-bit
// window-local index for every nonzero.  The executor then
//   * copies the tile's x lines into LDS with fully coalesced 128-byte reads (an FEM tile of 128 rows touches ~90
//     lines = 11 KB instead of issuing 3456 scattered 8-byte gathers -- the gather was measured at 22 % of the plain
//     kernel's time, profiles/r01_spmv_sweep.txt),
//   * streams val (8 B) + local index (2 B) per nonzero instead of val + 4-byte column: 10 B/nnz instead of 12,
//   * multiplies out of LDS and reduces rows exactly like the plain kernel (spmv.hip).
// The plan costs 2 B/nnz + 4 B per (tile, line) of extra device memory, is built once per pattern and is reused
// for every Newton step / Krylov iteration (values change, the pattern does not).  Patterns whose tiles touch
// more lines than the LDS window holds (e.g. random matrices) keep the plain kernel.
//
// STATUS (round 1): correct (tests/test_gpu_primitives.py::test_spmv_window_plan_equals_plain_kernel) but measured
// SLOWER than the plain kernel on the 256^3 hex-8 matrix -- 1.44-1.50 ms vs 1.17-1.28 ms (profiles/r01_spmv_sweep.txt):
// the third barrier per tile, the two-level dependent fill (line ids -> x lines) and 110 VGPRs (4 workgroups per CU
// instead of 5) cost more than the 2 B/nnz and the gather save.  It is therefore DISABLED by default
// (mfem_debug_set_spmv_window(1, cap, mult) enables it for plans created afterwards) and kept as the starting point
// for a double-buffered version.
#include "blas1.h"

typedef double d2_t __attribute__((ext_vector_type(2)));

#define WIN_LINE 16          // doubles per x line (128 B)
#define WIN_UNROLL 8

template <int CAP> struct WinCfg;
template <> struct WinCfg<4032> { static constexpr int NS = 4096; static constexpr int MAXLINES = 128; };
template <> struct WinCfg<2016> { static constexpr int NS = 2048; static constexpr int MAXLINES = 128; };

// ---- inspector ---------------------------------------------------------------------------------------------
// mode 0: count distinct lines per tile (nlines, global max); mode 1: also write the line lists and the
// window-local indices.
template <typename RP, int CAP>
__global__ __launch_bounds__(MFEM_BLOCK) void k_win_plan(int64_t n, const RP* __restrict__ rowptr,
                                                           const int32_t* __restrict__ col, int base, int R, int64_t ntiles,
                                                           int mode, int lstride, int32_t* __restrict__ nlines,
                                                           int32_t* __restrict__ maxlines, uint32_t* __restrict__ lines,
                                                           uint16_t* __restrict__ idx16) {
  constexpr int NS = WinCfg<CAP>::NS;
  constexpr int PER = NS / MFEM_BLOCK;
  __shared__ uint32_t key[NS];
  __shared__ uint32_t uniq[NS];
  __shared__ int scan[MFEM_BLOCK + 1];
  const int tid = threadIdx.x;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t r0 = tile * R;
    const int64_t r1 = (r0 + R < n) ? r0 + R : n;
    const int64_t s = (int64_t)rowptr[r0] - base;
    const int cnt = (int)((int64_t)rowptr[r1] - base - s);
    for (int i = tid; i < NS; i += MFEM_BLOCK) key[i] = (i < cnt) ? ((uint32_t)(col[s + i] - base) >> 4) : 0xFFFFFFFFu;
    __syncthreads();
    // bitonic sort, ascending
    for (int k = 2; k <= NS; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < NS; i += MFEM_BLOCK) {
          const int ixj = i ^ j;
          if (ixj > i) {
            const uint32_t a = key[i], b = key[ixj];
            const bool up = (i & k) == 0;
            if ((a > b) == up) {
              key[i] = b;
              key[ixj] = a;
            }
          }
        }
        __syncthreads();
      }
    // unique: each thread owns PER consecutive sorted keys
    int heads = 0;
    for (int t = 0; t < PER; ++t) {
      const int i = tid * PER + t;
      const uint32_t v = key[i];
      if (v != 0xFFFFFFFFu && (i == 0 || key[i - 1] != v)) ++heads;
    }
    scan[tid + 1] = heads;
    if (tid == 0) scan[0] = 0;
    __syncthreads();
    if (tid == 0)
      for (int i = 1; i <= MFEM_BLOCK; ++i) scan[i] += scan[i - 1];  // 256 adds, once per tile of a one-off pass
    __syncthreads();
    int pos = scan[tid];
    const int nl = scan[MFEM_BLOCK];
    for (int t = 0; t < PER; ++t) {
      const int i = tid * PER + t;
      const uint32_t v = key[i];
      if (v != 0xFFFFFFFFu && (i == 0 || key[i - 1] != v)) uniq[pos++] = v;
    }
    __syncthreads();
    if (tid == 0) {
      nlines[tile] = nl;
      atomicMax(maxlines, nl);
    }
    if (mode == 1 && nl <= lstride) {
      // pad the list with its first line: the executor always fills `lstride` lines (fixed-shape, fully unrolled)
      for (int i = tid; i < lstride; i += MFEM_BLOCK) lines[tile * (int64_t)lstride + i] = (i < nl) ? uniq[i] : (nl > 0 ? uniq[0] : 0u);
      for (int i = tid; i < cnt; i += MFEM_BLOCK) {
        const uint32_t c = (uint32_t)(col[s + i] - base);
        const uint32_t ln = c >> 4;
        int lo = 0, hi = nl - 1;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (uniq[mid] < ln) lo = mid + 1; else hi = mid;
        }
        idx16[s + i] = (uint16_t)(lo * WIN_LINE + (c & 15u));
      }
    }
    __syncthreads();
  }
}

// ---- executor ----------------------------------------------------------------------------------------------
template <typename RP, int CAP>
__global__ __launch_bounds__(MFEM_BLOCK) void k_spmv_win(
    int64_t n, int64_t nnz, const RP* __restrict__ rowptr, const uint16_t* __restrict__ idx16,
    const uint32_t* __restrict__ lines, const int32_t* __restrict__ nlines, int lstride, int64_t xlen,
    const double* __restrict__ vals, const double* __restrict__ x, double* __restrict__ y, double alpha, double beta,
    int base, int R, int tpr_log2, int64_t ntiles, const double* __restrict__ dotw, double* __restrict__ partials,
    const int32_t* __restrict__ done_flag) {
  extern __shared__ double wlds[];
  double* prod = wlds;                 // [CAP + 4]
  double* win = wlds + CAP + 4;        // [lstride * 16]
  __shared__ double red[4];
  if (done_flag && done_flag[0]) return;
  const int tid = threadIdx.x;
  const int tpr = 1 << tpr_log2;
  const int g = tid & (tpr - 1);
  double dot_acc = 0.0;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t r0 = tile * R;
    const int64_t r1 = (r0 + R < n) ? r0 + R : n;
    const int64_t s = (int64_t)rowptr[r0] - base;
    const int64_t e = (int64_t)rowptr[r1] - base;
    const int64_t sa = s & ~(int64_t)1;
    const int cnt = (int)(e - sa);
    // stream loads first: they fly while the x window is being filled
    d2_t v[WIN_UNROLL];
    uint32_t ix[WIN_UNROLL];
#pragma unroll
    for (int u = 0; u < WIN_UNROLL; ++u) {
      const int i = 2 * tid + u * 2 * MFEM_BLOCK;
      v[u] = (d2_t){0.0, 0.0};
      ix[u] = 0u;
      if (i < cnt) {
        if (sa + i + 1 < nnz) {
          v[u] = __builtin_nontemporal_load(reinterpret_cast<const d2_t*>(vals + sa + i));
          ix[u] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(idx16 + sa + i));
        } else {
          v[u].x = vals[sa + i];
          ix[u] = idx16[sa + i];
        }
      }
    }
    const int64_t rmine = r0 + (tid >> tpr_log2);
    int lo_pre = 0, hi_pre = 0;
    if (rmine < r1) {
      lo_pre = (int)((int64_t)rowptr[rmine] - base - sa);
      hi_pre = (int)((int64_t)rowptr[rmine + 1] - base - sa);
    }
    // x window: the tile's line list is padded to `lstride` entries by the inspector, so the fill has a fixed shape:
    // all line ids are requested first, then all x lines (one 128-byte line per 16 consecutive lanes), then LDS.
    constexpr int KMAX = WinCfg<CAP>::MAXLINES * WIN_LINE / MFEM_BLOCK;
    const uint32_t* tl = lines + tile * (int64_t)lstride;
    const int nwin = lstride * WIN_LINE;
    uint32_t ln[KMAX];
    double xv[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      const int t = tid + k * MFEM_BLOCK;
      ln[k] = (t < nwin) ? tl[t >> 4] : 0u;
    }
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      const int t = tid + k * MFEM_BLOCK;
      const int64_t gi = (int64_t)ln[k] * WIN_LINE + (t & 15);
      xv[k] = (t < nwin && gi < xlen) ? x[gi] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      const int t = tid + k * MFEM_BLOCK;
      if (t < nwin) win[t] = xv[k];
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < WIN_UNROLL; ++u) {
      const int i = 2 * tid + u * 2 * MFEM_BLOCK;
      if (i < cnt) {
        // entry sa+i < s belongs to the previous tile (its index refers to that tile's window): never read
        const double x0 = (sa + i >= s) ? win[ix[u] & 0xFFFFu] : 0.0;
        const double x1 = (i + 1 < cnt) ? win[ix[u] >> 16] : 0.0;
        *reinterpret_cast<d2_t*>(&prod[i]) = (d2_t){v[u].x * x0, v[u].y * x1};
      }
    }
    __syncthreads();
    for (int64_t r = rmine; r < r1; r += (MFEM_BLOCK >> tpr_log2)) {
      const int lo = (r == rmine) ? lo_pre : (int)((int64_t)rowptr[r] - base - sa);
      const int hi = (r == rmine) ? hi_pre : (int)((int64_t)rowptr[r + 1] - base - sa);
      double sum = 0.0;
      for (int j = lo + g; j < hi; j += tpr) sum += prod[j];
      for (int off = tpr >> 1; off > 0; off >>= 1) sum += __shfl_xor(sum, off, MFEM_WAVE);
      if (g == 0) {
        double yv = alpha * sum;
        if (beta != 0.0) yv += beta * y[r];
        y[r] = yv;
        if (dotw) dot_acc += yv * dotw[r];
      }
    }
    __syncthreads();
  }
  if (partials) {
    const double b = block_reduce_sum(dot_acc, red);
    if (tid == 0) partials[blockIdx.x] = b;
  }
}

// ---- host ------------------------------------------------------------------------------------------------
static int g_win_enable = 0;  // measured SLOWER than the plain kernel on hex-8 256^3 (1.44-1.50 ms vs 1.17-1.28 ms): off by default
static int g_win_cap = 4032;
static int g_win_grid_mult = 8;

extern "C" int mfem_debug_set_spmv_window(int enable, int cap, int grid_mult) {
  ++mfem_debug_epoch;
  g_win_enable = enable;
  if (cap == 4032 || cap == 2016) g_win_cap = cap;
  if (grid_mult > 0) g_win_grid_mult = grid_mult;
  return MFEM_OK;
}

template <typename RP, int CAP>
static int build_plan_t(mfem_context_s* ctx, mfem_csr_s* A) {
  int R = MFEM_BLOCK;
  while (R > 1 && (int64_t)R * A->max_row_nnz > CAP - 2) R >>= 1;
  if ((int64_t)R * A->max_row_nnz > CAP - 2) return MFEM_OK;  // rows too long: no plan
  const int64_t ntiles = (A->n + R - 1) / R;
  int32_t *d_nl = nullptr, *d_max = ctx->d_flags + 9;
  MFEM_CHECK_HIP(hipMalloc(&d_nl, sizeof(int32_t) * ntiles));
  MFEM_CHECK_HIP(hipMemsetAsync(d_max, 0, sizeof(int32_t), ctx->stream));
  const int grid = (int)(ntiles < (int64_t)ctx->num_cus * 4 ? ntiles : (int64_t)ctx->num_cus * 4);
  hipLaunchKernelGGL((k_win_plan<RP, CAP>), dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, A->n, (const RP*)A->rowptr, A->colidx,
                     A->index_base, R, ntiles, 0, 0, d_nl, d_max, (uint32_t*)nullptr, (uint16_t*)nullptr);
  MFEM_CHECK_LAUNCH();
  MFEM_CHECK_HIP(hipMemcpyAsync(ctx->h_flags + 9, d_max, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  MFEM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  const int maxl = ctx->h_flags[9];
  if (maxl <= 0 || maxl > WinCfg<CAP>::MAXLINES) {
    hipFree(d_nl);
    return MFEM_OK;  // the window would not fit: keep the plain kernel
  }
  const int lstride = (maxl + 7) & ~7;
  uint32_t* d_lines = nullptr;
  uint16_t* d_idx = nullptr;
  MFEM_CHECK_HIP(hipMalloc(&d_lines, sizeof(uint32_t) * ntiles * lstride));
  MFEM_CHECK_HIP(hipMalloc(&d_idx, sizeof(uint16_t) * (A->nnz + 2)));
  hipLaunchKernelGGL((k_win_plan<RP, CAP>), dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, A->n, (const RP*)A->rowptr, A->colidx,
                     A->index_base, R, ntiles, 1, lstride, d_nl, d_max, d_lines, d_idx);
  MFEM_CHECK_LAUNCH();
  MFEM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  A->win_ready = 1;
  A->win_cap = CAP;
  A->win_R = R;
  A->win_lstride = lstride;
  A->win_nlines = d_nl;
  A->win_lines = d_lines;
  A->win_idx = d_idx;
  return MFEM_OK;
}

// xlen: max column + 1 (the window fill must not read past the caller's x)
__global__ __launch_bounds__(MFEM_BLOCK) void k_max_col(int64_t nnz, const int32_t* __restrict__ col, int base,
                                                          int32_t* __restrict__ out) {
  int m = -1;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = col[i] - base;
    m = c > m ? c : m;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const int o = __shfl_down(m, off, MFEM_WAVE);
    m = o > m ? o : m;
  }
  if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

int mfem_spmv_window_plan(mfem_context_s* ctx, mfem_csr_s* A) {
  A->win_ready = 0;
  if (!g_win_enable || A->nnz == 0 || A->max_row_nnz <= 0) return MFEM_OK;
  if ((((uintptr_t)A->colidx) & 3) != 0) return MFEM_OK;
  int32_t* d_max = ctx->d_flags + 10;
  MFEM_CHECK_HIP(hipMemsetAsync(d_max, 0xFF, sizeof(int32_t), ctx->stream));  // -1
  hipLaunchKernelGGL(k_max_col, dim3(mfem_grid_for(A->nnz, MFEM_BLOCK, ctx->num_cus * 8)), dim3(MFEM_BLOCK), 0, ctx->stream, A->nnz,
                     A->colidx, A->index_base, d_max);
  MFEM_CHECK_LAUNCH();
  MFEM_CHECK_HIP(hipMemcpyAsync(ctx->h_flags + 10, d_max, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  MFEM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  A->win_xlen = (int64_t)ctx->h_flags[10] + 1;
  if (g_win_cap == 2016) {
    if (A->rowptr_bits == 64) return build_plan_t<int64_t, 2016>(ctx, A);
    return build_plan_t<int32_t, 2016>(ctx, A);
  }
  if (A->rowptr_bits == 64) return build_plan_t<int64_t, 4032>(ctx, A);
  return build_plan_t<int32_t, 4032>(ctx, A);
}

void mfem_spmv_window_free(mfem_csr_s* A) {
  if (A->win_nlines) hipFree(A->win_nlines);
  if (A->win_lines) hipFree(A->win_lines);
  if (A->win_idx) hipFree(A->win_idx);
  A->win_nlines = nullptr;
  A->win_lines = nullptr;
  A->win_idx = nullptr;
  A->win_ready = 0;
}

// returns 1 if launched, 0 if the plain kernel should be used, <0 on error
int mfem_spmv_window_launch(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, const double* x, double* y, double alpha,
                            double beta, const double* dotw, double* partials, int* n_partials, const int32_t* done_flag) {
  if (!A->win_ready || !g_win_enable) return 0;
  if ((((uintptr_t)vals) & 15) != 0) return 0;
  const int R = A->win_R;
  int tpr_log2 = 0;
  while ((MFEM_BLOCK >> (tpr_log2 + 1)) >= R) ++tpr_log2;
  const int64_t ntiles = (A->n + R - 1) / R;
  int cap = ctx->num_cus * g_win_grid_mult;
  if (cap > MFEM_MAX_PARTIALS) cap = MFEM_MAX_PARTIALS;
  const int grid = (int)(ntiles < cap ? ntiles : cap);
  const size_t lds = sizeof(double) * ((size_t)A->win_cap + 4 + (size_t)A->win_lstride * WIN_LINE);
#define LAUNCH_WIN(RP, CAP)                                                                                          \
  hipLaunchKernelGGL((k_spmv_win<RP, CAP>), dim3(grid), dim3(MFEM_BLOCK), lds, ctx->stream, A->n, A->nnz,            \
                     (const RP*)A->rowptr, A->win_idx, A->win_lines, A->win_nlines, A->win_lstride, A->win_xlen, vals, x, \
                     y, alpha, beta, A->index_base, R, tpr_log2, ntiles, dotw, partials, done_flag)
  if (A->win_cap == 4032) {
    if (A->rowptr_bits == 64) LAUNCH_WIN(int64_t, 4032); else LAUNCH_WIN(int32_t, 4032);
  } else {
    if (A->rowptr_bits == 64) LAUNCH_WIN(int64_t, 2016); else LAUNCH_WIN(int32_t, 2016);
  }
#undef LAUNCH_WIN
  MFEM_CHECK_LAUNCH();
  if (n_partials && partials) *n_partials = grid;
  return 1;
}
