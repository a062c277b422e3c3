// CSR SpMV for gfx950: replaces CUSPARSE mv!('N') behind mul! (reference misc/04_GPU_Utils.jl:131).
//
// Design (HBM-bound: 12 B per nonzero + 16 B per row):
//   * a workgroup owns a run of R consecutive rows whose nonzeros fit an LDS tile (CAP doubles);
//     R is a power of two chosen from the pattern's longest row, so FEM matrices with 27 / 81 /
//     125-wide rows all take this path;
//   * phase 1 streams val/col of the tile with 16-byte (val) + 8-byte (col) per-lane loads that
//     are contiguous across the whole workgroup -- coalescing does not depend on row length --
//     gathers x[col] (L2-resident: a hex mesh row touches 3 node planes) and parks the products
//     in LDS;
//   * phase 2 gives each row 256/R lanes that sum the row's products from LDS and combine with a
//     sub-wave shuffle; y is written once, coalesced;
//   * an optional fused dot product (w . y) is reduced per workgroup into ctx partials so the
//     Krylov loop needs no separate dot kernel or host sync for p.Ap;
//   * the grid is persistent (<= MFEM_MAX_PARTIALS workgroups, grid-stride over row tiles) and
//     the tile -> workgroup map is XCD-aware: workgroups with equal blockIdx % 8 share an XCD L2
//     (dispatch is round-robin over the 8 XCDs), so each XCD walks its own contiguous eighth of
//     the rows and x planes are fetched into one L2 instead of eight.
#include "blas1.h"

// Tile variants: CAP doubles of LDS product tile, UNROLL = 16-byte loads in flight per lane and batch.
// 4032 doubles = 31.5 KiB -> 5 workgroups per CU; 2016 -> 8 (wave-limited).
#define SPMV_CAP_MAX 4032

typedef double d2_t __attribute__((ext_vector_type(2)));
typedef int i2_t __attribute__((ext_vector_type(2)));

template <typename RP>
__global__ void k_max_row_nnz(int64_t n, const RP* __restrict__ rowptr, int32_t* __restrict__ out) {
  int m = 0;
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
    int len = (int)(rowptr[r + 1] - rowptr[r]);
    m = len > m ? len : m;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    int o = __shfl_down(m, off, MFEM_WAVE);
    m = o > m ? o : m;
  }
  if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

__device__ __forceinline__ int64_t tile_of(int64_t it, int64_t ntiles, int xcd_chunk) {
  // it = logical sequence number of this workgroup's next tile in dispatch order.  Workgroups with equal
  // it % 8 share an XCD (round-robin dispatch); give each XCD runs of `xcd_chunk` consecutive tiles, the
  // runs of the 8 XCDs interleaved so the chip as a whole still walks one contiguous window of the matrix.
  if (xcd_chunk <= 0) return it;
  const int64_t xcd = it & 7, local = it >> 3;
  const int64_t run = local / xcd_chunk, within = local % xcd_chunk;
  return (run * 8 + xcd) * xcd_chunk + within;  // may be >= ntiles near the end: caller skips
}

template <typename RP, bool VEC, int SPMV_CAP, int SPMV_UNROLL, int BLK = MFEM_BLOCK>
__global__ __launch_bounds__(BLK) void k_spmv_lds(
    int64_t n, int64_t nnz, const RP* __restrict__ rowptr, const int32_t* __restrict__ col,
    const double* __restrict__ vals, const double* __restrict__ x, double* __restrict__ y, double alpha,
    double beta, int base, int R, int tpr_log2, int64_t ntiles, int64_t ntiles_padded, int xcd_aware,
    const double* __restrict__ dotw, double* __restrict__ partials, const int32_t* __restrict__ done_flag, SpmvPart part) {
  __shared__ double prod[SPMV_CAP + 4];
  __shared__ double red[BLK / 64];
  if (done_flag && done_flag[0]) return;
  const int tid = threadIdx.x;
  const int tpr = 1 << tpr_log2;
  double dot_acc = 0.0;

  for (int64_t it = blockIdx.x; it < ntiles_padded; it += gridDim.x) {
    const int64_t tile = tile_of(it, ntiles, xcd_aware & 0xFFFF);
    if (tile >= ntiles) continue;  // uniform per workgroup
    const int64_t r0 = tile * R;
    const int64_t r1 = (r0 + R < n) ? r0 + R : n;
    if (spmv_part_skip(part, r0, r1)) continue;  // uniform per workgroup
    const int64_t s = (int64_t)rowptr[r0] - base;
    const int64_t e = (int64_t)rowptr[r1] - base;

    if (VEC) {
      const int64_t sa = s & ~(int64_t)1;  // 16-byte aligned start (vals/col bases are 16-B aligned)
      const int cnt = (int)(e - sa);
      // phase-2 row bounds of this lane's first row: issued now so the HBM latency hides under phase 1
      const int64_t rmine = r0 + (tid >> tpr_log2);
      int lo_pre = 0, hi_pre = 0;
      if (rmine < r1) {
        lo_pre = (int)((int64_t)rowptr[rmine] - base - sa);
        hi_pre = (int)((int64_t)rowptr[rmine + 1] - base - sa);
      }
      for (int i0 = 2 * tid; i0 < cnt; i0 += 2 * BLK * SPMV_UNROLL) {
        d2_t v[SPMV_UNROLL];
        i2_t c[SPMV_UNROLL];
#pragma unroll
        for (int u = 0; u < SPMV_UNROLL; ++u) {
          const int i = i0 + u * 2 * BLK;
          v[u] = (d2_t){0.0, 0.0};
          c[u] = (i2_t){base, base};
          if (i < cnt) {
            if (sa + i + 1 < nnz) {
              v[u] = __builtin_nontemporal_load(reinterpret_cast<const d2_t*>(vals + sa + i));
              c[u] = __builtin_nontemporal_load(reinterpret_cast<const i2_t*>(col + sa + i));
            } else {  // last odd entry of the whole matrix
              v[u].x = vals[sa + i];
              c[u].x = col[sa + i];
            }
          }
        }
#pragma unroll
        for (int u = 0; u < SPMV_UNROLL; ++u) {
          const int i = i0 + u * 2 * BLK;
          if (i < cnt) {
            // entry i+1 may belong to the next tile (i + 1 == cnt): its product is never read
            double x0, x1;
            if (xcd_aware & (1 << 30)) {  // timing probe: no gather
              x0 = (double)c[u].x;
              x1 = (double)c[u].y;
            } else {
              x0 = x[c[u].x - base];
              x1 = (i + 1 < cnt) ? x[c[u].y - base] : 0.0;
            }
            *reinterpret_cast<d2_t*>(&prod[i]) = (d2_t){v[u].x * x0, v[u].y * x1};
          }
        }
      }
      __syncthreads();
      // phase 2: tpr lanes per row
      const int g = tid & (tpr - 1);
      for (int64_t r = rmine; r < r1; r += (BLK >> tpr_log2)) {
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
    } else {
      const int cnt = (int)(e - s);
      for (int i0 = tid; i0 < cnt; i0 += BLK * SPMV_UNROLL) {
        double v[SPMV_UNROLL];
        int c[SPMV_UNROLL];
#pragma unroll
        for (int u = 0; u < SPMV_UNROLL; ++u) {
          const int i = i0 + u * BLK;
          v[u] = 0.0;
          c[u] = base;
          if (i < cnt) {
            v[u] = vals[s + i];
            c[u] = col[s + i];
          }
        }
#pragma unroll
        for (int u = 0; u < SPMV_UNROLL; ++u) {
          const int i = i0 + u * BLK;
          if (i < cnt) prod[i] = v[u] * x[c[u] - base];
        }
      }
      __syncthreads();
      const int g = tid & (tpr - 1);
      for (int64_t r = r0 + (tid >> tpr_log2); r < r1; r += (BLK >> tpr_log2)) {
        const int lo = (int)((int64_t)rowptr[r] - base - s);
        const int hi = (int)((int64_t)rowptr[r + 1] - base - s);
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
  }
  if (partials) {
    const double b = block_reduce_sum(dot_acc, red);
    if (tid == 0) partials[blockIdx.x] = b;
  }
}

// Fallback for patterns whose longest row does not fit the LDS tile: one wave per row.
template <typename RP>
__global__ __launch_bounds__(MFEM_BLOCK) void k_spmv_wave_per_row(
    int64_t n, const RP* __restrict__ rowptr, const int32_t* __restrict__ col, const double* __restrict__ vals,
    const double* __restrict__ x, double* __restrict__ y, double alpha, double beta, int base,
    const double* __restrict__ dotw, double* __restrict__ partials, const int32_t* __restrict__ done_flag, SpmvPart part) {
  __shared__ double red[4];
  if (done_flag && done_flag[0]) return;
  const int lane = threadIdx.x & 63;
  const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  double dot_acc = 0.0;
  for (int64_t r = wave; r < n; r += nwaves) {
    if (spmv_part_skip(part, r, r + 1)) continue;
    const int64_t lo = (int64_t)rowptr[r] - base, hi = (int64_t)rowptr[r + 1] - base;
    double sum = 0.0;
    for (int64_t j = lo + lane; j < hi; j += 64) sum += vals[j] * x[col[j] - base];
    sum = wave_reduce_sum(sum);
    if (lane == 0) {
      double yv = alpha * sum;
      if (beta != 0.0) yv += beta * y[r];
      y[r] = yv;
      if (dotw) dot_acc += yv * dotw[r];
    }
  }
  if (partials) {
    const double b = block_reduce_sum(dot_acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = b;
  }
}

// ---- host side ----------------------------------------------------------------------------
// Defaults from the 256^3 hex-8 sweep on MI355X (profiles/r01_spmv_sweep.txt): the round-robin tile map
// beat the XCD-contiguous one by ~4 %, and issuing the whole tile's loads up front (x8) beat x4 by ~6 %.
static int g_spmv_xcd_aware = 0;
static int g_spmv_grid_mult = 8;  // workgroups per CU of the persistent grid
// 0: CAP 4032 x4, 1: CAP 4032 x8 (256 threads);  2: CAP 2016 x8, 128 threads;  3: CAP 2016 x8, 64 threads;  7: CAP 1008 x8,
// 64 threads (wave-private tiles: the barrier degenerates);  4: CAP 4032 x4, 5: CAP 4032 x2 with 512 threads,  6: CAP 4032 x2
// with 1024 threads (same LDS tile shared by more waves: the tile kernel is LDS-limited to 4 workgroups per CU)
static int g_spmv_variant = 1;
static int g_spmv_nogather = 0;   // diagnostic only: replace x[col] by col-derived constants (WRONG results, timing probe)

extern "C" int mfem_debug_set_spmv(int xcd_aware, int grid_mult) {  // tuning hook for bench/profiling
  ++mfem_debug_epoch;
  g_spmv_xcd_aware = xcd_aware & 0xFFFF;   // tiles per XCD run (0 = plain round-robin)
  g_spmv_variant = (xcd_aware >> 16) & 7;
  g_spmv_nogather = (xcd_aware >> 20) & 1;
  if (grid_mult > 0) g_spmv_grid_mult = grid_mult;
  return MFEM_OK;
}

int mfem_csr_plan(mfem_context_s* ctx, mfem_csr_s* A) {
  A->serial = mfem_next_csr_serial();  // every creation path (mfem_csr_create, mfem_brick_pattern, mfem_pattern_build) plans once
  int32_t* d_max = ctx->d_flags + 8;
  MFEM_CHECK_HIP(hipMemsetAsync(d_max, 0, sizeof(int32_t), ctx->stream));
  const int grid = mfem_grid_for(A->n, MFEM_BLOCK, 4096);
  if (A->n > 0) {
    if (A->rowptr_bits == 64)
      hipLaunchKernelGGL(k_max_row_nnz<int64_t>, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, A->n,
                         (const int64_t*)A->rowptr, d_max);
    else
      hipLaunchKernelGGL(k_max_row_nnz<int32_t>, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, A->n,
                         (const int32_t*)A->rowptr, d_max);
    MFEM_CHECK_LAUNCH();
  }
  MFEM_CHECK_HIP(hipMemcpyAsync(ctx->h_flags + 8, d_max, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  MFEM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  A->max_row_nnz = ctx->h_flags[8];
  A->rows_per_block = (A->max_row_nnz > 0 && A->max_row_nnz <= SPMV_CAP_MAX - 2) ? 1 : 0;  // LDS path usable
  return MFEM_OK;
}

extern "C" int mfem_csr_create(mfem_context ctx, int64_t n, int64_t nnz, const void* rowptr, int rowptr_bits,
                               const int32_t* colidx, int index_base, mfem_csr* out) {
  MFEM_REQUIRE(ctx && out, "null argument");
  MFEM_REQUIRE(n >= 0 && nnz >= 0, "negative size");
  MFEM_REQUIRE(rowptr_bits == 32 || rowptr_bits == 64, "rowptr_bits must be 32 or 64");
  MFEM_REQUIRE(index_base == 0 || index_base == 1, "index_base must be 0 or 1");
  MFEM_REQUIRE(n == 0 || (rowptr && (nnz == 0 || colidx)), "null pattern arrays");
  MFEM_REQUIRE(rowptr_bits == 64 || nnz < ((int64_t)1 << 31), "nnz >= 2^31 needs 64-bit rowptr");
  mfem_csr_s* A = new mfem_csr_s();
  memset(A, 0, sizeof(*A));
  A->ctx = ctx;
  A->n = n;
  A->nnz = nnz;
  A->rowptr = rowptr;
  A->rowptr_bits = rowptr_bits;
  A->colidx = colidx;
  A->index_base = index_base;
  int rc = mfem_csr_plan(ctx, A);
  if (rc != MFEM_OK) {
    delete A;
    return rc;
  }
  *out = A;
  return MFEM_OK;
}

extern "C" int mfem_csr_destroy(mfem_csr A) {
  if (!A) return MFEM_OK;
  // a cached cycle graph holds this pattern's arrays in its kernel arguments
  if (A->ctx) mfem_graphs_invalidate(A->ctx);
  mfem_ell_free(A);
  mfem_sell_free(A);
  if (A->owned_rowptr) hipFree(A->owned_rowptr);
  if (A->owned_colidx) hipFree(A->owned_colidx);
  delete A;
  return MFEM_OK;
}

extern "C" const int64_t* mfem_csr_rowptr64(mfem_csr A) {
  return (A && A->rowptr_bits == 64) ? (const int64_t*)A->rowptr : nullptr;
}
extern "C" const int32_t* mfem_csr_colidx(mfem_csr A) { return A ? A->colidx : nullptr; }
extern "C" int64_t mfem_csr_nnz(mfem_csr A) { return A ? A->nnz : -1; }
extern "C" int64_t mfem_csr_n(mfem_csr A) { return A ? A->n : -1; }
extern "C" int64_t mfem_csr_ncols(mfem_csr A) { return A ? (A->ncols > 0 ? A->ncols : A->n) : -1; }

// Internal launcher: y = alpha*A*x + beta*y, optionally partial sums of (dotw . y) into `partials`
// (*n_partials receives the number written).
static int spmv_launch_inner(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, const double* x, double* y,
                             double alpha, double beta, const double* dotw, double* partials, int* n_partials,
                             const int32_t* done_flag, const SpmvPart& part);

static const SpmvPart kAllRows = {0, 0, {0}, {0}};

struct ProfScope {  // optional hip-event bracket around one SpMV (bench.py's roofline): a split SpMV counts as one launch
  mfem_context_s* ctx;
  int k;
  int begin() {
    k = -1;
    if (!ctx->prof_on) return MFEM_OK;
    if (ctx->prof_used == MFEM_PROF_PAIRS) {
      int rc = mfem_prof_flush(ctx);
      if (rc) return rc;
    }
    k = ctx->prof_used;
    MFEM_CHECK_HIP(hipEventRecord(ctx->prof_ev[2 * k], ctx->stream));
    return MFEM_OK;
  }
  int end() {
    if (k < 0) return MFEM_OK;
    MFEM_CHECK_HIP(hipEventRecord(ctx->prof_ev[2 * k + 1], ctx->stream));
    ctx->prof_used = k + 1;
    return MFEM_OK;
  }
};

int mfem_spmv_launch(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, const double* x, double* y,
                     double alpha, double beta, const double* dotw, double* partials, int* n_partials,
                     const int32_t* done_flag) {
  ProfScope prof{ctx, -1};
  int rc = prof.begin();
  if (rc) return rc;
  rc = spmv_launch_inner(ctx, A, vals, x, y, alpha, beta, dotw, partials, n_partials, done_flag, kAllRows);
  if (rc) return rc;
  return prof.end();
}

// The SpMV of a Krylov loop on a slab: y = alpha A x + beta y where x carries ghost blocks that the neighbours' boundary
// planes must fill first.  The exchange is started on the communicator's stream, the rows that reference no ghost column
// run beside it, the few planes of rows that do run after it has arrived (one extra small launch).  Layouts without a row
// split (the row-sorted sliced layout) wait for the exchange first.  Without a communicator this is mfem_spmv_launch.
static int g_halo_overlap = 1;
extern "C" int mfem_debug_set_halo_overlap(int on) {
  ++mfem_debug_epoch;
  g_halo_overlap = on ? 1 : 0;
  return MFEM_OK;
}

int mfem_spmv_halo(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, double* x, double* y, double alpha, double beta,
                   const double* dotw, double* partials, int* n_partials, const int32_t* done_flag) {
  if (mfem_comm_world(ctx) == 1) return mfem_spmv_launch(ctx, A, vals, x, y, alpha, beta, dotw, partials, n_partials, done_flag);
  ProfScope prof{ctx, -1};
  int rc = prof.begin();
  if (rc) return rc;
  rc = mfem_comm_halo_begin(ctx, x);
  if (rc) return rc;
  SpmvPart P;
  memset(&P, 0, sizeof(P));
  const int F = ctx->halo_fields;
  const bool split = g_halo_overlap && A->n > 0 && 2 * F <= MFEM_MAX_ZONES && !mfem_sell_bound(A, vals) &&
                     A->n == (int64_t)F * mfem_comm_owned_nodes(ctx);
  if (split) {
    const int64_t NO = mfem_comm_owned_nodes(ctx), PL = ctx->halo_plane_len;
    const int rank = mfem_comm_rank(ctx), world = mfem_comm_world(ctx);
    for (int f = 0; f < F; ++f) {
      if (rank > 0) { P.lo[P.nz] = f * NO; P.hi[P.nz] = f * NO + PL; ++P.nz; }
      if (rank < world - 1) { P.lo[P.nz] = (f + 1) * NO - PL; P.hi[P.nz] = (f + 1) * NO; ++P.nz; }
    }
    int np1 = 0, np2 = 0;
    P.part = 1;
    rc = spmv_launch_inner(ctx, A, vals, x, y, alpha, beta, dotw, partials, &np1, done_flag, P);
    if (rc) return rc;
    rc = mfem_comm_halo_end(ctx);
    if (rc) return rc;
    P.part = 2;
    rc = spmv_launch_inner(ctx, A, vals, x, y, alpha, beta, dotw, partials ? partials + np1 : nullptr, &np2, done_flag, P);
    if (rc) return rc;
    if (n_partials) *n_partials = np1 + np2;
    if (np1 + np2 > MFEM_MAX_PARTIALS) {
      mfem_set_error("split SpMV wrote %d partial sums (> %d)", np1 + np2, MFEM_MAX_PARTIALS);
      return MFEM_ERR_INVALID;
    }
  } else {
    rc = mfem_comm_halo_end(ctx);
    if (rc) return rc;
    rc = spmv_launch_inner(ctx, A, vals, x, y, alpha, beta, dotw, partials, n_partials, done_flag, kAllRows);
    if (rc) return rc;
  }
  return prof.end();
}

static int spmv_launch_inner(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, const double* x, double* y,
                             double alpha, double beta, const double* dotw, double* partials, int* n_partials,
                             const int32_t* done_flag, const SpmvPart& part) {
  if (n_partials) *n_partials = 0;
  if (A->n == 0) return MFEM_OK;
  {
    const int e = mfem_spmv_ell_launch(ctx, A, vals, x, y, alpha, beta, dotw, partials, n_partials, done_flag, part);
    if (e != 0) return e < 0 ? e : MFEM_OK;
    if (part.part != 0 && mfem_sell_bound(A, vals)) {
      mfem_set_error("the row-sorted sliced layout has no row split");
      return MFEM_ERR_UNSUPPORTED;
    }
    const int sl = mfem_spmv_sell_launch(ctx, A, vals, x, y, alpha, beta, dotw, partials, n_partials, done_flag);
    if (sl != 0) return sl < 0 ? sl : MFEM_OK;
  }
  const int base = A->index_base;
  const int cap_doubles = (g_spmv_variant == 7 && A->max_row_nnz <= 1008 - 2) ? 1008
                          : ((g_spmv_variant == 2 || g_spmv_variant == 3) && A->max_row_nnz <= 2016 - 2) ? 2016 : 4032;
  if (A->rows_per_block > 0) {
    const bool vec = ((((uintptr_t)vals) & 15) == 0) && ((((uintptr_t)A->colidx) & 7) == 0);
    const int blk = !vec ? MFEM_BLOCK : g_spmv_variant == 6 ? 1024 : (g_spmv_variant == 4 || g_spmv_variant == 5) ? 512
                    : (g_spmv_variant == 7 && cap_doubles == 1008) ? 64 : (g_spmv_variant == 3 && cap_doubles == 2016) ? 64
                    : (g_spmv_variant == 2 && cap_doubles == 2016) ? 128 : MFEM_BLOCK;
    int R = blk;
    while (R > 1 && (int64_t)R * A->max_row_nnz > cap_doubles - 2) R >>= 1;
    int tpr_log2 = 0;
    while ((blk >> (tpr_log2 + 1)) >= R) ++tpr_log2;  // tpr = blk / R
    const int64_t ntiles = (A->n + R - 1) / R;
    int cap = ctx->num_cus * g_spmv_grid_mult;
    if (cap > MFEM_MAX_PARTIALS) cap = MFEM_MAX_PARTIALS;
    if (part.part != 0) cap /= 2;  // the two parts of a split SpMV share one partial-sum array
    if (part.part == 2) {
      int64_t rows = 0;
      for (int z = 0; z < part.nz; ++z) rows += part.hi[z] - part.lo[z];
      const int64_t want = rows / R + 2 * part.nz + 8;
      if (want < cap) cap = (int)want;
    }
    cap &= ~7;  // multiple of 8 so blockIdx % 8 is a stable XCD label along the grid-stride loop
    if (cap < 8) cap = 8;
    int grid = (int)(ntiles < cap ? ((ntiles + 7) & ~(int64_t)7) : cap);
    int xcd = (ntiles >= 64) ? g_spmv_xcd_aware : 0;
    const int xch = xcd & 0xFFFF;
    const int64_t span = (int64_t)8 * (xch > 0 ? xch : 1);
    const int64_t ntiles_padded = xch ? (ntiles + span - 1) / span * span : ntiles;
    if (g_spmv_nogather) xcd |= (1 << 30);
#define LAUNCH_LDS(RP, VEC, CAP, UNR, BLK)                                                                 \
  hipLaunchKernelGGL((k_spmv_lds<RP, VEC, CAP, UNR, BLK>), dim3(grid), dim3(BLK), 0, ctx->stream, A->n,    \
                     A->nnz, (const RP*)A->rowptr, A->colidx, vals, x, y, alpha, beta, base, R, tpr_log2,  \
                     ntiles, ntiles_padded, xcd, dotw, partials, done_flag, part)
#define LAUNCH_VARIANT(RP)                                                   \
  do {                                                                       \
    if (!vec) LAUNCH_LDS(RP, false, 4032, 4, MFEM_BLOCK);                    \
    else if (cap_doubles == 1008) LAUNCH_LDS(RP, true, 1008, 8, 64);          \
    else if (cap_doubles == 2016 && g_spmv_variant == 2) LAUNCH_LDS(RP, true, 2016, 8, 128); \
    else if (cap_doubles == 2016) LAUNCH_LDS(RP, true, 2016, 8, 64);         \
    else if (g_spmv_variant == 1) LAUNCH_LDS(RP, true, 4032, 8, MFEM_BLOCK); \
    else if (g_spmv_variant == 4) LAUNCH_LDS(RP, true, 4032, 4, 512);        \
    else if (g_spmv_variant == 5) LAUNCH_LDS(RP, true, 4032, 2, 512);        \
    else if (g_spmv_variant == 6) LAUNCH_LDS(RP, true, 4032, 2, 1024);       \
    else LAUNCH_LDS(RP, true, 4032, 4, MFEM_BLOCK);                          \
  } while (0)
    if (A->rowptr_bits == 64) LAUNCH_VARIANT(int64_t); else LAUNCH_VARIANT(int32_t);
#undef LAUNCH_VARIANT
#undef LAUNCH_LDS
    MFEM_CHECK_LAUNCH();
    if (n_partials && partials) *n_partials = grid;
  } else {
    const int grid = mfem_grid_for(A->n, 4, ctx->num_cus * 8 < MFEM_MAX_PARTIALS ? ctx->num_cus * 8 : MFEM_MAX_PARTIALS);
    if (A->rowptr_bits == 64)
      hipLaunchKernelGGL(k_spmv_wave_per_row<int64_t>, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, A->n,
                         (const int64_t*)A->rowptr, A->colidx, vals, x, y, alpha, beta, base, dotw, partials, done_flag, part);
    else
      hipLaunchKernelGGL(k_spmv_wave_per_row<int32_t>, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, A->n,
                         (const int32_t*)A->rowptr, A->colidx, vals, x, y, alpha, beta, base, dotw, partials, done_flag, part);
    MFEM_CHECK_LAUNCH();
    if (n_partials && partials) *n_partials = grid;
  }
  return MFEM_OK;
}

extern "C" int mfem_spmv_csr(mfem_context ctx, mfem_csr A, const double* vals, const double* x, double* y,
                             double alpha, double beta) {
  MFEM_REQUIRE(ctx && A, "null handle");
  MFEM_REQUIRE(A->n == 0 || (x && y && (A->nnz == 0 || vals)), "null vector");
  return mfem_spmv_launch(ctx, A, vals, x, y, alpha, beta, nullptr, nullptr, nullptr, nullptr);
}
