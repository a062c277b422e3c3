// CSR SpMV with per-tile x windows (behind mul!: mfem_spmv_csr on the caller's CSR arrays, values not copied).
//
// What bounds the other CSR kernels on mesh matrices is the gather x[col]: one texture-addresser pass per 64 lanes whatever the bytes
// (profiles/r02_ta_gather_probe.txt), ~40 % of them for rows of uneven length (hex-27: 27 .. 125 entries per row, where the
// row-transposing tiles of spmv.hip do not apply).  The columns a tile of consecutive rows touches are a few CONTIGUOUS runs of x
// (the neighbouring lattice lines / the neighbouring nodes of a mesh numbering with locality).  An inspector pass (once per pattern,
// like cusparse's analysis) records per tile those runs -- (first column, length) -- and per nonzero the 16-bit position of its column
// inside the tile's window; the executor stages the window into LDS with coalesced loads of the runs and multiplies from there:
//   * no gather instruction on x at all, 2 B instead of 4 B of index per nonzero (10 B per nonzero from memory instead of 12);
//   * values and positions are read with the same unit-stride 16-byte / 4-byte loads as before, row sums as in the product-tile kernel
//     (same products, same order: bitwise the same y as k_spmv_lds).
// Tiles whose window does not fit (more than XW_WIN distinct columns or XW_MAXSEG runs: rows without locality) keep the column gather, tile
// by tile; nothing about the mesh is assumed.
#include "blas1.h"

#define XW_CAP 2016     // nonzeros per tile (LDS: 16 KB of products)
#define XW_N 2048       // sort size
#define XW_WIN 1536     // distinct columns per tile (LDS: 12 KB of x)
#define XW_MAXSEG 62
#define XW_SEGSTRIDE 128  // ints per tile in the segment table: [0] = runs (-1: gather tile), [1] = distinct columns, then (start, offset) pairs

typedef double xw_d2 __attribute__((ext_vector_type(2)));
typedef int xw_i2 __attribute__((ext_vector_type(2)));

// ---- inspector: one workgroup per tile
template <typename RP>
__global__ __launch_bounds__(MFEM_BLOCK) void k_xw_inspect(int64_t n, const RP* __restrict__ rowptr, const int32_t* __restrict__ col, int base,
                                                             int R, int64_t ntiles, int32_t* __restrict__ seg, uint16_t* __restrict__ loc,
                                                             int32_t* __restrict__ n_window_tiles) {
  __shared__ int32_t key[XW_N];
  __shared__ int32_t uniq[XW_N];
  __shared__ int32_t scan[MFEM_BLOCK + 1];
  __shared__ int32_t s_cnt[2];
  const int tid = threadIdx.x;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t r0 = tile * R, r1 = (r0 + R < n) ? r0 + R : n;
    const int64_t s = (int64_t)rowptr[r0] - base, e = (int64_t)rowptr[r1] - base;
    const int cnt = (int)(e - s);  // <= XW_CAP by the choice of R
    for (int i = tid; i < XW_N; i += MFEM_BLOCK) key[i] = i < cnt ? col[s + i] - base : INT32_MAX;
    __syncthreads();
    // bitonic sort of the tile's columns
    for (int k = 2; k <= XW_N; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < XW_N; i += MFEM_BLOCK) {
          const int l = i ^ j;
          if (l > i) {
            const int32_t a = key[i], b = key[l];
            const bool up = (i & k) == 0;
            if ((a > b) == up) {
              key[i] = b;
              key[l] = a;
            }
          }
        }
        __syncthreads();
      }
    // distinct columns: flags -> exclusive scan (8 elements per thread, then across the workgroup)
    int flag[XW_N / MFEM_BLOCK], local = 0;
#pragma unroll
    for (int u = 0; u < XW_N / MFEM_BLOCK; ++u) {
      const int i = tid * (XW_N / MFEM_BLOCK) + u;
      flag[u] = (i < cnt && (i == 0 || key[i] != key[i - 1])) ? 1 : 0;
      local += flag[u];
    }
    scan[tid + 1] = local;
    if (tid == 0) scan[0] = 0;
    __syncthreads();
    if (tid == 0)
      for (int i = 1; i <= MFEM_BLOCK; ++i) scan[i] += scan[i - 1];
    __syncthreads();
    const int D = scan[MFEM_BLOCK];
    {
      int pos = scan[tid];
#pragma unroll
      for (int u = 0; u < XW_N / MFEM_BLOCK; ++u) {
        const int i = tid * (XW_N / MFEM_BLOCK) + u;
        if (flag[u]) uniq[pos++] = key[i];
      }
    }
    if (tid == 0) s_cnt[0] = 0;
    __syncthreads();
    // runs of consecutive columns
    int32_t* st = seg + tile * XW_SEGSTRIDE;
    bool window = D <= XW_WIN && D > 0;
    if (window) {
      for (int i = tid; i < D; i += MFEM_BLOCK)
        if (i == 0 || uniq[i] != uniq[i - 1] + 1) atomicAdd(&s_cnt[0], 1);
    }
    __syncthreads();
    const int nseg = s_cnt[0];
    window = window && nseg <= XW_MAXSEG;
    if (window) {
      // the runs in ascending order: run q starts at the q-th flagged position (serial over <= 1536 entries: once per pattern)
      if (tid == 0) {
        int q = 0;
        for (int i = 0; i < D; ++i)
          if (i == 0 || uniq[i] != uniq[i - 1] + 1) {
            st[2 + 2 * q] = uniq[i];
            st[3 + 2 * q] = i;
            ++q;
          }
        st[0] = nseg;
        st[1] = D;
        atomicAdd(n_window_tiles, 1);
      }
      // position of every nonzero's column in the window = its rank among the distinct columns (binary search)
      for (int i = tid; i < cnt; i += MFEM_BLOCK) {
        const int32_t c = col[s + i] - base;
        int lo = 0, hi = D - 1;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (uniq[mid] < c) lo = mid + 1;
          else hi = mid;
        }
        loc[s + i] = (uint16_t)lo;
      }
    } else if (tid == 0) {
      st[0] = -1;
      st[1] = 0;
    }
    __syncthreads();
  }
}

// ---- executor
// Persistent workgroups; everything tile t + G needs from memory (its values, positions and window entries) is requested while tile t
// is being multiplied, its run table one tile earlier still: a tile costs two barriers and no exposed round trip.
struct XwTile {
  int64_t s, e;   // nonzero range
  int nseg, D;    // runs (-1: gather tile), distinct columns
};
template <typename RP>
__global__ __launch_bounds__(MFEM_BLOCK) __attribute__((amdgpu_waves_per_eu(4))) void k_spmv_xw(int64_t n, int64_t nnz, const RP* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                          const double* __restrict__ vals, const int32_t* __restrict__ seg,
                                                          const uint16_t* __restrict__ loc, const double* __restrict__ x, double* __restrict__ y,
                                                          double alpha, double beta, int base, int R, int tpr_log2, int64_t ntiles,
                                                          const double* __restrict__ dotw, double* __restrict__ partials,
                                                          const int32_t* __restrict__ done_flag, SpmvPart part) {
  constexpr int U = (XW_CAP / 2 + MFEM_BLOCK - 1) / MFEM_BLOCK;  // value pairs per thread
  constexpr int XU = XW_WIN / MFEM_BLOCK;                        // window entries per thread
  __shared__ double prod[XW_CAP + 4];
  __shared__ double xw[XW_WIN];
  __shared__ int32_t s_start[2][XW_MAXSEG + 2], s_off[2][XW_MAXSEG + 2];
  __shared__ double red[MFEM_BLOCK / 64];
  if (done_flag && done_flag[0]) return;
  const int tid = threadIdx.x, tpr = 1 << tpr_log2;
  const int64_t G = gridDim.x;
  double dot_acc = 0.0;
  auto tile_meta = [&](int64_t tile) -> XwTile {
    XwTile T{0, 0, 0, 0};
    if (tile < ntiles) {
      const int64_t r0 = tile * R, r1 = (r0 + R < n) ? r0 + R : n;
      T.s = (int64_t)rowptr[r0] - base;
      T.e = (int64_t)rowptr[r1] - base;
      T.nseg = seg[tile * XW_SEGSTRIDE];
      T.D = seg[tile * XW_SEGSTRIDE + 1];
    }
    return T;
  };
  // run table of a tile -> registers of the first XW_MAXSEG threads (entries past the tile's runs are never used)
  auto seg_request = [&](int64_t tile, int32_t& st, int32_t& of) {
    st = 0;
    of = 0;
    if (tile < ntiles && tid < XW_MAXSEG) {
      st = seg[tile * XW_SEGSTRIDE + 2 + 2 * tid];
      of = seg[tile * XW_SEGSTRIDE + 3 + 2 * tid];
    }
  };
  auto seg_store = [&](int buf, const XwTile& T, int32_t st, int32_t of) {
    if (tid < XW_MAXSEG) {
      s_start[buf][tid] = st;
      s_off[buf][tid] = (T.nseg > 0 && tid < T.nseg) ? of : 0x7fffffff;  // sentinel: window positions are below every unused offset
    }
  };
  // values / positions of a window tile and its window entries (run table in LDS buffer `buf`)
  auto request = [&](const XwTile& T, int buf, xw_d2 (&v)[U], uint32_t (&lc)[U], double (&xv)[XU]) {
    const int64_t sa = T.s & ~(int64_t)1;
    const int cnt = (int)(T.e - sa);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = 2 * (tid + u * MFEM_BLOCK);
      v[u] = (xw_d2){0.0, 0.0};
      lc[u] = 0u;
      if (T.nseg > 0 && i < cnt) {
        if (sa + i + 1 < nnz) {
          v[u] = __builtin_nontemporal_load(reinterpret_cast<const xw_d2*>(vals + sa + i));
          lc[u] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(loc + sa + i));
        } else {  // last odd entry of the whole matrix
          v[u].x = vals[sa + i];
          lc[u] = loc[sa + i];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < XU; ++u) {
      const int i = tid + u * MFEM_BLOCK;
      xv[u] = 0.0;
      if (T.nseg > 0 && i < T.D) {
        int lo = 0, hi = T.nseg - 1;  // the run holding window position i
        while (lo < hi) {
          const int mid = (lo + hi + 1) >> 1;
          if (s_off[buf][mid] <= i) lo = mid;
          else hi = mid - 1;
        }
        xv[u] = x[s_start[buf][lo] + (i - s_off[buf][lo])];
      }
    }
  };
  // ---- prologue: tile t0's run table and data, tile t0 + G's run table
  int64_t tile = blockIdx.x;
  XwTile T0 = tile_meta(tile), T1 = tile_meta(tile + G);
  int32_t sst, sof;
  seg_request(tile, sst, sof);
  seg_store(0, T0, sst, sof);
  seg_request(tile + G, sst, sof);
  seg_store(1, T1, sst, sof);
  __syncthreads();
  xw_d2 v[U], vn[U];
  uint32_t lc[U], lcn[U];
  double xv[XU], xvn[XU];
  request(T0, 0, v, lc, xv);
  int buf = 0;
  for (; tile < ntiles; tile += G) {
    const int64_t r0 = tile * R, r1 = (r0 + R < n) ? r0 + R : n;
    const bool skip = spmv_part_skip(part, r0, r1);  // uniform per workgroup
    const int64_t sa = T0.s & ~(int64_t)1;
    const int cnt = (int)(T0.e - sa);
    // 1. this tile's window into LDS
    if (T0.nseg > 0) {
#pragma unroll
      for (int u = 0; u < XU; ++u) {
        const int i = tid + u * MFEM_BLOCK;
        if (i < T0.D) xw[i] = xv[u];
      }
    }
    __syncthreads();  // window complete; the next tile's run table (stored at the end of the previous iteration) visible
    // 2. the next tile's data and the run table of the tile after it are requested now
    const XwTile T2 = tile_meta(tile + 2 * G);
    request(T1, buf ^ 1, vn, lcn, xvn);
    seg_request(tile + 2 * G, sst, sof);
    // phase-2 row bounds of this lane's first row
    const int64_t rmine = r0 + (tid >> tpr_log2);
    int lo_pre = 0, hi_pre = 0;
    if (rmine < r1) {
      lo_pre = (int)((int64_t)rowptr[rmine] - base - sa);
      hi_pre = (int)((int64_t)rowptr[rmine + 1] - base - sa);
    }
    // 3. products
    if (!skip) {
      if (T0.nseg > 0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int i = 2 * (tid + u * MFEM_BLOCK);
          if (i < cnt) {
            // an entry in front of the tile (i = 0 when s is odd) or behind it carries a position of a neighbouring tile: its product is
            // never read; keep the LDS read in range
            const uint32_t p0 = lc[u] & 0xFFFFu, p1 = lc[u] >> 16;
            const double x0 = xw[p0 < (uint32_t)T0.D ? p0 : 0], x1 = xw[p1 < (uint32_t)T0.D ? p1 : 0];
            *reinterpret_cast<xw_d2*>(&prod[i]) = (xw_d2){v[u].x * x0, v[u].y * x1};
          }
        }
      } else {
        // gather tile (no locality in this tile's columns): the product-tile path of spmv.hip, not pipelined
        for (int i0 = 2 * tid; i0 < cnt; i0 += 2 * MFEM_BLOCK) {
          xw_d2 vv = {0.0, 0.0};
          xw_i2 c = {base, base};
          if (sa + i0 + 1 < nnz) {
            vv = __builtin_nontemporal_load(reinterpret_cast<const xw_d2*>(vals + sa + i0));
            c = __builtin_nontemporal_load(reinterpret_cast<const xw_i2*>(col + sa + i0));
          } else {
            vv.x = vals[sa + i0];
            c.x = col[sa + i0];
          }
          const double x0 = x[c.x - base];
          const double x1 = (i0 + 1 < cnt) ? x[c.y - base] : 0.0;
          *reinterpret_cast<xw_d2*>(&prod[i0]) = (xw_d2){vv.x * x0, vv.y * x1};
        }
      }
    }
    __syncthreads();
    // 4. row sums: tpr lanes per row
    if (!skip) {
      const int g = tid & (tpr - 1);
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
    }
    // 5. rotate: the run table of tile + 2 G takes this tile's buffer (every lane passed the barrier after its last use in step 2 of the
    //    previous iteration... its readers were the window requests of THIS tile, issued one iteration ago)
    seg_store(buf, T2, sst, sof);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      v[u] = vn[u];
      lc[u] = lcn[u];
    }
#pragma unroll
    for (int u = 0; u < XU; ++u) xv[u] = xvn[u];
    T0 = T1;
    T1 = T2;
    buf ^= 1;
  }
  if (partials) {
    const double b = block_reduce_sum(dot_acc, red);
    if (tid == 0) partials[blockIdx.x] = b;
  }
}

// ---- host side
static int64_t g_xw_launches = 0;
extern "C" int64_t mfem_debug_xw_spmv_count(void) { return g_xw_launches; }

static int xw_rows_per_tile(const mfem_csr_s* A) {
  int R = MFEM_BLOCK;
  while (R > 1 && (int64_t)R * A->max_row_nnz > XW_CAP - 2) R >>= 1;
  return R;
}

// xw_state: 0 not inspected, -1 not used, 1 ready
int mfem_xw_plan(mfem_context_s* ctx, mfem_csr_s* A) {
  if (A->xw_state != 0) return MFEM_OK;
  A->xw_state = -1;
  if (A->n < 65536 || A->nnz < 1 || A->max_row_nnz < 1 || A->max_row_nnz > XW_CAP - 2) return MFEM_OK;  // small systems are launch-bound
  const int R = xw_rows_per_tile(A);
  const int64_t ntiles = (A->n + R - 1) / R;
  if (ntiles * XW_SEGSTRIDE >= ((int64_t)1 << 40)) return MFEM_OK;
  if (hipMalloc(&A->xw_seg, sizeof(int32_t) * (size_t)ntiles * XW_SEGSTRIDE) != hipSuccess) {
    (void)hipGetLastError();
    A->xw_seg = nullptr;
    return MFEM_OK;
  }
  if (hipMalloc(&A->xw_loc, sizeof(uint16_t) * (size_t)(A->nnz + 2)) != hipSuccess) {
    (void)hipGetLastError();
    hipFree(A->xw_seg);
    A->xw_seg = nullptr;
    A->xw_loc = nullptr;
    return MFEM_OK;
  }
  int32_t* d_cnt = ctx->d_flags + 9;
  MFEM_CHECK_HIP(hipMemsetAsync(d_cnt, 0, sizeof(int32_t), ctx->stream));
  MFEM_CHECK_HIP(hipMemsetAsync(A->xw_loc, 0, sizeof(uint16_t) * (size_t)(A->nnz + 2), ctx->stream));
  const int grid = (int)(ntiles < (int64_t)ctx->num_cus * 16 ? ntiles : (int64_t)ctx->num_cus * 16);
  if (A->rowptr_bits == 64)
    hipLaunchKernelGGL(k_xw_inspect<int64_t>, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, A->n, (const int64_t*)A->rowptr, A->colidx,
                       A->index_base, R, ntiles, A->xw_seg, A->xw_loc, d_cnt);
  else
    hipLaunchKernelGGL(k_xw_inspect<int32_t>, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, A->n, (const int32_t*)A->rowptr, A->colidx,
                       A->index_base, R, ntiles, A->xw_seg, A->xw_loc, d_cnt);
  MFEM_CHECK_LAUNCH();
  MFEM_CHECK_HIP(hipMemcpyAsync(ctx->h_flags + 9, d_cnt, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  MFEM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  A->xw_window_tiles = ctx->h_flags[9];
  if ((double)A->xw_window_tiles < 0.5 * (double)ntiles) {  // mostly gather tiles: nothing gained over the other kernels
    hipFree(A->xw_seg);
    hipFree(A->xw_loc);
    A->xw_seg = nullptr;
    A->xw_loc = nullptr;
    return MFEM_OK;
  }
  A->xw_R = R;
  A->xw_state = 1;
  return MFEM_OK;
}

void mfem_xw_free(mfem_csr_s* A) {
  if (A->xw_seg) hipFree(A->xw_seg);
  if (A->xw_loc) hipFree(A->xw_loc);
  A->xw_seg = nullptr;
  A->xw_loc = nullptr;
  A->xw_state = 0;
}

// returns 1 if launched, 0 if another kernel should be used, < 0 on error
int mfem_xw_launch(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, const double* x, double* y, double alpha, double beta,
                   const double* dotw, double* partials, int* n_partials, const int32_t* done_flag, const SpmvPart& part) {
  if (A->xw_state != 1 || (((uintptr_t)vals) & 15) != 0) return 0;
  const int R = A->xw_R;
  int tpr_log2 = 0;
  while ((MFEM_BLOCK >> (tpr_log2 + 1)) >= R) ++tpr_log2;  // tpr = 256 / R lanes per row
  const int64_t ntiles = (A->n + R - 1) / R;
  int cap = ctx->num_cus * 5;  // 28.5 KB of LDS per workgroup: five resident per CU
  if (cap > MFEM_MAX_PARTIALS) cap = MFEM_MAX_PARTIALS;
  if (part.part != 0) cap /= 2;  // the two parts of a split SpMV share one partial-sum array
  if (part.part == 2) {
    int64_t rows = 0;
    for (int z = 0; z < part.nz; ++z) rows += part.hi[z] - part.lo[z];
    const int64_t want = rows / R + 2 * part.nz + 8;
    if (want < cap) cap = (int)want;
  }
  const int grid = (int)(ntiles < cap ? ntiles : cap);
  ++g_xw_launches;
  if (A->rowptr_bits == 64)
    hipLaunchKernelGGL(k_spmv_xw<int64_t>, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, A->n, A->nnz, (const int64_t*)A->rowptr, A->colidx, vals,
                       A->xw_seg, A->xw_loc, x, y, alpha, beta, A->index_base, R, tpr_log2, ntiles, dotw, partials, done_flag, part);
  else
    hipLaunchKernelGGL(k_spmv_xw<int32_t>, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, A->n, A->nnz, (const int32_t*)A->rowptr, A->colidx, vals,
                       A->xw_seg, A->xw_loc, x, y, alpha, beta, A->index_base, R, tpr_log2, ntiles, dotw, partials, done_flag, part);
  MFEM_CHECK_LAUNCH();
  if (n_partials && partials) *n_partials = grid;
  return 1;
}
