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

  // the row-pointer pair of a tile is requested one tile ahead: otherwise every tile starts with a dependent HBM round trip
  // (rowptr -> addresses of the value / column streams) that nothing in the workgroup can hide
  int64_t s_next = 0, e_next = 0;
  {
    const int64_t tile = tile_of(blockIdx.x, ntiles, xcd_aware & 0xFFFF);
    if (blockIdx.x < ntiles_padded && tile < ntiles) {
      const int64_t r0 = tile * R, r1 = (r0 + R < n) ? r0 + R : n;
      s_next = (int64_t)rowptr[r0] - base;
      e_next = (int64_t)rowptr[r1] - base;
    }
  }
  for (int64_t it = blockIdx.x; it < ntiles_padded; it += gridDim.x) {
    const int64_t tile = tile_of(it, ntiles, xcd_aware & 0xFFFF);
    const int64_t s = s_next, e = e_next;
    {
      const int64_t itn = it + gridDim.x;
      const int64_t tn = tile_of(itn, ntiles, xcd_aware & 0xFFFF);
      if (itn < ntiles_padded && tn < ntiles) {
        const int64_t q0 = tn * R, q1 = (q0 + R < n) ? q0 + R : n;
        s_next = (int64_t)rowptr[q0] - base;
        e_next = (int64_t)rowptr[q1] - base;
      }
    }
    if (tile >= ntiles) continue;  // uniform per workgroup
    const int64_t r0 = tile * R;
    const int64_t r1 = (r0 + R < n) ? r0 + R : n;
    if (spmv_part_skip(part, r0, r1)) continue;  // uniform per workgroup

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
            const double x0 = x[c[u].x - base];
            const double x1 = (i + 1 < cnt) ? x[c[u].y - base] : 0.0;
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

// Row-transposing tile kernel.  The product-tile kernel above gathers x[col] in CSR order: the 128 nonzeros of one wave
// instruction span ~5 rows x 27 entries, i.e. ~12 different cache lines of x per instruction, and that instruction stream --
// not bytes -- is what it loses its time on (tools/gather_probe.hip).  Here the tile's val/col streams are staged RAW in LDS
// (same coalesced 16-byte / 8-byte loads), and after the barrier a lane walks ITS ROW's entries from LDS: the lanes of a wave
// then hold neighbouring rows at the same position of the row, whose columns are neighbouring entries of x (2-4 cache lines
// per gather instruction) -- the access order of the slot-major solver layouts, without a copy of the matrix.  tpr lanes
// share a row (entries lo + g, lo + g + tpr, ...) and combine by sub-wave shuffle; rows of any length (general CSR).
template <typename RP, int CAP, int BLK, int GU>
__global__ __launch_bounds__(BLK) void k_spmv_csr_t(
    int64_t n, int64_t nnz, const RP* __restrict__ rowptr, const int32_t* __restrict__ col,
    const double* __restrict__ vals, const double* __restrict__ x, double* __restrict__ y, double alpha,
    double beta, int base, int R, int tpr_log2, int64_t ntiles, const double* __restrict__ dotw,
    double* __restrict__ partials, const int32_t* __restrict__ done_flag, SpmvPart part) {
  __shared__ __attribute__((aligned(16))) double sv[CAP + 4];
  __shared__ __attribute__((aligned(16))) int32_t sc[CAP + 4];
  __shared__ double red[BLK / 64 < 4 ? 4 : BLK / 64];
  if (done_flag && done_flag[0]) return;
  const int tid = threadIdx.x;
  const int tpr = 1 << tpr_log2;
  const int g = tid & (tpr - 1);
  constexpr int LU = (CAP / 2 + BLK - 1) / BLK;  // 16-byte loads per lane that cover a full tile
  double dot_acc = 0.0;
  int64_t s_next = 0, e_next = 0;  // row-pointer pair of the next tile, requested one tile ahead
  if (blockIdx.x < ntiles) {
    const int64_t r0 = (int64_t)blockIdx.x * R, r1 = (r0 + R < n) ? r0 + R : n;
    s_next = (int64_t)rowptr[r0] - base;
    e_next = (int64_t)rowptr[r1] - base;
  }
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t r0 = tile * R;
    const int64_t r1 = (r0 + R < n) ? r0 + R : n;
    const int64_t s = s_next, e = e_next;
    if (tile + gridDim.x < ntiles) {
      const int64_t q0 = (tile + gridDim.x) * R, q1 = (q0 + R < n) ? q0 + R : n;
      s_next = (int64_t)rowptr[q0] - base;
      e_next = (int64_t)rowptr[q1] - base;
    }
    if (spmv_part_skip(part, r0, r1)) continue;  // uniform per workgroup
    const int64_t sa = s & ~(int64_t)1;  // 16-byte aligned start (vals / col bases are 16-byte / 8-byte aligned)
    const int cnt = (int)(e - sa);
    const int64_t rmine = r0 + (tid >> tpr_log2);
    int lo_pre = 0, hi_pre = 0;
    if (rmine < r1) {  // requested now: the latency hides under the tile loads
      lo_pre = (int)((int64_t)rowptr[rmine] - base - sa);
      hi_pre = (int)((int64_t)rowptr[rmine + 1] - base - sa);
    }
    {
      d2_t v[LU];
      i2_t c[LU];
#pragma unroll
      for (int u = 0; u < LU; ++u) {
        const int i = 2 * tid + u * 2 * BLK;
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
      for (int u = 0; u < LU; ++u) {
        const int i = 2 * tid + u * 2 * BLK;
        if (i < cnt) {
          *reinterpret_cast<d2_t*>(&sv[i]) = v[u];
          *reinterpret_cast<i2_t*>(&sc[i]) = c[u];
        }
      }
    }
    __syncthreads();
    for (int64_t r = rmine; r < r1; r += (BLK >> tpr_log2)) {
      const int lo = (r == rmine) ? lo_pre : (int)((int64_t)rowptr[r] - base - sa);
      const int hi = (r == rmine) ? hi_pre : (int)((int64_t)rowptr[r + 1] - base - sa);
      double sum = 0.0;
      int j = lo + g;
      for (; j + (GU - 1) * tpr < hi; j += GU * tpr) {
        double vv[GU], xx[GU];
#pragma unroll
        for (int u = 0; u < GU; ++u) {
          vv[u] = sv[j + u * tpr];
          xx[u] = x[sc[j + u * tpr] - base];
        }
#pragma unroll
        for (int u = 0; u < GU; ++u) sum += vv[u] * xx[u];
      }
      for (; j < hi; j += tpr) sum += sv[j] * x[sc[j] - base];
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

static std::atomic<int> g_csr_w_strips{0};                       // mfem_debug_set_csr_strips -- OFF: measured, no gain (profiles/r05_csr_strips.txt)
static std::atomic<int64_t> g_csr_w_strip_min_bytes{3 << 20};  // two lattice planes of x beyond this many bytes -> XCD strips
extern "C" int mfem_debug_set_csr_strips(int on, int64_t min_bytes) try {
  ++mfem_debug_epoch;
  g_csr_w_strips = on ? 1 : 0;
  if (min_bytes >= 0) g_csr_w_strip_min_bytes = min_bytes;
  return MFEM_OK;
} MFEM_API_CATCH("mfem_debug_set_csr_strips")

// Wave-private row-transposing tiles.  What bounds the two kernels above is the texture addresser: a 64-lane gather costs
// ~17 cycles when the lanes read consecutive entries of x and ~100 cycles in CSR order with a nonzero pair per lane (42 distinct
// cache lines per instruction; tools/ta_probe.hip), i.e. ~1.2 ms of addresser time per SpMV at 256^3.  Here a WAVE owns a run
// of R = 64 / tpr consecutive rows: it stages their val / col streams raw in its own LDS block (coalesced 16-byte / 8-byte
// loads, CSR order) and then lane l walks row l / tpr -- with tpr = 1 (rows of <= 31 entries) the 64 lanes of a gather hold
// the same position of 64 consecutive rows, which for a mesh matrix are consecutive entries of x.  No workgroup barrier, no
// cross-lane reduction for tpr = 1, y written unit-stride.
template <typename RP, int CAPW, int WAVES, int NG>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(CAPW > 2048 ? 1 : 2))) void k_spmv_csr_w(
    int64_t n, int64_t nnz, const RP* __restrict__ rowptr, const int32_t* __restrict__ col,
    const double* __restrict__ vals, const double* __restrict__ x, double* __restrict__ y, double alpha,
    double beta, int base, int R, int tpr_log2, int64_t ntiles, const double* __restrict__ dotw,
    double* __restrict__ partials, const int32_t* __restrict__ done_flag, SpmvPart part, const uint8_t* __restrict__ elide, int64_t strip_tp) {
  constexpr int LU = (CAPW / 2 + 63) / 64;  // (16 B + 8 B) loads per lane that cover a full tile
  static_assert(CAPW % 128 == 0, "the staging loop stores whole 128-entry groups");
  // NG = gathers a lane issues up front (rows of up to NG * tpr entries have none left over)
  __shared__ __attribute__((aligned(16))) double sv_all[WAVES][CAPW + 2];
  __shared__ __attribute__((aligned(16))) int32_t sc_all[WAVES][CAPW + 4];
  __shared__ double red[4];
  if (done_flag && done_flag[0]) return;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // (scalar: the tile index and what is loaded with it then are, too)
  double* sv = sv_all[w];
  int32_t* sc = sc_all[w];
  const int tpr = 1 << tpr_log2;
  const int g = lane & (tpr - 1);
  const int rsel = lane >> tpr_log2;  // row of the tile this lane group walks
  double dot_acc = 0.0;
  const int64_t tstride = (int64_t)gridDim.x * WAVES;
  // x as a buffer resource (byte offsets are 32-bit: the host side uses this kernel only while 8 * columns < 4 GiB)
  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(x), 0, 0xFFFFFFFF, 0x00020000);
  // Round 5 -- the pipeline below only works when the NUMBER of loads between a load and its first use is the same on every path: the compiler's
  // s_waitcnt vmcnt(N) for "the gathers have returned" counts the loads issued behind them, and where paths with different counts meet (a request behind
  // `if (t_next < ntiles)`, a column stream behind `if (!el)`, a gather behind `if (j < hi)`) it must assume the smallest -- vmcnt(0), i.e. the row sums waited
  // for the next tile's streams as well, and a value read back with readfirstlane right behind its load (row pointers, the elision flag) waited for every
  // gather in front of it.  Now: tile-level scalars come through scalar loads (the tile index is wave-uniform), every vector load is issued on every path --
  // a column stream that is not needed aims past the end of its bounds-checked buffer (returns zero, moves no data), a lane without an entry gathers x[0].
  // Software pipeline per wave: the tile after the current one sits in registers (requested while the current tile's
  // gathers were in flight), the row-pointer pair of the tile after that is requested one step earlier still.
  d2_t pv[LU];
  i2_t pc[LU];
  // Which tile a wave takes next.  Default: tiles round-robin over the grid -- all XCDs move along ONE front through the matrix, and a line of x is held by
  // an L2 from its first use (as the upper neighbour plane of a row) to its last (lower neighbour plane): two lattice planes of x, 1 MB at 256^3 but 4.2 MB at
  // 512^3 -- more than the 4 MB L2 of an XCD, so x came in three times (counter traffic 1.12x the design bytes, round 4).
  // strip_tp > 0 (round 5, an experiment kept behind mfem_debug_set_csr_strips, OFF by default: 8.34 against 8.23 ms at 512^3 -- the re-read x comes from the
  // Infinity Cache and is not what the kernel's time follows): tiles per lattice plane, rounded up.  The workgroups of XCD c (blockIdx % 8:
  // round-robin dispatch) then take, in every plane, the tiles [tp c / 8, tp (c + 1) / 8) -- an eighth of the plane swept through all planes, whose x window
  // (3 planes x 1 / 8 plane + two lines) stays in that XCD's L2.  Same tiles, same sums, another order of the walk: bitwise the same y.
  const int64_t xc = blockIdx.x & 7;
  const int64_t sb = strip_tp > 0 ? strip_tp * xc / 8 : 0, sx = strip_tp > 0 ? strip_tp * (xc + 1) / 8 - sb : 1;
  const int64_t qstride = strip_tp > 0 ? (int64_t)(gridDim.x >> 3) * WAVES : tstride;
  int64_t q_cur = strip_tp > 0 ? (int64_t)(blockIdx.x >> 3) * WAVES + w : (int64_t)blockIdx.x * WAVES + w;
  auto tile_at = [&](int64_t q) -> int64_t {  // tile of walk position q; >= ntiles: past the end (and so is every later position)
    if (strip_tp <= 0) return q;
    const int64_t p = q / sx, t = p * strip_tp + sb + (q - p * sx);
    return p * strip_tp >= ntiles ? ntiles : (t < ntiles ? t : -1);  // -1: this position holds no tile (the last, partial plane), later ones may
  };
  // skip positions without a tile and tiles that belong to the other part of a split SpMV (wave-uniform)
  auto next_tile = [&](int64_t& q) -> int64_t {
    for (;;) {
      const int64_t t = tile_at(q);
      if (t >= ntiles) return ntiles;
      if (t >= 0 && !spmv_part_skip(part, t * R, (t * R + R < n) ? t * R + R : n)) return t;
      q += qstride;
    }
  };
  int64_t t_cur = next_tile(q_cur);
  int64_t sa_cur = 0;
  int cnt_cur = 0, lo_cur = 0, hi_cur = 0;
  auto uniform64 = [](int64_t v) -> int64_t {  // the value is the same in every lane: keep it in scalar registers
    const uint32_t lo32 = __builtin_amdgcn_readfirstlane((uint32_t)v), hi32 = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi32 << 32) | lo32);
  };
  // elide[t] != 0: every row of tile t repeats the column offsets of the tile's first row (inspected once per pattern, k_csr_w_elide): only
  // that row's columns are read (the first 128 staged entries hold them)
  auto request = [&](int64_t tq, int64_t& sa, int& cnt, int& lo, int& hi, int& el) {  // issue the loads of tile t into pv / pc
    const int64_t t = uniform64(tq);  // (wave-uniform: the loads below that depend on it alone are scalar loads)
    const int64_t r0 = t * R, r1 = (r0 + R < n) ? r0 + R : n;
    const int64_t s = (int64_t)rowptr[r0] - base, e = (int64_t)rowptr[r1] - base;
    sa = s & ~(int64_t)1;
    cnt = (int)(e - sa);
    el = elide ? (int)elide[t] : 0;
    const int64_t r = r0 + rsel, rr = r < r1 ? r : r1 - 1;  // (a lane group behind the tile's last row reads that row's pointers and keeps an empty range)
    const int lo_r = (int)((int64_t)rowptr[rr] - base - sa), hi_r = (int)((int64_t)rowptr[rr + 1] - base - sa);
    lo = r < r1 ? lo_r : 0;
    hi = r < r1 ? hi_r : 0;
    // the tile's two streams as bounds-checked buffers (base in scalar registers, one offset register per lane, entries past
    // the tile's end read as zero): no per-load address pairs, no masks.  A tile whose rows repeat their first row's column offsets (el) reads 128 columns:
    // its column buffer ends there, the loads behind it return zeros without touching memory
    const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(vals + sa), 0, cnt * 8, 0x00020000);
    const __amdgpu_buffer_rsrc_t cr = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(col + sa), 0, (el && cnt > 128 ? 128 : cnt) * 4, 0x00020000);
#pragma unroll
    for (int u = 0; u < LU; ++u) {
      pv[u] = __builtin_bit_cast(d2_t, __builtin_amdgcn_raw_buffer_load_b128(vr, lane * 16, u * 1024, 2));
      pc[u] = __builtin_bit_cast(i2_t, __builtin_amdgcn_raw_buffer_load_b64(cr, lane * 8, u * 512, 2));
    }
  };
  int el_cur = 0;
  if (t_cur < ntiles) request(t_cur, sa_cur, cnt_cur, lo_cur, hi_cur, el_cur);
  while (t_cur < ntiles) {
    // ---- the requested tile goes to the wave's LDS block
#pragma unroll
    for (int u = 0; u < LU; ++u) {
      const int i = 2 * lane + u * 128;  // < CAPW: entries past the tile's end are zeros nobody reads
      *reinterpret_cast<d2_t*>(&sv[i]) = pv[u];
      if (!el_cur || u == 0) *reinterpret_cast<i2_t*>(&sc[i]) = pc[u];
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the wave's LDS stores have landed
    __builtin_amdgcn_wave_barrier();
    const int64_t r0 = t_cur * R, r1 = (r0 + R < n) ? r0 + R : n;
    const int64_t r = r0 + rsel;
    const int lo = lo_cur, hi = hi_cur;
    const int el = el_cur, lo0 = __builtin_amdgcn_readfirstlane(lo_cur);  // (lane 0 walks the tile's first row)
    // the column of entry j: staged, or -- el -- that of the same entry of the tile's first row, + the row's distance from it
    auto colof = [&](int j) -> int { return el ? sc[lo0 + (j - lo)] + rsel : sc[j]; };
    // ---- all gathers of the lane's row first ...
    double xx[NG];
    const int j0 = lo + g;
#pragma unroll
    for (int u = 0; u < NG; ++u) {
      const int j = j0 + u * tpr;
      // buffer form of the load: one 32-bit offset register per gather instead of a 64-bit address pair (28 gathers in flight); a lane whose row has no
      // entry j reads x[first column of the vector] instead (the product is dropped below) -- every lane issues every gather
      const int off = j < hi ? (colof(j) - base) * 8 : 0;
      xx[u] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(xres, off, 0, 0));
    }
    // ---- ... then the request for the next tile: it returns behind the gathers (loads return in order), so the row sums
    //      below do not wait for it, and it is in flight while they run
    q_cur += qstride;
    const int64_t t_next = next_tile(q_cur);
    int64_t sa_n = 0;
    int cnt_n = 0, lo_n = 0, hi_n = 0, el_n = 0;
    // (the LDS block is still being read below: the next tile stays in registers until the top of the loop; behind the last tile the current one is
    // requested once more -- its registers are never used)
    request(t_next < ntiles ? t_next : t_cur, sa_n, cnt_n, lo_n, hi_n, el_n);
    double sum = 0.0;
#pragma unroll
    for (int u = 0; u < NG; ++u) {
      const int j = j0 + u * tpr;
      sum += j < hi ? sv[j] * xx[u] : 0.0;
    }
    for (int j = j0 + NG * tpr; j < hi; j += tpr) sum += sv[j] * x[colof(j) - base];  // rows longer than NG * tpr entries
    for (int off = tpr >> 1; off > 0; off >>= 1) sum += __shfl_xor(sum, off, MFEM_WAVE);
    if (g == 0 && r < r1) {
      double yv = alpha * sum;
      if (beta != 0.0) yv += beta * y[r];
      y[r] = yv;
      if (dotw) dot_acc += yv * dotw[r];
    }
    __builtin_amdgcn_wave_barrier();  // every lane is done reading the block before the next tile's stores
    t_cur = t_next;
    sa_cur = sa_n;
    cnt_cur = cnt_n;
    lo_cur = lo_n;
    hi_cur = hi_n;
    el_cur = el_n;
  }
  if (partials) {
    const double b = block_reduce_sum(dot_acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = b;
  }
}

// Wave-private tiles cut by NONZEROS (rows of uneven length: hex-27's 27 / 45 / 75 / 125-entry rows, unstructured meshes).  The kernel
// above spends about the same time on a tile whatever it holds (one round of staging loads + gathers per tile and wave), and tiles of
// a fixed row count must be sized for the longest row: on the hex-27 matrix they are 0.3 - 0.5 full.  Here tile t holds the rows
// [rs[t], rs[t + 1]) with rs[t] = first row whose nonzeros start at or behind t * C (mfem_csr_plan_rowblocks, once per pattern,
// C = capacity - longest row - 2): every tile is C +- one row of nonzeros, 0.9+ full.  The rows of a tile are walked in groups of
// 64 / tpr (tpr chosen per tile from its row count), a row by tpr lanes in chunks of NG gathers; the tile's row pointers are staged
// in LDS next to its val / col streams.  Same software pipeline as above (next tile's streams requested behind the first chunk of
// gathers); the tile's row range is requested one tile earlier still.
#define RB_ROWS 128  // row pointers staged per tile (tiles with more rows -- runs of very short rows -- read the rest from memory)
#ifndef RB_WAVES_PER_EU
#define RB_WAVES_PER_EU 2
#endif
template <typename RP, int CAPW, int NG>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RB_WAVES_PER_EU))) void k_spmv_csr_rb(
    int64_t n, int64_t nnz, const RP* __restrict__ rowptr, const int32_t* __restrict__ col,
    const double* __restrict__ vals, const double* __restrict__ x, double* __restrict__ y, double alpha,
    double beta, int base, int64_t ntiles, const int32_t* __restrict__ rs, const double* __restrict__ dotw,
    double* __restrict__ partials, const int32_t* __restrict__ done_flag, int xcd_runs) {
  constexpr int LU = (CAPW / 2 + 63) / 64;
  static_assert(CAPW % 128 == 0, "the staging loop stores whole 128-entry groups");
  __shared__ __attribute__((aligned(16))) double sv[CAPW + 2];
  __shared__ __attribute__((aligned(16))) int32_t sc[CAPW + 4];
  __shared__ int32_t srp[RB_ROWS];
  __shared__ double sred[64];
  if (done_flag && done_flag[0]) return;
  const int lane = threadIdx.x;
  double dot_acc = 0.0;
  // xcd_runs: workgroups with equal blockIdx % 8 share an XCD (round-robin dispatch) and walk one contiguous eighth of the tiles
  const int64_t tstride = xcd_runs ? gridDim.x >> 3 : gridDim.x;
  const int64_t t_begin = xcd_runs ? ntiles * (blockIdx.x & 7) / 8 : 0, t_end = xcd_runs ? ntiles * ((blockIdx.x & 7) + 1) / 8 : ntiles;
  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(x), 0, 0xFFFFFFFF, 0x00020000);
  d2_t pv[LU];
  i2_t pc[LU];
  int32_t prp[2];  // row pointers r0 + lane, r0 + 64 + lane of the requested tile, relative to its first staged entry
  auto uniform32 = [](int32_t v) -> int32_t { return __builtin_amdgcn_readfirstlane(v); };
  auto uniform64 = [](int64_t v) -> int64_t {
    const uint32_t lo32 = __builtin_amdgcn_readfirstlane((uint32_t)v), hi32 = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi32 << 32) | lo32);
  };
  // request the streams of the tile with rows [r0, r1)
  auto request = [&](int32_t r0, int32_t r1, int64_t& sa, int el) {
    const int64_t s = uniform64((int64_t)rowptr[r0] - base), e = uniform64((int64_t)rowptr[r1] - base);
    sa = s & ~(int64_t)1;
    const int cnt = (int)(e - sa);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int64_t r = (int64_t)r0 + 64 * h + lane;
      prp[h] = r <= r1 ? (int32_t)((int64_t)rowptr[r] - base - sa) : 0;
    }
    const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(vals + sa), 0, cnt * 8, 0x00020000);
    const __amdgpu_buffer_rsrc_t cr = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(col + sa), 0, cnt * 4, 0x00020000);
#pragma unroll
    for (int u = 0; u < LU; ++u) {
      pv[u] = __builtin_bit_cast(d2_t, __builtin_amdgcn_raw_buffer_load_b128(vr, lane * 16, u * 1024, 2));
      // a tile whose rows repeat the column offsets of its first two rows (el): only those two rows' columns are read (<= 256 entries)
      if (!el || u < 2) pc[u] = __builtin_bit_cast(i2_t, __builtin_amdgcn_raw_buffer_load_b64(cr, lane * 8, u * 512, 2));
    }
  };
  int64_t t_cur = xcd_runs ? t_begin + (blockIdx.x >> 3) : blockIdx.x;
  int32_t r0 = 0, r1 = 0, r0n = 0, r1n = 0;  // rows of the current tile / of the tile after it
  int el = 0, eln = 0;                         // ... and their column-elision flags (bit 31 of rs[t])
  int64_t sa_cur = 0;
  if (t_cur < t_end) {
    const uint32_t w0 = (uint32_t)uniform32(rs[t_cur]);
    r0 = (int32_t)(w0 & 0x7fffffffu);
    el = (int)(w0 >> 31);
    r1 = uniform32(rs[t_cur + 1]) & 0x7fffffff;
    request(r0, r1, sa_cur, el);
  }
  if (t_cur + tstride < t_end) {
    const uint32_t w0 = (uint32_t)uniform32(rs[t_cur + tstride]);
    r0n = (int32_t)(w0 & 0x7fffffffu);
    eln = (int)(w0 >> 31);
    r1n = uniform32(rs[t_cur + tstride + 1]) & 0x7fffffff;
  }
  while (t_cur < t_end) {
#pragma unroll
    for (int u = 0; u < LU; ++u) {
      const int i = 2 * lane + u * 128;
      *reinterpret_cast<d2_t*>(&sv[i]) = pv[u];
      if (!el || u < 2) *reinterpret_cast<i2_t*>(&sc[i]) = pc[u];
    }
    srp[lane] = prp[0];
    srp[64 + lane] = prp[1];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    const int nr = r1 - r0;
    // The rows of the tile in two passes -- rows 0, 2, 4, .. then 1, 3, 5, .. (on an order-2 lattice neighbouring rows alternate between
    // node types with different stencil sizes, rows two apart share theirs) -- each pass with all 64 lanes: nc rows get tpr = 64 / nc
    // lanes each (any quotient, not only powers of two), lane = g * nc + slot, so that the lanes of a gather instruction hold the same
    // stencil position of neighbouring same-type rows and (almost) none of them idles while a longer row finishes.
    const int64_t t_next = t_cur + tstride;
    int64_t sa_n = 0;
    bool requested = false;
    for (int c = 0; c < 2; ++c) {
      const int ncl = (nr + 1 - c) >> 1;  // rows c, c + 2, ...
      for (int b0 = 0; b0 < ncl || !requested; b0 += 64) {
        const int nc = ncl - b0 < 64 ? (ncl - b0 > 0 ? ncl - b0 : 1) : 64;
        const int tpr = 64 / nc, g = lane / nc, slot = lane - g * nc;
        const int ri = c + 2 * (b0 + slot);
        int lo = 0, hi = 0;
        if (g < tpr && b0 + slot < ncl) {
          if (ri + 1 < RB_ROWS) {
            lo = srp[ri];
            hi = srp[ri + 1];
          } else {  // beyond the staged row pointers
            lo = (int)((int64_t)rowptr[(int64_t)r0 + ri] - base - sa_cur);
            hi = (int)((int64_t)rowptr[(int64_t)r0 + ri + 1] - base - sa_cur);
          }
        }
        const int cb = el ? srp[c] : 0, dcol = 2 * (b0 + slot);
        double sum = 0.0;
        int j = lo + g;
        do {
          double xx[NG];
#pragma unroll
          for (int u = 0; u < NG; ++u) {
            const int jj = j + u * tpr;
            xx[u] = 0.0;
            // el: the column of entry e of row c + 2 m is that of entry e of row c (the first row of the parity class), + 2 m
            if (jj < hi) xx[u] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(xres, ((el ? sc[cb + (jj - lo)] + dcol : sc[jj]) - base) * 8, 0, 0));
          }
          if (!requested) {  // behind the tile's first gathers: the next tile's streams
            requested = true;
            if (t_next < t_end) request(r0n, r1n, sa_n, eln);
          }
#pragma unroll
          for (int u = 0; u < NG; ++u) {
            const int jj = j + u * tpr;
            sum += (jj < hi ? sv[jj] : 0.0) * xx[u];
          }
          j += NG * tpr;
        } while (__any(j < hi));
        // the tpr partial sums of a row meet in LDS (tpr is any quotient: no butterfly)
        __builtin_amdgcn_wave_barrier();
        sred[lane] = sum;
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        if (lane < nc && b0 + lane < ncl) {
          double tot = 0.0;
          for (int q = 0; q < tpr; ++q) tot += sred[q * nc + lane];
          const int64_t r = (int64_t)r0 + c + 2 * (b0 + lane);
          double yv = alpha * tot;
          if (beta != 0.0) yv += beta * y[r];
          y[r] = yv;
          if (dotw) dot_acc += yv * dotw[r];
        }
      }
    }
    __builtin_amdgcn_wave_barrier();  // every lane is done reading the block before the next tile's stores
    t_cur = t_next;
    sa_cur = sa_n;
    r0 = r0n;
    r1 = r1n;
    el = eln;
    if (t_cur + tstride < t_end) {
      const uint32_t w0 = (uint32_t)uniform32(rs[t_cur + tstride]);
      r0n = (int32_t)(w0 & 0x7fffffffu);
      eln = (int)(w0 >> 31);
      r1n = uniform32(rs[t_cur + tstride + 1]) & 0x7fffffff;
    }
  }
  if (partials) {
    const double w = wave_reduce_sum(dot_acc);
    if (lane == 0) partials[blockIdx.x] = w;
  }
}

// rs[t] = first row whose nonzeros start at or behind t * C (t = 0 .. ntiles - 1), rs[ntiles] = n
template <typename RP>
__global__ __launch_bounds__(MFEM_BLOCK) void k_csr_rowblocks(int64_t n, const RP* __restrict__ rowptr, int base, int64_t C, int64_t ntiles,
                                                               int32_t* __restrict__ rs) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t <= ntiles; t += (int64_t)gridDim.x * blockDim.x) {
    if (t == ntiles) {
      rs[t] = (int32_t)n;
      continue;
    }
    const int64_t target = t * C;
    int64_t lo = 0, hi = n;  // first r in [0, n] with rowptr[r] - base >= target
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if ((int64_t)rowptr[mid] - base >= target) hi = mid;
      else lo = mid + 1;
    }
    rs[t] = (int32_t)lo;
  }
}

// Column elision: tile t is marked (bit 31 of rs[t]) when its rows of equal parity all repeat the column OFFSETS (col - row) of the
// tile's first row of that parity -- interior rows of a lattice stencil do; rows next to the mesh boundary, or a tile that straddles two
// lattice lines of different node types, do not.  The SpMV then reads the columns of the tile's first two rows only.  One wave per tile,
// a lane per row; the pattern is read once when it is created.
template <typename RP>
__global__ __launch_bounds__(MFEM_BLOCK) void k_csr_rb_elide(int64_t ntiles, const RP* __restrict__ rowptr, const int32_t* __restrict__ col, int base,
                                                              int32_t* __restrict__ rs, int32_t* __restrict__ count) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t t = wave; t < ntiles; t += nwaves) {
    const int32_t r0 = rs[t] & 0x7fffffff, r1 = rs[t + 1] & 0x7fffffff;
    const int nr = r1 - r0;
    bool ok = nr >= 1 && nr <= 64;  // (wave-uniform)
    int64_t lo = 0;
    int len = 0;
    if (ok && lane < nr) {
      lo = (int64_t)rowptr[r0 + lane] - base;
      len = (int)((int64_t)rowptr[r0 + lane + 1] - base - lo);
    }
    const int c = lane & 1;
    const int64_t lob = __shfl(lo, c, 64);
    const int lenb = __shfl(len, c, 64);
    const int len0 = __shfl(len, 0, 64), len1 = __shfl(len, 1, 64);
    bool match = true;
    if (ok && lane < nr) {
      match = len == lenb;
      const int d = lane - c;
      for (int e = 0; match && e < len; ++e) match = col[lo + e] - col[lob + e] == d;
    }
    ok = ok && len0 + len1 <= 254 && __all(match);
    if (ok && lane == 0) {
      rs[t] = (int32_t)((uint32_t)r0 | 0x80000000u);
      atomicAdd(count, 1);
    }
  }
}

// The same inspection for the tiles of a fixed row count (k_spmv_csr_w): flag[t] = 1 when every row of tile t repeats the column offsets
// of the tile's first row (one stencil for all rows: hex-8 operators away from the lattice line ends).
template <typename RP>
__global__ __launch_bounds__(MFEM_BLOCK) void k_csr_w_elide(int64_t n, int R, int64_t ntiles, const RP* __restrict__ rowptr,
                                                             const int32_t* __restrict__ col, int base, uint8_t* __restrict__ flag) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t t = wave; t < ntiles; t += nwaves) {
    const int64_t r0 = t * R, r1 = (r0 + R < n) ? r0 + R : n;
    const int nr = (int)(r1 - r0);  // <= 64
    int64_t lo = 0;
    int len = 0;
    if (lane < nr) {
      lo = (int64_t)rowptr[r0 + lane] - base;
      len = (int)((int64_t)rowptr[r0 + lane + 1] - base - lo);
    }
    const int64_t lo0 = __shfl(lo, 0, 64);
    const int len0 = __shfl(len, 0, 64);
    bool match = true;
    if (lane < nr) {
      match = len == len0;
      for (int e = 0; match && e < len; ++e) match = col[lo + e] - col[lo0 + e] == lane;
    }
    const bool ok = len0 >= 1 && len0 <= 126 && __all(match);
    if (lane == 0) flag[t] = ok ? 1 : 0;
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
static std::atomic<int> g_spmv_xcd_aware{0};
static std::atomic<int> g_rb_elide{1};  // row-block kernel: tiles whose rows repeat the column offsets of their first two rows read only those columns; bit 25 turns the inspection off (set before the pattern is created)
static std::atomic<int> g_rb_xcd{1};  // row-block kernel: an XCD walks a contiguous eighth of the tiles (hex-27 128^3: 2.89 against 3.01 ms with round-robin tiles); bit 26 turns it off
static std::atomic<int> g_spmv_tile2688{1};  // bit 27 of mfem_debug_set_spmv's first argument turns the 2688-entry wave tile off
static std::atomic<int> g_spmv_grid_mult{8};  // workgroups per CU of the persistent grid
static std::atomic<int> g_spmv_grid_mult_set{0};  // the caller chose it (mfem_debug_set_spmv): also applies to the wave-private kernel, which otherwise sizes its grid from what is resident
// Kernel variant (bits 16-18 of mfem_debug_set_spmv's first argument):
//   0 library default: 7 where tiles of a fixed row count fill their LDS block, 3 otherwise
//   1 product tile, CAP 4032, a nonzero PAIR per lane and load (16-byte / 8-byte loads), 8 pairs in flight, 256 threads
//   (2: as 1; a product tile with ONE nonzero per lane and load -- gathers over 64 consecutive nonzeros -- measured 1.48-1.58 ms
//   against 1.19 ms and was removed)
//   3 wave tiles cut by nonzeros (k_spmv_csr_rb; set before the pattern is created, the row blocks are planned then); the product tile
//     where they do not apply (split SpMV, unaligned arrays, fewer than 16 entries per row)
//   4 row-transposing workgroup tile, CAP 4032, 256 threads
//   6 wave-private row-transposing tiles (1792 / 2048 entries per wave), 2 waves per workgroup   7 (and 5) the same, 1 wave
static std::atomic<int> g_spmv_variant{0};

extern "C" int mfem_debug_set_spmv(int xcd_aware, int grid_mult) try {  // tuning hook for bench/profiling
  ++mfem_debug_epoch;
  g_spmv_xcd_aware = xcd_aware & 0xFFFF;   // tiles per XCD run (0 = plain round-robin)
  g_spmv_variant = (xcd_aware >> 16) & 7;
  g_spmv_tile2688 = ((xcd_aware >> 27) & 1) ? 0 : 1;
  g_rb_xcd = ((xcd_aware >> 26) & 1) ? 0 : 1;
  g_rb_elide = ((xcd_aware >> 25) & 1) ? 0 : 1;
  g_spmv_grid_mult_set = grid_mult > 0;
  g_spmv_grid_mult = grid_mult > 0 ? grid_mult : 8;
  return MFEM_OK;
} MFEM_API_CATCH("mfem_debug_set_spmv")

// Measured on the hex-27 128^3 matrix (capacity, gathers in flight, workgroups per CU): (2048, 16, 6) 2.99 ms, (1792, 16, 7) 2.86,
// (1536, 16, 8) 2.62, (1536, 20, 8) 3.02, (1280, 16, 8) 2.82, (1024, 12, 12) 4.47 -- two waves on every SIMD, the largest tile that allows it
// (overridable with -D for the sweep of tools/rb_sweep.sh: profiles/r05_csr_rb_sweep.txt)
#ifndef RB_CAP
#define RB_CAP 1536  // entries per tile of the row-block kernel (19.5 KB of LDS: eight one-wave workgroups per CU)
#endif
#ifndef RB_NG
#define RB_NG 16    // gathers a lane has in flight
#endif
#ifndef RB_WG_PER_CU
#define RB_WG_PER_CU 8
#endif
// do wave tiles of a fixed row count fill their LDS block (>= 0.65)?  (rows of near-uniform length: k_spmv_csr_w)
static bool csr_w_fills(const mfem_csr_s* A) {
  if (!(A->max_row_nnz > 0 && A->max_row_nnz <= 2048 - 2) || A->n == 0) return false;
  int tl = 0;
  while (tl < 6 && (int64_t)(64 >> tl) * A->max_row_nnz > 1792 - 2) ++tl;
  const double fill = (double)(64 >> tl) * ((double)A->nnz / (double)A->n) / 1792.0;
  if (tl <= 3 && fill >= 0.65) return true;
  return tl >= 1 && (int64_t)(128 >> tl) * A->max_row_nnz <= 2688 - 2 && 2.0 * fill * 1792.0 / 2688.0 >= 0.65;  // the 2688-entry tile
}

// the default kernel: wave tiles of a fixed row count where they fill AND the rows are short (<= 64 entries: hex-8 scalar, 256^3 1.21 ms
// against 1.30 ms for the row blocks); row blocks for wide rows of uniform length too (three fields, 81 entries: 1.33 against 1.40 ms --
// with more columns per row the x window of a tile range is what an XCD-contiguous walk keeps in one L2)
static bool csr_w_default(const mfem_csr_s* A) { return csr_w_fills(A) && A->max_row_nnz <= 64; }

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
  A->nb_F = 0;  // (asked by the layout plan: mfem_node_block_fields)
  A->nb_checked = 0;
  // rows of uneven length (tiles of a fixed row count would be less than 0.65 full): tiles cut by nonzeros, k_spmv_csr_rb
  A->rb_state = -1;
  if (A->rows_per_block > 0 && (g_spmv_variant == 3 || !csr_w_default(A)) && A->max_row_nnz <= RB_CAP / 4 && A->nnz >= 16 * A->n && A->n < ((int64_t)1 << 31) - 1) {
    const int64_t C = RB_CAP - 2 - A->max_row_nnz, ntiles = (A->nnz + C - 1) / C;
    MFEM_CHECK_HIP(hipMalloc(&A->rb_rows, sizeof(int32_t) * (size_t)(ntiles + 1)));
    const int g = mfem_grid_for(ntiles + 1, MFEM_BLOCK, 4096);
    if (A->rowptr_bits == 64)
      hipLaunchKernelGGL(k_csr_rowblocks<int64_t>, dim3(g), dim3(MFEM_BLOCK), 0, ctx->stream, A->n, (const int64_t*)A->rowptr, A->index_base, C,
                         ntiles, A->rb_rows);
    else
      hipLaunchKernelGGL(k_csr_rowblocks<int32_t>, dim3(g), dim3(MFEM_BLOCK), 0, ctx->stream, A->n, (const int32_t*)A->rowptr, A->index_base, C,
                         ntiles, A->rb_rows);
    MFEM_CHECK_LAUNCH();
    A->rb_ntiles = ntiles;
    A->rb_state = 1;
    A->rb_elided = 0;
    if (g_rb_elide) {
      int32_t* d_cnt = ctx->d_flags + 8;
      MFEM_CHECK_HIP(hipMemsetAsync(d_cnt, 0, sizeof(int32_t), ctx->stream));
      const int ge = mfem_grid_for(ntiles * 64, MFEM_BLOCK, ctx->num_cus * 16);
      if (A->rowptr_bits == 64)
        hipLaunchKernelGGL(k_csr_rb_elide<int64_t>, dim3(ge), dim3(MFEM_BLOCK), 0, ctx->stream, ntiles, (const int64_t*)A->rowptr, A->colidx,
                           A->index_base, A->rb_rows, d_cnt);
      else
        hipLaunchKernelGGL(k_csr_rb_elide<int32_t>, dim3(ge), dim3(MFEM_BLOCK), 0, ctx->stream, ntiles, (const int32_t*)A->rowptr, A->colidx,
                           A->index_base, A->rb_rows, d_cnt);
      MFEM_CHECK_LAUNCH();
      MFEM_CHECK_HIP(hipMemcpyAsync(ctx->h_flags + 8, d_cnt, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
      MFEM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
      A->rb_elided = ctx->h_flags[8];
    }
  }
  // tiles of a fixed row count (the default for short rows of uniform length): the same inspection, one flag per tile of Rw rows
  A->cw_R = 0;
  if (g_rb_elide && A->rows_per_block > 0 && csr_w_default(A)) {
    auto tpr_for = [&](int capw) {
      int tl = 0;
      while (tl < 6 && (int64_t)(64 >> tl) * A->max_row_nnz > capw - 2) ++tl;
      return tl;
    };
    const bool big = tpr_for(2048) < tpr_for(1792);
    const int Rw = 64 >> (big ? tpr_for(2048) : tpr_for(1792));  // (rows of up to 64 entries never take the 2688-entry tile)
    const int64_t ntw = (A->n + Rw - 1) / Rw;
    MFEM_CHECK_HIP(hipMalloc(&A->cw_elide, (size_t)ntw));
    const int ge = mfem_grid_for(ntw * 64, MFEM_BLOCK, ctx->num_cus * 16);
    if (A->rowptr_bits == 64)
      hipLaunchKernelGGL(k_csr_w_elide<int64_t>, dim3(ge), dim3(MFEM_BLOCK), 0, ctx->stream, A->n, Rw, ntw, (const int64_t*)A->rowptr, A->colidx,
                         A->index_base, A->cw_elide);
    else
      hipLaunchKernelGGL(k_csr_w_elide<int32_t>, dim3(ge), dim3(MFEM_BLOCK), 0, ctx->stream, A->n, Rw, ntw, (const int32_t*)A->rowptr, A->colidx,
                         A->index_base, A->cw_elide);
    MFEM_CHECK_LAUNCH();
    A->cw_R = Rw;
  }
  return MFEM_OK;
}

// everything the handle derived from the borrowed pattern arrays (not the arrays themselves)
static void csr_drop_plans(mfem_csr_s* A) {
  mfem_ell_unbind(A);
  mfem_sell_unbind(A);
  mfem_lat27_unbind(A);
  mfem_lat8_unbind(A);
  A->lat8_state = 0;
  A->lat_refused = 0;
  if (A->lat_inferred) {  // (a hint read off the arrays goes with them)
    A->lat_fields = A->lat_m0 = A->lat_m1 = A->lat_m2 = A->lat_plo = A->lat_gw = 0;
    A->lat_inferred = 0;
  }
  mfem_ell_free(A);
  mfem_sell_free(A);
  mfem_rem_free(A);
  A->lat27_state = 0;
  A->sym_state = 0;
  A->symp_state = 0;
  if (A->rb_rows) hipFree(A->rb_rows);
  if (A->cw_elide) hipFree(A->cw_elide);
  if (A->diag_off) hipFree(A->diag_off);
  A->rb_rows = nullptr;
  A->cw_elide = nullptr;
  A->diag_off = nullptr;
  A->rb_state = 0;
  A->rb_ntiles = A->rb_elided = 0;
  A->cw_R = 0;
}

extern "C" int mfem_csr_create(mfem_context ctx, int64_t n, int64_t nnz, const void* rowptr, int rowptr_bits,
                               const int32_t* colidx, int index_base, mfem_csr* out) try {
  MFEM_REQUIRE(ctx && out, "null argument");
  MFEM_REQUIRE(n >= 0 && nnz >= 0, "negative size");
  MFEM_REQUIRE(rowptr_bits == 32 || rowptr_bits == 64, "rowptr_bits must be 32 or 64");
  MFEM_REQUIRE(index_base == 0 || index_base == 1, "index_base must be 0 or 1");
  MFEM_REQUIRE(n == 0 || (rowptr && (nnz == 0 || colidx)), "null pattern arrays");
  MFEM_REQUIRE(rowptr_bits == 64 || nnz < ((int64_t)1 << 31), "nnz >= 2^31 needs 64-bit rowptr");
  mfem_host_alloc_probe();
  mfem_csr_s* A = new mfem_csr_s();
  memset(A, 0, sizeof(*A));
  A->ctx = ctx;
  A->n = n;
  A->nnz = nnz;
  A->rowptr = rowptr;
  A->rowptr_bits = rowptr_bits;
  A->colidx = colidx;
  A->index_base = index_base;
  int rc = MFEM_OK;
  try {
    rc = mfem_csr_plan(ctx, A);
  } catch (...) {  // (a host allocation of the inspection failed: nothing half-planned is left behind; the entry point's handler reports it)
    csr_drop_plans(A);
    delete A;
    throw;
  }
  if (rc != MFEM_OK) {
    csr_drop_plans(A);  // whatever the failed plan step left behind (row blocks, elision flags)
    delete A;
    return rc;
  }
  *out = A;
  return MFEM_OK;
} MFEM_API_CATCH("mfem_csr_create")

// The handle caches what it learnt from the borrowed rowptr / colidx (longest row, row blocks, which tiles repeat one column-offset
// list, the solver layouts).  A caller that has rewritten those arrays in place (same n, same nnz) re-runs the inspection here.
extern "C" int mfem_csr_replan(mfem_context ctx, mfem_csr A) try {
  MFEM_REQUIRE(ctx && A, "null argument");
  MFEM_REQUIRE(A->ctx == ctx, "the pattern belongs to another context");
  mfem_graphs_invalidate(ctx);
  csr_drop_plans(A);
  return mfem_csr_plan(ctx, A);
} MFEM_API_CATCH("mfem_csr_replan")

// Column entries (4 bytes each) one launch of the default CSR kernel reads by design: all of them in a tile whose rows do not repeat one
// offset list, the leading 128 / 256 staged entries (the first row / the first two rows) in a tile that does.
template <typename RP>
__global__ __launch_bounds__(MFEM_BLOCK) void k_csr_cols_read(int64_t n, int64_t ntiles, int R, const int32_t* __restrict__ rs,
                                                               const uint8_t* __restrict__ flag, const RP* __restrict__ rowptr, int base,
                                                               unsigned long long* __restrict__ total) {
  unsigned long long acc = 0;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < ntiles; t += (int64_t)gridDim.x * blockDim.x) {
    int64_t r0, r1;
    int el, lead;
    if (rs) {
      const uint32_t w0 = (uint32_t)rs[t];
      r0 = (int64_t)(w0 & 0x7fffffffu);
      r1 = (int64_t)(rs[t + 1] & 0x7fffffff);
      el = (int)(w0 >> 31);
      lead = 256;
    } else {
      r0 = t * R;
      r1 = (r0 + R < n) ? r0 + R : n;
      el = flag ? (int)flag[t] : 0;
      lead = 128;
    }
    const int64_t s0 = (int64_t)rowptr[r0] - base, e = (int64_t)rowptr[r1] - base;
    const int64_t staged = e - (s0 & ~(int64_t)1);  // the staged run starts on an even entry
    acc += (unsigned long long)(el ? (staged < lead ? staged : lead) : e - s0);
  }
  acc = (unsigned long long)wave_reduce_sum((double)acc);  // exact below 2^53
  if ((threadIdx.x & 63) == 0 && acc) atomicAdd(total, acc);
}

extern "C" int mfem_csr_spmv_bytes(mfem_context ctx, mfem_csr A, int64_t* bytes, int64_t* column_entries_read) try {
  MFEM_REQUIRE(ctx && A && bytes, "null argument");
  int64_t cols = A->nnz, table = 0;
  const bool rb = A->rb_state == 1 && A->rb_rows, cw = !rb && A->cw_R > 0 && A->cw_elide;
  if ((rb || cw) && A->n > 0) {
    unsigned long long* d_tot = reinterpret_cast<unsigned long long*>(ctx->d_flags + 8);  // (8-byte aligned: d_flags is hipMalloc'ed)
    MFEM_CHECK_HIP(hipMemsetAsync(d_tot, 0, sizeof(unsigned long long), ctx->stream));
    const int64_t nt = rb ? A->rb_ntiles : (A->n + A->cw_R - 1) / A->cw_R;
    const int g = mfem_grid_for(nt, MFEM_BLOCK, 4096);
    if (A->rowptr_bits == 64)
      hipLaunchKernelGGL(k_csr_cols_read<int64_t>, dim3(g), dim3(MFEM_BLOCK), 0, ctx->stream, A->n, nt, A->cw_R, rb ? A->rb_rows : nullptr,
                         rb ? nullptr : A->cw_elide, (const int64_t*)A->rowptr, A->index_base, d_tot);
    else
      hipLaunchKernelGGL(k_csr_cols_read<int32_t>, dim3(g), dim3(MFEM_BLOCK), 0, ctx->stream, A->n, nt, A->cw_R, rb ? A->rb_rows : nullptr,
                         rb ? nullptr : A->cw_elide, (const int32_t*)A->rowptr, A->index_base, d_tot);
    MFEM_CHECK_LAUNCH();
    unsigned long long h = 0;
    MFEM_CHECK_HIP(hipMemcpyAsync(&h, d_tot, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    MFEM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    cols = (int64_t)h;
    table = rb ? (nt + 1) * 4 : nt;  // the tile table itself (first rows + flag bit / one flag byte per tile)
  }
  if (column_entries_read) *column_entries_read = cols;
  // values once, the columns the kernel reads, x once (gathers of one entry by several rows are cache hits by design), y once, row pointers once
  *bytes = A->nnz * 8 + cols * 4 + A->n * 16 + (A->n + 1) * (A->rowptr_bits / 8) + table;
  return MFEM_OK;
} MFEM_API_CATCH("mfem_csr_spmv_bytes")

extern "C" int mfem_csr_destroy(mfem_csr A) try {
  if (!A) return MFEM_OK;
  // a cached cycle graph holds this pattern's arrays in its kernel arguments
  if (A->ctx && mfem_context_alive(A->ctx)) mfem_graphs_invalidate(A->ctx);
  mfem_ell_free(A);
  mfem_sell_free(A);
  mfem_rem_free(A);
  if (A->rb_rows) hipFree(A->rb_rows);
  if (A->cw_elide) hipFree(A->cw_elide);
  if (A->diag_off) hipFree(A->diag_off);
  if (A->owned_rowptr) hipFree(A->owned_rowptr);
  if (A->owned_colidx) hipFree(A->owned_colidx);
  delete A;
  return MFEM_OK;
} MFEM_API_CATCH("mfem_csr_destroy")

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
// run beside it, the few planes of rows that do run after it has arrived (one extra small launch).  The row-sorted sliced layout
// splits by blocks instead of zones: its ghost-reading rows are sorted behind all others when the pattern is planned.  Without a
// communicator this is mfem_spmv_launch.
static std::atomic<int> g_halo_overlap{1};
extern "C" int mfem_debug_set_halo_overlap(int on) try {
  ++mfem_debug_epoch;
  g_halo_overlap = on ? 1 : 0;
  return MFEM_OK;
} MFEM_API_CATCH("mfem_debug_set_halo_overlap")

int mfem_spmv_halo(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, double* x, double* y, double alpha, double beta,
                   const double* dotw, double* partials, int* n_partials, const int32_t* done_flag) {
  if (mfem_comm_world(ctx) == 1) return mfem_spmv_launch(ctx, A, vals, x, y, alpha, beta, dotw, partials, n_partials, done_flag);
  ProfScope prof{ctx, -1};
  int rc = prof.begin();
  if (rc) return rc;
  if (mfem_lat8_bound(A, vals) || mfem_lat27_bound(A, vals)) {
    // lattice tiles split by i-LAYERS of tiles, not by row zones: the layers that stage no ghost plane run beside the exchange (part 1), the top
    // layers and the gather pass -- which reads the lower ghost planes for the first owned rows -- after it (part 2)
    rc = mfem_comm_halo_begin(ctx, x);
    if (rc) return rc;
    SpmvPart P;
    memset(&P, 0, sizeof(P));
    if (g_halo_overlap) {
      P.part = 1;
      rc = spmv_launch_inner(ctx, A, vals, x, y, alpha, beta, dotw, partials, n_partials, done_flag, P);
      if (rc) {
        mfem_comm_halo_end(ctx);
        return rc;
      }
      P.part = 2;
    }
    rc = mfem_comm_halo_end(ctx);
    if (rc) return rc;
    rc = spmv_launch_inner(ctx, A, vals, x, y, alpha, beta, dotw, partials, n_partials, done_flag, P);
    if (rc) return rc;
    return prof.end();
  }
  rc = mfem_comm_halo_begin(ctx, x);
  if (rc) return rc;
  // an error between begin and end must not leave the exchange "in flight": the communicator would refuse every later one
  auto fail = [&](int code) {
    mfem_comm_halo_end(ctx);
    return code;
  };
  SpmvPart P;
  memset(&P, 0, sizeof(P));
  const int F = ctx->halo_fields;
  const bool split = g_halo_overlap && A->n > 0 && 2 * F <= MFEM_MAX_ZONES && A->n == (int64_t)F * mfem_comm_owned_nodes(ctx);
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
    if (rc) return fail(rc);
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
  if (!ctx->force_csr) {
    const int l8 = mfem_spmv_lat8_launch(ctx, A, vals, x, y, alpha, beta, dotw, partials, n_partials, done_flag, part.part);
    if (l8 != 0) return l8 < 0 ? l8 : MFEM_OK;
    const int e = mfem_spmv_ell_launch(ctx, A, vals, x, y, alpha, beta, dotw, partials, n_partials, done_flag, part);
    if (e != 0) return e < 0 ? e : MFEM_OK;
    const int lt = mfem_spmv_lat27_launch(ctx, A, vals, x, y, alpha, beta, dotw, partials, n_partials, done_flag, part.part);
    if (lt != 0) return lt < 0 ? lt : MFEM_OK;
    // (the sliced layout splits by BLOCKS: its ghost-reading rows are sorted behind all others, whatever the zones say)
    const int sl = mfem_spmv_sell_launch(ctx, A, vals, x, y, alpha, beta, dotw, partials, n_partials, done_flag, part.part);
    if (sl != 0) return sl < 0 ? sl : MFEM_OK;
  }
  const int base = A->index_base;
  // default: wave-private row-transposing tiles of a fixed row count when a wave's 64 / tpr rows fill its LDS block reasonably (rows of
  // near-uniform length: 256^3 hex-8 1.06 ms against 1.19 ms for the product tile); wave tiles cut by nonzeros otherwise (hex-27's
  // 27..125-entry rows: 2.6 - 3.0 ms against 3.4 - 3.6)
  int variant = g_spmv_variant;
  if (variant == 0) variant = csr_w_default(A) ? 7 : 3;
  if (variant == 3 && A->rb_state == 1 && part.part == 0 && ((((uintptr_t)vals) & 15) == 0) && ((((uintptr_t)A->colidx) & 7) == 0) &&
      (A->ncols > 0 ? A->ncols : A->n) < ((int64_t)1 << 29)) {
    // tiles cut by nonzeros (rows of uneven length); the grid is what is resident
    // (persistent grid = what is resident at once: one workgroup more per CU than fits runs as a second round and doubles the time -- the runtime says how
    // many of these one-wave workgroups a CU holds, RB_WG_PER_CU is the upper bound)
    static std::atomic<int> rb_resident[2] = {{0}, {0}};
    std::atomic<int>& res = rb_resident[A->rowptr_bits == 64 ? 1 : 0];
    if (res == 0) {
      int occ = 0;
      const hipError_t eo = A->rowptr_bits == 64
          ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(&k_spmv_csr_rb<int64_t, RB_CAP, RB_NG>), 64, 0)
          : hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(&k_spmv_csr_rb<int32_t, RB_CAP, RB_NG>), 64, 0);
      res = (eo == hipSuccess && occ >= 1 && occ <= RB_WG_PER_CU) ? occ : RB_WG_PER_CU;
    }
    int grid = ctx->num_cus * (g_spmv_grid_mult_set ? g_spmv_grid_mult.load() : res.load());
    if (grid > MFEM_MAX_PARTIALS) grid = MFEM_MAX_PARTIALS;
    if ((int64_t)grid > A->rb_ntiles) grid = (int)A->rb_ntiles;
    if (A->rowptr_bits == 64)
      hipLaunchKernelGGL((k_spmv_csr_rb<int64_t, RB_CAP, RB_NG>), dim3(grid), dim3(64), 0, ctx->stream, A->n, A->nnz, (const int64_t*)A->rowptr,
                         A->colidx, vals, x, y, alpha, beta, base, A->rb_ntiles, A->rb_rows, dotw, partials, done_flag, (g_rb_xcd && (grid & 7) == 0) ? 1 : 0);
    else
      hipLaunchKernelGGL((k_spmv_csr_rb<int32_t, RB_CAP, RB_NG>), dim3(grid), dim3(64), 0, ctx->stream, A->n, A->nnz, (const int32_t*)A->rowptr,
                         A->colidx, vals, x, y, alpha, beta, base, A->rb_ntiles, A->rb_rows, dotw, partials, done_flag, (g_rb_xcd && (grid & 7) == 0) ? 1 : 0);
    MFEM_CHECK_LAUNCH();
    if (n_partials && partials) *n_partials = grid;
    return MFEM_OK;
  }
  if (variant == 3) variant = 1;  // (split SpMV, unaligned arrays, no row blocks planned: the product tile)
  if (A->rows_per_block > 0) {
    const bool vec = ((((uintptr_t)vals) & 15) == 0) && ((((uintptr_t)A->colidx) & 7) == 0);
    if (vec && variant >= 5 && A->max_row_nnz <= 2048 - 2 && (A->ncols > 0 ? A->ncols : A->n) < ((int64_t)1 << 29)) {
      // wave-private tiles: R = 64 / tpr rows per wave, tpr the smallest power of two with R * max_row_nnz <= capacity - 2.
      // 1792 entries per wave (21.5 KB of LDS, 7 waves per CU) unless 2048 (24.6 KB, 6 waves) lets a wave own twice the rows.
      auto tpr_for = [&](int capw) {
        int tl = 0;
        while (tl < 6 && (int64_t)(64 >> tl) * A->max_row_nnz > capw - 2) ++tl;
        return tl;
      };
      const bool big = tpr_for(2048) < tpr_for(1792);
      // 2688 entries (32.3 KB, 4 waves): rows of 64..83 entries -- three fields on a 27-point stencil -- fill 0.99 of it with 32 rows,
      // 0.72 of a 1792-entry block with 16
      const bool huge = !big && g_spmv_tile2688 && tpr_for(2688) < tpr_for(1792);
      const int tl = huge ? tpr_for(2688) : big ? tpr_for(2048) : tpr_for(1792);
      const int Rw = 64 >> tl;
      const int64_t ntw = (A->n + Rw - 1) / Rw;
      const int waves = variant == 6 ? 2 : 1;
      // persistent grid = what is resident at once (LDS-limited; other counts leave a ragged last round: 8 per CU measured
      // 1.43 ms against 1.06 ms with 7 or 14 at 256^3)
      const int resident = (huge ? 4 : big ? 6 : 7) / waves;  // the 2688-entry tile keeps 42 gathers + the next tile in registers: one wave per SIMD
      int capw = ctx->num_cus * (g_spmv_grid_mult_set ? g_spmv_grid_mult.load() : resident);
      if (capw > MFEM_MAX_PARTIALS) capw = MFEM_MAX_PARTIALS;
      if (part.part != 0 && capw > MFEM_MAX_PARTIALS / 2) capw = MFEM_MAX_PARTIALS / 2;
      if (part.part == 2) {
        int64_t rows = 0;
        for (int z = 0; z < part.nz; ++z) rows += part.hi[z] - part.lo[z];
        const int64_t want = rows / (Rw * waves) + 2 * part.nz + 8;
        if (want < capw) capw = (int)want;
      }
      const int gridw = (int)((ntw + waves - 1) / waves < capw ? (ntw + waves - 1) / waves : capw);
      // XCD strips (see the kernel): a one-field lattice pattern whose two planes of x outgrow an XCD's L2 (4 MB) -- 512^3, not 256^3
      int64_t strip_tp = 0;
      if (g_csr_w_strips && part.part == 0 && (gridw & 7) == 0 && A->lat_fields == 1 && A->lat_m1 > 0 && A->lat_m2 > 0 && A->ncols <= A->n) {
        const int64_t PL = (int64_t)A->lat_m1 * A->lat_m2;
        if (PL * 16 > g_csr_w_strip_min_bytes && A->n >= 4 * PL) strip_tp = (PL + Rw - 1) / Rw;
      }
#define LAUNCH_W(RP, CAPW, WV, NG)                                                                               \
  hipLaunchKernelGGL((k_spmv_csr_w<RP, CAPW, WV, NG>), dim3(gridw), dim3(64 * WV), 0, ctx->stream, A->n, A->nnz, \
                     (const RP*)A->rowptr, A->colidx, vals, x, y, alpha, beta, base, Rw, tl, ntw, dotw, partials, \
                     done_flag, part, (A->cw_elide && A->cw_R == Rw) ? A->cw_elide : nullptr, strip_tp)
#define LAUNCH_WV(RP)                                   \
  do {                                                  \
    if (huge && waves == 2) LAUNCH_W(RP, 2688, 2, 42);  \
    else if (huge) LAUNCH_W(RP, 2688, 1, 42);           \
    else if (big && waves == 2) LAUNCH_W(RP, 2048, 2, 32);   \
    else if (big) LAUNCH_W(RP, 2048, 1, 32);            \
    else if (waves == 2) LAUNCH_W(RP, 1792, 2, 28);     \
    else LAUNCH_W(RP, 1792, 1, 28);                     \
  } while (0)
      if (A->rowptr_bits == 64) LAUNCH_WV(int64_t); else LAUNCH_WV(int32_t);
#undef LAUNCH_WV
#undef LAUNCH_W
      MFEM_CHECK_LAUNCH();
      if (n_partials && partials) *n_partials = gridw;
      return MFEM_OK;
    }
    const bool transposing = vec && variant >= 4;
    int cap_doubles = 4032, blk = MFEM_BLOCK;
    if (vec) switch (variant) {
        default: break;
      }
    while (cap_doubles < 4032 && A->max_row_nnz > cap_doubles - 2) cap_doubles *= 2;  // the longest row must fit the tile
    if (cap_doubles == 4032 && blk < 128) blk = MFEM_BLOCK;
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
    int xcd = (ntiles >= 64) ? g_spmv_xcd_aware.load() : 0;
    const int xch = xcd & 0xFFFF;
    const int64_t span = (int64_t)8 * (xch > 0 ? xch : 1);
    const int64_t ntiles_padded = xch ? (ntiles + span - 1) / span * span : ntiles;
#define LAUNCH_LDS(RP, VEC, CAP, UNR, BLK)                                                                 \
  hipLaunchKernelGGL((k_spmv_lds<RP, VEC, CAP, UNR, BLK>), dim3(grid), dim3(BLK), 0, ctx->stream, A->n,    \
                     A->nnz, (const RP*)A->rowptr, A->colidx, vals, x, y, alpha, beta, base, R, tpr_log2,  \
                     ntiles, ntiles_padded, xcd, dotw, partials, done_flag, part)
#define LAUNCH_T(RP, CAP, BLK)                                                                              \
  hipLaunchKernelGGL((k_spmv_csr_t<RP, CAP, BLK, 7>), dim3(grid), dim3(BLK), 0, ctx->stream, A->n, A->nnz,  \
                     (const RP*)A->rowptr, A->colidx, vals, x, y, alpha, beta, base, R, tpr_log2, ntiles, dotw, \
                     partials, done_flag, part)
#define LAUNCH_VARIANT(RP)                                                          \
  do {                                                                              \
    if (!vec) LAUNCH_LDS(RP, false, 4032, 4, MFEM_BLOCK);                           \
    else if (transposing) LAUNCH_T(RP, 4032, MFEM_BLOCK);                           \
    else LAUNCH_LDS(RP, true, 4032, 8, MFEM_BLOCK);                                 \
  } while (0)
    if (A->rowptr_bits == 64) LAUNCH_VARIANT(int64_t); else LAUNCH_VARIANT(int32_t);
#undef LAUNCH_VARIANT
#undef LAUNCH_T
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
                             double alpha, double beta) try {
  MFEM_REQUIRE(ctx && A, "null handle");
  MFEM_REQUIRE(A->n == 0 || (x && y && (A->nnz == 0 || vals)), "null vector");
  return mfem_spmv_launch(ctx, A, vals, x, y, alpha, beta, nullptr, nullptr, nullptr, nullptr);
} MFEM_API_CATCH("mfem_spmv_csr")
