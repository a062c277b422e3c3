// Row-sorted sliced ELL (SELL-128-sigma with sigma = n) for the Krylov loop on matrices whose rows are NOT of near-uniform
// length -- hex-27 (27 / 45 / 75 / 125 entries per row depending on the node type) and every unstructured mesh.  Solver layout
// mode 3; the uniform case is spmv_ell.hip (modes 1 and 2), the caller-facing contract stays CSR.
//
//   * inspector (once per pattern): rows are stably sorted by decreasing length (hipCUB radix sort of (max_len - len, row));
//     rows of equal length keep their mesh order, so the x gathers of a 128-row block stay as local as in CSR order.  Block b
//     stores K_b = its longest (= first) row's length slots, slot-major: element (row r', slot s) at ptr[b] + s * 128 + (r' & 127).
//     Padding is what is left of length changes inside a block: 0.0-0.3 % for hex-27.
//     Rows of equal length are further grouped by the signature of their diagonal list (a hash of col - row over the row): a
//     block whose 128 rows share ONE list stores that list once (K_b relative offsets) and its column stream is never read --
//     col = row + off[s].  For hex-27 that is every block away from the mesh boundary (8 node types = 8 lists).
//   * per solve: the working values are copied into that layout (one pass, like the reference's K_total[K_val_ids] gather).
//   * SpMV: a wave owns a block, a lane two neighbouring sorted rows; value / column streams are unit-stride 16-byte / 8-byte
//     loads, the row sum runs in registers in slot (= column) order, and y is written through the row permutation.
#include <hipcub/hipcub.hpp>

#include "blas1.h"

#define SELL_B 128
typedef double s_d2 __attribute__((ext_vector_type(2)));
typedef int s_i2 __attribute__((ext_vector_type(2)));

extern std::atomic<int64_t> g_layout_min_rows_cols;  // spmv_ell.hip
static std::atomic<int> g_sell_enable{1};
static std::atomic<int> g_sell_offsets{1};  // blocks with one diagonal list skip their column stream
static std::atomic<int> g_sell_window_log2{0};
static std::atomic<int> g_sell_unroll{5};   // slots in flight per lane (bits 16-20 of mfem_debug_set_sell; 0 = default)
static std::atomic<int> g_sell_wg_per_cu{8};
static std::atomic<int> g_sell_region{0};   // edge of the lattice regions of the row sort (bits 4-7 of mfem_debug_set_sell x 8; 0 = global sort)
static std::atomic<int> g_sell_xcd{0};      // bit 2: every XCD walks a contiguous eighth of the block list
static std::atomic<int> g_sell_per_u{0};    // bits 21-22: node slots in flight in a field-periodic block of three fields (0: 3 -- the default --, 1: 2, 2: 4)
static std::atomic<int> g_sell_periodic{1}; // bit 3: 0 = field-periodic blocks read their whole column stream (round 6 A/B)
extern "C" int64_t mfem_debug_sell_periodic_blocks(mfem_csr A) { return A ? (int64_t)A->sell_periodic_blocks * (A->sell_fields > 0 ? 1 : 0) : -1; }
extern "C" int mfem_debug_set_sell(int enable) try {  // bit 0: layout on/off; bit 1: always read explicit columns
  ++mfem_debug_epoch;
  g_sell_enable = enable & 1;
  g_sell_offsets = (enable & 2) ? 0 : 1;
  g_sell_xcd = (enable & 4) ? 1 : 0;
  g_sell_periodic = (enable & 8) ? 0 : 1;
  g_sell_per_u = (enable >> 21) & 3;
  g_sell_region = ((enable >> 4) & 15) * 8;
  g_sell_window_log2 = (enable >> 8) & 63;
  g_sell_unroll = ((enable >> 16) & 31) ? ((enable >> 16) & 31) : 5;
  g_sell_wg_per_cu = ((enable >> 24) & 31) ? ((enable >> 24) & 31) : 8;  // rows are sorted within windows of 2^w consecutive rows (0 = over the whole matrix)
  return MFEM_OK;
} MFEM_API_CATCH("mfem_debug_set_sell")

struct SellRegions {  // lattice regions of the row sort (R = 0: none)
  int R;
  int64_t n_nodes, PL, m2, nri, nrj, nrk;
};

template <typename RP>
__global__ __launch_bounds__(MFEM_BLOCK) void k_sell_keys(int64_t n, const RP* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                            int base, int maxlen, int wshift, int lenbits,
                                                            uint64_t* __restrict__ keys, int32_t* __restrict__ ids,
                                                            int32_t* __restrict__ n_ghost_rows, SellRegions G) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n; r += stride) {
    const int64_t lo = (int64_t)rowptr[r] - base, hi = (int64_t)rowptr[r + 1] - base;
    uint32_t h = 2166136261u;  // FNV-1a over the diagonal list col - row
    bool ghost = false;        // the row reads a ghost column of a slab pattern (columns numbered behind the n owned ones)
    for (int64_t j = lo; j < hi; ++j) {
      const int64_t c = (int64_t)col[j] - base;
      ghost = ghost || c >= n;
      uint32_t d = (uint32_t)(c - r);
      for (int k = 0; k < 4; ++k) {
        h = (h ^ (d & 255u)) * 16777619u;
        d >>= 8;
      }
    }
    // ghost-reading rows sort behind all others (bit 63): the leading blocks can run while the halo exchange is in flight
    // window of the sort: 2^wshift consecutive rows, or -- lattice hint -- a cube of R^3 lattice points: the rows of all lengths (node
    // types) of one region are neighbours in the block list, so the x entries they share are fetched while they are still in the L2s
    uint64_t win = (uint64_t)(r >> wshift);
    if (G.R > 0) {
      const int64_t node = r % G.n_nodes, f = r / G.n_nodes;
      const int64_t pi = node / G.PL, rem = node - pi * G.PL;
      const int64_t pj = rem / G.m2, pk = rem - pj * G.m2;
      win = (uint64_t)(((f * G.nri + pi / G.R) * G.nrj + pj / G.R) * G.nrk + pk / G.R);
    }
    keys[r] = ((uint64_t)(ghost ? 1 : 0) << 63) | (win << (32 + lenbits)) |
              ((uint64_t)(uint32_t)(maxlen - (int32_t)(hi - lo)) << 32) | h;
    ids[r] = (int32_t)r;
    if (ghost) atomicAdd(n_ghost_rows, 1);
  }
}

// Do the diagonal-list signatures repeat?  On a lattice a row shares its list with its first or second neighbour (hex-8: every interior row; hex-27: the
// node types alternate with period 2); on an unstructured pattern practically never.  The signature is the LOW word of the sort key: where it does not
// repeat it must not take part in the sort -- rows of one length would be shuffled by a hash and the x gathers of a 128-row block, local in mesh order,
// would come from all over the vector (round 6: the hex-20 meshes of every shipped example ran this layout at 0.18 of HBM, 3 x slower than the CSR kernel).
__global__ __launch_bounds__(MFEM_BLOCK) void k_sell_sig_repeats(int64_t n, const uint64_t* __restrict__ keys, unsigned long long* __restrict__ count) {
  __shared__ double red_unused[1];
  (void)red_unused;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned long long c = 0;
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r + 2 < n; r += stride) {
    const uint32_t h = (uint32_t)keys[r];
    c += (h == (uint32_t)keys[r + 1] || h == (uint32_t)keys[r + 2]) ? 1 : 0;
  }
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);
}
__global__ __launch_bounds__(MFEM_BLOCK) void k_sell_clear_sig(int64_t n, uint64_t* __restrict__ keys) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n; r += stride) keys[r] &= 0xFFFFFFFF00000000ull;
}

// flags[b] = 1 and off[ptr[b] / 128 + s] = the common diagonal list when all 128 rows of block b have the list of its first row
template <typename RP>
__global__ __launch_bounds__(SELL_B) void k_sell_block_flags(int64_t n, int64_t nblk, const RP* __restrict__ rowptr,
                                                               const int32_t* __restrict__ col, int base,
                                                               const int32_t* __restrict__ rowid, const int64_t* __restrict__ ptr,
                                                               int32_t* __restrict__ flags, int32_t* __restrict__ off,
                                                               int32_t* __restrict__ nreg) {
  __shared__ int bad;
  for (int64_t b = blockIdx.x; b < nblk; b += gridDim.x) {
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    const int64_t rs = b * SELL_B + threadIdx.x;
    const int64_t r0 = rowid[b * SELL_B];
    const int64_t lo0 = (int64_t)rowptr[r0] - base;
    const int K = (int)((int64_t)rowptr[r0 + 1] - base - lo0);
    int fail = 0;
    if (rs < n) {
      const int64_t r = rowid[rs];
      const int64_t lo = (int64_t)rowptr[r] - base;
      if ((int)((int64_t)rowptr[r + 1] - base - lo) != K) fail = 1;
      for (int s = 0; s < K && !fail; ++s)
        if ((int64_t)col[lo + s] - r != (int64_t)col[lo0 + s] - r0) fail = 1;
    } else {
      fail = 1;
    }
    if (fail) bad = 1;
    __syncthreads();
    if (!bad) {
      const int64_t o0 = ptr[b] / SELL_B;
      for (int s = threadIdx.x; s < K; s += SELL_B) off[o0 + s] = (int32_t)((int64_t)col[lo0 + s] - base - r0);
    }
    if (threadIdx.x == 0) {
      flags[b] = bad ? 0 : 1;
      if (!bad) atomicAdd(nreg, 1);
    }
    __syncthreads();
  }
}

// K_b * 128 for every block, K_b = its longest row: the first row, except in the one block where the ghost-reading rows (sorted behind
// all others, again by decreasing length) begin
template <typename RP>
__global__ __launch_bounds__(MFEM_BLOCK) void k_sell_block_sizes(int64_t n, int64_t nblk, const RP* __restrict__ rowptr,
                                                                   const int32_t* __restrict__ rowid, int64_t* __restrict__ sizes) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; b < nblk; b += stride) {
    int64_t K = 0;
    for (int64_t rs = b * SELL_B; rs < (b + 1) * SELL_B && rs < n; ++rs) {
      const int64_t r = rowid[rs];
      const int64_t len = (int64_t)rowptr[r + 1] - (int64_t)rowptr[r];
      K = len > K ? len : K;
    }
    sizes[b] = K * SELL_B;
  }
}

// columns (0-based; padding = the row itself) or values (padding = 0) into the sliced layout.  Four lanes per sorted row, 16 rows per
// wave pass: a lane quad reads 32 contiguous bytes of its row per step, so a row's cache line is used up by four consecutive loads of the
// same wave (with a lane per row the 64 row cursors of a wave advance 8 bytes at a time over a 64 KB working set: 10.5 ms per bind of the
// hex-27 128^3 matrix against 4.9 ms here); a store covers full 128-byte lines of four slots.
template <typename RP, typename T, bool COLS>
__global__ __launch_bounds__(MFEM_BLOCK) void k_sell_fill(int64_t n, int64_t nblk, const RP* __restrict__ rowptr,
                                                            const int32_t* __restrict__ rowid, const int64_t* __restrict__ ptr,
                                                            const T* __restrict__ src, int base, T* __restrict__ out,
                                                            const int32_t* __restrict__ col, const double* __restrict__ dsc) {
  // values only -- dsc != nullptr: entry / dsc[its column] (right Jacobi scaling folded into the copy)
  const int lane = threadIdx.x & 63, g = lane & 3;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t t = wave; t < 8 * nblk; t += nwaves) {  // a wave fills 16 rows of a block
    const int64_t b = t >> 3;
    const int64_t rs = b * SELL_B + (t & 7) * 16 + (lane >> 2);  // sorted row
    const int64_t p0 = ptr[b];
    const int Kb = (int)((ptr[b + 1] - p0) / SELL_B);
    int64_t lo = 0;
    int len = 0;
    T pad = (T)0;
    if (rs < n) {
      const int64_t r = rowid[rs];
      lo = (int64_t)rowptr[r] - base;
      len = (int)((int64_t)rowptr[r + 1] - base - lo);
      if (COLS) pad = (T)r;
    }
    T* o = out + p0 + (rs & (SELL_B - 1));
    if (!COLS && dsc) {
      for (int s = g; s < Kb; s += 4) o[(int64_t)s * SELL_B] = s < len ? (T)((double)src[lo + s] / dsc[col[lo + s] - base]) : pad;
    } else {
      // four loads of a lane in flight before the first store (round 6: one at a time -- load, wait, store -- copied the 15 GB of the hex-20 elasticity
      // matrix at 0.9 TB/s: 33.6 ms per solve)
      int s = g;
      for (; s + 12 < Kb; s += 16) {
        T t4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int su = s + 4 * u;
          t4[u] = su < len ? (COLS ? (T)(src[lo + su] - base) : src[lo + su]) : pad;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) o[(int64_t)(s + 4 * u) * SELL_B] = t4[u];
      }
      for (; s < Kb; s += 4) o[(int64_t)s * SELL_B] = s < len ? (COLS ? (T)(src[lo + s] - base) : src[lo + s]) : pad;
    }
  }
}

__global__ __launch_bounds__(MFEM_BLOCK) void k_sell_clear_flag2(int64_t nblk, int32_t* __restrict__ flags) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; b < nblk; b += stride)
    if (flags[b] == 2) flags[b] = 0;
}
// flags[b] = 2 when block b (not regular) is FIELD-PERIODIC: all 128 rows have K = F * P entries and col[f * P + t] = col[t] + f * shift for every row.
// A field-major multi-field matrix on ANY mesh has that form in its full blocks (row (g, i) lists the nodes coupled to i once per column field): the SpMV
// then reads P column slots instead of K -- a third of the column stream for three fields, 22 % of the bytes of a hex-20 elasticity product.
template <typename RP>
__global__ __launch_bounds__(SELL_B) void k_sell_block_periodic(int64_t n, int64_t nblk, const RP* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                                  int base, const int32_t* __restrict__ rowid, const int64_t* __restrict__ ptr, int F,
                                                                  int64_t shift, int32_t* __restrict__ flags, int32_t* __restrict__ nper) {
  __shared__ int bad;
  for (int64_t b = blockIdx.x; b < nblk; b += gridDim.x) {
    if (flags[b] != 0) continue;  // (block-uniform)
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    const int K = (int)((ptr[b + 1] - ptr[b]) / SELL_B);
    const int64_t rs = b * SELL_B + threadIdx.x;
    int fail = (K % F != 0 || K == 0) ? 1 : 0;
    if (!fail) {
      if (rs < n) {
        const int64_t r = rowid[rs];
        const int64_t lo = (int64_t)rowptr[r] - base;
        if ((int)((int64_t)rowptr[r + 1] - base - lo) != K) fail = 1;
        const int P = K / F;
        for (int f = 1; f < F && !fail; ++f)
          for (int t = 0; t < P && !fail; ++t)
            if ((int64_t)col[lo + (int64_t)f * P + t] != (int64_t)col[lo + t] + f * shift) fail = 1;
      } else {
        fail = 1;
      }
    }
    if (fail) bad = 1;
    __syncthreads();
    if (threadIdx.x == 0 && !bad) {
      flags[b] = 2;
      atomicAdd(nper, 1);
    }
    __syncthreads();
  }
}

// one field-periodic block: U node slots at a time, their F x U values and x entries in flight (the column slot of a node is read once for its F fields)
template <int F, int U>
__device__ __forceinline__ void sell_periodic_block(int Kb, int64_t shift, const double* __restrict__ v, const int32_t* __restrict__ c,
                                                    const double* __restrict__ x, double& acc0, double& acc1) {
  const int P = Kb / F;
  int t = 0;
  for (; t + U <= P; t += U) {
    int32_t c0[U], c1[U];
    double v0[F][U], v1[F][U], x0[F][U], x1[F][U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      c0[u] = __builtin_nontemporal_load(c + (int64_t)(t + u) * SELL_B);
      c1[u] = __builtin_nontemporal_load(c + (int64_t)(t + u) * SELL_B + 64);
    }
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        v0[f][u] = __builtin_nontemporal_load(v + (int64_t)(f * P + t + u) * SELL_B);
        v1[f][u] = __builtin_nontemporal_load(v + (int64_t)(f * P + t + u) * SELL_B + 64);
      }
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        x0[f][u] = x[(int64_t)c0[u] + f * shift];
        x1[f][u] = x[(int64_t)c1[u] + f * shift];
      }
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        acc0 += v0[f][u] * x0[f][u];
        acc1 += v1[f][u] * x1[f][u];
      }
  }
  for (; t < P; ++t) {
    const int64_t ca = c[(int64_t)t * SELL_B], cb = c[(int64_t)t * SELL_B + 64];
#pragma unroll
    for (int f = 0; f < F; ++f) {
      acc0 += __builtin_nontemporal_load(v + (int64_t)(f * P + t) * SELL_B) * x[ca + f * shift];
      acc1 += __builtin_nontemporal_load(v + (int64_t)(f * P + t) * SELL_B + 64) * x[cb + f * shift];
    }
  }
}

// =====================================================================================================================================================
// Node-blocked sliced layout ("BSELL", round 6) for field-major multi-field matrices on unstructured meshes: unknown (f, i) = f * ncp + i, and the rows
// (0, i) .. (F - 1, i) of node i all list the nodes coupled to i, once per column field (what mfem_pattern_build makes for n_fields fields; checked entry
// by entry by k_bsell_check).  The row-sorted layout above gives every ROW a lane: per value it reads a third of a column index (field-periodic blocks)
// and gathers one x entry -- on hex-20 elasticity 96^3 the product moved 21.2 GB for 17.7 by design (x gathers that miss the L2s) at the HBM copy rate.
// Here a lane owns a NODE: per coupled node ONE column index, F gathers of x and F x F values for the node's F row sums -- a ninth of the column stream,
// a third of the gathers.  Nodes are stably sorted by their number of coupled nodes; a block is 64 nodes; slot t of a block holds, for each of its nodes,
// the F x F values towards the node's t-th coupled node as F * F unit-stride runs of 64 doubles.
// =====================================================================================================================================================
static std::atomic<int> g_bsell_fill_quads{0};  // mfem_debug_set("bsell", 3): the layout copy by lane quads per row (the first form) instead of the LDS transpose
static std::atomic<int> g_bsell_enable{1};  // bit 9 of mfem_debug_set_sell's word... (own key: mfem_debug_set("bsell", on))
static std::atomic<long long> g_bsell_spmv_count{0};
extern "C" int mfem_debug_set_bsell(int on) {
  ++mfem_debug_epoch;
  g_bsell_enable = (on & 1) ? 1 : 0;
  g_bsell_fill_quads = (on & 2) ? 1 : 0;
  return MFEM_OK;
}
extern "C" long long mfem_debug_bsell_spmv_count(void) { return g_bsell_spmv_count; }
extern "C" int mfem_debug_bsell_fields(mfem_csr A) { return A ? A->bsell_F : -1; }

// bad[0] != 0: some node's rows do not have the node-blocked form
template <typename RP>
__global__ __launch_bounds__(MFEM_BLOCK) void k_bsell_check(int64_t ncp, int F, const RP* __restrict__ rowptr, const int32_t* __restrict__ col, int base,
                                                              int32_t* __restrict__ bad) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < ncp; i += stride) {
    const int64_t lo0 = (int64_t)rowptr[i] - base;
    const int len = (int)((int64_t)rowptr[i + 1] - base - lo0);
    bool ok = len % F == 0;
    const int L = len / F;
    for (int t = 0; ok && t < L; ++t) {
      const int64_t c = (int64_t)col[lo0 + t] - base;
      ok = c >= 0 && c < ncp;
    }
    for (int f = 0; ok && f < F; ++f) {
      const int64_t lo = (int64_t)rowptr[(int64_t)f * ncp + i] - base;
      ok = (int)((int64_t)rowptr[(int64_t)f * ncp + i + 1] - base - lo) == len;
      for (int g = 0; ok && g < F; ++g)
        for (int t = 0; ok && t < L; ++t) ok = (int64_t)col[lo + (int64_t)g * L + t] == (int64_t)col[lo0 + t] + (int64_t)g * ncp;
    }
    if (!ok) *bad = 1;
  }
}
template <typename RP>
__global__ __launch_bounds__(MFEM_BLOCK) void k_bsell_keys(int64_t ncp, int F, int maxL, const RP* __restrict__ rowptr, uint32_t* __restrict__ keys,
                                                             int32_t* __restrict__ ids) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < ncp; i += stride) {
    keys[i] = (uint32_t)(maxL - (int)(((int64_t)rowptr[i + 1] - (int64_t)rowptr[i]) / F));
    ids[i] = (int32_t)i;
  }
}
// node slots x 64 of every block (its first node is its longest)
template <typename RP>
__global__ __launch_bounds__(MFEM_BLOCK) void k_bsell_block_sizes(int64_t nblk, int F, const RP* __restrict__ rowptr, const int32_t* __restrict__ nodeid,
                                                                    int64_t* __restrict__ sizes) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; b < nblk; b += stride) {
    const int64_t i = nodeid[b * 64];
    sizes[b] = ((int64_t)rowptr[i + 1] - (int64_t)rowptr[i]) / F * 64;
  }
}
// node-level columns, 0-based (padding: the node itself, with zero values)
template <typename RP>
__global__ __launch_bounds__(MFEM_BLOCK) void k_bsell_cols(int64_t ncp, int64_t nblk, int F, const RP* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                             int base, const int32_t* __restrict__ nodeid, const int64_t* __restrict__ ptr,
                                                             int32_t* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t b = wave; b < nblk; b += nwaves) {
    const int64_t p0 = ptr[b];
    const int Kb = (int)((ptr[b + 1] - p0) / 64);
    const int64_t ns = b * 64 + lane;
    int64_t lo = 0, i = 0;
    int L = 0;
    if (ns < ncp) {
      i = nodeid[ns];
      lo = (int64_t)rowptr[i] - base;
      L = (int)(((int64_t)rowptr[i + 1] - base - lo) / F);
    }
    for (int t = 0; t < Kb; ++t) out[p0 + (int64_t)t * 64 + lane] = t < L ? col[lo + t] - base : (int32_t)i;
  }
}
// values into the node-blocked layout (once per solve).  A lane quad per CSR row, 16 consecutive sorted nodes of one row field per wave pass: the quad
// reads 32 contiguous bytes of its row per step, the 16 rows' stores of one slot are 128 contiguous bytes.  dsc != nullptr: entry / dsc[its column].
template <typename RP, int F>
__global__ __launch_bounds__(MFEM_BLOCK) void k_bsell_fill(int64_t ncp, int64_t nblk, const RP* __restrict__ rowptr, const int32_t* __restrict__ nodeid,
                                                             const int64_t* __restrict__ ptr, const double* __restrict__ src, int base,
                                                             double* __restrict__ out, const int32_t* __restrict__ col, const double* __restrict__ dsc) {
  const int lane = threadIdx.x & 63, g4 = lane & 3, q = lane >> 2;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t p = wave; p < nblk * F * 4; p += nwaves) {
    const int64_t b = p / (F * 4);
    const int rem = (int)(p - b * (F * 4)), f = rem >> 2, sub = rem & 3;
    const int64_t p0 = ptr[b];
    const int Kb = (int)((ptr[b + 1] - p0) / 64);
    const int nl = sub * 16 + q;  // the node's lane in the product kernel
    const int64_t ns = b * 64 + nl;
    int64_t lo = 0;
    int L = 0;
    if (ns < ncp) {
      const int64_t i = nodeid[ns];
      lo = (int64_t)rowptr[(int64_t)f * ncp + i] - base;
      L = (int)(((int64_t)rowptr[(int64_t)f * ncp + i + 1] - base - lo) / F);
    }
    double* o = out + p0 * (F * F) + (int64_t)(f * F) * 64 + nl;
#pragma unroll
    for (int g = 0; g < F; ++g) {
      const double* sg = src + lo + (int64_t)g * L;
      const int32_t* cg = col + lo + (int64_t)g * L;
      double* og = o + (int64_t)g * 64;
      int t = g4;
      for (; t + 28 < Kb; t += 32) {  // eight loads of a lane in flight before the first store (four: 17.0 ms per bind of the hex-20 elasticity matrix at 96^3)
        double t8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int tu = t + 4 * u;
          t8[u] = tu < L ? (dsc ? sg[tu] / dsc[cg[tu] - base] : sg[tu]) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) og[(int64_t)(t + 4 * u) * (64 * F * F)] = t8[u];
      }
      for (; t + 12 < Kb; t += 16) {
        double t4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int tu = t + 4 * u;
          t4[u] = tu < L ? (dsc ? sg[tu] / dsc[cg[tu] - base] : sg[tu]) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) og[(int64_t)(t + 4 * u) * (64 * F * F)] = t4[u];
      }
      for (; t < Kb; t += 4) og[(int64_t)t * (64 * F * F)] = t < L ? (dsc ? sg[t] / dsc[cg[t] - base] : sg[t]) : 0.0;
    }
  }
}

// The same copy through an LDS transpose (round 6, for coupling lists of up to BSELL_T_MAXL nodes): a workgroup takes (block, row field f, column field g);
// its waves read the 64 nodes' g-segments of row (f, node) -- L contiguous values each, unit-stride lanes -- into LDS [node][t], then every slot t leaves
// as ONE 512-byte run of 64 lanes.  The quad-per-row form above reads 32-byte pieces and writes 128-byte pieces, 8 bytes per lane: 16.3 ms for the 15 GB
// of the hex-20 elasticity matrix at 96^3 (1.8 TB/s).
#define BSELL_T_MAXL 127
// (a first version with a workgroup per (block, f, g) ran at 15.3 ms: the copy is bound by the 1.9e9 gathers of dsc[column], not by its access pattern --
// the divisor depends on the COLUMN (g, coupled node) alone, so a workgroup now takes (block, g), gathers the divisors once into registers and walks the F
// row fields with them: a third of the gathers)
template <typename RP, int F>
__global__ __launch_bounds__(MFEM_BLOCK) void k_bsell_fill_t(int64_t ncp, int64_t nblk, const RP* __restrict__ rowptr, const int32_t* __restrict__ nodeid,
                                                               const int64_t* __restrict__ ptr, const double* __restrict__ src, int base,
                                                               double* __restrict__ out, const int32_t* __restrict__ col, const double* __restrict__ dsc,
                                                               int ldl) {
  extern __shared__ double tl[];  // [64][ldl] values, then [F][64] segment starts (int64), then [64] lengths (int)
  int64_t* s_lo = reinterpret_cast<int64_t*>(tl + (size_t)64 * ldl);
  int* s_L = reinterpret_cast<int*>(s_lo + F * 64);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int64_t job = blockIdx.x; job < nblk * F; job += gridDim.x) {
    const int64_t b = job / F;
    const int g = (int)(job - b * F);
    const int64_t p0 = ptr[b];
    const int Kb = (int)((ptr[b + 1] - p0) / 64);
    if (tid < 64) {
      const int64_t ns = b * 64 + tid;
      int L = 0;
      int64_t i = 0;
      if (ns < ncp) {
        i = nodeid[ns];
        L = (int)(((int64_t)rowptr[i + 1] - (int64_t)rowptr[i]) / F);
      }
#pragma unroll
      for (int f = 0; f < F; ++f) s_lo[f * 64 + tid] = ns < ncp ? (int64_t)rowptr[(int64_t)f * ncp + i] - base + (int64_t)g * L : 0;
      s_L[tid] = L;
    }
    __syncthreads();
    // the divisors of this wave's 16 nodes x 2 entries per lane (columns: from the node's first row -- every row field lists the same)
    double dv[4][4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int nl = w * 4 + 16 * q + u;
        const int64_t lo = s_lo[nl];
        const int L = s_L[nl];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int t = lane + 64 * h;
          dv[q][u][h] = (dsc && t < L) ? dsc[col[lo + t] - base] : 1.0;
        }
      }
    for (int f = 0; f < F; ++f) {
      // phase 1: a wave per node, four nodes' loads in flight
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        double v[4][2];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int nl = w * 4 + 16 * q + u;
          const int64_t lo = s_lo[f * 64 + nl];
          const int L = s_L[nl];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int t = lane + 64 * h;
            v[u][h] = t < L ? src[lo + t] : 0.0;
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int t = lane + 64 * h;
            if (t < Kb) tl[(size_t)(w * 4 + 16 * q + u) * ldl + t] = dsc ? v[u][h] / dv[q][u][h] : v[u][h];  // (zeros behind a node's own list: the block's padding)
          }
      }
      __syncthreads();
      // phase 2: a slot per wave trip, lane = node
      double* o = out + p0 * (F * F) + (int64_t)(f * F + g) * 64 + lane;
      for (int t = w; t < Kb; t += 4) o[(int64_t)t * (64 * F * F)] = tl[(size_t)lane * ldl + t];
      __syncthreads();
    }
  }
}

// y = alpha A x + beta y: a wave per block of 64 nodes, a lane per node, U node slots (their F x F values, column and F x entries) in flight
template <int F, int U>
__global__ __launch_bounds__(MFEM_BLOCK) void k_spmv_bsell(int64_t ncp, int64_t nblk, const int64_t* __restrict__ ptr, const int32_t* __restrict__ nodeid,
                                                             const int32_t* __restrict__ cols, const double* __restrict__ vals, const double* __restrict__ x,
                                                             double* __restrict__ y, double alpha, double beta, const double* __restrict__ dotw,
                                                             double* __restrict__ partials, const int32_t* __restrict__ done_flag) {
  __shared__ double red[4];
  if (done_flag && done_flag[0]) return;
  double dot_acc = 0.0;
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  constexpr int64_t SS = 64 * F * F;  // doubles per node slot of a block
  for (int64_t b = wave; b < nblk; b += nwaves) {
    const int64_t p0 = ptr[b];
    const int Kb = (int)((ptr[b + 1] - p0) / 64);
    const double* v = vals + p0 * (F * F) + lane;
    const int32_t* c = cols + p0 + lane;
    const int64_t ns = b * 64 + lane;
    const int64_t node = ns < ncp ? nodeid[ns] : 0;
    double acc[F];
#pragma unroll
    for (int f = 0; f < F; ++f) acc[f] = 0.0;
    int t = 0;
    for (; t + U <= Kb; t += U) {
      int32_t cu[U];
      double vv[U][F * F], xx[U][F];
#pragma unroll
      for (int u = 0; u < U; ++u) cu[u] = __builtin_nontemporal_load(c + (int64_t)(t + u) * 64);
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int m = 0; m < F * F; ++m) vv[u][m] = __builtin_nontemporal_load(v + (int64_t)(t + u) * SS + m * 64);
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int g = 0; g < F; ++g) xx[u][g] = x[(int64_t)cu[u] + (int64_t)g * ncp];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int f = 0; f < F; ++f)
#pragma unroll
          for (int g = 0; g < F; ++g) acc[f] += vv[u][f * F + g] * xx[u][g];
    }
    for (; t < Kb; ++t) {
      const int64_t cc = c[(int64_t)t * 64];
#pragma unroll
      for (int g = 0; g < F; ++g) {
        const double xg = x[cc + (int64_t)g * ncp];
#pragma unroll
        for (int f = 0; f < F; ++f) acc[f] += __builtin_nontemporal_load(v + (int64_t)t * SS + (f * F + g) * 64) * xg;
      }
    }
    if (ns < ncp) {
#pragma unroll
      for (int f = 0; f < F; ++f) {
        const int64_t r = (int64_t)f * ncp + node;
        double yv = alpha * acc[f];
        if (beta != 0.0) yv += beta * y[r];
        y[r] = yv;
        if (dotw) dot_acc += yv * dotw[r];
      }
    }
  }
  if (partials) {
    const double bsum = block_reduce_sum(dot_acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = bsum;
  }
}

template <int SELL_U>
__global__ __launch_bounds__(MFEM_BLOCK) void k_spmv_sell(int64_t n, int64_t nblk, const int64_t* __restrict__ ptr,
                                                            const int32_t* __restrict__ rowid, const int32_t* __restrict__ flags,
                                                            const int32_t* __restrict__ off, const int32_t* __restrict__ cols,
                                                            const double* __restrict__ vals, const double* __restrict__ x,
                                                            double* __restrict__ y, double alpha, double beta,
                                                            const double* __restrict__ dotw, double* __restrict__ partials,
                                                            const int32_t* __restrict__ done_flag, int64_t b_lo, int64_t b_hi, int xcd, int pF,
                                                            int64_t pshift) {
  __shared__ double red[4];
  if (done_flag && done_flag[0]) return;
  double dot_acc = 0.0;
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  // xcd != 0: workgroups with equal blockIdx % 8 share an XCD (round-robin dispatch) and walk one contiguous eighth of the blocks
  int64_t b_first = b_lo + wave, b_stride = nwaves, b_end = b_hi;
  if (xcd) {
    const int64_t x = blockIdx.x & 7, wpw = blockDim.x >> 6, nb = b_hi - b_lo;
    b_first = b_lo + nb * x / 8 + ((int64_t)(blockIdx.x >> 3) * wpw + (threadIdx.x >> 6));
    b_end = b_lo + nb * (x + 1) / 8;
    b_stride = (int64_t)(gridDim.x >> 3) * wpw;
  }
  for (int64_t b = b_first; b < b_end; b += b_stride) {  // [b_lo, b_hi): all blocks, or one part of a split (multi-rank) SpMV
    const int64_t p0 = ptr[b];
    const int Kb = (int)((ptr[b + 1] - p0) / SELL_B);
    // lane l owns the block's rows l and l + 64: the value loads of a slot are two unit-stride 512-byte runs, and the x
    // loads of same-type rows (stride 2 along the fastest lattice direction for hex-27) touch half as many lines as with
    // two consecutive rows per lane
    const int64_t rs0 = b * SELL_B + lane, rs1 = rs0 + 64;
    const double* v = vals + p0 + lane;
    const int32_t* c = cols + p0 + lane;
    double acc0 = 0.0, acc1 = 0.0;
    const int bflag = flags ? __builtin_amdgcn_readfirstlane(flags[b]) : 0;
    const bool regular = bflag == 1;  // full block, one diagonal list
    const bool periodic = bflag == 2 && (pF & 15) > 1;  // full block of a field-major multi-field matrix: the node list repeats per column field
    int64_t rid0 = 0, rid1 = 0;
    if (rs0 < n) rid0 = rowid[rs0];
    if (rs1 < n) rid1 = rowid[rs1];
    // SELL_U slots in flight per lane (the kernel is latency-bound without: one slot at a time ran at 3.0 TB/s)
    if (regular) {
      const int32_t* ob = off + __builtin_amdgcn_readfirstlane((int)(p0 / SELL_B));
      int s = 0;
      for (; s + SELL_U <= Kb; s += SELL_U) {
        double v0[SELL_U], v1[SELL_U], x0[SELL_U], x1[SELL_U];
#pragma unroll
        for (int u = 0; u < SELL_U; ++u) {
          v0[u] = __builtin_nontemporal_load(v + (int64_t)(s + u) * SELL_B);
          v1[u] = __builtin_nontemporal_load(v + (int64_t)(s + u) * SELL_B + 64);
          const int64_t o = ob[s + u];
          x0[u] = x[rid0 + o];
          x1[u] = x[rid1 + o];
        }
#pragma unroll
        for (int u = 0; u < SELL_U; ++u) {
          acc0 += v0[u] * x0[u];
          acc1 += v1[u] * x1[u];
        }
      }
      for (; s < Kb; ++s) {
        const int64_t o = ob[s];
        acc0 += __builtin_nontemporal_load(v + (int64_t)s * SELL_B) * x[rid0 + o];
        acc1 += __builtin_nontemporal_load(v + (int64_t)s * SELL_B + 64) * x[rid1 + o];
      }
    } else if (periodic) {
      const int F_ = pF & 15, pu = pF >> 4;  // (fields; node slots in flight: 0 = the default)
      if (F_ == 3) {
        // (hex-20 elasticity 96^3, one box: 2 node slots in flight 3.69 ms, 3: 3.52, 4: 3.67; the whole column stream: 4.37)
        if (pu == 1) sell_periodic_block<3, 2>(Kb, pshift, v, c, x, acc0, acc1);
        else if (pu == 2) sell_periodic_block<3, 4>(Kb, pshift, v, c, x, acc0, acc1);
        else sell_periodic_block<3, 3>(Kb, pshift, v, c, x, acc0, acc1);
      } else if (F_ == 2) sell_periodic_block<2, 3>(Kb, pshift, v, c, x, acc0, acc1);
      else sell_periodic_block<4, 2>(Kb, pshift, v, c, x, acc0, acc1);
    } else {
      int s = 0;
      for (; s + SELL_U <= Kb; s += SELL_U) {
        double v0[SELL_U], v1[SELL_U], x0[SELL_U], x1[SELL_U];
#pragma unroll
        for (int u = 0; u < SELL_U; ++u) {
          v0[u] = __builtin_nontemporal_load(v + (int64_t)(s + u) * SELL_B);
          v1[u] = __builtin_nontemporal_load(v + (int64_t)(s + u) * SELL_B + 64);
          x0[u] = x[__builtin_nontemporal_load(c + (int64_t)(s + u) * SELL_B)];
          x1[u] = x[__builtin_nontemporal_load(c + (int64_t)(s + u) * SELL_B + 64)];
        }
#pragma unroll
        for (int u = 0; u < SELL_U; ++u) {
          acc0 += v0[u] * x0[u];
          acc1 += v1[u] * x1[u];
        }
      }
      for (; s < Kb; ++s) {
        acc0 += __builtin_nontemporal_load(v + (int64_t)s * SELL_B) * x[c[(int64_t)s * SELL_B]];
        acc1 += __builtin_nontemporal_load(v + (int64_t)s * SELL_B + 64) * x[c[(int64_t)s * SELL_B + 64]];
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if ((h ? rs1 : rs0) < n) {
        const int64_t r = h ? rid1 : rid0;
        double yv = alpha * (h ? acc1 : acc0);
        if (beta != 0.0) yv += beta * y[r];
        y[r] = yv;
        if (dotw) dot_acc += yv * dotw[r];
      }
    }
  }
  if (partials) {
    const double bsum = block_reduce_sum(dot_acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = bsum;
  }
}

#define SELL_CHECK(expr)                                                                    \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess) {                                                                 \
      mfem_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));  \
      rc = MFEM_ERR_HIP;                                                                    \
      goto done;                                                                            \
    }                                                                                       \
  } while (0)

// A->nb_F: the field count F (4, 3, 2 tried in that order: the largest that fits -- four fields also read as two super-fields of two) for which the
// pattern is node-blocked, 0 if none.  Once per pattern (the check reads every column index: 61 ms for the 1.9e9 entries of hex-20 elasticity at 96^3).
int mfem_node_block_fields(mfem_context_s* ctx, mfem_csr_s* A) {
  if (A->nb_F > 0 || A->nb_checked) return MFEM_OK;
  A->nb_checked = 1;
  A->nb_F = 0;
  if (A->ncols > A->n || A->n < 2 || A->max_row_nnz < 2) return MFEM_OK;
  const int cand[3] = {4, 3, 2};
  int32_t* d_bad = ctx->d_flags + 9;
  for (int ci = 0; ci < 3 && A->nb_F == 0; ++ci) {
    const int f = cand[ci];
    if (A->n % f != 0 || A->max_row_nnz % f != 0) continue;
    const int64_t ncp = A->n / f;
    MFEM_CHECK_HIP(hipMemsetAsync(d_bad, 0, sizeof(int32_t), ctx->stream));
    const int grid = mfem_grid_for(ncp, MFEM_BLOCK, ctx->num_cus * 16);
    if (A->rowptr_bits == 64)
      hipLaunchKernelGGL(k_bsell_check<int64_t>, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, ncp, f, (const int64_t*)A->rowptr, A->colidx, A->index_base, d_bad);
    else
      hipLaunchKernelGGL(k_bsell_check<int32_t>, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, ncp, f, (const int32_t*)A->rowptr, A->colidx, A->index_base, d_bad);
    MFEM_CHECK_LAUNCH();
    MFEM_CHECK_HIP(hipMemcpyAsync(ctx->h_flags + 9, d_bad, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    MFEM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->h_flags[9] == 0) A->nb_F = f;
  }
  return MFEM_OK;
}

// Plans the node-blocked layout if the pattern has its form (F = 4, 3, 2 tried in that order).  A->bsell_F > 0 afterwards: taken.
static int bsell_plan(mfem_context_s* ctx, mfem_csr_s* A) {
  A->bsell_F = 0;
  if (!g_bsell_enable || A->ncols > A->n || A->lat_fields > 0 || A->n < 64 * 4) return MFEM_OK;  // (slab patterns with ghost columns and lattice patterns keep the row-sorted form)
  int rc = mfem_node_block_fields(ctx, A);
  if (rc) return rc;
  const int F = A->nb_F;
  if (F == 0 || A->n / F < 64) return MFEM_OK;
  const int64_t ncp = A->n / F, nblk = (ncp + 63) / 64;
  const int maxL = A->max_row_nnz / F;
  uint32_t *keys = nullptr, *keys2 = nullptr;
  int32_t *ids = nullptr, *nodeid = nullptr, *cols = nullptr;
  int64_t *sizes = nullptr, *ptr = nullptr;
  void* tmp = nullptr;
  size_t tb = 0, tb2 = 0;
  int64_t slots = 0;
  const int grid = mfem_grid_for(ncp, MFEM_BLOCK, ctx->num_cus * 16);
  int bits = 1;
  while ((1 << bits) <= maxL && bits < 31) ++bits;
  SELL_CHECK(hipMalloc(&keys, sizeof(uint32_t) * (size_t)ncp));
  SELL_CHECK(hipMalloc(&keys2, sizeof(uint32_t) * (size_t)ncp));
  SELL_CHECK(hipMalloc(&ids, sizeof(int32_t) * (size_t)ncp));
  SELL_CHECK(hipMalloc(&nodeid, sizeof(int32_t) * (size_t)ncp));
  SELL_CHECK(hipMalloc(&sizes, sizeof(int64_t) * (size_t)(nblk + 1)));
  SELL_CHECK(hipMalloc(&ptr, sizeof(int64_t) * (size_t)(nblk + 1)));
  if (A->rowptr_bits == 64)
    hipLaunchKernelGGL(k_bsell_keys<int64_t>, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, ncp, F, maxL, (const int64_t*)A->rowptr, keys, ids);
  else
    hipLaunchKernelGGL(k_bsell_keys<int32_t>, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, ncp, F, maxL, (const int32_t*)A->rowptr, keys, ids);
  SELL_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, keys, keys2, ids, nodeid, (int)ncp, 0, bits, ctx->stream));
  SELL_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tb2, sizes, ptr, (int)(nblk + 1), ctx->stream));
  if (tb2 > tb) tb = tb2;
  SELL_CHECK(hipMalloc(&tmp, tb));
  SELL_CHECK(hipcub::DeviceRadixSort::SortPairs(tmp, tb, keys, keys2, ids, nodeid, (int)ncp, 0, bits, ctx->stream));  // stable: nodes of one length keep their mesh order
  SELL_CHECK(hipMemsetAsync(sizes, 0, sizeof(int64_t) * (size_t)(nblk + 1), ctx->stream));
  if (A->rowptr_bits == 64)
    hipLaunchKernelGGL(k_bsell_block_sizes<int64_t>, dim3(mfem_grid_for(nblk, MFEM_BLOCK, ctx->num_cus * 16)), dim3(MFEM_BLOCK), 0, ctx->stream, nblk, F,
                       (const int64_t*)A->rowptr, nodeid, sizes);
  else
    hipLaunchKernelGGL(k_bsell_block_sizes<int32_t>, dim3(mfem_grid_for(nblk, MFEM_BLOCK, ctx->num_cus * 16)), dim3(MFEM_BLOCK), 0, ctx->stream, nblk, F,
                       (const int32_t*)A->rowptr, nodeid, sizes);
  SELL_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp, tb, sizes, ptr, (int)(nblk + 1), ctx->stream));
  SELL_CHECK(hipMemcpyAsync(&slots, ptr + nblk, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
  SELL_CHECK(hipStreamSynchronize(ctx->stream));
  if ((double)slots * F * F <= 1.15 * (double)A->nnz + 128.0 * A->max_row_nnz * F) {
    SELL_CHECK(hipMalloc(&cols, sizeof(int32_t) * (size_t)(slots > 0 ? slots : 1)));
    const int g2 = mfem_grid_for(nblk * 64, MFEM_BLOCK, ctx->num_cus * 16);
    if (A->rowptr_bits == 64)
      hipLaunchKernelGGL(k_bsell_cols<int64_t>, dim3(g2), dim3(MFEM_BLOCK), 0, ctx->stream, ncp, nblk, F, (const int64_t*)A->rowptr, A->colidx, A->index_base,
                         nodeid, ptr, cols);
    else
      hipLaunchKernelGGL(k_bsell_cols<int32_t>, dim3(g2), dim3(MFEM_BLOCK), 0, ctx->stream, ncp, nblk, F, (const int32_t*)A->rowptr, A->colidx, A->index_base,
                         nodeid, ptr, cols);
    if (hipGetLastError() != hipSuccess) {
      mfem_set_error("k_bsell_cols launch failed");
      rc = MFEM_ERR_HIP;
      goto done;
    }
    A->sell_rowid = nodeid;
    A->sell_ptr = ptr;
    A->sell_cols = cols;
    A->sell_flags = nullptr;
    A->sell_off = nullptr;
    A->sell_total = slots * F * F;
    A->sell_nblk = nblk;
    A->sell_nb_int = nblk;
    A->sell_regular_blocks = 0;
    A->sell_fields = F;
    A->sell_shift = ncp;
    A->sell_periodic_blocks = (int32_t)nblk;
    A->sell_sig_sorted = 0;
    A->bsell_F = F;
    A->bsell_ncp = ncp;
    A->bsell_slots = slots;
    nodeid = nullptr;
    ptr = nullptr;
    cols = nullptr;
  }
done:
  if (keys) hipFree(keys);
  if (keys2) hipFree(keys2);
  if (ids) hipFree(ids);
  if (nodeid) hipFree(nodeid);
  if (sizes) hipFree(sizes);
  if (ptr) hipFree(ptr);
  if (cols) hipFree(cols);
  if (tmp) hipFree(tmp);
  return rc;
}


// sell_state: 0 not planned, -1 not eligible, 1 ready
int mfem_sell_plan(mfem_context_s* ctx, mfem_csr_s* A) {
  if (A->sell_state != 0) return MFEM_OK;
  if (A->n < g_layout_min_rows_cols) return MFEM_OK;  // launch-bound sizes stay on the CSR tile kernel: do not even sort
  A->sell_state = -1;
  if (A->n < SELL_B || A->nnz < 1 || A->max_row_nnz < 1 || A->n >= ((int64_t)1 << 31)) return MFEM_OK;
  int rc = bsell_plan(ctx, A);  // the node-blocked form first: a multi-field matrix on an unstructured mesh
  if (rc) return rc;
  if (A->bsell_F > 0) {
    A->sell_state = 1;
    return MFEM_OK;
  }
  const int64_t n = A->n, nblk = (n + SELL_B - 1) / SELL_B;
  uint64_t *keys = nullptr, *keys2 = nullptr;
  int32_t *ids = nullptr, *rowid = nullptr;
  int64_t *sizes = nullptr, *ptr = nullptr;
  void* tmp = nullptr;
  size_t tb = 0, tb2 = 0;
  int64_t total = 0;
  int32_t n_ghost = 0;
  int32_t* d_ghost = ctx->d_flags + 9;
  const bool has_ghosts = A->ncols > A->n;
  SellRegions G{};
  uint64_t nwin = 1;
  const int grid = mfem_grid_for(n, MFEM_BLOCK, ctx->num_cus * 16);
  int lenbits = 1;
  while ((1 << lenbits) <= A->max_row_nnz && lenbits < 31) ++lenbits;
  const int wshift = g_sell_window_log2 > 0 ? g_sell_window_log2.load() : 63;
  SELL_CHECK(hipMalloc(&keys, sizeof(uint64_t) * (size_t)n));
  SELL_CHECK(hipMalloc(&ids, sizeof(int32_t) * (size_t)n));
  SELL_CHECK(hipMalloc(&keys2, sizeof(uint64_t) * (size_t)n));
  SELL_CHECK(hipMalloc(&rowid, sizeof(int32_t) * (size_t)n));
  SELL_CHECK(hipMalloc(&sizes, sizeof(int64_t) * (size_t)(nblk + 1)));
  SELL_CHECK(hipMalloc(&ptr, sizeof(int64_t) * (size_t)(nblk + 1)));
  SELL_CHECK(hipMemsetAsync(d_ghost, 0, sizeof(int32_t), ctx->stream));
  if (g_sell_region > 0 && g_sell_window_log2 == 0 && A->lat_m1 > 0 && A->lat_m2 > 0 && A->lat_fields > 0 && n % A->lat_fields == 0 &&
      (n / A->lat_fields) % ((int64_t)A->lat_m1 * A->lat_m2) == 0) {
    G.R = g_sell_region;
    G.n_nodes = n / A->lat_fields;
    G.PL = (int64_t)A->lat_m1 * A->lat_m2;
    G.m2 = A->lat_m2;
    G.nri = (G.n_nodes / G.PL + G.R - 1) / G.R;
    G.nrj = (A->lat_m1 + G.R - 1) / G.R;
    G.nrk = (A->lat_m2 + G.R - 1) / G.R;
    nwin = (uint64_t)A->lat_fields * G.nri * G.nrj * G.nrk;
  }
  if (A->rowptr_bits == 64)
    hipLaunchKernelGGL(k_sell_keys<int64_t>, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, n, (const int64_t*)A->rowptr, A->colidx,
                       A->index_base, A->max_row_nnz, wshift, lenbits, keys, ids, d_ghost, G);
  else
    hipLaunchKernelGGL(k_sell_keys<int32_t>, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, n, (const int32_t*)A->rowptr, A->colidx,
                       A->index_base, A->max_row_nnz, wshift, lenbits, keys, ids, d_ghost, G);
  SELL_CHECK(hipMemcpyAsync(ctx->h_flags + 9, d_ghost, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  {
    // signatures that do not repeat (unstructured patterns) stay out of the sort: rows of one length keep their mesh order
    unsigned long long* d_rep = (unsigned long long*)(void*)sizes;  // (zeroed below before its own use)
    unsigned long long h_rep = 0;
    SELL_CHECK(hipMemsetAsync(d_rep, 0, sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL(k_sell_sig_repeats, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, n, keys, d_rep);
    SELL_CHECK(hipMemcpyAsync(&h_rep, d_rep, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    SELL_CHECK(hipStreamSynchronize(ctx->stream));
    if ((int64_t)h_rep < n / 8) {
      hipLaunchKernelGGL(k_sell_clear_sig, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, n, keys);
      A->sell_sig_sorted = 0;
    } else {
      A->sell_sig_sorted = 1;
    }
  }
  {
    int bits = 32 + lenbits;  // the low word is the signature of the diagonal list
    if (G.R > 0) {
      while (bits < 63 && ((nwin - 1) >> (bits - 32 - lenbits))) ++bits;                     // region index on top
    } else if (wshift < 63)
      while (bits < 64 && ((uint64_t)(n - 1) >> wshift) >> (bits - 32 - lenbits)) ++bits;  // window index on top
    if (has_ghosts) bits = 64;  // ... and the ghost-reading rows behind everything else
    SELL_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, keys, keys2, ids, rowid, (int)n, 0, bits, ctx->stream));
    SELL_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tb2, sizes, ptr, (int)(nblk + 1), ctx->stream));
    if (tb2 > tb) tb = tb2;
    SELL_CHECK(hipMalloc(&tmp, tb));
    SELL_CHECK(hipcub::DeviceRadixSort::SortPairs(tmp, tb, keys, keys2, ids, rowid, (int)n, 0, bits, ctx->stream));  // stable
  }
  SELL_CHECK(hipMemsetAsync(sizes, 0, sizeof(int64_t) * (size_t)(nblk + 1), ctx->stream));
  if (A->rowptr_bits == 64)
    hipLaunchKernelGGL(k_sell_block_sizes<int64_t>, dim3(mfem_grid_for(nblk, MFEM_BLOCK, ctx->num_cus * 16)), dim3(MFEM_BLOCK), 0,
                       ctx->stream, n, nblk, (const int64_t*)A->rowptr, rowid, sizes);
  else
    hipLaunchKernelGGL(k_sell_block_sizes<int32_t>, dim3(mfem_grid_for(nblk, MFEM_BLOCK, ctx->num_cus * 16)), dim3(MFEM_BLOCK), 0,
                       ctx->stream, n, nblk, (const int32_t*)A->rowptr, rowid, sizes);
  SELL_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp, tb, sizes, ptr, (int)(nblk + 1), ctx->stream));
  SELL_CHECK(hipMemcpyAsync(&total, ptr + nblk, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
  SELL_CHECK(hipStreamSynchronize(ctx->stream));
  n_ghost = ctx->h_flags[9];
  // (padding: up to one block of the longest rows at the tail; with ghost-reading rows sorted last, one more where they begin)
  // (... and with lattice regions every region pads each of its row lengths to whole blocks: the regions are sized so that this stays small)
  if ((double)total <= (G.R > 0 ? 1.25 : 1.15) * (double)A->nnz + (has_ghosts ? 256.0 : 128.0) * A->max_row_nnz) {
    SELL_CHECK(hipMalloc(&A->sell_cols, sizeof(int32_t) * (size_t)total));
    const int g2 = mfem_grid_for(8 * nblk * 64, MFEM_BLOCK, ctx->num_cus * 16);
    if (A->rowptr_bits == 64)
      hipLaunchKernelGGL((k_sell_fill<int64_t, int32_t, true>), dim3(g2), dim3(MFEM_BLOCK), 0, ctx->stream, n, nblk,
                         (const int64_t*)A->rowptr, rowid, ptr, A->colidx, A->index_base, A->sell_cols, (const int32_t*)nullptr, (const double*)nullptr);
    else
      hipLaunchKernelGGL((k_sell_fill<int32_t, int32_t, true>), dim3(g2), dim3(MFEM_BLOCK), 0, ctx->stream, n, nblk,
                         (const int32_t*)A->rowptr, rowid, ptr, A->colidx, A->index_base, A->sell_cols, (const int32_t*)nullptr, (const double*)nullptr);
    if (hipGetLastError() != hipSuccess) {
      mfem_set_error("k_sell_fill launch failed");
      rc = MFEM_ERR_HIP;
      goto done;
    }
    // blocks with a single diagonal list
    {
      int32_t* d_cnt = ctx->d_flags + 9;
      SELL_CHECK(hipMalloc(&A->sell_flags, sizeof(int32_t) * (size_t)nblk));
      SELL_CHECK(hipMalloc(&A->sell_off, sizeof(int32_t) * (size_t)(total / SELL_B + 1)));
      SELL_CHECK(hipMemsetAsync(d_cnt, 0, sizeof(int32_t), ctx->stream));
      const int g3 = (int)(nblk < (int64_t)ctx->num_cus * 64 ? nblk : (int64_t)ctx->num_cus * 64);
      if (A->rowptr_bits == 64)
        hipLaunchKernelGGL(k_sell_block_flags<int64_t>, dim3(g3), dim3(SELL_B), 0, ctx->stream, n, nblk, (const int64_t*)A->rowptr,
                           A->colidx, A->index_base, rowid, ptr, A->sell_flags, A->sell_off, d_cnt);
      else
        hipLaunchKernelGGL(k_sell_block_flags<int32_t>, dim3(g3), dim3(SELL_B), 0, ctx->stream, n, nblk, (const int32_t*)A->rowptr,
                           A->colidx, A->index_base, rowid, ptr, A->sell_flags, A->sell_off, d_cnt);
      SELL_CHECK(hipMemcpyAsync(ctx->h_flags + 9, d_cnt, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
      SELL_CHECK(hipStreamSynchronize(ctx->stream));
      A->sell_regular_blocks = ctx->h_flags[9];
      // field-periodic blocks among the others (k_sell_block_periodic): F = 3, 2, 4 fields of n / F rows each, the first F that covers a quarter of the blocks
      A->sell_fields = 0;
      A->sell_shift = 0;
      A->sell_periodic_blocks = 0;
      if (g_sell_periodic && !has_ghosts && A->sell_regular_blocks < nblk / 2) {
        const int cand[3] = {3, 2, 4};
        for (int ci = 0; ci < 3 && A->sell_fields == 0; ++ci) {
          const int F = cand[ci];
          if (n % F != 0 || A->max_row_nnz % F != 0) continue;
          SELL_CHECK(hipMemsetAsync(d_cnt, 0, sizeof(int32_t), ctx->stream));
          if (A->rowptr_bits == 64)
            hipLaunchKernelGGL(k_sell_block_periodic<int64_t>, dim3(g3), dim3(SELL_B), 0, ctx->stream, n, nblk, (const int64_t*)A->rowptr, A->colidx,
                               A->index_base, rowid, ptr, F, n / F, A->sell_flags, d_cnt);
          else
            hipLaunchKernelGGL(k_sell_block_periodic<int32_t>, dim3(g3), dim3(SELL_B), 0, ctx->stream, n, nblk, (const int32_t*)A->rowptr, A->colidx,
                               A->index_base, rowid, ptr, F, n / F, A->sell_flags, d_cnt);
          SELL_CHECK(hipMemcpyAsync(ctx->h_flags + 9, d_cnt, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
          SELL_CHECK(hipStreamSynchronize(ctx->stream));
          if (ctx->h_flags[9] >= nblk / 4) {
            A->sell_fields = F;
            A->sell_shift = n / F;
            A->sell_periodic_blocks = ctx->h_flags[9];
          } else if (ctx->h_flags[9] > 0) {  // (a few blocks happened to fit: not taken -- back to "generic")
            hipLaunchKernelGGL(k_sell_clear_flag2, dim3(g3), dim3(MFEM_BLOCK), 0, ctx->stream, nblk, A->sell_flags);
          }
        }
      }
    }
    A->sell_rowid = rowid;
    A->sell_ptr = ptr;
    A->sell_total = total;
    A->sell_nblk = nblk;
    // the rows that read ghost columns are the last n_ghost sorted rows: the blocks in front of the first of them form the interior part
    // of a split SpMV (the block that holds both kinds belongs to the boundary part)
    A->sell_nb_int = has_ghosts ? (n - (int64_t)n_ghost) / SELL_B : nblk;
    A->sell_state = 1;
    rowid = nullptr;
    ptr = nullptr;
  }
done:
  if (keys) hipFree(keys);
  if (ids) hipFree(ids);
  if (keys2) hipFree(keys2);
  if (rowid) hipFree(rowid);
  if (sizes) hipFree(sizes);
  if (ptr) hipFree(ptr);
  if (tmp) hipFree(tmp);
  return rc;
}

size_t mfem_sell_vals_bytes(const mfem_csr_s* A) {
  return (A->sell_state == 1 && g_sell_enable && A->n >= g_layout_min_rows_cols) ? sizeof(double) * (size_t)A->sell_total : 0;
}

int mfem_sell_bind(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, double* buf, const double* dsc) {
  A->sell_vals = nullptr;
  A->sell_src = nullptr;
  if (A->sell_state != 1 || !g_sell_enable || !buf) return MFEM_OK;
  if (A->bsell_F > 0) {
    const int maxL = A->max_row_nnz / A->bsell_F;
    if (maxL <= BSELL_T_MAXL && !g_bsell_fill_quads) {  // through an LDS transpose (k_bsell_fill_t)
      const int ldl = maxL | 1;  // (odd row stride: phase 2's lanes -- one node each -- spread over the banks)
      const size_t ldsb = sizeof(double) * (size_t)64 * ldl + (size_t)A->bsell_F * 64 * sizeof(int64_t) + 64 * sizeof(int);
      const int64_t jobs = A->sell_nblk * A->bsell_F;
      const int gt = (int)(jobs < (int64_t)ctx->num_cus * 12 ? jobs : (int64_t)ctx->num_cus * 12);
#define BSELL_FILL_T(RP, F)                                                                                                                     \
  hipLaunchKernelGGL((k_bsell_fill_t<RP, F>), dim3(gt), dim3(MFEM_BLOCK), ldsb, ctx->stream, A->bsell_ncp, A->sell_nblk, (const RP*)A->rowptr, \
                     A->sell_rowid, A->sell_ptr, vals, A->index_base, buf, A->colidx, dsc, ldl)
#define BSELL_FILL_TF(RP)                           \
  do {                                              \
    if (A->bsell_F == 3) BSELL_FILL_T(RP, 3);       \
    else if (A->bsell_F == 2) BSELL_FILL_T(RP, 2);  \
    else BSELL_FILL_T(RP, 4);                       \
  } while (0)
      if (A->rowptr_bits == 64) BSELL_FILL_TF(int64_t); else BSELL_FILL_TF(int32_t);
#undef BSELL_FILL_TF
#undef BSELL_FILL_T
      MFEM_CHECK_LAUNCH();
      A->sell_vals = buf;
      A->sell_src = vals;
      return MFEM_OK;
    }
    const int gb = mfem_grid_for(A->sell_nblk * A->bsell_F * 4 * 64, MFEM_BLOCK, ctx->num_cus * 16);
#define BSELL_FILL(RP, F)                                                                                                                      \
  hipLaunchKernelGGL((k_bsell_fill<RP, F>), dim3(gb), dim3(MFEM_BLOCK), 0, ctx->stream, A->bsell_ncp, A->sell_nblk, (const RP*)A->rowptr,    \
                     A->sell_rowid, A->sell_ptr, vals, A->index_base, buf, A->colidx, dsc)
#define BSELL_FILL_F(RP)                          \
  do {                                            \
    if (A->bsell_F == 3) BSELL_FILL(RP, 3);       \
    else if (A->bsell_F == 2) BSELL_FILL(RP, 2);  \
    else BSELL_FILL(RP, 4);                       \
  } while (0)
    if (A->rowptr_bits == 64) BSELL_FILL_F(int64_t); else BSELL_FILL_F(int32_t);
#undef BSELL_FILL_F
#undef BSELL_FILL
    MFEM_CHECK_LAUNCH();
    A->sell_vals = buf;
    A->sell_src = vals;
    return MFEM_OK;
  }
  const int g2 = mfem_grid_for(8 * A->sell_nblk * 64, MFEM_BLOCK, ctx->num_cus * 16);
  if (A->rowptr_bits == 64)
    hipLaunchKernelGGL((k_sell_fill<int64_t, double, false>), dim3(g2), dim3(MFEM_BLOCK), 0, ctx->stream, A->n, A->sell_nblk,
                       (const int64_t*)A->rowptr, A->sell_rowid, A->sell_ptr, vals, A->index_base, buf, A->colidx, dsc);
  else
    hipLaunchKernelGGL((k_sell_fill<int32_t, double, false>), dim3(g2), dim3(MFEM_BLOCK), 0, ctx->stream, A->n, A->sell_nblk,
                       (const int32_t*)A->rowptr, A->sell_rowid, A->sell_ptr, vals, A->index_base, buf, A->colidx, dsc);
  MFEM_CHECK_LAUNCH();
  A->sell_vals = buf;
  A->sell_src = vals;
  return MFEM_OK;
}

bool mfem_sell_bound(const mfem_csr_s* A, const double* vals) { return A->sell_vals && vals == A->sell_src; }

void mfem_sell_unbind(mfem_csr_s* A) {
  A->sell_vals = nullptr;
  A->sell_src = nullptr;
}

void mfem_sell_free(mfem_csr_s* A) {
  if (A->sell_cols) hipFree(A->sell_cols);
  if (A->sell_rowid) hipFree(A->sell_rowid);
  if (A->sell_ptr) hipFree(A->sell_ptr);
  if (A->sell_flags) hipFree(A->sell_flags);
  if (A->sell_off) hipFree(A->sell_off);
  A->sell_flags = nullptr;
  A->sell_off = nullptr;
  A->sell_cols = nullptr;
  A->sell_rowid = nullptr;
  A->sell_ptr = nullptr;
  A->sell_state = 0;
  A->bsell_F = 0;
}

// returns 1 if launched, 0 if another kernel should be used, <0 on error
int mfem_spmv_sell_launch(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, const double* x, double* y, double alpha,
                          double beta, const double* dotw, double* partials, int* n_partials, const int32_t* done_flag, int part) {
  if (!A->sell_vals || vals != A->sell_src) return 0;
  // part 1: the leading blocks, whose rows read no ghost column; part 2: the rest (mfem_spmv_halo)
  const int64_t b_lo = part == 2 ? A->sell_nb_int : 0, b_hi = part == 1 ? A->sell_nb_int : A->sell_nblk;
  if (n_partials) *n_partials = 0;
  if (A->bsell_F > 0) {  // node-blocked form (no ghost columns: never split)
    if (part == 2) return 1;
    int capb = ctx->num_cus * g_sell_wg_per_cu;
    if (capb > MFEM_MAX_PARTIALS) capb = MFEM_MAX_PARTIALS;
    const int gridb = mfem_grid_for(A->sell_nblk * 64, MFEM_BLOCK, capb);
#define BSELL_LAUNCH(F, U)                                                                                                                    \
  hipLaunchKernelGGL((k_spmv_bsell<F, U>), dim3(gridb), dim3(MFEM_BLOCK), 0, ctx->stream, A->bsell_ncp, A->sell_nblk, A->sell_ptr, A->sell_rowid, \
                     A->sell_cols, A->sell_vals, x, y, alpha, beta, dotw, partials, done_flag)
    const int bu = g_sell_per_u;  // (A/B: node slots in flight)
    if (A->bsell_F == 3) {  // (hex-20 elasticity 96^3, one box: 1 node slot in flight 2.98 ms, 2: 2.89, 3: 2.80)
      if (bu == 1) BSELL_LAUNCH(3, 2);
      else if (bu == 2) BSELL_LAUNCH(3, 4);
      else if (bu == 3) BSELL_LAUNCH(3, 1);
      else BSELL_LAUNCH(3, 3);
    } else if (A->bsell_F == 2) BSELL_LAUNCH(2, 4);
    else BSELL_LAUNCH(4, 2);
#undef BSELL_LAUNCH
    MFEM_CHECK_LAUNCH();
    ++g_bsell_spmv_count;
    if (n_partials && partials) *n_partials = gridb;
    return 1;
  }
  if (b_hi <= b_lo) return 1;
  int cap = ctx->num_cus * g_sell_wg_per_cu;
  if (cap > MFEM_MAX_PARTIALS) cap = MFEM_MAX_PARTIALS;
  if (part != 0) cap /= 2;  // the two parts of a split SpMV share one partial-sum array
  const int grid = mfem_grid_for((b_hi - b_lo) * 64, MFEM_BLOCK, cap);
#define SELL_LAUNCH(U)                                                                                                            \
  hipLaunchKernelGGL(k_spmv_sell<U>, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, A->n, A->sell_nblk, A->sell_ptr, A->sell_rowid, \
                     g_sell_offsets ? A->sell_flags : nullptr, A->sell_off, A->sell_cols, A->sell_vals, x, y, alpha, beta, dotw,   \
                     partials, done_flag, b_lo, b_hi, (g_sell_xcd && (grid & 7) == 0) ? 1 : 0, g_sell_periodic ? (A->sell_fields | (g_sell_per_u << 4)) : 0, A->sell_shift)
  switch (g_sell_unroll) {
    case 4: SELL_LAUNCH(4); break;
    case 8: SELL_LAUNCH(8); break;
    case 9: SELL_LAUNCH(9); break;
    case 10: SELL_LAUNCH(10); break;
    case 15: SELL_LAUNCH(15); break;
    default: SELL_LAUNCH(5); break;
  }
#undef SELL_LAUNCH
  MFEM_CHECK_LAUNCH();
  if (n_partials && partials) *n_partials = grid;
  return 1;
}
