// Inspector-executor SpMV for matrices whose rows have (nearly) the same length -- every matrix the structured assembly
// produces (27 entries per row for hex-8 thermal, 81 for 3-field elasticity, fewer only on the boundary).
//
// The caller's contract is unchanged: CSR pattern + values in CSR order (`mul!(b, A, x)`, 04_GPU_Utils.jl:131).  The
// reference's own linear solver starts every solve with a gather copy of the values (K_total[K_val_ids],
// 02_Preconditioner.jl:35); here that one pass per solve transposes the (already preconditioner-scaled) values into a
// slot-major padded layout  ell_vals[s][row]  next to a column table  ell_cols[s][row]  built once per pattern.
// In the Krylov loop lane <-> row, the slot loop runs in registers:
//   * the value / column streams are unit-stride across the lanes of a wave (512-byte and 256-byte runs), no LDS
//     staging, no workgroup barrier, no cross-lane reduction;
//   * the gather x[col] of one slot touches 64 CONSECUTIVE-ish entries (neighbouring rows have neighbouring columns):
//     4-5 cache lines per instruction instead of ~12 in CSR order (tools/gather_probe.hip, modes 0 vs 5);
//   * the row sum is accumulated in slot order = the plain sequential CSR row sum (deterministic).
// Padding entries carry value 0 and the row's own index as column.  Eligible when padding <= 10 % of nnz and
// max_row_nnz <= 128; rows of uneven length (hex-27: 27..125 entries) take the row-sorted sliced layout of spmv_sell.hip,
// small systems stay on the LDS-tile CSR kernel of spmv.hip (size thresholds below).
#include <vector>

#include "blas1.h"

// Blocked slot-major layout ("sliced ELL"): rows are grouped in blocks of ELL_B = 128 (the rows of one wave at two rows per
// lane); block b stores its K slots one after the other, element (row r, slot s) at  b * K * 128 + s * 128 + (r & 127).
// A wave therefore reads ONE contiguous K-kilobyte chunk per block and the kernel as a whole walks memory front to back
// like a copy, instead of K streams a full vector length apart.
#define ELL_B 128
__host__ __device__ __forceinline__ int64_t ell_base(int64_t r, int K) { return (r >> 7) * ((int64_t)K * ELL_B) + (r & (ELL_B - 1)); }

// Size thresholds: below them the Krylov loop is launch-bound and the CSR tile kernel, which spreads the nonzeros of few rows
// over many lanes, is as fast or faster than a lane-per-row layout whose slots are walked one dependent batch after the other
// (tools/probe_ell.py, tools/probe_small_solve.py: crossover ~3e5 rows with diagonal slots, ~1e6 rows with explicit columns).
std::atomic<int64_t> g_layout_min_rows_dia{262144};
std::atomic<int64_t> g_layout_min_rows_cols{1000000};
// hex-27 lattice tiles (spmv_lat27.hip): crossover against the CSR tile kernel + cycle graphs between 1.2e5 rows (41 / 45 us per CG iteration)
// and 2.7e5 (70 / 54 us); at 9.1e5 rows 183 / 110 us (tools/probe_lat_threshold.py, profiles/r03_lat_tiles_thresholds.txt)
#define LAT27_MIN_ROWS 180000
std::atomic<int64_t> g_layout_min_rows_lat27{LAT27_MIN_ROWS};
extern "C" int mfem_debug_set_layout_min_rows(int64_t diagonal_slots, int64_t explicit_columns) try {
  ++mfem_debug_epoch;
  g_layout_min_rows_dia = diagonal_slots;
  g_layout_min_rows_cols = explicit_columns;
  g_layout_min_rows_lat27 = explicit_columns < LAT27_MIN_ROWS ? explicit_columns : LAT27_MIN_ROWS;
  return MFEM_OK;
} MFEM_API_CATCH("mfem_debug_set_layout_min_rows")

static std::atomic<int> g_ell_enable{1};
static std::atomic<int> g_dia_enable{1};
// 0 (default): 2 rows x 3 diagonals, sharing the x loads of a run of three consecutive offsets when the diagonals come in such
// runs; 8: 2 rows x3 without that sharing; 1: 2 rows x2, 3: 2 rows x9, 4: 4 rows x1, 5: 4 rows x3, 6: 2 rows x1
static std::atomic<int> g_dia_variant{0};
static std::atomic<int> g_dia_block{MFEM_BLOCK};  // threads per workgroup of the default diagonal-slotted kernel (tuning: 256 / 512 / 1024)
static std::atomic<int> g_dia_sym{1};      // bit 22 of mfem_debug_set_ell turns the symmetric sweep kernels off
static std::atomic<int> g_symp_direct{1};  // bit 27: 0 = patch-major copy made from the slot-major copy in a second pass (k_symp_bind) instead of by k_dia_vals
static std::atomic<int> g_symp_tail{1};    // bit 26: 0 = the rows outside the swept planes in a launch of their own (as in a split SpMV)
static std::atomic<int> g_dia_symp{1};     // bit 23: the workgroup-tile sweep (k_spmv_sym27) instead of the wave-private patch sweep (k_spmv_symp)
static std::atomic<int> g_dia_pipe{1};     // bit 28: 0 = the layout copy (k_dia_vals) without its software pipeline
static std::atomic<int> g_dia_fast{1};     // bit 29: 0 = the layout copy without its fast path for full swept tiles (then pipelined as in bit 28)
static std::atomic<int> g_dia_xcd{0};      // 1: each XCD walks a contiguous eighth of the rows (needs a grid that is a multiple of 8)
// kernel variant (rows per lane x slots per batch, see the switch in mfem_spmv_ell_launch) and persistent workgroups per CU.
// Measured inside CG at 256^3 (profiles/r01_spmv_sweep.txt): 2 rows x 1 slot, 6 or 8 workgroups per CU is the fastest;
// workgroup counts that are not fully resident (10, 12 per CU) lose 15 %.
static std::atomic<int> g_ell_variant{6};
static std::atomic<int> g_ell_grid_mult{6};
static std::atomic<int> g_symp_fingerprint{1};
static std::atomic<long long> g_symp_fp_checks{0};  // binds whose symmetry verdict came from the fill's fingerprint (tests)
extern "C" long long mfem_debug_symp_fingerprint_count(void) { return g_symp_fp_checks; }
extern "C" int mfem_debug_set_ell(int enable) try {  // bit 0: enable; bits 4-7: kernel variant; bits 8-15: workgroups per CU
  ++mfem_debug_epoch;
  g_ell_enable = enable & 1;
  g_dia_enable = (enable & 2) ? 0 : 1;   // bit 1: keep explicit columns even when the matrix is diagonal-structured
  g_dia_variant = (enable >> 16) & 15;
  g_dia_xcd = (enable >> 20) & 1;
  g_dia_sym = ((enable >> 22) & 1) ? 0 : 1;
  g_dia_symp = ((enable >> 23) & 1) ? 0 : 1;
  g_symp_tail = ((enable >> 26) & 1) ? 0 : 1;
  g_symp_direct = ((enable >> 27) & 1) ? 0 : 1;
  g_dia_pipe = ((enable >> 28) & 1) ? 0 : 1;
  g_dia_fast = ((enable >> 29) & 1) ? 0 : 1;
  g_symp_fingerprint = ((enable >> 30) & 1) ? 0 : 1;  // bit 30: the symmetry of the swept rows by the separate check pass (k_spmv_symp<1>) instead of the fill's fingerprint
  g_dia_block = ((enable >> 24) & 3) == 1 ? 512 : ((enable >> 24) & 3) == 2 ? 1024 : ((enable >> 24) & 3) == 3 ? 128 : MFEM_BLOCK;  // bit 20: XCD-contiguous chunks
  g_ell_variant = (enable >> 4) & 15;
  if ((enable >> 8) & 255) g_ell_grid_mult = (enable >> 8) & 255;
  return MFEM_OK;
} MFEM_API_CATCH("mfem_debug_set_ell")

// cols[s][r] (0-based) for s < K; pad: the row index itself
template <typename RP>
__global__ __launch_bounds__(MFEM_BLOCK) void k_ell_cols(int64_t n, int64_t npad, int K, const RP* __restrict__ rowptr,
                                                           const int32_t* __restrict__ col, int base, int32_t* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < npad; r += stride) {
    int64_t lo = 0;
    int len = 0;
    if (r < n) {
      lo = (int64_t)rowptr[r] - base;
      len = (int)((int64_t)rowptr[r + 1] - base - lo);
    }
    const int32_t self = (int32_t)(r < n ? r : 0);
    for (int s = 0; s < K; ++s) out[ell_base(r, K) + s * ELL_B] = s < len ? col[lo + s] - base : self;
  }
}

// Transposition of the values through LDS: a wave owns 64 consecutive rows, whose CSR values are one contiguous run -> read with
// unit-stride lanes into the wave's LDS block, written out slot by slot with lane <-> row (both sides coalesced).
template <typename RP>
__global__ __launch_bounds__(MFEM_BLOCK) void k_ell_vals_lds(int64_t n, int64_t npad, int K, const RP* __restrict__ rowptr,
                                                               const double* __restrict__ vals, int base, double* __restrict__ out,
                                                               const int32_t* __restrict__ col, const double* __restrict__ dsc) {
  // dsc != nullptr: the copy is the right-Jacobi-scaled matrix, entry / dsc[its column] (Mat_Div_Jacobi folded into this pass)
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  double* T = lds + (size_t)w * 64 * K;
  const int64_t ntiles = npad >> 6;
  for (int64_t tile = (int64_t)blockIdx.x * nw + w; tile < ntiles; tile += (int64_t)gridDim.x * nw) {
    const int64_t r0 = tile << 6, r = r0 + lane;
    const int64_t rend = (r0 + 64 < n) ? r0 + 64 : n;
    int64_t lo = 0;
    int len = 0;
    if (r < n) {
      lo = (int64_t)rowptr[r] - base;
      len = (int)((int64_t)rowptr[r + 1] - base - lo);
    }
    const int64_t s0 = r0 < n ? (int64_t)rowptr[r0] - base : 0;
    const int cnt = r0 < n ? (int)((int64_t)rowptr[rend] - base - s0) : 0;  // <= 64 K
    constexpr int NB = 28;  // loads in flight per lane: a tile of 27-entry rows in ONE round trip (with 8 the kernel waited 79 % of its wave cycles)
    for (int i0 = lane; i0 < cnt; i0 += 64 * NB) {
      double tv[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int i = i0 + 64 * u;
        tv[u] = i < cnt ? vals[s0 + i] : 0.0;
        if (dsc && i < cnt) tv[u] /= dsc[col[s0 + i] - base];
      }
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int i = i0 + 64 * u;
        if (i < cnt) T[i] = tv[u];
      }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    const int off = (int)(lo - s0);
    for (int s = 0; s < K; ++s) out[ell_base(r, K) + s * ELL_B] = s < len ? T[off + s] : 0.0;
    __builtin_amdgcn_wave_barrier();
  }
}

typedef double e_d2 __attribute__((ext_vector_type(2)));
typedef int e_i2 __attribute__((ext_vector_type(2)));

// RPT rows per lane (1: 8-byte value / 4-byte column loads; 2: 16-byte / 8-byte loads of two neighbouring rows), U slots in
// flight per batch.  npad is a multiple of 64, so row pairs (even r) are 16-byte aligned in every slot plane.
template <int RPT, int U>
__global__ __launch_bounds__(MFEM_BLOCK) void k_spmv_ell(int64_t n, int64_t npad, int K, const int32_t* __restrict__ cols,
                                                           const double* __restrict__ vals, const double* __restrict__ x,
                                                           double* __restrict__ y, double alpha, double beta,
                                                           const double* __restrict__ dotw, double* __restrict__ partials,
                                                           const int32_t* __restrict__ done_flag, SpmvPart part) {
  __shared__ double red[4];
  if (done_flag && done_flag[0]) return;
  double dot_acc = 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * RPT;
  for (int64_t r = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) * RPT; r < n; r += stride) {
    if (spmv_part_skip(part, r, r + RPT)) continue;
    const double* v = vals + ell_base(r, K);
    const int32_t* c = cols + ell_base(r, K);
    if (RPT == 1) {
      double acc = 0.0;
      int s = 0;
      for (; s + U <= K; s += U) {
        double vv[U];
        int32_t cc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          vv[u] = __builtin_nontemporal_load(v + (s + u) * ELL_B);
          cc[u] = __builtin_nontemporal_load(c + (s + u) * ELL_B);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += vv[u] * x[cc[u]];
      }
      for (; s < K; ++s) acc += __builtin_nontemporal_load(v + s * ELL_B) * x[__builtin_nontemporal_load(c + s * ELL_B)];
      double yv = alpha * acc;
      if (beta != 0.0) yv += beta * y[r];
      y[r] = yv;
      if (dotw) dot_acc += yv * dotw[r];
    } else {
      // rows r, r + 1 (r even; row r + 1 may be the pad row n when n is odd: its slots are zero-valued and point at row 0)
      e_d2 acc = {0.0, 0.0};
      int s = 0;
      for (; s + U <= K; s += U) {
        e_d2 vv[U];
        e_i2 cc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          vv[u] = __builtin_nontemporal_load(reinterpret_cast<const e_d2*>(v + (s + u) * ELL_B));
          cc[u] = __builtin_nontemporal_load(reinterpret_cast<const e_i2*>(c + (s + u) * ELL_B));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          acc.x += vv[u].x * x[cc[u].x];
          acc.y += vv[u].y * x[cc[u].y];
        }
      }
      for (; s < K; ++s) {
        const e_d2 v1 = __builtin_nontemporal_load(reinterpret_cast<const e_d2*>(v + s * ELL_B));
        const e_i2 c1 = __builtin_nontemporal_load(reinterpret_cast<const e_i2*>(c + s * ELL_B));
        acc.x += v1.x * x[c1.x];
        acc.y += v1.y * x[c1.y];
      }
      double y0 = alpha * acc.x, y1 = alpha * acc.y;
      const bool two = r + 1 < n;
      if (beta != 0.0) {
        y0 += beta * y[r];
        if (two) y1 += beta * y[r + 1];
      }
      y[r] = y0;
      if (two) y[r + 1] = y1;
      if (dotw) {
        dot_acc += y0 * dotw[r];
        if (two) dot_acc += y1 * dotw[r + 1];
      }
    }
  }
  if (partials) {
    const double b = block_reduce_sum(dot_acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = b;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Diagonal-slotted blocks.  When every entry of the matrix sits on one of D <= 32 diagonals (col - row in a fixed sorted
// offset list: any lattice stencil -- 27 for the hex-8 scalar operator), slot s of a row is "the entry on diagonal s"
// (zero if the row has none) instead of "the s-th entry".  A 128-row block whose rows only touch in-range positions is
// then REGULAR: the column of (row r, slot s) is r + off[s], the column stream is not read at all, and the gather is a
// unit-stride 16-byte load.  Blocks that contain rows pointing outside [0, n_x) on some diagonal (first / last rows) or
// entries off the diagonal list (ghost columns of a slab) stay on the generic slot-major path with explicit columns.
// The detection is an inspection of the caller's CSR pattern; nothing about the mesh is assumed.
// ---------------------------------------------------------------------------------------------------------------
// Several diagonal lists ("classes") may coexist: a 3-field matrix in field-major numbering has one list per row field
// ((g - f) * n_nodes + stencil offset).  Each regular block belongs to one class.
#define DIA_MAXD 96
#define DIA_MAXC 4
struct DiaOffsets {
  int ncls;
  int D[DIA_MAXC];
  int32_t off[DIA_MAXC][DIA_MAXD];
};

// flags[b] = c + 1 when every row of the 128-row block b is regular for class c: all its entries sit on the class's
// diagonals and r + off[s] is a valid x index for EVERY listed diagonal (so the kernel may load x there even where the
// row has no entry); 0 otherwise.  nreg counts the regular blocks.
template <typename RP>
__global__ __launch_bounds__(128) void k_dia_flags(int64_t n, int64_t nx, const RP* __restrict__ rowptr,
                                                     const int32_t* __restrict__ col, int base,
                                                     const DiaOffsets* __restrict__ Op, int32_t* __restrict__ flags,
                                                     int32_t* __restrict__ nreg) {
  const DiaOffsets& O = *Op;
  __shared__ int ok_mask;
  for (int64_t blk = blockIdx.x; blk * 128 < n; blk += gridDim.x) {
    if (threadIdx.x == 0) ok_mask = (1 << O.ncls) - 1;
    __syncthreads();
    const int64_t r = blk * 128 + threadIdx.x;
    int mask = 0;
    if (r < n) {
      const int64_t lo = (int64_t)rowptr[r] - base, hi = (int64_t)rowptr[r + 1] - base;
      for (int c = 0; c < O.ncls; ++c) {
        const int D = O.D[c];
        bool ok = r + O.off[c][0] >= 0 && r + O.off[c][D - 1] < nx;
        int s = 0;
        int64_t last = INT64_MIN;
        for (int64_t j = lo; j < hi && ok; ++j) {  // columns strictly ascending, offsets ascending: merge
          const int64_t d = (int64_t)col[j] - base - r;
          if (d <= last) ok = false;  // unsorted or duplicate columns: only the explicit-column path sums every entry
          last = d;
          while (s < D && O.off[c][s] < d) ++s;
          if (s == D || O.off[c][s] != d) ok = false;
        }
        if (ok) mask |= 1 << c;
      }
    }  // rows past n (partial last block): mask 0 -> generic path
    atomicAnd(&ok_mask, mask);
    __syncthreads();
    if (threadIdx.x == 0) {
      const int m = ok_mask;
      flags[blk] = m ? __ffs(m) : 0;
      if (m) atomicAdd(nreg, 1);
    }
    __syncthreads();
  }
}

// geometry of the wave-private patch sweep (k_spmv_symp below)
#include "spmv_symp.h"
struct SympGeom {
  int64_t PL, nx;
  int m1, m2, p0, p1, NS, NPk;
  int nseg;  // runs per patch: a run = one patch swept through nplanes / nseg consecutive planes
};

// values in diagonal slots + per-block regular flag (regular: every row r of the block has 0 <= r + off[s] < nx for all s).
// With pv != nullptr the rows of the swept lattice planes [Gm.p0, Gm.p1) go straight to the patch-major copy of the patch sweep (layout:
// k_spmv_symp) -- their 27 slots, the edge block entries they own, and the diagonal alone to the slot-major copy (k_ell_diag reads it
// there) -- instead of through the slot-major copy and a second pass (k_symp_bind): 1.97 + 1.85 ms -> one pass at 256^3.
// the copies are written as full coalesced streams and not read again by this kernel: nontemporal stores (per-solve work of C2 3.65 -> 3.3 ms)
#if defined(DV_ABL) && DV_ABL == 1   // timing-only ablation builds (tools/ab_libs.sh; never in the product library): 1 = no stores (one per lane and tile
#define DIA_ST(p, v) do { if ((v) == 1.2345e300) __builtin_nontemporal_store((v), (p)); } while (0)  // keeps the loads alive), 3 = plain instead of nontemporal stores
#elif defined(DV_ABL) && DV_ABL == 3
#define DIA_ST(p, v) (*(p) = (v))
#else
#define DIA_ST(p, v) __builtin_nontemporal_store((v), (p))
#endif
// LPR = lanes per row: 1 (64 rows per wave tile) or 2 (32 rows); SYM: the symmetrically scaled copy -- its own instantiation, so that the plain
// copy's code is what it was (2.4 ms at 256^3; 2.7 with the test for the scaling in it)
template <typename RP, int LPR, bool SYM, bool PIPE>
__global__ __launch_bounds__(MFEM_BLOCK) void k_dia_vals(int64_t n, int64_t npad, int K, const RP* __restrict__ rowptr,
                                                           const int32_t* __restrict__ col, const double* __restrict__ vals,
                                                           int base, const DiaOffsets* __restrict__ Op,
                                                           const int32_t* __restrict__ flags, double* __restrict__ out, SympGeom Gm,
                                                           double* __restrict__ pv, const double* __restrict__ dsc,
                                                           const double* __restrict__ ssym, int fast) {
  // ssym != nullptr: the copy is S^-1 A S^-1 with ssym = 1 / S, entry * (ssym[row] * ssym[column]) with the PRODUCT of the two factors formed
  // first -- a mirrored pair is then multiplied by the same number, so a bitwise symmetric matrix stays bitwise symmetric (the scaled CG,
  // cg_variant 4).  (Multiplying by reciprocals, not dividing: 27 divisions per row cost more than the rest of the placement.)
  // dsc != nullptr: the copy is the right-Jacobi-scaled matrix, entry / dsc[its column] (Mat_Div_Jacobi folded into this pass; the
  // columns are then read for every tile)
  const DiaOffsets& O = *Op;
  extern __shared__ double lds[];
  const int64_t slo = pv ? (int64_t)Gm.p0 * Gm.PL : 0, shi = pv ? (int64_t)Gm.p1 * Gm.PL : 0;  // swept rows
  const int spNP = Gm.NS * Gm.NPk;
  const int64_t spT = (int64_t)spNP * (Gm.p1 - Gm.p0);
  // LPR = 2: a wave takes 32 rows at a time, two lanes per row -- lanes 0..31 walk their row's entries forward through the first half of
  // the diagonal list, lanes 32..63 walk them backward through the second half.  A lane per row (64 rows per tile) needs 12 x 64 x K
  // bytes of staging per wave: two waves per CU on 81-entry rows (4.0 ms per bind at C3 against 3.4 ms with two lanes); on 27-entry
  // rows six waves per CU are enough and the lane per row is faster (2.4 against 2.7 ms at 256^3).
  constexpr int RT = 64 / LPR, SH = LPR == 2 ? 5 : 6;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;  // (w: wave-uniform, known to the compiler)
  const int half = LPR == 2 ? lane >> 5 : 0, rl = lane & (RT - 1);
  double* T = lds + (size_t)w * RT * K;
  // columns are staged only for the Jacobi scaling pass (dsc); the placement below needs them for the few tiles that are not `full`
  // (mesh boundary) and reads those from memory -- 8 instead of 12 bytes of LDS per staged entry, half as many more waves per CU
  int32_t* Tc = reinterpret_cast<int32_t*>(lds + (size_t)nw * RT * K) + (size_t)w * RT * K;
  const bool stage_cols = !PIPE && dsc != nullptr;
  const int64_t ntiles = npad >> SH;
  constexpr int NB = 28;
  // Software pipeline (round 4; a lane per row, rows of at most NB entries, no scaling pass -- the 27-diagonal lattice copies of C2): the NEXT tile's
  // values are loaded into registers before the current tile is placed, so a wave keeps one tile of loads in flight while it reads LDS and issues
  // its 27 scattered stores -- each wave had one memory round trip per tile with nothing else outstanding (10 waves per CU).
  constexpr bool pipe = PIPE;  // (chosen at the launch: LPR == 1, no scaling pass, K <= NB)
  const int64_t tstride = (int64_t)gridDim.x * nw;
  double tvn[NB];
  int64_t s0n = 0;
  int cntn = 0;
  auto tile_span = [&](int64_t t, int64_t& s0_, int& cnt_) {
    const int64_t q0 = t << SH, qend = (q0 + RT < n) ? q0 + RT : n;
    s0_ = q0 < n ? (int64_t)rowptr[q0] - base : 0;
    cnt_ = q0 < n ? (int)((int64_t)rowptr[qend] - base - s0_) : 0;
  };
  auto prefetch = [&]() {
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int i = lane + 64 * u;
#if defined(DV_ABL) && DV_ABL == 2   // (ablation: no value loads)
      tvn[u] = (double)i;
#else
      tvn[u] = i < cntn ? __builtin_nontemporal_load(vals + s0n + i) : 0.0;
#endif
    }
  };
  if (pipe) {
    const int64_t t0 = (int64_t)blockIdx.x * nw + w;
    if (t0 < ntiles) {
      tile_span(t0, s0n, cntn);
      prefetch();
    }
  }
  // fast != 0: the swept rows are filled by k_symp_fill (below) -- this launch visits only the tiles that hold other rows (the tiles [sk0, sk1) lie inside the
  // swept range and are stepped over) and leaves the swept rows of the tiles it visits alone
  const int64_t sk0 = fast ? (slo + RT - 1) >> SH : 0, sk1 = fast ? (shi >> SH > sk0 ? shi >> SH : sk0) : 0;
  for (int64_t tix = (int64_t)blockIdx.x * nw + w; tix < ntiles - (sk1 - sk0); tix += tstride) {
    const int64_t tile = tix < sk0 ? tix : tix + (sk1 - sk0);
    const int64_t r0 = tile << SH, r = r0 + rl;
    const int64_t rend = (r0 + RT < n) ? r0 + RT : n;
    int64_t lo = 0;
    int len = 0;
    if (r < n) {
      lo = (int64_t)rowptr[r] - base;
      len = (int)((int64_t)rowptr[r + 1] - base - lo);
    }
    int64_t s0;
    int cnt;  // <= RT D
    int64_t s0_next = 0;
    int cnt_next = 0;
    if (pipe) {
      s0 = s0n;
      cnt = cntn;
      if (tile + tstride < ntiles) tile_span(tile + tstride, s0_next, cnt_next);  // (these row pointers arrive beside the values in flight)
    } else {
      s0 = r0 < n ? (int64_t)rowptr[r0] - base : 0;
      cnt = r0 < n ? (int)((int64_t)rowptr[rend] - base - s0) : 0;
    }
    const int cls = __builtin_amdgcn_readfirstlane(flags[tile >> (7 - SH)]) - 1;
    // a tile of a regular block whose rows all have every diagonal of the class (cnt = RT D: away from the mesh boundary, nearly all
    // tiles): entry s of a row IS its slot s -- the columns are not needed, a third of the kernel's reads
    const bool full = cls >= 0 && cnt == RT * O.D[cls];
    if (pipe) {
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int i = lane + 64 * u;
        if (i < cnt) T[i] = tvn[u];
      }
      s0n = s0_next;
      cntn = cnt_next;
      if (tile + tstride < ntiles) prefetch();  // in flight until the next trip's LDS stores
    }
    // staging: all loads of a lane are issued before the first LDS store.  28 in flight per lane: a 64-row tile of 27-entry rows (27 per lane) is
    // ONE memory round trip, a 32-row tile of 81-entry rows two (SQ counters of the version with batches of 8: 79 % of the wave cycles waiting,
    // ~10 waves per CU with 4 KB in flight each)
    if constexpr (!PIPE)
    for (int i0 = lane; i0 < cnt; i0 += 64 * NB) {
      double tv[NB];
      int32_t tc[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int i = i0 + 64 * u;
        tv[u] = i < cnt ? vals[s0 + i] : 0.0;
        tc[u] = (i < cnt && stage_cols) ? col[s0 + i] : 0;
      }
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int i = i0 + 64 * u;
        if (i < cnt) {
          T[i] = tv[u];
          if (stage_cols) Tc[i] = tc[u] - base;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    if (!PIPE && dsc) {
      // right Jacobi scaling of the staged tile, entry / dsc[column]: the gathers of 16 entries per lane are in flight together -- one more
      // memory round trip per batch of 1024 entries (dividing inside the staging loop above made every batch of its loads wait twice)
      __builtin_amdgcn_wave_barrier();
      for (int i0 = lane; i0 < cnt; i0 += 64 * 16) {
        double dd[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int i = i0 + 64 * u;
          dd[u] = i < cnt ? dsc[Tc[i]] : 1.0;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int i = i0 + 64 * u;
          if (i < cnt) T[i] /= dd[u];
        }
      }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_s_waitcnt(0xC07F);
    }
    const int off0 = (int)(lo - s0);
    const int dir = half ? -1 : 1;
    double sr = 1.0;
    if constexpr (SYM) sr = r < n ? ssym[r] : 1.0;
    auto sym_scaled = [&](double a, int64_t c) -> double {
      if constexpr (SYM) return a * (sr * ssym[c]);
      else return a;
    };
    auto colat = [&](int j) -> int64_t { return stage_cols ? (int64_t)Tc[off0 + j] : (int64_t)col[lo + j] - base; };
    if (cls >= 0 && r0 < shi && r0 + RT > slo) {  // a tile with swept rows (all of them in regular blocks of the 27-diagonal lattice class)
      const bool sw = r >= slo && r < shi;
      if (fast && sw) len = 0;  // (k_symp_fill's row: nothing is stored for it below)
      int line = 0, pcol = 0;
      int64_t mainoff = 0, lowoff = 0, edgeoff = 0;
      if (sw) {
        const int p = (int)(r / Gm.PL), rem = (int)(r - (int64_t)p * Gm.PL), jj = rem / Gm.m2, kk = rem - jj * Gm.m2;
        line = jj % SP_L;
        pcol = kk % SP_W;
        const int64_t step = (int64_t)(p - Gm.p0) * spNP + (jj / SP_L) * Gm.NPk + kk / SP_W;
        mainoff = step * SP_MAIN + line * SP_W + pcol;
        lowoff = spT * SP_MAIN + step * SP_LOW + line * SP_W + pcol;
        edgeoff = step * SP_MAIN + 14 * SP_ROWS;
      }
      int j = half ? len - 1 : 0;
#pragma unroll
      for (int t = 0; t < (LPR == 2 ? 14 : 27); ++t) {  // forward lanes: slots 0..13 (all 27 with a lane per row), backward lanes: slots 26..14
        const int sl = half ? 26 - t : t;
        const bool act = half == 0 || t < 13;
        double v = 0.0;
        if (act && j >= 0 && j < len && (full || colat(j) - r == O.off[cls][sl])) {
          v = sym_scaled(T[off0 + j], r + O.off[cls][sl]);
          j += dir;
        }
        if (!act) continue;
        if (!sw) {
          DIA_ST(out + ell_base(r, K) + sl * ELL_B, v);
        } else if (!fast) {
          DIA_ST(pv + (sl < 13 ? lowoff + sl * SP_ROWS : mainoff + (sl - 13) * SP_ROWS), v);
          if (sl == 13) out[ell_base(r, K) + 13 * ELL_B] = v;  // the diagonal (offset 0 is the 14th of the 27 lattice offsets)
          if (half == 0 && t < 13) {                           // the edge block entry the row owns for this lower slot, if any
            const int e = sp_edge_of(t, line, pcol);
            if (e >= 0) pv[edgeoff + e] = v;
          }
        }
      }
    } else if (cls >= 0) {  // regular 128-row block of class cls: slot s = diagonal s
      const int D = O.D[cls], Dh = LPR == 2 ? (D + 1) >> 1 : D;
      if (half == 0) {
        int j = 0;
        for (int sl = 0; sl < Dh; ++sl) {
          double v = 0.0;
          if (j < len && (full || colat(j) - r == O.off[cls][sl])) {
            v = sym_scaled(T[off0 + j], r + O.off[cls][sl]);
            ++j;
          }
          DIA_ST(out + ell_base(r, K) + sl * ELL_B, v);
        }
      } else {
        int j = len - 1;
        for (int sl = D - 1; sl >= Dh; --sl) {
          double v = 0.0;
          if (j >= 0 && (full || colat(j) - r == O.off[cls][sl])) {
            v = sym_scaled(T[off0 + j], r + O.off[cls][sl]);
            --j;
          }
          DIA_ST(out + ell_base(r, K) + sl * ELL_B, v);
        }
      }
    } else {                 // generic block: slot s = s-th entry, columns come from ell_cols
      for (int sl = half; sl < K; sl += LPR) out[ell_base(r, K) + sl * ELL_B] = sl < len ? sym_scaled(T[off0 + sl], colat(sl)) : 0.0;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// The swept rows of the patch-major copy: a workgroup of two waves per patch step (plane, strip of SP_L lines, patch column), a wave per pair of lattice lines,
// lane = line * SP_W + column -- 64 rows whose CSR values are two contiguous runs (one per line: 6.9 KB when all 32 rows have their 27 entries).  Every slot store of
// a wave is ONE aligned 512-byte piece of the copy and the two halves of each 1 KB slot are written by the same workgroup; the step's edge block (318 entries owned by
// rows of both waves) is gathered in LDS and written as one contiguous piece.  (k_dia_vals' tiles of 64 consecutive rows drift against the patch columns -- a
// 513-point line is 16 patches + 1 point -- and wrote two or three unaligned pieces per slot and the edge entries one by one; tools/copy_probe.hip, 512^3, all stores:
// 14.8 ms in that shape, 10.7 ms in this one; a plain aligned copy of the same bytes 9.4 ms.)  Two memory round trips per full tile -- the four row pointers of the
// runs (scalar loads), then the 27 values + the 27 factors of the symmetric scaling per lane, all issued before the first wait; tiles with short rows (lattice edge)
// or missing lines / columns take the general path: per-lane row pointers, masked staging, the short rows' columns decoded into slots (offset = di PL + dj m2 + dk,
// guaranteed by the class test in mfem_ell_plan) and every row expanded to its 27 slots in LDS.
template <typename RP, bool SYM>
__global__ __launch_bounds__(128) void k_symp_fill(int64_t n, int K, const RP* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                    const double* __restrict__ vals, int base, const DiaOffsets* __restrict__ Op, int cls,
                                                    double* __restrict__ out, SympGeom Gm, double* __restrict__ pv, const double* __restrict__ ssym,
                                                    unsigned long long* __restrict__ fp) {
  const DiaOffsets& O = *Op;
  extern __shared__ double lds[];
  constexpr int RUN = SP_W * 27;  // entries of a full 32-row run
  // Symmetry fingerprint (round 6; fp != nullptr): are the values this pass WRITES bitwise symmetric among the swept rows?  Every stored entry (r, c), c != r,
  // both rows swept, adds  sign(c - r) * m(min, max) * bits(v)  to a 64-bit sum in wrap-around arithmetic, m a 64-bit hash of the unordered pair.  A
  // symmetric copy cancels pair by pair -- exactly, in any order (integer sums commute: no atomics on doubles, no second pass); a copy with v_rc != v_cr
  // anywhere leaves a non-zero sum unless the pairs' hashed multipliers conspire (2^-63 for a given matrix).  It replaces the separate check pass over the
  // copy (k_spmv_symp<1>: 4.6 ms of the 21 ms a 512^3 solve spends outside its iterations); the pass is still there (bit 30 of the "ell" knob) and the
  // test-suite compares the two verdicts.  Stricter than the pass (which looks at the pairs the sweep mirrors): never the other way round.
  unsigned long long fsum = 0;
  const int64_t sw_lo = (int64_t)Gm.p0 * Gm.PL, sw_hi = (int64_t)Gm.p1 * Gm.PL;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // (launched with two waves per workgroup)
  double* T = lds + (size_t)w * (2 * RUN);
  double* E = lds + 2 * (2 * RUN);  // the step's edge block (SP_EPAD entries; zero between steps: entries of rows outside the lattice and the padding stay 0)
  const int spNP = Gm.NS * Gm.NPk;
  const int64_t spT = (int64_t)spNP * (Gm.p1 - Gm.p0);
  const int h = lane >> 5, c = lane & (SP_W - 1);
  const int32_t PL = (int32_t)Gm.PL, m2 = Gm.m2;
  auto uni64 = [](int64_t x) -> int64_t {  // (the same value in every lane: into scalar registers, so that what depends on it stays scalar)
    const uint32_t xl = __builtin_amdgcn_readfirstlane((uint32_t)(uint64_t)x), xh = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)x >> 32));
    return (int64_t)(((uint64_t)xh << 32) | xl);
  };
  for (int i = threadIdx.x; i < SP_EPAD; i += 128) E[i] = 0.0;
  __syncthreads();
  for (int64_t step = blockIdx.x; step < spT; step += gridDim.x) {  // (every barrier below is reached by both waves: the trip count is the workgroup's)
    const int kp = (int)(step % Gm.NPk);
    const int64_t q = step / Gm.NPk;
    const int strip = (int)(q % Gm.NS), pl = (int)(q / Gm.NS);
    const int jj0 = strip * SP_L + 2 * w, kk0 = kp * SP_W;
    const int ncol = m2 - kk0 < SP_W ? m2 - kk0 : SP_W, nlines = Gm.m1 - jj0 < 2 ? (Gm.m1 - jj0 < 1 ? 0 : 1) : 2;  // (the last strip may end before this wave's lines)
    const int64_t rb = (int64_t)(Gm.p0 + pl) * Gm.PL + (int64_t)jj0 * m2 + kk0;  // lane 0's row
    const int64_t rB = nlines == 2 ? rb + m2 : rb;                              // lane 32's row (no second line: the first again, nothing of it is used)
    int64_t a0 = 0, a1 = 0, b0 = 0, b1 = 0;
    if (nlines > 0) {
      a0 = uni64((int64_t)rowptr[rb]) - base, a1 = uni64((int64_t)rowptr[rb + ncol]) - base;
      b0 = uni64((int64_t)rowptr[rB]) - base, b1 = uni64((int64_t)rowptr[rB + ncol]) - base;
    }
    const bool full = ncol == SP_W && nlines == 2 && a1 - a0 == RUN && b1 - b0 == RUN;
    const bool valid = c < ncol && h < nlines;
    const int64_t r = rb + (int64_t)h * m2 + c;
    const int line = 2 * w + h;
    double* const pm = pv + step * SP_MAIN + line * SP_W + c;
    double* const plo = pv + spT * SP_MAIN + step * SP_LOW + line * SP_W + c;
    double* const pe = pv + step * SP_MAIN + 14 * SP_ROWS;
    const double* Tr = T + lane * 27;
    double sc[27];
    double srow = 1.0;
    uint32_t present = 0x7FFFFFFu;
    if (full) {
      double tv[27];
      const double* vA = vals + a0 + lane;
      const double* vB = vals + b0 + lane - RUN;
      const double* v13 = h ? vB : vA;  // entries 832 .. 895 of the tile: the first run ends at 864
#pragma unroll
      for (int u = 0; u < 27; ++u) tv[u] = __builtin_nontemporal_load((u < 13 ? vA : u == 13 ? v13 : vB) + 64 * u);
      if constexpr (SYM) {
        srow = ssym[r];
#pragma unroll
        for (int u = 0; u < 27; ++u) sc[u] = ssym[r + O.off[cls][u]];
      }
#pragma unroll
      for (int u = 0; u < 27; ++u) T[lane + 64 * u] = tv[u];
      __builtin_amdgcn_wave_barrier();
    } else {
      int64_t lo = 0;
      int len = 0;
      if (valid) {
        lo = (int64_t)rowptr[r] - base;
        len = (int)((int64_t)rowptr[r + 1] - base - lo);
      }
      const int cntA = (int)(a1 - a0), cntB = nlines == 2 ? (int)(b1 - b0) : 0;  // <= RUN each (regular blocks: at most 27 entries per row)
      {
        double ta[14], tb[14];
#pragma unroll
        for (int u = 0; u < 14; ++u) {
          const int i = lane + 64 * u;
          ta[u] = i < cntA ? __builtin_nontemporal_load(vals + a0 + i) : 0.0;
          tb[u] = i < cntB ? __builtin_nontemporal_load(vals + b0 + i) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 14; ++u) {
          const int i = lane + 64 * u;
          if (i < cntA) T[i] = ta[u];
          if (i < cntB) T[RUN + i] = tb[u];
        }
      }
      // the columns of the short rows (their entries are decoded into slots below; a row of 27 entries has entry s in slot s)
      int32_t cj[27];
      const bool shortrow = valid && len < 27;
#pragma unroll
      for (int j = 0; j < 27; ++j) cj[j] = (shortrow && j < len) ? col[lo + j] - base : 0;
      __builtin_amdgcn_wave_barrier();
      const int off0 = h * RUN + (int)(lo - (h ? b0 : a0));
      double ev[27];
#pragma unroll
      for (int j = 0; j < 27; ++j) ev[j] = j < len ? T[off0 + j] : 0.0;
      __builtin_amdgcn_wave_barrier();  // every lane holds its entries: the staging area may now be overwritten by the expanded rows
      present = 0;
      if (shortrow) {
#pragma unroll
        for (int sl = 0; sl < 27; ++sl) T[lane * 27 + sl] = 0.0;
      }
#pragma unroll
      for (int j = 0; j < 27; ++j) {
        if (j < len) {
          int sl = j;
          if (shortrow) {
            const int32_t d = cj[j] - (int32_t)r;
            const int di = (2 * d > PL) - (2 * d < -PL);
            const int32_t d1 = d - di * PL;
            const int dj = (2 * d1 > m2) - (2 * d1 < -m2);
            sl = 9 * (di + 1) + 3 * (dj + 1) + (d1 - dj * m2 + 1);
          }
          T[lane * 27 + sl] = ev[j];
          present |= 1u << sl;
        }
      }
      __builtin_amdgcn_wave_barrier();
      if constexpr (SYM) {
        srow = valid ? ssym[r] : 1.0;
#pragma unroll
        for (int u = 0; u < 27; ++u) sc[u] = valid ? ssym[r + O.off[cls][u]] : 1.0;
      }
    }
    if (valid) {
#pragma unroll
      for (int sl = 0; sl < 27; ++sl) {
        double v = 0.0;
        if (present >> sl & 1u) {
          v = Tr[sl];
          if constexpr (SYM) v = v * (srow * sc[sl]);  // (the product of the two factors first: a mirrored pair is multiplied by the same number)
          if (fp && sl != 13) {
            const int64_t cc = r + O.off[cls][sl];
            if (cc >= sw_lo && cc < sw_hi) {
              const uint64_t lo_ = (uint64_t)(sl < 13 ? cc : r), hi_ = (uint64_t)(sl < 13 ? r : cc);
              uint64_t z = lo_ * 0x9E3779B97F4A7C15ull + hi_ * 0xD1B54A32D192ED03ull + 0x2545F4914F6CDD1Dull;
              z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
              z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
              z = (z ^ (z >> 31)) | 1ull;
              const unsigned long long t = z * (unsigned long long)__double_as_longlong(v);
              fsum += sl < 13 ? (0ull - t) : t;
            }
          }
        }
        if (sl < 13) {
          DIA_ST(plo + sl * SP_ROWS, v);
          const int e = sp_edge_of(sl, line, c);
          if (e >= 0) E[e] = v;
        } else {
          DIA_ST(pm + (sl - 13) * SP_ROWS, v);
          if (sl == 13) DIA_ST(out + ell_base(r, K) + 13 * ELL_B, v);  // the diagonal also to the slot-major copy (k_ell_diag reads it there)
        }
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SP_EPAD; i += 128) {
      DIA_ST(pe + i, E[i]);
      E[i] = 0.0;
    }
    __syncthreads();
  }
  if (fp) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) fsum += __shfl_down(fsum, o, MFEM_WAVE);
    if (lane == 0 && fsum) atomicAdd(fp, fsum);
  }
}

// lane <-> RPT (2 or 4) neighbouring rows; a wave covers one aligned block of 64 RPT rows; U diagonals per batch
typedef double u_d2 __attribute__((ext_vector_type(2), aligned(8)));
// The RPT rows r .. r + RPT - 1 of one lane (r a multiple of RPT; a wave covers aligned 128-row blocks): regular blocks by
// diagonal, the others through their explicit columns.  Shared by the plain kernel and the symmetric sweep kernel (which
// sends the chunks outside its regular range here).
template <int RPT, int U, bool TRIPLES>
__device__ __forceinline__ void dia_rows(int64_t r, int64_t n, int64_t npad, int K, const DiaOffsets& O,
                                         const int32_t* __restrict__ flags, const int32_t* __restrict__ cols,
                                         const double* __restrict__ vals, const double* __restrict__ x, double* __restrict__ y,
                                         double alpha, double beta, const double* __restrict__ dotw, int xcd, double& dot_acc,
                                         int64_t skip_lo = 0, int64_t skip_hi = 0) {  // rows in [skip_lo, skip_hi) belong to another launch
  constexpr int H = RPT / 2;  // 16-byte pairs per lane
  const double* v = vals + ell_base(r, K);
  e_d2 acc[H];
  double xself0 = 0.0, xself1 = 0.0;  // x[r], x[r + 1] when the kernel has loaded them anyway (fused w.y with w == x, as in CG)
  bool have_self = false;
#pragma unroll
  for (int h = 0; h < H; ++h) acc[h] = (e_d2){0.0, 0.0};
  // the wave's rows [b0, b0 + 64 RPT) are RPT / 2 aligned 128-row blocks: regular only if all of them are (wave-uniform)
  const int64_t blk = r / (64 * RPT) * (RPT / 2);
  const int cls = __builtin_amdgcn_readfirstlane(flags[blk]) - 1;  // wave-uniform: keeps the offset reads scalar
  bool interior = cls >= 0;
  if (RPT == 4) interior = interior && ((blk + 1) * 128 < npad) && flags[blk + 1] == cls + 1;
  const int32_t* off = O.off[cls < 0 ? 0 : cls];
  const int D = O.D[cls < 0 ? 0 : cls];
  if (interior && TRIPLES && RPT == 2) {
    // the diagonals come in runs of three consecutive offsets (o - 1, o, o + 1: the fastest lattice direction): the two
    // rows of the lane need x[r + o - 1 .. r + o + 2] for the whole run -- two 16-byte loads instead of three
    for (int s = 0; s < D; s += 3) {
      const e_d2 va = __builtin_nontemporal_load(reinterpret_cast<const e_d2*>(v + s * ELL_B));
      const e_d2 vb = __builtin_nontemporal_load(reinterpret_cast<const e_d2*>(v + (s + 1) * ELL_B));
      const e_d2 vc = __builtin_nontemporal_load(reinterpret_cast<const e_d2*>(v + (s + 2) * ELL_B));
      const u_d2* xp = reinterpret_cast<const u_d2*>(x + r + off[s]);
      const u_d2 xa = xp[0], xb = xp[1];
      acc[0].x += va.x != 0.0 ? va.x * xa.x : 0.0;
      acc[0].y += va.y != 0.0 ? va.y * xa.y : 0.0;
      acc[0].x += vb.x != 0.0 ? vb.x * xa.y : 0.0;
      acc[0].y += vb.y != 0.0 ? vb.y * xb.x : 0.0;
      acc[0].x += vc.x != 0.0 ? vc.x * xb.x : 0.0;
      acc[0].y += vc.y != 0.0 ? vc.y * xb.y : 0.0;
      if (off[s + 1] == 0) {  // the main diagonal's run: x[r], x[r + 1] are the lane's own entries (wave-uniform test)
        xself0 = xa.y;
        xself1 = xb.x;
        have_self = true;
      }
    }
  } else if (interior) {
    int s = 0;
    for (; s + U <= D; s += U) {
      e_d2 vv[U][H];
      u_d2 xx[U][H];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int h = 0; h < H; ++h) {
          vv[u][h] = __builtin_nontemporal_load(reinterpret_cast<const e_d2*>(v + (s + u) * ELL_B) + h);
          xx[u][h] = *(reinterpret_cast<const u_d2*>(x + r + off[s + u]) + h);
        }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int h = 0; h < H; ++h) {
          // a zero slot stands for "no entry": it must not pick up a non-finite x from a position the CSR row never reads
          acc[h].x += vv[u][h].x != 0.0 ? vv[u][h].x * xx[u][h].x : 0.0;
          acc[h].y += vv[u][h].y != 0.0 ? vv[u][h].y * xx[u][h].y : 0.0;
        }
    }
    for (; s < D; ++s)
#pragma unroll
      for (int h = 0; h < H; ++h) {
        const e_d2 vv = __builtin_nontemporal_load(reinterpret_cast<const e_d2*>(v + s * ELL_B) + h);
        const u_d2 xx = *(reinterpret_cast<const u_d2*>(x + r + off[s]) + h);
        acc[h].x += vv.x != 0.0 ? vv.x * xx.x : 0.0;
        acc[h].y += vv.y != 0.0 ? vv.y * xx.y : 0.0;
      }
  } else {  // generic block (boundary rows, ghost columns): explicit columns, compact slots
    const int32_t* c = cols + ell_base(r, K);
    for (int s = 0; s < K; ++s)
#pragma unroll
      for (int h = 0; h < H; ++h) {
        if (r + 2 * h >= npad) continue;
        const e_d2 vv = __builtin_nontemporal_load(reinterpret_cast<const e_d2*>(v + s * ELL_B) + h);
        const e_i2 cc = __builtin_nontemporal_load(reinterpret_cast<const e_i2*>(c + s * ELL_B) + h);
        acc[h].x += vv.x * x[cc.x];
        acc[h].y += vv.y * x[cc.y];
      }
  }
#pragma unroll
  for (int h = 0; h < H; ++h) {
    const int64_t rr = r + 2 * h;
    if (rr >= n) break;
    double y0 = alpha * acc[h].x, y1 = alpha * acc[h].y;
    const bool one = rr < skip_lo || rr >= skip_hi;
    const bool two = rr + 1 < n && (rr + 1 < skip_lo || rr + 1 >= skip_hi);
    if (beta != 0.0) {
      if (one) y0 += beta * y[rr];
      if (two) y1 += beta * y[rr + 1];
    }
    if (one) y[rr] = y0;
    if (two) y[rr + 1] = y1;
    if (dotw) {
      if (RPT == 2 && have_self && dotw == x) {  // p.Ap of CG: p[r], p[r + 1] are already in registers
        if (one) dot_acc += y0 * xself0;
        if (two) dot_acc += y1 * xself1;
      } else {
        if (one) dot_acc += y0 * dotw[rr];
        if (two) dot_acc += y1 * dotw[rr + 1];
      }
    }
  }
}

template <int RPT, int U, bool TRIPLES = false>
__global__ __launch_bounds__(1024) void k_spmv_dia(int64_t n, int64_t npad, int K, const DiaOffsets* __restrict__ Op,
                                                           const int32_t* __restrict__ flags, const int32_t* __restrict__ cols,
                                                           const double* __restrict__ vals, const double* __restrict__ x,
                                                           double* __restrict__ y, double alpha, double beta,
                                                           const double* __restrict__ dotw, double* __restrict__ partials,
                                                           const int32_t* __restrict__ done_flag, int xcd, SpmvPart part) {
  __shared__ double red[16];
  if (done_flag && done_flag[0]) return;
  const DiaOffsets& O = *Op;
  double dot_acc = 0.0;
  // xcd > 0: workgroups with equal blockIdx % 8 share an XCD (round-robin dispatch); XCD x walks its own contiguous eighth of
  // the rows, so the x window an L2 has to hold is an eighth of the vector instead of all of it
  const int64_t rows_per_wg = (int64_t)blockDim.x * RPT;
  const int64_t nchunks = (n + rows_per_wg - 1) / rows_per_wg;
  int64_t chunk = blockIdx.x, chunk_end = nchunks, chunk_step = gridDim.x;
  if (xcd & 1) {
    const int64_t per = (nchunks + 7) / 8;
    chunk = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    chunk_end = ((blockIdx.x & 7) + 1) * per < nchunks ? ((blockIdx.x & 7) + 1) * per : nchunks;
    chunk_step = gridDim.x >> 3;
  }
  for (; chunk < chunk_end; chunk += chunk_step) {
    const int64_t r = chunk * rows_per_wg + (int64_t)threadIdx.x * RPT;
    if (r >= n) continue;
    if (spmv_part_skip(part, chunk * rows_per_wg, (chunk + 1) * rows_per_wg)) continue;  // workgroup-uniform
    dia_rows<RPT, U, TRIPLES>(r, n, npad, K, O, flags, cols, vals, x, y, alpha, beta, dotw, xcd, dot_acc);
  }
  if (partials) {
    const double b = block_reduce_sum(dot_acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = b;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Symmetric sweep kernel for the 27-point lattice stencil (offsets di PL + dj m2 + dk).  For a symmetric matrix the entry of
// row r on a lower diagonal -o equals the entry of row r - o on the upper diagonal +o.  A workgroup owns an in-plane tile
// of 512 rows and sweeps it through consecutive lattice planes (chunks c, c + S, c + 2 S, ...): the nine upper diagonals
// that point to the next plane are kept in LDS when they are loaded, and the next plane's rows read their nine
// previous-plane (lower) diagonals from there instead of from HBM.  Same products, same summation order as the plain
// diagonal-slotted kernel: the result is bitwise the same whenever the matrix is bitwise symmetric (checked at bind time).
// ---------------------------------------------------------------------------------------------------------------
#define SYM_ROWS 512                 // rows of a tile = 4 blocks of 128 (768 rows / 384 threads / 2 workgroups per CU mirror more but run at 1.07 instead of 0.93 ms per CG iteration)
#define SYM_THREADS (SYM_ROWS / 2)
#define SYM_WG_PER_CU 3              // 53 KB of LDS per workgroup
#define SYM_LD(p) __builtin_nontemporal_load(p)  // plain loads measured slower: 0.954 vs 0.928 ms per CG iteration at 256^3
__global__ __launch_bounds__(SYM_THREADS) void k_spmv_sym27(int64_t n, int64_t npad, int K, const DiaOffsets* __restrict__ Op,
                                                             const int32_t* __restrict__ flags, const int32_t* __restrict__ cols,
                                                             const double* __restrict__ vals, const double* __restrict__ x,
                                                             double* __restrict__ y, double alpha, double beta,
                                                             const double* __restrict__ dotw, double* __restrict__ partials,
                                                             const int32_t* __restrict__ done_flag, int64_t c0, int64_t c1, int S, int nsteps, int cls, int gs,
                                                             int part) {  // 0: sweep + the chunks outside it; 1: sweep only; 2: only the chunks outside the sweep (every row that reads a ghost column of a slab is among them)
  __shared__ __attribute__((aligned(16))) double hist[9][SYM_ROWS];  // diagonals 18..26 (into the next plane) of the previous chunk
  __shared__ __attribute__((aligned(16))) double exch[4][SYM_ROWS];  // diagonals 14..17 (+z, +y) of this chunk
  __shared__ double red[16];
  if (done_flag && done_flag[0]) return;
  const int32_t* off = Op->off[cls];
  const int tid = threadIdx.x;
  double dot_acc = 0.0;
  // gs workgroups over S tiles: tile t is swept by nseg (+ 1 for the first gs % S tiles) workgroups, each taking a contiguous
  // range of the tile's nsteps plane steps
  const int tile = blockIdx.x % S, seg = blockIdx.x / S;
  const int nseg = gs / S + (tile < gs % S ? 1 : 0);
  const int seg_len = part == 2 ? 0 : (nsteps + nseg - 1) / nseg;
  bool have_hist = false;
  e_d2 up_next[4] = {{0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}};
  {
    const int64_t chunk = c0 + tile + (int64_t)S * ((int64_t)seg * seg_len);
    if (part != 2 && seg * seg_len < nsteps && chunk < c1) {
      const double* v = vals + ell_base(chunk * SYM_ROWS + 2 * tid, K);
#pragma unroll
      for (int u = 0; u < 4; ++u) up_next[u] = SYM_LD(reinterpret_cast<const e_d2*>(v + (14 + u) * ELL_B));
    }
  }
  for (int it = 0; it < seg_len; ++it) {
    const int step = seg * seg_len + it;
    const int64_t chunk = c0 + tile + (int64_t)S * step;
    if (step >= nsteps || chunk >= c1) break;  // workgroup-uniform
    const int64_t r = chunk * SYM_ROWS + 2 * tid;
    const double* v = vals + ell_base(r, K);
    e_d2 acc = {0.0, 0.0};
    // ---- the lane's own +z / +y diagonals first: the rows behind it in this chunk read them as their -z / -y diagonals
    e_d2 up[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      up[u] = up_next[u];  // requested during the previous chunk (or before the loop)
      *reinterpret_cast<e_d2*>(&exch[u][2 * tid]) = up[u];
    }
    __syncthreads();  // exch complete; also: every wave has finished writing the previous chunk's history
    // a slot's value pair, mirrored from LDS (row `lp` of table `tab`) when both source rows are in it, else from the row itself
    auto slot = [&](int s, const double* tab, int lp, bool ok) -> e_d2 {
      e_d2 w;
      if (ok && lp >= 0 && lp + 1 < SYM_ROWS) {
        w.x = tab[lp];
        w.y = tab[lp + 1];
      } else {
        w = SYM_LD(reinterpret_cast<const e_d2*>(v + s * ELL_B));
      }
      return w;
    };
#define SYM_RUN(va, vb, vc, s0)                                          \
  {                                                                      \
    const u_d2* xp = reinterpret_cast<const u_d2*>(x + r + off[s0]);     \
    const u_d2 xa = xp[0], xb = xp[1];                                   \
    acc.x += va.x != 0.0 ? va.x * xa.x : 0.0;                            \
    acc.y += va.y != 0.0 ? va.y * xa.y : 0.0;                            \
    acc.x += vb.x != 0.0 ? vb.x * xa.y : 0.0;                            \
    acc.y += vb.y != 0.0 ? vb.y * xb.x : 0.0;                            \
    acc.x += vc.x != 0.0 ? vc.x * xb.x : 0.0;                            \
    acc.y += vc.y != 0.0 ? vc.y * xb.y : 0.0;                            \
    if (s0 == 12) {                                                      \
      xself0 = xa.y;                                                     \
      xself1 = xb.x;                                                     \
    }                                                                    \
  }
    double xself0 = 0.0, xself1 = 0.0;
    // ---- the nine diagonals into the previous plane: entry (r, r + o) = entry (r + o, r) on diagonal 26 - s of row r + o,
    //      kept in `hist` if that row was in the previous chunk of this sweep
    const int lph = 2 * tid + S * SYM_ROWS;
#pragma unroll
    for (int s = 0; s < 9; s += 3) {
      const e_d2 va = slot(s, hist[8 - s], lph + off[s], have_hist);
      const e_d2 vb = slot(s + 1, hist[7 - s], lph + off[s + 1], have_hist);
      const e_d2 vc = slot(s + 2, hist[6 - s], lph + off[s + 2], have_hist);
      SYM_RUN(va, vb, vc, s);
    }
    {  // -y diagonals 9..11 <- +y diagonals 17..15 of the rows one lattice line behind, if those are in this chunk
      const e_d2 va = slot(9, exch[3], 2 * tid + off[9], true);
      const e_d2 vb = slot(10, exch[2], 2 * tid + off[10], true);
      const e_d2 vc = slot(11, exch[1], 2 * tid + off[11], true);
      SYM_RUN(va, vb, vc, 9);
    }
    {  // -z (12) <- +z (14) of the row before; main diagonal 13; +z from the registers
      e_d2 va;
      va.y = up[0].x;  // row r + 1: entry (r + 1, r) = entry (r, r + 1)
      if (tid > 0) va.x = exch[0][2 * tid - 1];
      else va.x = v[12 * ELL_B];
      const e_d2 vb = SYM_LD(reinterpret_cast<const e_d2*>(v + 13 * ELL_B));
      SYM_RUN(va, vb, up[0], 12);
    }
    SYM_RUN(up[1], up[2], up[3], 15);
    if (it + 1 < seg_len && step + 1 < nsteps && chunk + S < c1) {  // the next chunk's +z / +y diagonals: their latency hides behind
      const double* vn = vals + ell_base((chunk + S) * SYM_ROWS + 2 * tid, K);  // the rest of this chunk
#pragma unroll
      for (int u = 0; u < 4; ++u) up_next[u] = SYM_LD(reinterpret_cast<const e_d2*>(vn + (14 + u) * ELL_B));
    }
    __syncthreads();  // every wave is done reading hist and exch
    // ---- the nine diagonals into the next plane: they also go to LDS for the next chunk of the sweep
#pragma unroll
    for (int s = 18; s < 27; s += 3) {
      const e_d2 va = SYM_LD(reinterpret_cast<const e_d2*>(v + s * ELL_B));
      const e_d2 vb = SYM_LD(reinterpret_cast<const e_d2*>(v + (s + 1) * ELL_B));
      const e_d2 vc = SYM_LD(reinterpret_cast<const e_d2*>(v + (s + 2) * ELL_B));
      SYM_RUN(va, vb, vc, s);
      *reinterpret_cast<e_d2*>(&hist[s - 18][2 * tid]) = va;
      *reinterpret_cast<e_d2*>(&hist[s - 17][2 * tid]) = vb;
      *reinterpret_cast<e_d2*>(&hist[s - 16][2 * tid]) = vc;
    }
#undef SYM_RUN
    have_hist = true;
    double y0 = alpha * acc.x, y1 = alpha * acc.y;
    if (beta != 0.0) {
      y0 += beta * y[r];
      y1 += beta * y[r + 1];
    }
    y[r] = y0;
    y[r + 1] = y1;
    if (dotw) {
      if (dotw == x) dot_acc += y0 * xself0 + y1 * xself1;
      else dot_acc += y0 * dotw[r] + y1 * dotw[r + 1];
    }
  }
  // after its sweep every workgroup takes a share of the chunks outside the regular range (first / last lattice planes, ghost
  // planes of a slab) through the plain per-row code: no extra workgroups, no tail behind the sweeps
  if (part != 1) {
    const int64_t nchunks = (n + SYM_ROWS - 1) / SYM_ROWS;
    for (int64_t q = blockIdx.x;; q += gridDim.x) {
      const int64_t ch = q < c0 ? q : c1 + (q - c0);
      if (ch >= nchunks) break;
      const int64_t r = ch * SYM_ROWS + 2 * tid;
      if (r < n) dia_rows<2, 3, true>(r, n, npad, K, *Op, flags, cols, vals, x, y, alpha, beta, dotw, 0, dot_acc);
    }
  }
  if (partials) {
    const double b = block_reduce_sum(dot_acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = b;
  }
}

// bad[0] |= 1 unless, for every row r of the regular chunk range and every lower diagonal s < 13 whose source row r + off[s]
// is in the range too, entry (r, s) equals entry (r + off[s], 26 - s) bitwise: exactly the substitutions k_spmv_sym27 makes
__global__ __launch_bounds__(MFEM_BLOCK) void k_sym27_check(int K, const DiaOffsets* __restrict__ Op, const double* __restrict__ vals,
                                                              int64_t row_lo, int64_t row_hi, int cls, int32_t* __restrict__ bad) {
  const int32_t* off = Op->off[cls];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int fail = 0;
  for (int64_t r = row_lo + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < row_hi; r += stride) {
    const double* v = vals + ell_base(r, K);
    for (int s = 0; s < 13; ++s) {
      const int64_t rs = r + off[s];
      if (rs < row_lo) continue;
      const double a = v[s * ELL_B], b = vals[ell_base(rs, K) + (26 - s) * ELL_B];
      if (__double_as_longlong(a) != __double_as_longlong(b) && !(a == 0.0 && b == 0.0)) fail = 1;
    }
  }
  if (fail) atomicOr(bad, 1);
}

// ---------------------------------------------------------------------------------------------------------------
// Symmetric sweep, wave-private patches (k_spmv_symp; default when the lattice form is recognised).  The workgroup-tile kernel
// above cuts a lattice plane into contiguous 512-row ranges: a lattice line longer than the tile (512^3: 513 points) leaves
// only the in-line diagonals mirrorable (31 %), and two workgroup barriers per chunk bound what three workgroups per CU can keep
// in flight.  Here a WAVE owns a (j, k) patch of 4 lattice lines x 32 points (lane <-> two neighbouring points of one line) and
// sweeps it through consecutive lattice planes with no workgroup barrier at all:
//   * the matrix values of the swept planes live in a patch-major copy, made by k_symp_bind when the solve binds its values: what every
//     step reads -- slots 13..26 and the edge block, 16.9 KB -- contiguous per step [plane][patch], the lower slots (read where a run
//     starts and by the symmetry check) behind;
//   * the upper diagonals of a step go to the wave's LDS block when they are loaded: +z / +y (slots 14..17) for the rows behind
//     them in this plane, the nine next-plane diagonals (18..26) for the same patch one plane on -- 10.5 of the 13 lower
//     diagonals of a row are mirrored from there whatever the line length (the rest: patch edges, read from the row's own slot);
//   * x is staged per plane: the (4 + 2) x (32 + 2) neighbourhood of the patch enters LDS once and serves the 27 products of three
//     consecutive steps -- four global loads per lane and step instead of eighteen gathers.
// Same products, same summation order as the plain diagonal-slotted kernel (v_mul_f64 + v_add_f64, no contraction): y is bitwise the
// same whenever the mirrored pairs are bitwise equal (MODE 1 checks exactly those pairs when the values are bound).
//   * A halo cell of a mirror table (its source row belongs to another patch) is filled from the step's EDGE BLOCK -- the own
//     slot-s entries of the rows at the patch rim, 318 doubles stored behind the step's 27 slots -- so that every lane reads every
//     lower slot with the same two LDS loads; a step reads 14 slots x 1 KB + 2.5 KB of edge block + 1.6 KB of x.
//   * Everything a step needs from memory is requested one step ahead into registers (its loads are in flight during the products
//     of the current step); two wave-level barriers per step order the LDS phases.
//   * Runs: a run = one patch through nplanes / nseg consecutive planes, one wave (one-wave workgroups, 7 per CU: 22.8 KB of LDS
//     each); a run's first step has no history and fills the previous-plane tables from the rows' own slots.  XCD c (workgroups
//     with blockIdx % 8 == c) sweeps a contiguous eighth of the patches, segment by segment, so neighbouring patches advance
//     through the planes together on one L2 (512^3: CG iteration 7.29 -> 6.66 ms against arbitrary equal cuts of the step list).
//   * Rows outside the swept planes (first / last lattice plane, the planes next to the ghost planes of a slab) come from the slot-major
//     copy through the per-row code: the sweep's waves take them in 128-row units after their runs (unsplit SpMV: one launch, no tail);
//     in a split (multi-rank) SpMV they are the boundary part, a launch of their own (k_spmv_dia_outside) after the halo has arrived.
// Measured (CG iteration, tools/probe_sym.py): see symp_wanted().  What bounds it: 2.78 GB of fabric traffic per SpMV at 256^3
// (2.61 GB by the count above) in 0.59 ms = 4.7 TB/s; the time does not depend on the number of resident waves (2 .. 7 per CU), the
// y stores cost 0.1 ms of it (non-temporal 16-byte stores: -1.5 %), the edge block 0.07 ms, the x staging 0.03 ms
// (profiles/r02_symp_experiments.txt).
// ---------------------------------------------------------------------------------------------------------------

// the rows outside the swept planes, taken by the sweep's waves after their runs (unsplit SpMV): per-row code on the slot-major copy
struct SympTail {
  int on, K;
  int64_t n, npad, lo, hi;  // rows [lo, hi) are the sweep's
  const DiaOffsets* Op;
  const int32_t* flags;
  const int32_t* cols;
  const double* ell;
};
__constant__ int32_t c_sp_ecell[SP_EPAD];  // edge block entry -> LDS cell of its mirror table (the two padding entries: a spare cell)
static int symp_upload_tables(int device) {  // __constant__ data is per device
  static bool done[64] = {};
  if (device >= 0 && device < 64 && done[device]) return MFEM_OK;
  int32_t h[SP_EPAD];
  for (int e = 0; e < SP_EPAD; ++e) {
    int s_, l_, c_, cell = 0;
    h[e] = sp_edge(e, s_, l_, c_, cell) ? cell : SP_TAB;
  }
  MFEM_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_sp_ecell), h, sizeof(h)));
  if (device >= 0 && device < 64) done[device] = true;
  return MFEM_OK;
}

template <int MODE>
__global__ __launch_bounds__(64) void k_spmv_symp(SympGeom Gm, const double* __restrict__ pv, const double* __restrict__ x,
                                                   double* __restrict__ y, double alpha, double beta,
                                                   const double* __restrict__ dotw, double* __restrict__ partials,
                                                   const int32_t* __restrict__ done_flag, int32_t* __restrict__ bad, SympTail tail) {
  __shared__ __attribute__((aligned(16))) double xs[3][SP_XL][SP_XW];
  __shared__ __attribute__((aligned(16))) double tab[SP_TAB + 2];
  if (done_flag && done_flag[0]) return;
  const int lane = threadIdx.x, lj = lane / SP_PW, pk = lane % SP_PW, lb = lj * SP_LS + 2 * pk;
  const int NP = Gm.NS * Gm.NPk, nplanes = Gm.p1 - Gm.p0;
  // Runs and XCDs: workgroups with equal blockIdx % 8 share an XCD (round-robin dispatch; gridDim.x is a multiple of 8).  XCD c sweeps
  // a contiguous eighth of the patches, segment by segment, so that the runs resident on it at any time are neighbouring patches at
  // about the same plane: their overlapping x neighbourhoods meet in that XCD's L2.
  const int xcd = blockIdx.x & 7, pc = NP / 8, prem = NP % 8, pcnt = pc + (xcd < prem ? 1 : 0), pfirst = xcd * pc + (xcd < prem ? xcd : prem);
  // the LDS cells this lane fills from the edge block (5 entries per lane; table made on the host once: decoding 320 entries with
  // sp_edge() at the top of every launch cost every wave a few thousand instructions)
  int ecell[SP_EU];
#pragma unroll
  for (int u = 0; u < SP_EU; ++u) ecell[u] = c_sp_ecell[lane + 64 * u];
  int cur_patch = -1, bp = 0, bc = 1, bn = 2;  // x ring: previous / current / next plane
  bool have_hist = false, vx = false, vy = false;
  int64_t rin = 0;       // in-plane row offset j * m2 + k of the lane's first row
  int xo[SP_XU], xa[SP_XU];  // x staging: in-plane offset (may be negative) and LDS slot of the lane's neighbourhood points
  double dot_acc = 0.0;
  int fail = 0;
  e_d2 cur[14];          // slots 13..26 of the step, requested one step ahead
  double ed[SP_EU], xr[SP_XU];  // its edge block entries and the x neighbourhood of the plane after it
  // x neighbourhood entry of plane `plane`: positions outside the vector's owned entries (beyond the last lattice line of the last
  // plane) are only ever multiplied by structurally absent entries -- any finite value serves: clamp
  auto xidx = [&](int plane, int u) -> int64_t {
    int64_t idx = (int64_t)plane * Gm.PL + xo[u];
    idx = idx < 0 ? 0 : idx;
    return idx < Gm.nx ? idx : Gm.nx - 1;
  };
  auto request = [&](const double* v, int pnext) {
    if (vx) {
#pragma unroll
      for (int u = 0; u < 14; ++u) cur[u] = SYM_LD(reinterpret_cast<const e_d2*>(v + 2 * lane + u * SP_ROWS));
    } else {
#pragma unroll
      for (int u = 0; u < 14; ++u) cur[u] = (e_d2){0.0, 0.0};
    }
    if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < SP_EU; ++u) ed[u] = SYM_LD(v + 14 * SP_ROWS + lane + 64 * u);  // the block is padded to SP_EPAD entries
#pragma unroll
      for (int u = 0; u < SP_XU - 1; ++u) xr[u] = x[xidx(pnext, u)];
      xr[SP_XU - 1] = lane < SP_XN - 64 * (SP_XU - 1) ? x[xidx(pnext, SP_XU - 1)] : 0.0;
    }
  };
  auto stage_x = [&](int buf, int plane) {
    double* dst = &xs[buf][0][0];
#pragma unroll
    for (int u = 0; u < SP_XU - 1; ++u) dst[xa[u]] = x[xidx(plane, u)];
    if (lane < SP_XN - 64 * (SP_XU - 1)) dst[xa[SP_XU - 1]] = x[xidx(plane, SP_XU - 1)];
  };
  for (int run = blockIdx.x >> 3; run < pcnt * Gm.nseg; run += gridDim.x >> 3) {
  const int patch = pfirst + run % pcnt, seg = run / pcnt;
  const int64_t t0 = (int64_t)patch * nplanes + (int64_t)nplanes * seg / Gm.nseg, t1 = (int64_t)patch * nplanes + (int64_t)nplanes * (seg + 1) / Gm.nseg;
  cur_patch = -1;
  for (int64_t t = t0; t < t1; ++t) {
    const int p = Gm.p0 + (int)(t - (int64_t)patch * nplanes);
    const int64_t step = (int64_t)(p - Gm.p0) * NP + patch;  // [plane][patch]: the runs of a segment advance plane by plane together
    const double* v = pv + step * SP_MAIN;                                         // slots 13..26 + edge block of the step
    const double* vlow = pv + (int64_t)NP * nplanes * SP_MAIN + step * SP_LOW;    // its slots 0..12
    if (patch != cur_patch) {  // wave-uniform: a run or a patch starts -- nothing was requested ahead, no history
      cur_patch = patch;
      const int j0 = (patch / Gm.NPk) * SP_L, k0 = (patch % Gm.NPk) * SP_W;
      const int j = j0 + lj, k = k0 + 2 * pk;
      vx = j < Gm.m1 && k < Gm.m2;
      vy = j < Gm.m1 && k + 1 < Gm.m2;
      rin = (int64_t)j * Gm.m2 + k;
#pragma unroll
      for (int u = 0; u < SP_XU; ++u) {
        const int tt = lane + 64 * u, xl = tt / SP_XC, xc = tt - SP_XC * xl;
        xo[u] = (j0 - 1 + xl) * Gm.m2 + (k0 - 1 + xc);
        xa[u] = xl * SP_XW + xc;  // (u = 3: only lanes < 12 belong to the neighbourhood)
      }
      have_hist = false;
      __syncthreads();  // the previous patch's last products may still be reading the x ring
      request(v, p + 1);
      if (MODE == 0) {
        stage_x(bp, p - 1);
        stage_x(bc, p);
        // no history: the row's own previous-plane slots go where the mirror reads would look for them
#pragma unroll
        for (int s = 0; s < 9; ++s) {
          const e_d2 w = vx ? SYM_LD(reinterpret_cast<const e_d2*>(vlow + 2 * lane + s * SP_ROWS)) : (e_d2){0.0, 0.0};
          double* c = tab + sp_tbase(s) + (sp_dj(s) + sp_adj(s)) * SP_LS + sp_dk(s) + 2 + lb;
          c[0] = w.x;
          c[1] = w.y;
        }
      }
    }
    // ---- phase B: this step's +z / +y slots, its edge entries and the next plane's x go to LDS
#pragma unroll
    for (int s = 9; s < 13; ++s) *reinterpret_cast<e_d2*>(tab + sp_tbase(s) + sp_adj(s) * SP_LS + 2 + lb) = cur[26 - s - 13];
    e_d2 low[13];
    if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < SP_EU; ++u) tab[ecell[u]] = ed[u];
      double* dst = &xs[bn][0][0];
#pragma unroll
      for (int u = 0; u < SP_XU - 1; ++u) dst[xa[u]] = xr[u];
      if (lane < SP_XN - 64 * (SP_XU - 1)) dst[xa[SP_XU - 1]] = xr[SP_XU - 1];
    } else {
#pragma unroll
      for (int s = 0; s < 13; ++s) low[s] = vx ? SYM_LD(reinterpret_cast<const e_d2*>(vlow + 2 * lane + s * SP_ROWS)) : (e_d2){0.0, 0.0};
    }
    // the step's own upper slots stay in `mine`; the next step of the same sweep is requested now and arrives during the products
    e_d2 mine[14];
#pragma unroll
    for (int u = 0; u < 14; ++u) mine[u] = cur[u];
    const bool more = t + 1 < t1 && p + 1 < Gm.p1;  // the next step continues this sweep
    __syncthreads();  // one wave: orders its LDS writes before the reads of other lanes
    if (more) request(v + (int64_t)NP * SP_MAIN, p + 2);
    auto mirrored = [&](int s) -> e_d2 {
      const double* c = tab + sp_tbase(s) + (sp_dj(s) + sp_adj(s)) * SP_LS + sp_dk(s) + 2 + lb;
      e_d2 w;
      w.x = c[0];
      w.y = c[1];
      return w;
    };
    e_d2 acc = {0.0, 0.0};
    double xself0 = 0.0, xself1 = 0.0;
    // products rounded, then added, in slot order: what the plain kernel computes.  A structurally absent entry is an explicit zero
    // and every staged x is an owned entry of the vector (finite whenever x is), so its product is a signed zero that leaves the sum
    // unchanged -- no select needed here.
    auto run = [&](const e_d2& va, const e_d2& vb, const e_d2& vc, int buf, int dj, bool self) {
#pragma clang fp contract(off)  // v_mul_f64 + v_add_f64 like the plain kernel, not v_fma_f64
      const double* xp = &xs[buf][lj + 1 + dj][2 * pk];
      const e_d2 xa2 = *reinterpret_cast<const e_d2*>(xp), xb2 = *reinterpret_cast<const e_d2*>(xp + 2);
      acc.x = acc.x + va.x * xa2.x;
      acc.y = acc.y + va.y * xa2.y;
      acc.x = acc.x + vb.x * xa2.y;
      acc.y = acc.y + vb.y * xb2.x;
      acc.x = acc.x + vc.x * xb2.x;
      acc.y = acc.y + vc.y * xb2.y;
      if (self) {
        xself0 = xa2.y;
        xself1 = xb2.x;
      }
    };
    if (MODE == 1) {
      // exactly the pairs the sweep mirrors: source rows inside the patch, previous-plane slots only where a history exists
#pragma unroll
      for (int s = 0; s < 13; ++s) {
        const int dj = sp_dj(s), dk = sp_dk(s);
        const bool in = lj + dj >= 0 && lj + dj < SP_L && (dk < 0 ? pk > 0 : dk > 0 ? pk < SP_PW - 1 : true);
        if (in && vx && (s >= 9 || have_hist)) {
          const e_d2 m = mirrored(s);
          if (__double_as_longlong(m.x) != __double_as_longlong(low[s].x) && !(m.x == 0.0 && low[s].x == 0.0)) fail = 1;
          if (vy && __double_as_longlong(m.y) != __double_as_longlong(low[s].y) && !(m.y == 0.0 && low[s].y == 0.0)) fail = 1;
        }
      }
    } else {
      run(mirrored(0), mirrored(1), mirrored(2), bp, -1, false);
      run(mirrored(3), mirrored(4), mirrored(5), bp, 0, false);
      run(mirrored(6), mirrored(7), mirrored(8), bp, 1, false);
      run(mirrored(9), mirrored(10), mirrored(11), bc, -1, false);
      run(mirrored(12), mine[0], mine[1], bc, 0, true);
      run(mine[2], mine[3], mine[4], bc, 1, false);
    }
    __syncthreads();  // every lane is done with the tables
#pragma unroll
    for (int s = 0; s < 9; ++s) *reinterpret_cast<e_d2*>(tab + sp_tbase(s) + sp_adj(s) * SP_LS + 2 + lb) = mine[26 - s - 13];
    have_hist = true;
    if (MODE == 0) {
      run(mine[5], mine[6], mine[7], bn, -1, false);
      run(mine[8], mine[9], mine[10], bn, 0, false);
      run(mine[11], mine[12], mine[13], bn, 1, false);
      const int64_t r = (int64_t)p * Gm.PL + rin;
      double y0 = alpha * acc.x, y1 = alpha * acc.y;
      if (beta != 0.0) {
        if (vx) y0 += beta * y[r];
        if (vy) y1 += beta * y[r + 1];
      }
      {
        // one 16-byte store (8-byte aligned), non-temporal: 0.910 -> 0.896 ms per CG iteration at 256^3
        if (vy) __builtin_nontemporal_store((u_d2){y0, y1}, reinterpret_cast<u_d2*>(y + r));
        else if (vx) __builtin_nontemporal_store(y0, y + r);
      }
      if (dotw) {
        if (dotw == x) {
          if (vx) dot_acc += y0 * xself0;
          if (vy) dot_acc += y1 * xself1;
        } else {
          if (vx) dot_acc += y0 * dotw[r];
          if (vy) dot_acc += y1 * dotw[r + 1];
        }
      }
      const int b = bp;
      bp = bc;
      bc = bn;
      bn = b;
    }
    if (!more) cur_patch = -1;  // nothing requested: the next step (if any) starts like a run
  }
  }
  if (MODE == 0 && tail.on) {
    // 128-row units in front of and behind the swept planes (a unit straddling the boundary is masked row by row), shared among the waves
    const int64_t UA = (tail.lo + 127) / 128, ub = tail.hi / 128, UB = (tail.n + 127) / 128 - ub;
    for (int64_t u = blockIdx.x; u < UA + UB; u += gridDim.x) {
      const int64_t r = (u < UA ? u : ub + (u - UA)) * 128 + 2 * lane;
      if (r < tail.n)
        dia_rows<2, 3, true>(r, tail.n, tail.npad, tail.K, *tail.Op, tail.flags, tail.cols, tail.ell, x, y, alpha, beta, dotw, 0, dot_acc,
                             tail.lo, tail.hi);
    }
  }
  if (MODE == 1) {
    if (fail) atomicOr(bad, 1);
  } else if (partials) {
    const double w = wave_reduce_sum(dot_acc);
    if (lane == 0) partials[blockIdx.x] = w;
  }
}

// patch-major copy of the swept planes from the slot-major copy: per step [plane - p0][patch] the slots 13..26 + the edge block (main
// part) and, behind all main parts, the slots 0..12 (low part), zero
// where the patch sticks out of the lattice; one wave per (plane, patch)
__global__ __launch_bounds__(MFEM_BLOCK) void k_symp_bind(SympGeom Gm, int K, const double* __restrict__ ell, double* __restrict__ pv) {
  const int lane = threadIdx.x & 63, lj = lane / SP_PW, pk = lane % SP_PW;
  const int NP = Gm.NS * Gm.NPk;
  const int64_t T = (int64_t)NP * (Gm.p1 - Gm.p0);
  for (int64_t t = (int64_t)blockIdx.x * (MFEM_BLOCK / 64) + (threadIdx.x >> 6); t < T; t += (int64_t)gridDim.x * (MFEM_BLOCK / 64)) {
    const int nplanes = Gm.p1 - Gm.p0, patch = (int)(t / nplanes), p = Gm.p0 + (int)(t % nplanes);
    const int j0 = (patch / Gm.NPk) * SP_L, k0 = (patch % Gm.NPk) * SP_W;
    const int j = j0 + lj, k = k0 + 2 * pk;
    const bool vx = j < Gm.m1 && k < Gm.m2, vy = j < Gm.m1 && k + 1 < Gm.m2;
    const int64_t r = (int64_t)p * Gm.PL + (int64_t)j * Gm.m2 + k;
    const int64_t b0 = ell_base(r, K), b1 = ell_base(r + 1, K);
    const int64_t step = (int64_t)(p - Gm.p0) * NP + patch;
    double* out = pv + step * SP_MAIN;
    double* outlow = pv + T * SP_MAIN + step * SP_LOW;
    for (int s0 = 0; s0 < 27; s0 += 9) {
      e_d2 w[9];
#pragma unroll
      for (int u = 0; u < 9; ++u) {
        w[u].x = vx ? ell[b0 + (s0 + u) * ELL_B] : 0.0;
        w[u].y = vy ? ell[b1 + (s0 + u) * ELL_B] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 9; ++u) {
        const int sl = s0 + u;  // slots 0..12 to the low part, 13..26 to the main part
        double* dst = sl < 13 ? outlow + sl * SP_ROWS : out + (sl - 13) * SP_ROWS;
        *reinterpret_cast<e_d2*>(dst + 2 * lane) = w[u];
      }
    }
    for (int e = lane; e < SP_EPAD; e += 64) {
      int s = 0, line = 0, col = 0, cell = 0;
      double val = 0.0;
      if (sp_edge(e, s, line, col, cell) && j0 + line < Gm.m1 && k0 + col < Gm.m2)
        val = ell[ell_base((int64_t)p * Gm.PL + (int64_t)(j0 + line) * Gm.m2 + k0 + col, K) + s * ELL_B];
      out[14 * SP_ROWS + e] = val;
    }
  }
}

// lattice lines of odd length: the lane pair at the line's end holds the last point and a cell outside the lattice, which no row writes
// and the sweep reads as a structurally absent entry -- an explicit zero in all 27 slots of every step of the last patch column
__global__ __launch_bounds__(MFEM_BLOCK) void k_symp_zero_odd(SympGeom Gm, double* __restrict__ pv) {
  const int NP = Gm.NS * Gm.NPk, nplanes = Gm.p1 - Gm.p0;
  const int64_t T = (int64_t)NP * nplanes, cells = (int64_t)nplanes * Gm.NS * SP_L * 27;
  const int col = Gm.m2 - (Gm.NPk - 1) * SP_W;  // first column past the line in the last patch column (odd, < SP_W)
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < cells; t += (int64_t)gridDim.x * blockDim.x) {
    const int s = (int)(t % 27), line = (int)((t / 27) % SP_L);
    const int64_t q = t / (27 * SP_L);
    const int strip = (int)(q % Gm.NS), pl = (int)(q / Gm.NS);
    const int64_t step = (int64_t)pl * NP + (int64_t)strip * Gm.NPk + (Gm.NPk - 1);
    const int idx = line * SP_W + col;
    if (s < 13) pv[T * SP_MAIN + step * SP_LOW + s * SP_ROWS + idx] = 0.0;
    else pv[step * SP_MAIN + (s - 13) * SP_ROWS + idx] = 0.0;
  }
}

// the rows outside [skip_lo, skip_hi) through the plain per-row code on the slot-major copy (chunks that lie inside the range are not
// visited, rows of straddling chunks are masked)
template <bool TRIPLES>
__global__ __launch_bounds__(MFEM_BLOCK) void k_spmv_dia_outside(int64_t n, int64_t npad, int K, const DiaOffsets* __restrict__ Op,
                                                                   const int32_t* __restrict__ flags, const int32_t* __restrict__ cols,
                                                                   const double* __restrict__ vals, const double* __restrict__ x,
                                                                   double* __restrict__ y, double alpha, double beta,
                                                                   const double* __restrict__ dotw, double* __restrict__ partials,
                                                                   const int32_t* __restrict__ done_flag, int64_t skip_lo, int64_t skip_hi) {
  __shared__ double red[16];
  if (done_flag && done_flag[0]) return;
  double dot_acc = 0.0;
  const int64_t R = 2 * MFEM_BLOCK, nchunks = (n + R - 1) / R;
  int64_t cA = (skip_lo + R - 1) / R, cB = skip_hi / R;
  if (cB < cA) cB = cA;
  for (int64_t q = blockIdx.x; q < cA + (nchunks - cB); q += gridDim.x) {
    const int64_t ch = q < cA ? q : cB + (q - cA);
    const int64_t r = ch * R + 2 * (int64_t)threadIdx.x;
    if (r < n) dia_rows<2, 3, TRIPLES>(r, n, npad, K, *Op, flags, cols, vals, x, y, alpha, beta, dotw, 0, dot_acc, skip_lo, skip_hi);
  }
  if (partials) {
    const double b = block_reduce_sum(dot_acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = b;
  }
}

// Decide eligibility and build the column table (once per pattern).  A->max_row_nnz must be known (mfem_csr_plan).
static int ell_plan_body(mfem_context_s* ctx, mfem_csr_s* A);
int mfem_ell_plan(mfem_context_s* ctx, mfem_csr_s* A) {
  // The inspection allocates on the host (std::vector, mfem_host_alloc_probe) after it has begun to record its verdict: an exception on the way
  // must leave the pattern UNPLANNED -- the next solve inspects again -- not half-planned on the slower path for good.  Same for an error return.
  if (A->ell_state != 0) return MFEM_OK;
  struct Undo {
    mfem_csr_s* A;
    bool armed;
    ~Undo() {
      if (!armed) return;
      mfem_ell_free(A);  // (also resets ell_state / dia_state to "not inspected")
      A->sym_state = 0;
      A->symp_state = 0;
    }
  } undo{A, true};
  const int rc = ell_plan_body(ctx, A);
  if (!rc) undo.armed = false;
  return rc;
}
static int ell_plan_body(mfem_context_s* ctx, mfem_csr_s* A) {
  if (A->ell_state != 0) return MFEM_OK;
  if (A->n < g_layout_min_rows_dia && A->n < g_layout_min_rows_cols) return MFEM_OK;  // launch-bound sizes: CSR tile kernel
  A->ell_state = -1;
  const int K = A->max_row_nnz;
  if (A->n < 1 || K < 1 || K > 128) return MFEM_OK;
  const int64_t npad = (A->n + ELL_B - 1) & ~(int64_t)(ELL_B - 1);
  if ((double)K * (double)npad > 1.10 * (double)A->nnz + 64.0 * K) return MFEM_OK;  // > 10 % padding
  MFEM_CHECK_HIP(hipMalloc(&A->ell_cols, sizeof(int32_t) * (size_t)K * (size_t)npad));
  const int grid = mfem_grid_for(npad, MFEM_BLOCK, ctx->num_cus * 16);
  if (A->rowptr_bits == 64)
    hipLaunchKernelGGL(k_ell_cols<int64_t>, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, A->n, npad, K, (const int64_t*)A->rowptr,
                       A->colidx, A->index_base, A->ell_cols);
  else
    hipLaunchKernelGGL(k_ell_cols<int32_t>, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, A->n, npad, K, (const int32_t*)A->rowptr,
                       A->colidx, A->index_base, A->ell_cols);
  MFEM_CHECK_LAUNCH();
  A->ell_K = K;
  A->ell_npad = npad;
  A->ell_state = 1;
  // diagonal structure?  Candidate diagonal lists come from full-length rows sampled in 8 windows along the matrix (a
  // field-major multi-field matrix has one list per field); per-128-row-block flags say which list, if any, a block obeys.
  A->dia_state = -1;
  if (K <= DIA_MAXD) {
    DiaOffsets O;
    memset(&O, 0, sizeof(O));
    for (int wdw = 0; wdw < 8 && O.ncls < DIA_MAXC; ++wdw) {
      const int64_t centre = A->n * (2 * wdw + 1) / 16;
      const int64_t w0 = centre > 1024 ? centre - 1024 : 0;
      const int64_t wn = (A->n - w0) < 2048 ? (A->n - w0) : 2048;  // rows in the window
      if (wn <= 0) continue;
      mfem_host_alloc_probe();
      std::vector<int64_t> win((size_t)wn + 1);
      if (A->rowptr_bits == 64) {
        MFEM_CHECK_HIP(hipMemcpyAsync(win.data(), (const char*)A->rowptr + w0 * 8, (size_t)(wn + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
        MFEM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
      } else {
        std::vector<int32_t> w32((size_t)wn + 1);
        MFEM_CHECK_HIP(hipMemcpyAsync(w32.data(), (const char*)A->rowptr + w0 * 4, (size_t)(wn + 1) * 4, hipMemcpyDeviceToHost, ctx->stream));
        MFEM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        for (int64_t i = 0; i <= wn; ++i) win[(size_t)i] = w32[(size_t)i];
      }
      int64_t rm = -1;
      for (int64_t i = 0; i < wn && rm < 0; ++i)
        if (win[(size_t)i + 1] - win[(size_t)i] == K) rm = i;
      if (rm < 0) continue;
      int32_t cbuf[DIA_MAXD];
      MFEM_CHECK_HIP(hipMemcpyAsync(cbuf, A->colidx + (win[(size_t)rm] - A->index_base), sizeof(int32_t) * K, hipMemcpyDeviceToHost, ctx->stream));
      MFEM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
      int32_t cand[DIA_MAXD];
      for (int i = 0; i < K; ++i) cand[i] = (int32_t)((int64_t)cbuf[i] - A->index_base - (w0 + rm));
      bool seen = false;
      for (int c = 0; c < O.ncls && !seen; ++c) seen = memcmp(O.off[c], cand, sizeof(int32_t) * K) == 0;
      if (!seen) {
        memcpy(O.off[O.ncls], cand, sizeof(int32_t) * K);
        O.D[O.ncls] = K;
        ++O.ncls;
      }
    }
    if (O.ncls > 0) {
      const int64_t nblk = npad / ELL_B;
      const int64_t nx = A->n + (ctx->comm ? 2 * ctx->halo_plane_len * ctx->halo_fields : 0);  // length of the local x
      MFEM_CHECK_HIP(hipMalloc(&A->dia_flags, sizeof(int32_t) * (size_t)nblk));
      MFEM_CHECK_HIP(hipMalloc(&A->dia_dev, sizeof(DiaOffsets)));
      MFEM_CHECK_HIP(hipMemcpyAsync(A->dia_dev, &O, sizeof(DiaOffsets), hipMemcpyHostToDevice, ctx->stream));
      MFEM_CHECK_HIP(hipMemsetAsync(A->dia_flags, 0, sizeof(int32_t) * (size_t)nblk, ctx->stream));
      int32_t* d_cnt = ctx->d_flags + 9;
      MFEM_CHECK_HIP(hipMemsetAsync(d_cnt, 0, sizeof(int32_t), ctx->stream));
      const int g2 = (int)(nblk < (int64_t)ctx->num_cus * 64 ? nblk : (int64_t)ctx->num_cus * 64);
      if (A->rowptr_bits == 64)
        hipLaunchKernelGGL(k_dia_flags<int64_t>, dim3(g2), dim3(128), 0, ctx->stream, A->n, nx, (const int64_t*)A->rowptr, A->colidx,
                           A->index_base, (const DiaOffsets*)A->dia_dev, A->dia_flags, d_cnt);
      else
        hipLaunchKernelGGL(k_dia_flags<int32_t>, dim3(g2), dim3(128), 0, ctx->stream, A->n, nx, (const int32_t*)A->rowptr, A->colidx,
                           A->index_base, (const DiaOffsets*)A->dia_dev, A->dia_flags, d_cnt);
      MFEM_CHECK_LAUNCH();
      MFEM_CHECK_HIP(hipMemcpyAsync(ctx->h_flags + 9, d_cnt, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
      MFEM_CHECK_HIP(hipStreamSynchronize(ctx->stream));  // also orders the H2D copy of the stack object O
      if ((double)ctx->h_flags[9] >= 0.5 * (double)nblk) {  // the other blocks run the explicit-column loop, as in mode 1
        A->dia_state = 1;
        A->dia_classes = O.ncls;
        A->dia_regular_blocks = ctx->h_flags[9];
        A->dia_triples = (K % 3 == 0);
        for (int c = 0; c < O.ncls && A->dia_triples; ++c)
          for (int i = 0; i + 2 < K && A->dia_triples; i += 3)
            if (O.off[c][i + 1] != O.off[c][i] + 1 || O.off[c][i + 2] != O.off[c][i] + 2) A->dia_triples = 0;
        // 27-point lattice stencil with one class: candidate for the symmetric sweep kernel
        A->sym_state = -1;
        int lc = -1;  // the class with the lattice form (a slab has further classes for the rows next to its ghost planes)
        int64_t m2 = 0, PL = 0;
        for (int c = 0; c < O.ncls && lc < 0 && K == 27; ++c) {
          if (O.D[c] != 27 || O.off[c][13] != 0 || O.off[c][14] != 1) continue;
          m2 = O.off[c][16];
          PL = O.off[c][22];
          bool lattice = m2 > 2 && PL > 2 * m2;
          for (int q = 0; q < 27 && lattice; ++q)
            if (O.off[c][q] != (q / 9 - 1) * PL + ((q / 3) % 3 - 1) * m2 + (q % 3 - 1)) lattice = false;
          if (lattice) lc = c;
        }
        if (lc >= 0) {
          const bool lattice = true;
          const int64_t Sc = (PL + SYM_ROWS / 2) / SYM_ROWS;  // chunks per plane, rounded: the tiles drift by PL - Sc * 512 rows per plane
          if (lattice && Sc >= 8 && Sc <= MFEM_MAX_PARTIALS / 2) {  // any drift between tile and plane: the mirrored-fraction rule below decides
            std::vector<int32_t> hf((size_t)nblk);
            MFEM_CHECK_HIP(hipMemcpy(hf.data(), A->dia_flags, sizeof(int32_t) * (size_t)nblk, hipMemcpyDeviceToHost));
            // longest run of regular blocks, cut to whole chunks (4 blocks)
            int64_t best_lo = 0, best_hi = 0, lo = -1;
            for (int64_t b = 0; b <= nblk; ++b) {
              const bool reg = b < nblk && hf[(size_t)b] == lc + 1 && (b + 1) * ELL_B <= A->n;
              if (reg && lo < 0) lo = b;
              if (!reg && lo >= 0) {
                if (b - lo > best_hi - best_lo) { best_lo = lo; best_hi = b; }
                lo = -1;
              }
            }
            const int64_t bpc = SYM_ROWS / ELL_B, c0 = (best_lo + bpc - 1) / bpc, c1 = best_hi / bpc;
            // entries per chunk that k_spmv_sym27 mirrors instead of loading (same lane pattern in every chunk)
            int64_t mx = 0, myz = 0;
            if (c1 - c0 >= 4 * Sc) {
              for (int t = 0; t < SYM_ROWS / 2; ++t) {
                for (int q = 0; q < 9; ++q) {
                  const int64_t lp = 2 * t + Sc * SYM_ROWS + O.off[lc][q];
                  if (lp >= 0 && lp + 1 < SYM_ROWS) mx += 2;
                }
                for (int q = 9; q < 12; ++q) {
                  const int64_t lp = 2 * t + O.off[lc][q];
                  if (lp >= 0 && lp + 1 < SYM_ROWS) myz += 2;
                }
                myz += t > 0 ? 2 : 1;
              }
            }
            // worth it from a quarter of the 13 lower diagonals mirrored (hex-8 256^3: 65 %; 512^3, where a 513-point lattice line is
            // longer than the tile and only the dj = 0 and -z diagonals qualify: 31 %, CG iteration 8.70 -> 7.91 ms)
            if (c1 - c0 >= 4 * Sc && 20 * (mx + myz) >= 5 * 13 * SYM_ROWS) {
              A->sym_state = 1;
              A->sym_c0 = c0;
              A->sym_c1 = c1;
              A->sym_S = (int)Sc;
              A->sym_cls = lc;
              A->sym_mx = mx;
              A->sym_myz = myz;
            }
          }
          // wave-private patch sweep: the lattice planes that lie entirely in the longest run of regular blocks
          A->symp_state = -1;
          if (PL % m2 == 0 && PL / m2 >= 2 && PL < (int64_t)1 << 30) {
            std::vector<int32_t> hf((size_t)nblk);
            MFEM_CHECK_HIP(hipMemcpy(hf.data(), A->dia_flags, sizeof(int32_t) * (size_t)nblk, hipMemcpyDeviceToHost));
            int64_t best_lo = 0, best_hi = 0, lo = -1;
            for (int64_t b = 0; b <= nblk; ++b) {
              const bool reg = b < nblk && hf[(size_t)b] == lc + 1 && (b + 1) * ELL_B <= A->n;
              if (reg && lo < 0) lo = b;
              if (!reg && lo >= 0) {
                if (b - lo > best_hi - best_lo) { best_lo = lo; best_hi = b; }
                lo = -1;
              }
            }
            const int64_t p0 = (best_lo * ELL_B + PL - 1) / PL, p1 = best_hi * ELL_B / PL;
            // a swept row reads x[r - PL - m2 - 1 .. r + PL + m2 + 1]: plane p0 >= 1 and p1 <= (rows / PL) - 1 follow from the
            // regular-block test (r + off in [0, nx) for every row of the block)
            if (p1 - p0 >= 4 && p0 >= 1) {
              A->symp_state = 1;
              A->symp_m2 = (int)m2;
              A->symp_m1 = (int)(PL / m2);
              A->symp_PL = PL;
              A->symp_p0 = (int)p0;
              A->symp_p1 = (int)p1;
              A->symp_NS = (A->symp_m1 + SP_L - 1) / SP_L;
              A->symp_NPk = (A->symp_m2 + SP_W - 1) / SP_W;
              A->sym_cls = lc;
            }
          }
        }
      } else {
        hipFree(A->dia_flags);
        hipFree(A->dia_dev);
        A->dia_flags = nullptr;
        A->dia_dev = nullptr;
      }
    }
  }
  return MFEM_OK;
}

static int sym27_grid(const mfem_context_s* ctx, const mfem_csr_s* A, int64_t* nsteps_out) {
  // 53 KB of LDS per workgroup: three per CU; equal segments for every tile and all workgroups resident in one round
  // (645 workgroups of 51 steps beat 768 of 43 / 51 at 256^3: the longest segment sets the time)
  const int64_t nsteps = (A->sym_c1 - A->sym_c0 + A->sym_S - 1) / A->sym_S;
  const int resident = SYM_WG_PER_CU * ctx->num_cus;
  int nseg = resident / A->sym_S;
  if (nseg < 1) nseg = 1;
  // tiles that cannot fill the resident slots in whole rounds (512^3: 514 tiles on 768 slots) are cut into ~2.7 rounds of shorter
  // segments instead: 8.92 -> 7.91 ms per CG iteration there; at 256^3 (645 of 768) more segments change nothing
  if ((int64_t)A->sym_S * nseg * 10 < (int64_t)resident * 8) nseg = (8 * ctx->num_cus + A->sym_S - 1) / A->sym_S;
  while (nseg > 1 && (int64_t)A->sym_S * nseg > MFEM_MAX_PARTIALS - 512) --nseg;  // one partial sum per workgroup (+ <= 512 of the boundary part of a split SpMV)
  if (nseg > nsteps / 8) nseg = (int)(nsteps / 8);  // a segment's first step has no history: keep segments >= 8 steps long
  if (nseg < 1) nseg = 1;
  if (nsteps_out) *nsteps_out = nsteps;
  return A->sym_S * nseg;
}
static std::atomic<int64_t> g_sym_launches{0};
// the sweep kernel needs ~2 workgroups per CU of >= 8 steps each to beat the plain kernel: chunk ranges below ~2700 chunks (1.4 M rows) stay on the
// plain kernel (mfem_debug_set_layout_min_rows(0, ...) lifts the limit for the parity tests)
static bool sym27_wanted(const mfem_csr_s* A) {
  return A->sym_state == 1 && g_dia_sym && A->dia_triples && (g_layout_min_rows_dia == 0 || A->sym_c1 - A->sym_c0 >= 2700);  // measured crossover between 96^3 and 112^3
}
extern "C" int64_t mfem_debug_sym_spmv_count(void) { return g_sym_launches; }

// the patch sweep is used from the same size on as the workgroup-tile sweep was (launch-bound below; the parity tests lift the limit)
// Which sweep: measured CG iteration, workgroup-tile sweep / patch sweep (tools/probe_sym.py): 128^3 0.142 / 0.173 ms, 192^3 0.369 / 0.397,
// 256^3 0.896 / 0.896, 320^3 1.82 / 1.72, 384^3 3.28 / 2.89, 512^3 7.95 / 6.66 -- the patch sweep from 2.4e7 swept rows on, or where a
// lattice line no longer fits the 512-row tile twice (the parity tests lift all size limits and then always take it)
static bool symp_wanted(const mfem_csr_s* A) {
  if (!(A->symp_state == 1 && g_dia_sym && g_dia_symp && A->dia_triples)) return false;
  if (g_layout_min_rows_dia == 0) return true;
  return (int64_t)(A->symp_p1 - A->symp_p0) * A->symp_PL >= 24000000 || A->symp_m2 > 256;
}
static int symp_nseg(const mfem_context_s* ctx, const mfem_csr_s* A);
bool mfem_symp_wanted(const mfem_csr_s* A) { return symp_wanted(A); }
bool mfem_dia_layout_planned(const mfem_csr_s* A) { return A->ell_state == 1 && g_ell_enable && A->dia_state == 1 && g_dia_enable; }
static SympGeom symp_geom(const mfem_context_s* ctx, const mfem_csr_s* A) {
  SympGeom G;
  G.PL = A->symp_PL;
  G.nx = A->n;  // the sweep stages owned entries of x only (swept rows reference no ghost column)
  G.m1 = A->symp_m1;
  G.m2 = A->symp_m2;
  G.p0 = A->symp_p0;
  G.p1 = A->symp_p1;
  G.NS = A->symp_NS;
  G.NPk = A->symp_NPk;
  G.nseg = symp_nseg(ctx, A);
  return G;
}
static int64_t symp_steps(const mfem_csr_s* A) { return (int64_t)A->symp_NS * A->symp_NPk * (A->symp_p1 - A->symp_p0); }
// runs per patch: the smallest count that fills >= 90 % of the resident one-wave workgroups in whole rounds (a run's first step has no
// history: runs stay >= 16 planes long)
static int symp_nseg(const mfem_context_s* ctx, const mfem_csr_s* A) {
  const int64_t NP = (int64_t)A->symp_NS * A->symp_NPk, slots = (int64_t)SP_WG_PER_CU * ctx->num_cus;
  const int nplanes = A->symp_p1 - A->symp_p0;
  int best = 1;
  double best_eff = 0.0;
  for (int ns = 1; ns <= (nplanes / 16 > 1 ? nplanes / 16 : 1) && ns <= 64; ++ns) {
    const int64_t R = NP * ns, rounds = (R + slots - 1) / slots;
    const double eff = (double)R / (double)(rounds * slots);
    if (eff > best_eff + 1e-9) { best_eff = eff; best = ns; }
    if (eff >= 0.9) { best = ns; break; }
  }
  return best;
}
static int symp_grid(const mfem_context_s* ctx, const mfem_csr_s* A) {
  const int64_t NP = (int64_t)A->symp_NS * A->symp_NPk;
  int64_t g = 8 * ((NP + 7) / 8) * symp_nseg(ctx, A);  // every XCD's share of the runs, padded to the largest share
  int64_t cap = (int64_t)SP_WG_PER_CU * ctx->num_cus;
  if (cap > MFEM_MAX_PARTIALS - 1024) cap = MFEM_MAX_PARTIALS - 1024;
  cap &= ~(int64_t)7;
  if (g > cap) g = cap;
  return g < 8 ? 8 : (int)g;
}
// matrix values (8 B) one sweep SpMV reads from memory with `grid` runs: the 14 upper slots of every valid lane pair and the edge block
// per step + the nine previous-plane slots wherever a run or a patch starts
static int64_t symp_count_entries(const mfem_csr_s* A, int nseg) {
  const int NP = A->symp_NS * A->symp_NPk, nplanes = A->symp_p1 - A->symp_p0;
  int64_t e = 0;
  for (int patch = 0; patch < NP; ++patch) {
    int64_t nv = 0;
    for (int lane = 0; lane < 64; ++lane) {
      const int j = (patch / A->symp_NPk) * SP_L + lane / SP_PW, k = (patch % A->symp_NPk) * SP_W + 2 * (lane % SP_PW);
      if (j < A->symp_m1 && k < A->symp_m2) ++nv;
    }
    e += (28 * nv + SP_NE) * nplanes + 18 * nv * nseg;
  }
  return e;
}

size_t mfem_ell_vals_bytes(const mfem_csr_s* A) {
  if (A->ell_state != 1 || !g_ell_enable) return 0;
  const bool dia = A->dia_state == 1 && g_dia_enable;
  if (A->n < (dia ? g_layout_min_rows_dia : g_layout_min_rows_cols)) return 0;
  size_t bytes = sizeof(double) * (size_t)A->ell_K * (size_t)A->ell_npad;
  if (dia && symp_wanted(A)) bytes += sizeof(double) * SP_STEP * (size_t)symp_steps(A);  // patch-major copy of the swept planes
  return bytes;
}

// Transpose CSR-ordered values into `buf` and route subsequent mfem_spmv_launch calls with these `vals` to the ELL kernel.
// dsc (optional): right Jacobi column scaling applied on the way (copy = vals[j] / dsc[col[j]]): the Krylov loop then runs on the scaled
// matrix without a scaled CSR copy ever existing (solve_inner).  `vals` stays the identity of the bound values.
int mfem_ell_bind(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, double* buf, const double* dsc, const double* ssym) {
  A->ell_vals = nullptr;
  A->ell_src = nullptr;
  A->ell_bound_mode = 0;
  A->sym_bound = 0;
  if (A->ell_state != 1 || !g_ell_enable || !buf) return MFEM_OK;
  if (A->dia_state == 1 && g_dia_enable) {
    const DiaOffsets* O = (const DiaOffsets*)A->dia_dev;
    // a lane per row while at least two waves of 64-row tiles fit 64 KB of staging (K <= 42: the 27-diagonal lattice); two lanes per row,
    // 32-row tiles, beyond (the 81 diagonals of three fields)
    const size_t eb = dsc ? 12 : 8;  // 8 B value (+ 4 B column for the scaling pass) per staged entry (<= rt K per tile)
    const int lpr = eb * 64 * (size_t)A->ell_K * 2 > 64 * 1024 ? 2 : 1;
    const int rt = 64 / lpr;
    int wv = 2;  // two-wave workgroups: what fits a CU is then decided in steps of two waves (27 diagonals: 27.6 KB per workgroup, 5 per CU)
    while (wv > 1 && eb * rt * (size_t)A->ell_K * wv > 64 * 1024) wv >>= 1;
    const size_t ldsb = eb * rt * (size_t)A->ell_K * wv;
    const int64_t nt = A->ell_npad / rt;
    int g = (int)((nt + wv - 1) / wv);
    if (g > ctx->num_cus * 16) g = ctx->num_cus * 16;
    // the slot-major copy; with the patch sweep wanted (and not the two-pass knob) the swept planes go straight to the patch-major copy
    bool fp_made = false;  // the last dia_vals call left the symmetry fingerprint of the swept rows in d_flags[16..17]
    auto dia_vals = [&](const SympGeom& G, double* pvals, bool want_fp = false) -> int {
      fp_made = false;
#define DV_LAUNCH_(RP, LPR_, SYM_, PIPE_)                                                                                                  \
  hipLaunchKernelGGL((k_dia_vals<RP, LPR_, SYM_, PIPE_>), dim3(g), dim3(64 * wv), ldsb, ctx->stream, A->n, A->ell_npad, A->ell_K, (const RP*)A->rowptr, \
                     A->colidx, vals, A->index_base, O, A->dia_flags, buf, G, pvals, dsc, ssym, fast ? 1 : 0)
#define DV_LAUNCH(RP, LPR_, PIPE_) do { if (ssym) DV_LAUNCH_(RP, LPR_, true, PIPE_); else DV_LAUNCH_(RP, LPR_, false, PIPE_); } while (0)
      // (the software-pipelined staging: a lane per row, rows of at most 28 entries, no scaling pass; bit 28 of mfem_debug_set_ell turns it off)
      // the swept rows by k_symp_fill (patch-aligned tiles): a patch-major copy to fill, no column scaling pass, 32-bit row arithmetic, lattice lines and
      // planes long enough for its column decoding
      const bool fast = lpr == 1 && !dsc && pvals && A->n < ((int64_t)1 << 31) && G.PL < ((int64_t)1 << 30) && G.m1 >= 3 && G.m2 >= 3 && g_dia_fast;
      const bool pipe = lpr == 1 && !dsc && A->ell_K <= 28 && g_dia_pipe && !fast;
      if (A->rowptr_bits == 64) {
        if (lpr == 2) DV_LAUNCH(int64_t, 2, false); else if (pipe) DV_LAUNCH(int64_t, 1, true); else DV_LAUNCH(int64_t, 1, false);
      } else {
        if (lpr == 2) DV_LAUNCH(int32_t, 2, false); else if (pipe) DV_LAUNCH(int32_t, 1, true); else DV_LAUNCH(int32_t, 1, false);
      }
#undef DV_LAUNCH_
#undef DV_LAUNCH
      MFEM_CHECK_LAUNCH();
      if (fast) {
        const int64_t ft = (int64_t)(G.p1 - G.p0) * G.NS * G.NPk;  // patch steps, one workgroup of two waves each
        int gf = (int)(ft < (int64_t)ctx->num_cus * 20 ? ft : (int64_t)ctx->num_cus * 20);  // (5 workgroups are resident per CU: four rounds)
        if (gf < 1) gf = 1;
        const size_t lf = sizeof(double) * (2 * (2 * SP_W * 27) + SP_EPAD);
#define SF_LAUNCH(RP, SYM_)                                                                                                                   \
  hipLaunchKernelGGL((k_symp_fill<RP, SYM_>), dim3(gf), dim3(128), lf, ctx->stream, A->n, A->ell_K, (const RP*)A->rowptr, A->colidx, vals, A->index_base, O, \
                     A->sym_cls, buf, G, pvals, ssym, fpr)
        unsigned long long* fpr = want_fp ? (unsigned long long*)(ctx->d_flags + 16) : nullptr;
        if (fpr) MFEM_CHECK_HIP(hipMemsetAsync(fpr, 0, sizeof(unsigned long long), ctx->stream));
        fp_made = fpr != nullptr;
        if (A->rowptr_bits == 64) { if (ssym) SF_LAUNCH(int64_t, true); else SF_LAUNCH(int64_t, false); }
        else { if (ssym) SF_LAUNCH(int32_t, true); else SF_LAUNCH(int32_t, false); }
#undef SF_LAUNCH
        MFEM_CHECK_LAUNCH();
      }
      return MFEM_OK;
    };
    const bool sweep = symp_wanted(A), direct = sweep && g_symp_direct && (g_dia_variant == 0 || g_dia_variant == 7);  // (the other variants read all rows from the slot-major copy)
    double* pvals = buf + (size_t)A->ell_K * (size_t)A->ell_npad;
    SympGeom G{};
    if (sweep) G = symp_geom(ctx, A);
    if (direct && (G.m2 & 1)) {
      const int64_t cells = (int64_t)(G.p1 - G.p0) * G.NS * SP_L * 27;
      hipLaunchKernelGGL(k_symp_zero_odd, dim3(mfem_grid_for(cells, MFEM_BLOCK, ctx->num_cus * 8)), dim3(MFEM_BLOCK), 0, ctx->stream, G, pvals);
      MFEM_CHECK_LAUNCH();
    }
    { const int rt = dia_vals(G, direct ? pvals : nullptr, g_symp_fingerprint != 0); if (rt) return rt; }
    A->ell_vals = buf;
    A->ell_src = vals;
    A->ell_bound_mode = 2;
    A->sym_bound = 0;
    A->symp_bound = 0;
    A->symp_vals = nullptr;
    if (sweep) {  // patch-major copy of the swept planes; are the pairs the sweep mirrors bitwise equal?
      { const int rt = symp_upload_tables(ctx->device); if (rt) return rt; }
      if (!direct) {
        const int64_t T = symp_steps(A);
        const int gb = (int)(T / 4 + 1 < (int64_t)ctx->num_cus * 32 ? T / 4 + 1 : (int64_t)ctx->num_cus * 32);
        hipLaunchKernelGGL(k_symp_bind, dim3(gb), dim3(MFEM_BLOCK), 0, ctx->stream, G, A->ell_K, (const double*)buf, pvals);
        MFEM_CHECK_LAUNCH();
      }
      if (fp_made) {  // the fill summed the fingerprint of the values it wrote (k_symp_fill): zero = symmetric among the swept rows
        MFEM_CHECK_HIP(hipMemcpyAsync(ctx->h_flags + 16, ctx->d_flags + 16, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
        MFEM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        unsigned long long fpv;
        memcpy(&fpv, ctx->h_flags + 16, sizeof(fpv));
        ctx->h_flags[9] = fpv ? 1 : 0;
        ++g_symp_fp_checks;
      } else {
        int32_t* d_bad = ctx->d_flags + 9;
        MFEM_CHECK_HIP(hipMemsetAsync(d_bad, 0, sizeof(int32_t), ctx->stream));
        const int gs = symp_grid(ctx, A);
        hipLaunchKernelGGL(k_spmv_symp<1>, dim3(gs), dim3(64), 0, ctx->stream, G, (const double*)pvals, (const double*)nullptr,
                           (double*)nullptr, 0.0, 0.0, (const double*)nullptr, (double*)nullptr, (const int32_t*)nullptr, d_bad, SympTail{});
        MFEM_CHECK_LAUNCH();
        MFEM_CHECK_HIP(hipMemcpyAsync(ctx->h_flags + 9, d_bad, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        MFEM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
      }
      A->symp_vals = pvals;
      A->symp_bound = ctx->h_flags[9] ? 0 : 1;
      A->symp_pairs = symp_count_entries(A, G.nseg);
      if (A->symp_bound) return MFEM_OK;
      if (direct) {  // not symmetric: the plain kernel serves all rows from the slot-major copy -- the swept rows go there now
        A->symp_vals = nullptr;
        const int rt = dia_vals(SympGeom{}, nullptr);
        if (rt) return rt;
      }
    }
    if (sym27_wanted(A)) {  // are these values bitwise symmetric where the sweep kernel would mirror them?
      int32_t* d_bad = ctx->d_flags + 9;
      MFEM_CHECK_HIP(hipMemsetAsync(d_bad, 0, sizeof(int32_t), ctx->stream));
      hipLaunchKernelGGL(k_sym27_check, dim3(ctx->num_cus * 8), dim3(MFEM_BLOCK), 0, ctx->stream, A->ell_K, O, (const double*)buf,
                         A->sym_c0 * SYM_ROWS, A->sym_c1 * SYM_ROWS, A->sym_cls, d_bad);
      MFEM_CHECK_LAUNCH();
      MFEM_CHECK_HIP(hipMemcpyAsync(ctx->h_flags + 9, d_bad, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
      MFEM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
      A->sym_bound = ctx->h_flags[9] ? 0 : 1;
    }
    return MFEM_OK;
  }
  int waves = 4;
  while (waves > 1 && sizeof(double) * 64 * (size_t)A->ell_K * waves > 64 * 1024) waves >>= 1;
  const size_t lds = sizeof(double) * 64 * (size_t)A->ell_K * waves;
  const int64_t ntiles = A->ell_npad >> 6;
  int grid = (int)((ntiles + waves - 1) / waves);
  if (grid > ctx->num_cus * 16) grid = ctx->num_cus * 16;
  if (A->rowptr_bits == 64)
    hipLaunchKernelGGL(k_ell_vals_lds<int64_t>, dim3(grid), dim3(64 * waves), lds, ctx->stream, A->n, A->ell_npad, A->ell_K,
                       (const int64_t*)A->rowptr, vals, A->index_base, buf, A->colidx, dsc);
  else
    hipLaunchKernelGGL(k_ell_vals_lds<int32_t>, dim3(grid), dim3(64 * waves), lds, ctx->stream, A->n, A->ell_npad, A->ell_K,
                       (const int32_t*)A->rowptr, vals, A->index_base, buf, A->colidx, dsc);
  MFEM_CHECK_LAUNCH();
  A->ell_vals = buf;
  A->ell_src = vals;
  A->ell_bound_mode = 1;
  return MFEM_OK;
}

// d[r] = |A_rr| read from the bound slot-major copy (n values instead of a scan of all nonzeros); rows without a stored
// diagonal keep 1.0 (Jacobi_By_Diagonal, 02_Preconditioner.jl:122-130)
__global__ __launch_bounds__(MFEM_BLOCK) void k_ell_diag(int64_t n, int K, const DiaOffsets* __restrict__ Op,
                                                           const int32_t* __restrict__ flags, const int32_t* __restrict__ cols,
                                                           const double* __restrict__ vals, double* __restrict__ d) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n; r += stride) {
    const int64_t b = ell_base(r, K);
    const int cls = flags ? flags[r >> 7] - 1 : -1;
    double out = 1.0;
    if (cls >= 0) {
      const int D = Op->D[cls];
      for (int s = 0; s < D; ++s)
        if (Op->off[cls][s] == 0) {
          const double v = vals[b + s * ELL_B];
          if (v != 0.0) out = fabs(v);
          break;
        }
    } else {
      for (int s = 0; s < K; ++s)
        if (cols[b + s * ELL_B] == (int32_t)r) {
          const double v = vals[b + s * ELL_B];
          if (v != 0.0) out = fabs(v);  // a padding slot (col = self, value 0) is not a stored diagonal
          break;
        }
    }
    d[r] = out;
  }
}

int mfem_ell_diag(mfem_context_s* ctx, mfem_csr_s* A, double* d) {
  if (!A->ell_vals) return MFEM_ERR_INVALID;
  const int grid = mfem_grid_for(A->n, MFEM_BLOCK, ctx->num_cus * 16);
  hipLaunchKernelGGL(k_ell_diag, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, A->n, A->ell_K, (const DiaOffsets*)A->dia_dev,
                     A->ell_bound_mode == 2 ? A->dia_flags : nullptr, A->ell_cols, A->ell_vals, d);
  MFEM_CHECK_LAUNCH();
  return MFEM_OK;
}

void mfem_ell_unbind(mfem_csr_s* A) {
  A->sym_bound = 0;
  A->symp_bound = 0;
  A->symp_vals = nullptr;
  A->ell_vals = nullptr;
  A->ell_src = nullptr;
  A->ell_bound_mode = 0;
}

void mfem_ell_free(mfem_csr_s* A) {
  if (A->ell_cols) hipFree(A->ell_cols);
  if (A->dia_flags) hipFree(A->dia_flags);
  if (A->dia_dev) hipFree(A->dia_dev);
  A->ell_cols = nullptr;
  A->dia_flags = nullptr;
  A->dia_dev = nullptr;
  A->ell_state = 0;
  A->dia_state = 0;
}

// returns 1 if launched, 0 if the CSR kernel should be used, <0 on error
int mfem_spmv_ell_launch(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, const double* x, double* y, double alpha,
                         double beta, const double* dotw, double* partials, int* n_partials, const int32_t* done_flag,
                         const SpmvPart& part) {
  if (!A->ell_vals || vals != A->ell_src) return 0;
  int cap = ctx->num_cus * g_ell_grid_mult;
  if (cap > MFEM_MAX_PARTIALS) cap = MFEM_MAX_PARTIALS;
  if (part.part != 0) cap /= 2;  // the two parts of a split SpMV share one partial-sum array
  if (part.part == 2) {          // a few planes of rows: no point in a chip-filling persistent grid
    int64_t rows = 0;
    for (int z = 0; z < part.nz; ++z) rows += part.hi[z] - part.lo[z];
    const int64_t want = rows / 512 + 2 * part.nz + 1;
    if (want < cap) cap = (int)want;
  }
  if (A->ell_bound_mode == 2) {
    const DiaOffsets* O = (const DiaOffsets*)A->dia_dev;
    const int drpt = (g_dia_variant == 4 || g_dia_variant == 5) ? 4 : 2;
    const int gdb = mfem_grid_for((A->n + 1) / 2, g_dia_block, cap * MFEM_BLOCK / g_dia_block);
    const int gd = mfem_grid_for((A->n + drpt - 1) / drpt, MFEM_BLOCK, cap);
#define LAUNCH_DIA(RPT, U)                                                                                                \
  hipLaunchKernelGGL((k_spmv_dia<RPT, U>), dim3(gd), dim3(MFEM_BLOCK), 0, ctx->stream, A->n, A->ell_npad, A->ell_K, O,          \
                     A->dia_flags, A->ell_cols, A->ell_vals, x, y, alpha, beta, dotw, partials, done_flag, g_dia_xcd, part)
    switch (g_dia_variant) {
      case 1: LAUNCH_DIA(2, 2); break;
      case 3: LAUNCH_DIA(2, 9); break;
      case 4: LAUNCH_DIA(4, 1); break;
      case 5: LAUNCH_DIA(4, 3); break;
      case 6: LAUNCH_DIA(2, 1); break;
      case 0:
      case 7:
        if (A->symp_bound == 1 && A->symp_vals) {  // (decided when the values were bound: the swept rows may exist in the patch-major copy only)
          // swept planes: wave-private patch sweep on the patch-major copy; the other rows: per-row code on the slot-major copy.
          // part 1 of a split SpMV = the sweep (reads no ghost column), part 2 = the rest
          const SympGeom G = symp_geom(ctx, A);
          const int gs = symp_grid(ctx, A);
          int np = 0;
          const int64_t lo = (int64_t)G.p0 * G.PL, hi = (int64_t)G.p1 * G.PL;
          // unsplit SpMV: the rows outside the swept planes are taken by the sweep's waves after their runs (no second launch, no tail);
          // split SpMV (multi-rank): part 1 = the sweep alone (it reads no ghost column), part 2 = the other rows in a launch of their own
          SympTail tl{};
          if (part.part == 0 && g_symp_tail) {
            tl.on = 1;
            tl.K = A->ell_K;
            tl.n = A->n;
            tl.npad = A->ell_npad;
            tl.lo = lo;
            tl.hi = hi;
            tl.Op = O;
            tl.flags = A->dia_flags;
            tl.cols = A->ell_cols;
            tl.ell = A->ell_vals;
          }
          if (part.part != 2) {
            ++g_sym_launches;
            hipLaunchKernelGGL(k_spmv_symp<0>, dim3(gs), dim3(64), 0, ctx->stream, G, (const double*)A->symp_vals, x, y, alpha, beta, dotw,
                               partials, done_flag, (int32_t*)nullptr, tl);
            MFEM_CHECK_LAUNCH();
            np = gs;
          }
          if (part.part == 2 || (part.part == 0 && !tl.on)) {
            const int64_t outside = (lo + 511) / 512 + (A->n - hi + 511) / 512 + 2;
            const int go = (int)(outside < 1 ? 1 : outside < 1024 ? outside : 1024);
            double* pp = partials ? partials + np : nullptr;
            if (A->dia_triples)
              hipLaunchKernelGGL(k_spmv_dia_outside<true>, dim3(go), dim3(MFEM_BLOCK), 0, ctx->stream, A->n, A->ell_npad, A->ell_K, O,
                                 A->dia_flags, A->ell_cols, A->ell_vals, x, y, alpha, beta, dotw, pp, done_flag, lo, hi);
            else
              hipLaunchKernelGGL(k_spmv_dia_outside<false>, dim3(go), dim3(MFEM_BLOCK), 0, ctx->stream, A->n, A->ell_npad, A->ell_K, O,
                                 A->dia_flags, A->ell_cols, A->ell_vals, x, y, alpha, beta, dotw, pp, done_flag, lo, hi);
            MFEM_CHECK_LAUNCH();
            np += go;
          }
          if (n_partials && partials) *n_partials = np;
          return 1;
        }
        if (g_dia_variant != 8 && sym27_wanted(A) && A->sym_bound == 1 && g_dia_block == MFEM_BLOCK) {
          // rows of the regular chunk range: symmetric sweep kernel; the rest: the plain kernel with that range skipped
          int64_t nsteps = 0;
          const int gs = sym27_grid(ctx, A, &nsteps);
          // part 2 (the chunks outside the sweep, after the halo has arrived): as many workgroups as there are such chunks
          const int64_t outside = (A->n + SYM_ROWS - 1) / SYM_ROWS - (A->sym_c1 - A->sym_c0);
          const int gl = part.part == 2 ? (int)(outside < 1 ? 1 : outside < 512 ? outside : 512) : gs;
          if (part.part != 2) ++g_sym_launches;
          hipLaunchKernelGGL(k_spmv_sym27, dim3(gl), dim3(SYM_THREADS), 0, ctx->stream, A->n, A->ell_npad, A->ell_K, O, A->dia_flags,
                             A->ell_cols, A->ell_vals, x, y, alpha, beta, dotw, partials, done_flag, A->sym_c0, A->sym_c1, A->sym_S,
                             (int)nsteps, A->sym_cls, gs, part.part);
          MFEM_CHECK_LAUNCH();
          if (n_partials && partials) *n_partials = gl;
          return 1;
        }
        if (A->dia_triples && g_dia_variant != 8) {
          hipLaunchKernelGGL((k_spmv_dia<2, 3, true>), dim3(gdb), dim3(g_dia_block), 0, ctx->stream, A->n, A->ell_npad, A->ell_K, O,
                             A->dia_flags, A->ell_cols, A->ell_vals, x, y, alpha, beta, dotw, partials, done_flag, g_dia_xcd, part);
        } else {
          LAUNCH_DIA(2, 3);
        }
        break;
      default: LAUNCH_DIA(2, 3); break;
    }
#undef LAUNCH_DIA
    MFEM_CHECK_LAUNCH();
    if (n_partials && partials) *n_partials = (g_dia_variant == 0 || g_dia_variant == 7) && A->dia_triples ? gdb : gd;
    return 1;
  }
  const int rpt = (g_ell_variant == 0 || g_ell_variant == 2 || g_ell_variant == 4) ? 1 : 2;
  const int grid = mfem_grid_for((A->n + rpt - 1) / rpt, MFEM_BLOCK, cap);
#define LAUNCH_ELL(RPT, U)                                                                                               \
  hipLaunchKernelGGL((k_spmv_ell<RPT, U>), dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, A->n, A->ell_npad, A->ell_K,     \
                     A->ell_cols, A->ell_vals, x, y, alpha, beta, dotw, partials, done_flag, part)
  switch (g_ell_variant) {
    case 1: LAUNCH_ELL(2, 9); break;
    case 2: LAUNCH_ELL(1, 27); break;
    case 3: LAUNCH_ELL(2, 27); break;
    case 4: LAUNCH_ELL(1, 3); break;
    case 5: LAUNCH_ELL(2, 3); break;
    case 6: LAUNCH_ELL(2, 1); break;
    case 7: LAUNCH_ELL(2, 2); break;
    case 8: LAUNCH_ELL(2, 4); break;
    case 9: LAUNCH_ELL(2, 5); break;
    case 10: LAUNCH_ELL(2, 6); break;
    default: LAUNCH_ELL(1, 9); break;
  }
#undef LAUNCH_ELL
  MFEM_CHECK_LAUNCH();
  if (n_partials && partials) *n_partials = grid;
  return 1;
}

// Matrix entries (8-byte values) one SpMV of the planned solver layout reads from memory: K * padded rows for the slot-major
// layouts, less what the symmetric sweep kernel takes from LDS when the bound values are symmetric (*symmetric_sweep = 1 if the
// structure allows that kernel; whether it runs is decided per solve by the bitwise symmetry check of the values).
extern "C" int mfem_csr_solver_layout_entries(mfem_context ctx, mfem_csr A, int64_t* entries, int32_t* symmetric_sweep) try {
  MFEM_REQUIRE(ctx && A, "null argument");
  int32_t mode = 0;
  int rc = mfem_csr_solver_layout(ctx, A, &mode, nullptr, nullptr, nullptr);
  if (rc) return rc;
  int64_t e = A->nnz;
  int sym = 0;
  if (mode == 1 || mode == 2) e = (int64_t)A->ell_K * A->ell_npad;
  if (mode == 3) e = A->sell_total;
  if (mode == 4) {
    e = mfem_lat27_entries(A);
    sym = 3;
  }
  if (mode == 5) {
    e = mfem_lat8_entries(A);
    sym = 3;
  }
  if (mode == 2 && symp_wanted(A)) {
    sym = 2;
    // the rows outside the swept planes read their K slots; the sweep reads symp_pairs value pairs
    const int64_t swept = (int64_t)(A->symp_p1 - A->symp_p0) * A->symp_PL;
    e = (int64_t)A->ell_K * (A->ell_npad - swept) + symp_count_entries(A, symp_nseg(ctx, A));
  } else if (mode == 2 && sym27_wanted(A)) {
    sym = 1;
    const int gs = sym27_grid(ctx, A, nullptr);
    const int64_t nch = A->sym_c1 - A->sym_c0;
    e -= (nch - gs) * A->sym_mx + nch * A->sym_myz;
  }
  if (entries) *entries = e;
  if (symmetric_sweep) *symmetric_sweep = sym;
  return MFEM_OK;
} MFEM_API_CATCH("mfem_csr_solver_layout_entries")

// Bytes one SpMV of the planned solver layout moves by design (bench.py's roofline numerator): matrix entries, 4-byte columns where the
// kernel reads them, x as often as the kernel fetches it from memory by design, y once.
extern "C" int mfem_csr_solver_layout_bytes(mfem_context ctx, mfem_csr A, int64_t* bytes) try {
  MFEM_REQUIRE(ctx && A && bytes, "null argument");
  int32_t mode = 0, slots = 0, sym = 0;
  int64_t npad = 0, reg = 0, ent = 0;
  int rc = mfem_csr_solver_layout(ctx, A, &mode, &slots, &npad, &reg);
  if (rc) return rc;
  rc = mfem_csr_solver_layout_entries(ctx, A, &ent, &sym);
  if (rc) return rc;
  int64_t b = ent * 8 + A->n * 16;
  if (mode == 0) b = A->nnz * 12 + A->n * 16 + (A->n + 1) * (A->rowptr_bits / 8);
  if (mode == 1) b = ent * 12 + A->n * 16;
  if (mode == 3) {  // sliced layout: padded slots; blocks whose 128 rows share one diagonal list read it instead of their column stream
    const double reg = A->sell_nblk > 0 ? (double)A->sell_regular_blocks / (double)A->sell_nblk : 0.0;
    // field-periodic blocks (round 6) read one column slot per node and F values: 1 / F of their column stream
    const double per = (A->sell_nblk > 0 && A->sell_fields > 1) ? (double)A->sell_periodic_blocks / (double)A->sell_nblk : 0.0;
    const double colfrac = (1.0 - reg - per) + (A->sell_fields > 1 ? per / (double)A->sell_fields : 0.0);
    b = A->sell_total * 8 + (int64_t)(colfrac * (double)A->sell_total) * 4 + A->n * 16 + A->n * 4;  // + the row permutation
    if (A->bsell_F > 0) b = A->sell_total * 8 + A->bsell_slots * 4 + A->n * 16 + A->bsell_ncp * 4;  // node-blocked: one column per F x F values
  }
  if (mode == 4) b = mfem_lat27_design_bytes(A);
  if (mode == 5) b = mfem_lat8_design_bytes(A);
  if (mode == 2) {
    b += (A->n > reg ? A->n - reg : 0) * (int64_t)slots * 4;  // rows in generic blocks read their columns
    if (sym == 2) {  // the sweep stages a (4 + 2) x (32 + 2) neighbourhood of x per step instead of reading each swept entry once
      const int64_t swept = (int64_t)(A->symp_p1 - A->symp_p0) * A->symp_PL;
      b += symp_steps(A) * (int64_t)SP_XN * 8 - swept * 8;
    }
  }
  *bytes = b;
  return MFEM_OK;
} MFEM_API_CATCH("mfem_csr_solver_layout_bytes")

// What the Krylov loop of the next mfem_solve will run on this pattern: 0 = CSR tile kernel, 1 = slot-major copy with explicit
// columns, 2 = slot-major copy with diagonal-slotted regular blocks, 3 = row-sorted sliced layout, 4 / 5 = symmetric lattice tiles (one rank;
// the values of each solve decide, modes 3 / 2 serve it otherwise).  Plans what it reports if that has not happened yet.
extern "C" int mfem_csr_solver_layout(mfem_context ctx, mfem_csr A, int32_t* mode, int32_t* slots, int64_t* padded_rows,
                                      int64_t* regular_rows) try {
  MFEM_REQUIRE(ctx && A, "null handle");
  int rc = MFEM_OK, m = 0;
  // the lattice-tile layouts are looked at first: where they apply, the others are planned only on demand (krylov.hip)
  rc = mfem_lat27_plan(ctx, A);
  if (rc) return rc;
  if (mfem_lat27_bytes(A)) m = 4;
  if (m == 0 && mfem_lat8_for_method(A, true)) {  // (asked first, as mfem_solve does: a one-field brick pattern answers for cg! = mode 2, and the plan's
    rc = mfem_lat8_plan(ctx, A);                   //  entry-by-entry check of the pattern costs 27 ms at 512^3 -- VERDICT r4 item 8)
    if (rc) return rc;
    if (mfem_lat8_bytes(A) && mfem_lat8_for_method(A, true)) m = 5;
  }
  if (m == 0) {
    rc = mfem_ell_plan(ctx, A);
    if (rc) return rc;
    if (mfem_ell_vals_bytes(A)) m = (A->dia_state == 1 && g_dia_enable) ? 2 : 1;
    if (m == 0 && !(A->ell_state == 1 && g_ell_enable)) {
      rc = mfem_sell_plan(ctx, A);
      if (rc) return rc;
      if (mfem_sell_vals_bytes(A)) m = 3;
    }
  }
  if (mode) *mode = m;
  if (slots) *slots = (m == 1 || m == 2) ? A->ell_K : m >= 3 ? A->max_row_nnz : 0;
  if (padded_rows) *padded_rows = (m == 1 || m == 2) ? A->ell_npad : m == 3 ? (A->bsell_F > 0 ? A->sell_nblk * 64 * A->bsell_F : A->sell_nblk * 128) : 0;
  if (regular_rows) *regular_rows = m == 2 ? (int64_t)A->dia_regular_blocks * 128 : 0;
  return MFEM_OK;
} MFEM_API_CATCH("mfem_csr_solver_layout")

// y = alpha A x + beta y through the layout mfem_solve would use for this pattern (the one-off conversion of `vals` included):
// a test / diagnostic entry point -- production SpMVs of caller-supplied values go through mfem_spmv_csr.
extern "C" int mfem_spmv_solver_layout(mfem_context ctx, mfem_csr A, const double* vals, const double* x, double* y, double alpha,
                                       double beta) try {
  MFEM_REQUIRE(ctx && A, "null handle");
  MFEM_REQUIRE(A->n == 0 || (x && y && (A->nnz == 0 || vals)), "null vector");
  if (A->n == 0) return MFEM_OK;
  int rc = MFEM_OK;
  bool bound = false;
  {  // lattice tiles, if the structure allows them and these values are symmetric
    rc = mfem_lat27_plan(ctx, A);
    if (rc) return rc;
    size_t lb = mfem_lat27_bytes(A);
    const bool is27 = lb != 0;
    if (!lb && mfem_lat8_for_method(A, true)) {
      rc = mfem_lat8_plan(ctx, A);
      if (rc) return rc;
      if (mfem_lat8_for_method(A, true)) lb = mfem_lat8_bytes(A);
    }
    if (lb) {
      const size_t lay = (lb + 255) & ~(size_t)255;
      rc = mfem_ws_reserve(ctx, lay + (2 * (size_t)A->n + (size_t)(A->ncols > A->n ? A->ncols : A->n)) * sizeof(double));
      if (rc) return rc;
      double* scratch = (double*)((char*)ctx->ws + lay);
      rc = is27 ? mfem_lat27_bind(ctx, A, vals, (double*)ctx->ws, nullptr, scratch, mfem_rem_diag())
                : mfem_lat8_bind(ctx, A, vals, (double*)ctx->ws, nullptr, scratch, mfem_rem_diag());
      if (rc) return rc;
      bound = mfem_lat27_bound(A, vals) || mfem_lat8_bound(A, vals);
    }
  }
  if (!bound) {
    rc = mfem_ell_plan(ctx, A);
    if (rc) return rc;
    const size_t bytes = mfem_ell_vals_bytes(A);
    if (bytes) {
      rc = mfem_ws_reserve(ctx, bytes);
      if (rc) return rc;
      rc = mfem_ell_bind(ctx, A, vals, (double*)ctx->ws, nullptr, nullptr);
      if (rc) return rc;
    } else {
      rc = mfem_sell_plan(ctx, A);
      if (rc) return rc;
      const size_t sb = mfem_sell_vals_bytes(A);
      if (sb) {
        rc = mfem_ws_reserve(ctx, sb);
        if (rc) return rc;
        rc = mfem_sell_bind(ctx, A, vals, (double*)ctx->ws, nullptr);
        if (rc) return rc;
      }
    }
  }
  rc = mfem_spmv_launch(ctx, A, vals, x, y, alpha, beta, nullptr, nullptr, nullptr, nullptr);
  mfem_ell_unbind(A);
  mfem_sell_unbind(A);
  mfem_lat27_unbind(A);
  mfem_lat8_unbind(A);
  return rc;
} MFEM_API_CATCH("mfem_spmv_solver_layout")
