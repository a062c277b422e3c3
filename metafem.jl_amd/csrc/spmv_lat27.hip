// Symmetric lattice-tile layout for the Krylov loop on the hex-27 (order-2 Lagrange, one field) lattice matrix: solver layout mode 4.
// The caller-facing contract stays CSR (mul!, misc/04_GPU_Utils.jl:131; iterative_Solve!, solver/03_Iterative_Solvers.jl:31-49).
//
// Why: the row-sorted sliced layout (spmv_sell.hip) streams all 64 entries of an average hex-27 row and gathers x from global memory
// (10 L1 accesses per value line, 1.33 x the design bytes from HBM: profiles/r03_sell_memory_path.txt).  A hex-27 stiffness matrix
// that CG may be run on is symmetric, and its pattern is a lattice stencil whose reach depends on the parity of the node in each
// direction (even = element boundary: offsets -2..2, odd = element interior: -1..1; 8 node types, 27..125 entries per row).
//
//   * layout (per solve, one pass over the CSR values): only the diagonal and the entries with column > row are stored -- 14..63 of a
//     row's 27..125.  Rows are grouped in units of 4 x 4 x 8 lattice points = 16 rows of each of the 8 types; a wave owns a unit, four
//     lanes share a row and take every fourth stored entry, so a unit is 68 wave-wide 16-byte-per-lane... (8 bytes per lane per step,
//     two steps per 16-byte load) unit-stride steps: 34.8 KB instead of the 65.5 KB of its 8 192 entries.  No column stream: the column of
//     a slot is the row's lattice position plus a per-type table entry.
//   * SpMV, pass 1 (k_spmv_lat27): a workgroup owns a tile of 8 x 8 x 32 lattice points (16 units, 8 waves).  It stages x of the tile and
//     of the (+2, +-2, +-2) neighbourhood its stored entries reach in LDS (4 320 cells), and accumulates y in a second LDS block of
//     the same shape: for a stored entry a = A[r][c] the lane adds a x[c] to its register sum for row r and a x[r] to cell c
//     (ds_add_f64) -- the mirrored entry A[c][r] is never read.  The whole y block -- own cells and neighbourhood -- leaves as one
//     contiguous 34.6 KB run per tile.
//   * pass 2 (k_lat27_gather): row r sums the up to 18 tiles whose block covers it, in a fixed order, applies alpha / beta and the fused
//     dot product.  No global atomics; 2.1 cells per row written and read again (+ 12 % traffic).
//   * eligibility is decided in two steps: the pattern must BE the lattice stencil (checked entry by entry once per pattern), and the values
//     of this solve must be symmetric: measured per bind with a probe product (mfem_sym_probe below: the layout against the CSR kernel on one
//     vector); the sliced layout serves the solve when a row of the two products differs by more than 4e-13 of that row's diagonal entry.  A right Jacobi scaling (bicgstabl_GS!, idrs!, cgs2! work on A D^-1, which is not
//     symmetric) is applied to x while it is staged: (A D^-1) x = A (x / d), so the stored matrix stays the symmetric A.
//   * y differs from the CSR kernel's by round-off (other summation order), and the order in which the waves of a workgroup add into
//     an LDS cell is not fixed: results are reproducible to ~1e-16 relative, not bitwise (mfem_debug_set_lat27(0) selects the sliced layout).
#include "blas1.h"
#include "spmv_lat_tables.h"
#include "krylov.h"  // scalar / flag slots of the Krylov loop (the fused CG update below)

#define L27_TI 8
#define L27_TJ 8
#define L27_TK 32
#define L27_SJ (L27_TJ + 4)
#define L27_SK (L27_TK + 4)
#define L27_CELLS ((L27_TI + 2) * L27_SJ * L27_SK)  // 4320
#define L27_PI (L27_SJ * L27_SK + 8)                 // plane stride of the LDS blocks: 440 = 8 mod 16, so the 16 rows of a step (2 a PI + 2 b SK + 2 c) fall on 16 different bank pairs
#define L27_LDS_CELLS ((L27_TI + 2) * L27_PI)
#define L27_UNIT_D 4352                             // doubles per unit: 64 lanes x 68 steps

typedef double l_d2 __attribute__((ext_vector_type(2)));

extern std::atomic<int64_t> g_layout_min_rows_lat27;  // spmv_ell.hip
static std::atomic<int> g_lat27_enable{1};
static std::atomic<int> g_lat27_det{1};  // bit 3 of mfem_debug_set_lat27: 0 = pass 1 by the four-lanes-per-row kernel (not bitwise reproducible), 1 (default) = lane = row, phase-major
static std::atomic<int> g_lat27_cg_fused{1};  // bit 2 of mfem_debug_set_lat27: 0 = CG iterations as SpMV (pass 1 + pass 2) + k_cg_update instead of pass 1 + k_lat27_gather_cg
static std::atomic<int> g_lat27_gather_staged{1};  // bit 1 of mfem_debug_set_lat27: 0 = pass 2 by k_lat27_gather (masked blocks, a round trip per covering block)
static std::atomic<long long> g_lat27_count{0};
extern "C" long long mfem_debug_lat27_spmv_count(void) { return g_lat27_count; }  // SpMVs the layout has served (bench.py: which kernel ran)
// max |A[r][c] - A[c][r]| / max |A[r][c]| the layout pass of the last bind on this pattern measured (-1: no bind yet)
extern "C" double mfem_debug_lat27_asymmetry(mfem_csr A) { return A ? A->lat27_asym : -1.0; }
extern "C" int mfem_debug_set_lat27(int enable) try {
  ++mfem_debug_epoch;
  g_lat27_enable = enable & 1;
  g_lat27_gather_staged = ((enable >> 1) & 1) ? 0 : 1;
  g_lat27_cg_fused = ((enable >> 2) & 1) ? 0 : 1;
  g_lat27_det = ((enable >> 3) & 1) ? 0 : 1;
  return MFEM_OK;
} MFEM_API_CATCH("mfem_debug_set_lat27")

struct Lat27Geom {
  int m0, m1, m2;     // OWNED lattice points per direction (m0 = owned planes of a slab; m1, m2 odd)
  int nui, nuj, nuk;  // units of 4 x 4 x 8 points
  int nti, ntj, ntk;  // tiles of 8 x 8 x 32 points
  int64_t n;          // m0 * m1 * m2 owned rows
  // slab: the owned planes are [plo, plo + m0) (plo even: slabs are cut on element boundaries) of a lattice of mg planes; x carries, behind the n
  // owned entries, a low and a high block of gw = 2 ghost planes (brick_xindex); plo = 0, mg = m0 for a whole brick
  int plo, mg, gw;
};
// local x index at GLOBAL plane gi (owned or ghost), in-plane position ip
__device__ __forceinline__ int64_t l27_xindex(const Lat27Geom& G, int gi, int64_t ip) {
  const int64_t PL = (int64_t)G.m1 * G.m2;
  if (gi >= G.plo && gi < G.plo + G.m0) return (int64_t)(gi - G.plo) * PL + ip;
  const int side = gi < G.plo ? 0 : 1;
  const int off = side ? gi - (G.plo + G.m0) : gi - (G.plo - G.gw);
  return G.n + ((int64_t)side * G.gw + off) * PL + ip;
}

// type t = 4 (i odd) + 2 (j odd) + (k odd); steps per lane K4, stored slots Kup, group base inside a unit (doubles), table base (entries)
__constant__ int c_l27_Kup[8];
__constant__ int16_t c_l27_off[L27_TAB];   // [tb[t] + q * K4 + it]: LDS cell offset of slot it * 4 + q (0 for padding)
__constant__ int8_t c_l27_d[L27_TAB][4];   // same index: (di, dj, dk) of the slot; di = L27_PAD for padding
__constant__ int c_l27_K4[8];
__constant__ int c_l27_gb[8];
__constant__ int c_l27_tb[8];
static std::atomic<bool> g_l27_tables{false};

static int lat27_upload_tables() {
  static std::mutex mu;  // uploads from two host threads must not interleave (the tables themselves are process-wide: see the threading note in include/metafem_mi355x.h)
  std::lock_guard<std::mutex> lk(mu);
  if (g_l27_tables) return MFEM_OK;
  int Kup[8];
  int16_t off[L27_TAB];
  int8_t d[L27_TAB][4];
  if (!l27_build_tables(d, Kup)) return MFEM_ERR_INVALID;  // (spmv_lat_tables.h)
  for (int i = 0; i < L27_TAB; ++i) off[i] = d[i][0] == L27_PAD ? (int16_t)0 : (int16_t)(d[i][0] * L27_PI + d[i][1] * L27_SK + d[i][2]);
  if (hipMemcpyToSymbol(HIP_SYMBOL(c_l27_Kup), Kup, sizeof(Kup)) != hipSuccess ||
      hipMemcpyToSymbol(HIP_SYMBOL(c_l27_off), off, sizeof(off)) != hipSuccess ||
      hipMemcpyToSymbol(HIP_SYMBOL(c_l27_d), d, sizeof(d)) != hipSuccess ||
      hipMemcpyToSymbol(HIP_SYMBOL(c_l27_K4), l27_K4, sizeof(l27_K4)) != hipSuccess ||
      hipMemcpyToSymbol(HIP_SYMBOL(c_l27_gb), l27_gb, sizeof(l27_gb)) != hipSuccess ||
      hipMemcpyToSymbol(HIP_SYMBOL(c_l27_tb), l27_tb, sizeof(l27_tb)) != hipSuccess) {
    mfem_set_error("lattice-tile tables: hipMemcpyToSymbol failed");
    return MFEM_ERR_HIP;
  }
  g_l27_tables = true;
  return MFEM_OK;
}

// offsets a row at lattice coordinate g (of m points) has along one direction: [lo, lo + cnt)
__device__ __forceinline__ void l27_range(int g, int m, int& lo, int& cnt) {
  if (g & 1) {
    lo = -1;
    cnt = 3;
  } else {
    lo = g >= 2 ? -2 : -g;
    const int hi = (m - 1 - g) >= 2 ? 2 : (m - 1 - g);
    cnt = hi - lo + 1;
  }
}

// 1 in *bad if some row is not the lattice stencil row: length = product of the per-direction ranges, columns in lexicographic order
template <typename RP>
__global__ __launch_bounds__(MFEM_BLOCK) void k_l27_verify(Lat27Geom G, const RP* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                             int base, int32_t* __restrict__ bad) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t PL = (int64_t)G.m1 * G.m2;
  int fail = 0;
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < G.n; r += stride) {
    const int gi = (int)(r / PL) + G.plo;  // global plane
    const int64_t rem = r % PL;
    const int gj = (int)(rem / G.m2), gk = (int)(rem - (int64_t)gj * G.m2);
    int li, ni, lj, nj, lk, nk;
    l27_range(gi, G.mg, li, ni);
    l27_range(gj, G.m1, lj, nj);
    l27_range(gk, G.m2, lk, nk);
    const int64_t lo = (int64_t)rowptr[r] - base, hi = (int64_t)rowptr[r + 1] - base;
    if (hi - lo != (int64_t)ni * nj * nk) {
      fail = 1;
      continue;
    }
    int64_t j = lo;
    for (int a = 0; a < ni; ++a)
      for (int b = 0; b < nj; ++b) {
        const int64_t c0 = l27_xindex(G, gi + li + a, (int64_t)(gj + lj + b) * G.m2 + gk + lk);
        for (int c = 0; c < nk; ++c, ++j)
          if ((int64_t)col[j] - base != c0 + c) fail = 1;
      }
  }
  if (fail) bad[0] = 1;
}

// The layout pass: a wave per unit.  stats[1] = max |A[r][c]| over the stored entries (bit pattern of a non-negative double, which orders like
// an integer).
template <typename RP>
__global__ __launch_bounds__(MFEM_BLOCK) void k_l27_fill(Lat27Geom G, const RP* __restrict__ rowptr, int base,
                                                           const double* __restrict__ vals, double* __restrict__ out,
                                                           unsigned long long* __restrict__ stats) {
  const int lane = threadIdx.x & 63, q = lane & 3, rho = lane >> 2;
  const int ra = rho >> 3, rb = (rho >> 2) & 1, rc = rho & 3;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int64_t nunits = (int64_t)G.nui * G.nuj * G.nuk;
  double amax = 0.0;
  for (int64_t u = wave; u < nunits; u += nwaves) {
    const int uk = (int)(u % G.nuk);
    const int64_t u2 = u / G.nuk;
    const int uj = (int)(u2 % G.nuj), ui = (int)(u2 / G.nuj);
    double* ou = out + u * L27_UNIT_D;
    for (int t = 0; t < 8; ++t) {
      const int oi = ui * 4 + ((t >> 2) & 1) + 2 * ra, gj = uj * 4 + ((t >> 1) & 1) + 2 * rb, gk = uk * 8 + (t & 1) + 2 * rc;  // oi: owned plane
      const int gi = oi + G.plo;
      const bool valid = oi < G.m0 && gj < G.m1 && gk < G.m2;
      const int64_t r = ((int64_t)oi * G.m1 + gj) * G.m2 + gk;
      int li = 0, ni = 1, lj = 0, nj = 1, lk = 0, nk = 1;
      int64_t rp = 0;
      if (valid) {
        l27_range(gi, G.mg, li, ni);
        l27_range(gj, G.m1, lj, nj);
        l27_range(gk, G.m2, lk, nk);
        rp = (int64_t)rowptr[r] - base;
      }
      const int K4 = c_l27_K4[t], tb = c_l27_tb[t] + q * K4;
      double* og = ou + c_l27_gb[t] + lane * 2;
      for (int it = 0; it < K4; it += 2) {
        l_d2 pr;
        pr.x = 0.0;
        pr.y = 0.0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int di = c_l27_d[tb + it + h][0], dj = c_l27_d[tb + it + h][1], dk = c_l27_d[tb + it + h][2];
          const int ci = gi + di, cj = gj + dj, ck = gk + dk;
          if (valid && di != L27_PAD && ci < G.mg && cj >= 0 && cj < G.m1 && ck >= 0 && ck < G.m2) {  // (ci may be a ghost plane of a slab)
            const double v = vals[rp + ((int64_t)(di - li) * nj + (dj - lj)) * nk + (dk - lk)];
            double av = fabs(v);
            if (!(av == av)) av = __builtin_huge_val();  // NaN: fmax would drop it
            amax = fmax(amax, av);
            if (h) pr.y = v; else pr.x = v;
          }
        }
        *(l_d2*)(og + (int64_t)(it >> 1) * 128) = pr;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amax = fmax(amax, __shfl_down(amax, o, MFEM_WAVE));
  if (lane == 0) atomicMax(stats + 1, (unsigned long long)__double_as_longlong(amax));
}

// ---- the symmetry measure shared by the lattice-tile layouts (modes 4 and 5) ---------------------------------------------------------
// The layout stores one triangle and mirrors it; whether that is the caller's matrix is measured with a probe product: x with entries of magnitude
// in [0.75, 1.25) and a random SIGN each (zero mean: a skew part with zero row sums -- convection-like terms -- is not attenuated the way a
// nearly constant probe would), y1 = (layout) x, y2 = (CSR kernel on the caller's values) x.  y1 - y2 = (L - U^T) x: an entry pair that differs by
// delta shows up as >= 0.75 |delta| in its row (the other terms of that row are the other pairs' differences: no cancellation for a generic x).
// The two products round differently (a few 1e-15 of the row's entries for rows of up to 125 entries), so the layout is taken when
//     max over rows r of |y1 - y2|_r / |a_rr|  <=  4e-13
// -- the difference is weighed PER ROW by that row's diagonal entry (badly scaled matrices: a penalty or Robin row of 1e5 no longer hides an
// asymmetric pair in a row of 1e-3); rows without a stored non-zero diagonal are weighed by the global max |a|.
__global__ __launch_bounds__(MFEM_BLOCK) void k_probe_vector(int64_t n, double* __restrict__ x) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t z = (uint64_t)i + 0x9E3779B97F4A7C15ull;  // splitmix64
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const double mag = 0.75 + 0.5 * (double)(z >> 11) * (1.0 / 9007199254740992.0);
    x[i] = (z & 1ull) ? mag : -mag;
  }
}
__global__ __launch_bounds__(MFEM_BLOCK) void k_probe_diff(int64_t n, const double* __restrict__ a, const double* __restrict__ b,
                                                             const double* __restrict__ scale, unsigned long long* __restrict__ out) {
  double d = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double e = fabs(a[i] - b[i]) / scale[i];  // (scale > 0: |diagonal| or the preset max |a|)
    if (!(e == e)) e = __builtin_huge_val();
    d = fmax(d, e);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) d = fmax(d, __shfl_down(d, o, MFEM_WAVE));
  if ((threadIdx.x & 63) == 0) atomicMax(out, (unsigned long long)__double_as_longlong(d));
}

// scratch: ncols + 2 n doubles (x carries the ghost entries of a slab pattern).  The layout must be bound for `vals` with no column scaling; unbind() must leave the pattern without any bound layout
// (the second product then runs the CSR kernel).  *asym = max over rows of |y1 - y2|_r / |a_rr| (rows without a non-zero diagonal: / amax).
int mfem_sym_probe(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, double* scratch, double amax, void (*unbind)(mfem_csr_s*),
                   void (*rebind)(mfem_csr_s*, void*), void* cookie, double* asym, int rem_fields) {
  A->rem_active = 0;
  A->rem_asym_before = 0.0;
  A->rem_last_rows = A->rem_last_ent = 0;
  const int64_t n = A->n, nc = A->ncols > n ? A->ncols : n;
  double *x = scratch, *y1 = scratch + nc, *y2 = y1 + n;
  unsigned long long* d_stat = (unsigned long long*)(ctx->d_flags + 12);
  MFEM_CHECK_HIP(hipMemsetAsync(d_stat, 0, sizeof(unsigned long long), ctx->stream));
  const int prof = ctx->prof_on;
  ctx->prof_on = 0;  // (not SpMVs of the solve: bench.py's per-launch timing must not see them)
  ctx->probe_active = 1;
  hipLaunchKernelGGL(k_probe_vector, dim3(mfem_vec_grid(ctx, nc)), dim3(MFEM_BLOCK), 0, ctx->stream, nc, x);
  int rc = mfem_spmv_launch(ctx, A, vals, x, y1, 1.0, 0.0, nullptr, nullptr, nullptr);
  if (!rc) {
    unbind(A);
    rc = mfem_spmv_launch(ctx, A, vals, x, y2, 1.0, 0.0, nullptr, nullptr, nullptr);
    rebind(A, cookie);
  }
  ctx->prof_on = prof;
  ctx->probe_active = 0;
  if (rc) return rc;
  const bool finite = amax < __builtin_huge_val() && amax == amax;
  if (!(amax > 0.0) || !finite) {  // an all-zero matrix is symmetric; a non-finite one is not taken
    *asym = finite ? 0.0 : 1.0;
    return MFEM_OK;
  }
  // the rows' weights into x (the probe vector has served): |a_rr|, preset max |a| where no non-zero diagonal is stored
  int mfem_fill(mfem_context_s* ctx, int64_t n, double v, double* x);  // (krylov.hip)
  rc = mfem_fill(ctx, n, amax, x);
  if (!rc) rc = mfem_jacobi_diag_launch(ctx, A, vals, x, 0);
  if (rc) return rc;
  hipLaunchKernelGGL(k_probe_diff, dim3(mfem_vec_grid(ctx, n)), dim3(MFEM_BLOCK), 0, ctx->stream, n, y1, y2, x, d_stat);
  MFEM_CHECK_LAUNCH();
  MFEM_CHECK_HIP(hipMemcpyAsync(ctx->h_flags + 12, d_stat, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
  MFEM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  double dmax;
  memcpy(&dmax, ctx->h_flags + 12, sizeof(double));
  *asym = dmax;  // already relative: max_r |y1 - y2|_r / |a_rr|
  A->rem_asym_before = dmax;
  if (dmax <= 4e-13 || rem_fields <= 0 || !(dmax < __builtin_huge_val())) return MFEM_OK;
  // A = S + N (spmv_rem.hip): the rows above the gate get a remainder N[r][c] = A[r][c] - A[c][r] on their mirrored entries; accepted when the SAME
  // probe passes on S + N.  (x holds the rows' weights now, y1 / y2 the two products.)
  bool built = false;
  rc = mfem_rem_build(ctx, A, vals, rem_fields, y1, y2, x, 4e-13, &built);
  if (rc || !built) return rc;
  hipLaunchKernelGGL(k_probe_vector, dim3(mfem_vec_grid(ctx, nc)), dim3(MFEM_BLOCK), 0, ctx->stream, nc, x);  // the probe vector again (the weights took its place)
  MFEM_CHECK_LAUNCH();
  ctx->probe_active = 1;
  rc = mfem_rem_apply(ctx, A, x, nullptr, y1, 1.0, nullptr, nullptr, nullptr, nullptr);
  ctx->probe_active = 0;
  if (!rc) rc = mfem_fill(ctx, n, amax, x);
  if (!rc) rc = mfem_jacobi_diag_launch(ctx, A, vals, x, 0);
  if (rc) return rc;
  MFEM_CHECK_HIP(hipMemsetAsync(d_stat, 0, sizeof(unsigned long long), ctx->stream));
  hipLaunchKernelGGL(k_probe_diff, dim3(mfem_vec_grid(ctx, n)), dim3(MFEM_BLOCK), 0, ctx->stream, n, y1, y2, x, d_stat);
  MFEM_CHECK_LAUNCH();
  MFEM_CHECK_HIP(hipMemcpyAsync(ctx->h_flags + 12, d_stat, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
  MFEM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  double dmax2;
  memcpy(&dmax2, ctx->h_flags + 12, sizeof(double));
  if (dmax2 <= 4e-13) {
    *asym = dmax2;
    A->rem_active = 1;
    A->rem_last_rows = A->rem_nrows;
    A->rem_last_ent = A->rem_nent;
  }
  return MFEM_OK;
}

// A unit's 68 steps run as 5 chunks of 16 / 10 / 10 / 16 / 16 steps (type 0; type 1; type 2; types 3 + 4; types 5 + 6 + 7: neighbours in the unit's
// storage); the values of chunk k + 1 are in flight while chunk k is worked through LDS (two register buffers that swap roles; 9 smaller chunks left
// 32 - 64 bytes per lane in flight, this keeps 80 - 128).
template <int N>
__device__ __forceinline__ void l27_load(l_d2 (&v)[8], const double* __restrict__ gv) {
#pragma unroll
  for (int u = 0; u < N / 2; ++u) v[u] = __builtin_nontemporal_load((const l_d2*)gv + u * 64);
}

// N steps of one type from v[OFF ...] (OFF in 16-byte pairs)
template <int N, int OFF>
__device__ __forceinline__ void l27_proc(const l_d2 (&v)[8], const uint32_t* __restrict__ tq, int pos, int q, const double* xs, double* ys) {
  uint32_t w[N / 2];
#pragma unroll
  for (int u = 0; u < N / 2; ++u) w[u] = tq[u];
  const double xr = xs[pos];
  double acc = 0.0;
#pragma unroll
  for (int it = 0; it < N; ++it) {
    const int o = (it & 1) ? ((int)w[it >> 1] >> 16) : (int)(int16_t)(w[it >> 1] & 0xffffu);
    const double a = (it & 1) ? v[OFF + (it >> 1)].y : v[OFF + (it >> 1)].x;
    acc += a * xs[pos + o];
    double m = a * xr;
    if (it == 0) m = q == 0 ? 0.0 : m;  // slot 0 is the diagonal: nothing to mirror
    __builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double*)(ys + pos + o), m);
  }
  acc += __shfl_xor(acc, 1, MFEM_WAVE);
  acc += __shfl_xor(acc, 2, MFEM_WAVE);
  if (q == 0) __builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double*)(ys + pos), acc);
}

// on entry A holds the unit's first chunk; on exit B holds the first chunk of the unit at uv_next (if any)
__device__ __forceinline__ void l27_unit(l_d2 (&A)[8], l_d2 (&B)[8], const double* __restrict__ uv, const double* __restrict__ uv_next,
                                         int p0, int q, const uint32_t* __restrict__ tabs, const double* xs, double* ys) {
  const int PI = L27_PI, PJ = L27_SK;
  l27_load<10>(B, uv + 1024);
  l27_proc<16, 0>(A, tabs + q * 8, p0, q, xs, ys);                     // type 0
  l27_load<10>(A, uv + 1664);
  l27_proc<10, 0>(B, tabs + 32 + q * 5, p0 + 1, q, xs, ys);            // type 1 (k odd)
  l27_load<16>(B, uv + 2304);
  l27_proc<10, 0>(A, tabs + 52 + q * 5, p0 + PJ, q, xs, ys);           // type 2 (j odd)
  l27_load<16>(A, uv + 3328);
  l27_proc<6, 0>(B, tabs + 72 + q * 3, p0 + PJ + 1, q, xs, ys);        // type 3
  l27_proc<10, 3>(B, tabs + 84 + q * 5, p0 + PI, q, xs, ys);           // type 4 (i odd)
  if (uv_next) l27_load<16>(B, uv_next);
  l27_proc<6, 0>(A, tabs + 104 + q * 3, p0 + PI + 1, q, xs, ys);       // type 5
  l27_proc<6, 3>(A, tabs + 116 + q * 3, p0 + PI + PJ, q, xs, ys);      // type 6
  l27_proc<4, 6>(A, tabs + 128 + q * 2, p0 + PI + PJ + 1, q, xs, ys);  // type 7
}

// pass 1: one workgroup per tile.  dump[tile][cell] = what the tile's stored entries contribute to y on its own cells and on the
// (+2, +-2, +-2) neighbourhood.
// dsc != nullptr: the operator is A D^-1 (right Jacobi scaling, Mat_Div_Jacobi of 02_Preconditioner.jl:141-148): the stored matrix stays the
// symmetric A and x is divided by d while it is staged.
// dotp != nullptr (the fused CG iteration, no column scaling): dotp[tile] = sum over the tile's block of x(cell) * y-contribution(cell); summed over the tiles that is
// x . A x -- every contribution to y[r] sits in exactly one cell of one block, beside the x[r] the tile staged -- so the dot product of a CG iteration needs no pass 2.
__global__ __launch_bounds__(512, 4) void k_spmv_lat27(Lat27Geom G, const double* __restrict__ vals, const double* __restrict__ x,
                                                       const double* __restrict__ dsc, double* __restrict__ dump,
                                                       const int32_t* __restrict__ done_flag, int tile0, int tcount, double* __restrict__ dotp) {
  __shared__ double xs[L27_LDS_CELLS];
  __shared__ double ys[L27_LDS_CELLS];
  __shared__ uint32_t tabs[L27_TAB / 2];
  __shared__ double dred[8];
  if (done_flag && done_flag[0]) return;
  // workgroups with equal blockIdx % 8 share an XCD (round-robin dispatch): each XCD walks a contiguous eighth of the tiles, so the
  // neighbourhoods that overlap are staged through one L2
  // (this launch covers the tiles [tile0, tile0 + tcount) of the i-major tile list: all of them, or the interior / boundary part of a slab's SpMV)
  const int chunk = (tcount + 7) >> 3;
  const int tsub = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= chunk || tsub >= tcount) return;
  const int tile = tile0 + tsub;
  const int tk = tile % G.ntk, t2 = tile / G.ntk, tj = t2 % G.ntj, ti = t2 / G.ntj;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6, q = lane & 3, rho = lane >> 2;
  const int ra = rho >> 3, rb = (rho >> 2) & 1, rc = rho & 3;
  // the wave's two units: (0, b, c) and (1, b, c) of the tile's 2 x 2 x 4; the second exists only if the first does
  const int ub = (wv >> 2) & 1, uc = wv & 3;
  const int ui = ti * 2, uj = tj * 2 + ub, uk = tk * 4 + uc;
  const bool e0 = uj < G.nuj && uk < G.nuk, e1 = e0 && ui + 1 < G.nui;
  const double* uv0 = vals + (((int64_t)ui * G.nuj + uj) * G.nuk + uk) * L27_UNIT_D + lane * 2;
  const double* uv1 = uv0 + (int64_t)G.nuj * G.nuk * L27_UNIT_D;
  l_d2 A[8], B[8];
  if (e0) l27_load<16>(A, uv0);  // in flight while x is staged
  for (int e = tid; e < L27_TAB / 2; e += 512)
    tabs[e] = (uint32_t)(uint16_t)c_l27_off[2 * e] | ((uint32_t)(uint16_t)c_l27_off[2 * e + 1] << 16);
  const int i0 = ti * L27_TI, j0 = tj * L27_TJ - 2, k0 = tk * L27_TK - 2;
  for (int e = tid; e < L27_LDS_CELLS; e += 512) {
    const int li = e / L27_PI, r2 = e - li * L27_PI, lj = r2 / L27_SK, lk = r2 - lj * L27_SK;
    const int gi = G.plo + i0 + li, gj = j0 + lj, gk = k0 + lk;  // global plane: the two planes behind the last owned one are ghost planes
    double xv = 0.0;
    if (lj < L27_SJ && gi < G.mg && gi < G.plo + G.m0 + G.gw && gj >= 0 && gj < G.m1 && gk >= 0 && gk < G.m2) {
      const int64_t r = l27_xindex(G, gi, (int64_t)gj * G.m2 + gk);
      xv = dsc ? x[r] / dsc[r] : x[r];
    }
    xs[e] = xv;
    ys[e] = 0.0;
  }
  __syncthreads();
  if (e0) {
    // LDS cell of the lane's row of type (0, 0, 0) in the first unit; the other types are +1 in the odd directions, the second unit 4 planes on
    const int p0 = (2 * ra) * L27_PI + (ub * 4 + 2 * rb + 2) * L27_SK + (uc * 8 + 2 * rc + 2);
    l27_unit(A, B, uv0, e1 ? uv1 : nullptr, p0, q, tabs, xs, ys);
    if (e1) l27_unit(B, A, uv1, nullptr, p0 + 4 * L27_PI, q, tabs, xs, ys);
  }
  __syncthreads();
  double* dt = dump + (int64_t)tile * L27_CELLS;
  double dacc = 0.0;
  for (int e = tid; e < L27_CELLS; e += 512) {
    const int li = e / (L27_SJ * L27_SK);
    const double yv = ys[e + 8 * li];
    dt[e] = yv;
    dacc += yv * xs[e + 8 * li];
  }
  if (dotp) {  // (kernel argument: every thread of the workgroup takes the same way)
    const double d = block_reduce_sum(dacc, dred);
    if (tid == 0) dotp[tile] = d;
  }
}


// =====================================================================================================================================================
// Deterministic form of pass 1 (round 6; tables and the argument: spmv_lat_tables.h, "mode 4, deterministic order").  LANE = ROW: a wave owns the rows of
// four node types in a cube of 8 x 8 x 8 lattice points, the two waves of a cube split the types by the parity of their (j, k) column, a tile of
// 8 x 8 x 32 points = 4 cubes = 8 waves as before.  Every stored slot is one wave-wide step with a compile-time offset; the steps run phase-major -- a phase =
// the (dj, dk) of the offset -- with LDS-only barriers between phases, so every cell of the tile's y block receives its mirrored products from one wave
// per phase in program order: y is bitwise the same from run to run.  A cube's 260 steps are stored as its two waves' streams (138 + 122 steps of 64
// lanes), a pair of steps per lane side by side: each wave reads its 70 / 62 KB front to back with 16-byte loads.  bit 3 of the "lat27" knob selects
// the four-lanes-per-row kernel above (ds_add_f64 across waves: ~1e-16, not bitwise) for the A/B.
#define L27D_CUBE_D (L27D_CUBE_STEPS * 64)  // doubles per cube
__host__ __device__ constexpr int l27d_coff(int di, int dj, int dk) { return di * L27_PI + dj * L27_SK + dk; }
__host__ __device__ constexpr int l27d_toff(int t) { return ((t >> 2) & 1) * L27_PI + ((t >> 1) & 1) * L27_SK + (t & 1); }

template <int PG, int V0, int N>
__device__ __forceinline__ void l27d_load(double (&v)[8], const double* __restrict__ sb) {
  static_assert(V0 % 2 == 0 && N % 2 == 0, "stream steps leave in pairs");
#pragma unroll
  for (int i = 0; i < N; i += 2) {
    const l_d2 pr = __builtin_nontemporal_load((const l_d2*)sb + (int64_t)((V0 + i) >> 1) * 64);
    v[i] = pr.x;
    v[i + 1] = pr.y;
  }
}
// the barriers of `n` phase changes (a phase without a step of this wave's types still has its barrier: both waves of a cube, and all cubes, meet 24 times)
template <int n>
__device__ __forceinline__ void l27d_barriers() {
  if constexpr (n > 0) {
    mfem_lds_barrier();
    l27d_barriers<n - 1>();
  }
}
template <int PG, int V0, int I, int N>
__device__ __forceinline__ void l27d_proc(const double (&v)[8], int p0, bool act, const double (&xo)[4], double (&acc)[4], const double* xs, double* ys) {
  if constexpr (I < N) {
    constexpr L27DStream S = l27d_stream(PG);
    constexpr int vv = V0 + I, q = S.q[vv], t = l27d_type(PG, q);
    constexpr int coff = l27d_toff(t) + l27d_coff(S.di[vv], S.dj[vv], S.dk[vv]);
    l27d_barriers<(vv == 0 ? S.phase[0] : S.phase[vv] - S.phase[vv > 0 ? vv - 1 : 0])>();
    if (act) {  // (wave-uniform)
      const double a = v[I];
      acc[q] += a * xs[p0 + coff];
      if constexpr (!(S.di[vv] == 0 && S.dj[vv] == 0 && S.dk[vv] == 0))
        __builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double*)(ys + p0 + coff), a * xo[q]);  // (the diagonal has no mirror)
    }
    l27d_proc<PG, V0, I + 1, N>(v, p0, act, xo, acc, xs, ys);
  }
}
template <int PG, int C>
__device__ __forceinline__ void l27d_run(double (&A)[8], double (&B)[8], const double* __restrict__ sb, int p0, bool act, const double (&xo)[4],
                                         double (&acc)[4], const double* xs, double* ys) {
  constexpr int NV = l27d_stream(PG).n, NCH = (NV + 7) / 8, LAST = NCH - 1;
  constexpr int nthis = (C == LAST) ? NV - 8 * LAST : 8;
  if constexpr (C < LAST) {
    constexpr int nnext = (C + 1 == LAST) ? NV - 8 * LAST : 8;
    l27d_load<PG, (C + 1) * 8, nnext>((C & 1) ? A : B, sb);
  }
  __builtin_amdgcn_sched_barrier(0);
  l27d_proc<PG, C * 8, 0, nthis>((C & 1) ? B : A, p0, act, xo, acc, xs, ys);
#pragma unroll
  for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(acc[q]));
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (C < LAST) l27d_run<PG, C + 1>(A, B, sb, p0, act, xo, acc, xs, ys);
}
template <int PG>
__device__ __forceinline__ void l27d_wave(const double* __restrict__ sb, int p0, bool act, const double* xs, double* ys) {
  constexpr L27DStream S = l27d_stream(PG);
  double A[8], B[8];
  l27d_load<PG, 0, 8>(A, sb);
  double xo[4], acc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    xo[q] = xs[p0 + l27d_toff(l27d_type(PG, q))];
    acc[q] = 0.0;
  }
  l27d_run<PG, 0>(A, B, sb, p0, act, xo, acc, xs, ys);
  l27d_barriers<L27D_NPHASE - 1 - S.phase[S.n - 1]>();  // (none: the last phase holds the diagonal of every type)
  if (act) {  // the row sums: still the last phase (own cells; the other adds into them in this phase come from this wave)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      __builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double*)(ys + p0 + l27d_toff(l27d_type(PG, q))), acc[q]);
  }
}

// pass 1, deterministic: same tile, same dump, same arguments as k_spmv_lat27
__global__ __launch_bounds__(512, 4) void k_spmv_lat27d(Lat27Geom G, const double* __restrict__ vals, const double* __restrict__ x,
                                                        const double* __restrict__ dsc, double* __restrict__ dump,
                                                        const int32_t* __restrict__ done_flag, int tile0, int tcount, double* __restrict__ dotp) {
  __shared__ double xs[L27_LDS_CELLS];
  __shared__ double ys[L27_LDS_CELLS];
  __shared__ double dred[8];
  if (done_flag && done_flag[0]) return;
  const int chunk = (tcount + 7) >> 3;
  const int tsub = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= chunk || tsub >= tcount) return;  // (the whole workgroup leaves)
  const int tile = tile0 + tsub;
  const int tk = tile % G.ntk, t2 = tile / G.ntk, tj = t2 % G.ntj, ti = t2 / G.ntj;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int cu = wv >> 1, pg = wv & 1;
  const int nck = (G.m2 + 7) >> 3, ncj = G.ntj;  // cubes per direction (a tile is one cube in i and j, four in k)
  const int ck = tk * 4 + cu;
  const bool act = ck < nck;
  // a cube that does not exist (lattice edge in k) is read from a place that does and not worked on
  const double* sb = vals + (act ? (((int64_t)ti * ncj + tj) * nck + ck) * (int64_t)L27D_CUBE_D : 0) + (pg ? (int64_t)l27d_stream(0).n * 64 : 0) + lane * 2;
  const int i0 = ti * L27_TI, j0 = tj * L27_TJ - 2, k0 = tk * L27_TK - 2;
  for (int e = tid; e < L27_LDS_CELLS; e += 512) {
    const int li = e / L27_PI, r2 = e - li * L27_PI, lj = r2 / L27_SK, lk = r2 - lj * L27_SK;
    const int gi = G.plo + i0 + li, gj = j0 + lj, gk = k0 + lk;
    double xv = 0.0;
    if (lj < L27_SJ && gi < G.mg && gi < G.plo + G.m0 + G.gw && gj >= 0 && gj < G.m1 && gk >= 0 && gk < G.m2) {
      const int64_t r = l27_xindex(G, gi, (int64_t)gj * G.m2 + gk);
      xv = dsc ? x[r] / dsc[r] : x[r];
    }
    xs[e] = xv;
    ys[e] = 0.0;
  }
  __syncthreads();
  {
    const int a = lane >> 4, b = (lane >> 2) & 3, c = lane & 3;
    const int p0 = (2 * a) * L27_PI + (2 * b + 2) * L27_SK + (cu * 8 + 2 * c + 2);  // the lane's row of type (0, 0, 0)
    if (pg == 0) l27d_wave<0>(sb, p0, act, xs, ys);
    else l27d_wave<1>(sb, p0, act, xs, ys);
  }
  __syncthreads();
  double* dt = dump + (int64_t)tile * L27_CELLS;
  double dacc = 0.0;
  for (int e = tid; e < L27_CELLS; e += 512) {
    const int li = e / (L27_SJ * L27_SK);
    const double yv = ys[e + 8 * li];
    dt[e] = yv;
    dacc += yv * xs[e + 8 * li];
  }
  if (dotp) {
    const double d = block_reduce_sum(dacc, dred);
    if (tid == 0) dotp[tile] = d;
  }
}

// the layout pass of the deterministic form: a wave per (cube, parity group), lane = row.  A lane walks the UPPER HALF of its CSR row front to back (the
// stored slots of a type in lexicographic order = ascending columns: each 128-byte line of the row is used up by 16 consecutive loads of the lane while it
// sits in the L1) and drops every value at its place in the phase-major stream (8-byte stores; the two halves of a 16-byte pair come from two steps).  A
// first version read in STREAM order -- 64 rows per instruction, each row touched again and again over 60 steps: 11.6 ms per bind of the 128^3 matrix
// against 2.65 ms of the four-lanes-per-row fill.
__constant__ uint8_t c_l27d_lex[2][4][64][4];  // [pg][q][e]: (di, dj + 2, dk + 2, stream step v) of the e-th stored slot of the wave's q-th type, lexicographic
__constant__ int c_l27d_kup[2][4];
static std::atomic<bool> g_l27d_tables{false};
static int lat27d_upload_tables() {
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  if (g_l27d_tables) return MFEM_OK;
  static uint8_t h[2][4][64][4];
  int kup[2][4];
  memset(h, 0, sizeof(h));
  for (int pg = 0; pg < 2; ++pg) {
    const L27DStream S = l27d_stream(pg);
    for (int q = 0; q < 4; ++q) {
      const int t = l27d_type(pg, q);
      const int R0 = (t & 4) ? 1 : 2, R1 = (t & 2) ? 1 : 2, R2 = (t & 1) ? 1 : 2;
      int e = 0;
      for (int di = 0; di <= R0; ++di)
        for (int dj = -R1; dj <= R1; ++dj)
          for (int dk = -R2; dk <= R2; ++dk) {
            if (!(di > 0 || dj > 0 || (dj == 0 && dk >= 0))) continue;  // (the diagonal first: (0, 0, 0) is the smallest stored offset)
            int v = -1;
            for (int u = 0; u < S.n; ++u)
              if (S.q[u] == q && S.di[u] == di && S.dj[u] == dj && S.dk[u] == dk) v = u;
            if (v < 0 || e >= 64) {
              mfem_set_error("lattice-tile tables (deterministic form): a stored slot has no stream step");
              return MFEM_ERR_INTERNAL;
            }
            h[pg][q][e][0] = (uint8_t)di;
            h[pg][q][e][1] = (uint8_t)(dj + 2);
            h[pg][q][e][2] = (uint8_t)(dk + 2);
            h[pg][q][e][3] = (uint8_t)v;
            ++e;
          }
      kup[pg][q] = e;
    }
  }
  if (hipMemcpyToSymbol(HIP_SYMBOL(c_l27d_lex), h, sizeof(h)) != hipSuccess || hipMemcpyToSymbol(HIP_SYMBOL(c_l27d_kup), kup, sizeof(kup)) != hipSuccess) {
    mfem_set_error("lattice-tile tables (deterministic form): hipMemcpyToSymbol failed");
    return MFEM_ERR_HIP;
  }
  g_l27d_tables = true;
  return MFEM_OK;
}
template <typename RP>
__global__ __launch_bounds__(MFEM_BLOCK) void k_l27d_fill(Lat27Geom G, const RP* __restrict__ rowptr, int base, const double* __restrict__ vals,
                                                            double* __restrict__ out, unsigned long long* __restrict__ stats) {
  // a wave per (cube, node type): the type's 64 rows 16 at a time, FOUR LANES PER ROW reading four consecutive entries of the row's upper half (16 rows x
  // 32 bytes per load instruction: the lines in flight fit the L1 -- with a lane per row, 64 rows per instruction, the fill took 11.6 ms for the 128^3
  // matrix, 4.4 x the four-lanes-per-row fill of the other form); every value goes to its place in the phase-major stream of its row's lane
  const int lane = threadIdx.x & 63, qd = lane & 3, rho = lane >> 2;
  const int b = rho >> 2, c = rho & 3;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int nci = G.nti, ncj = G.ntj, nck = (G.m2 + 7) >> 3;
  const int64_t nwork = 8 * (int64_t)nci * ncj * nck;
  const int n0 = l27d_stream(0).n;
  double amax = 0.0;
  for (int64_t u = wave; u < nwork; u += nwaves) {
    const int t = (int)(u & 7);
    const int64_t cube = u >> 3;
    const int ck = (int)(cube % nck);
    const int64_t c2 = cube / nck;
    const int cj = (int)(c2 % ncj), ci = (int)(c2 / ncj);
    const int pg = (((t >> 1) & 1) + (t & 1)) & 1;
    const int q = pg ? (t == 1 ? 0 : t == 5 ? 1 : t == 2 ? 2 : 3) : (t == 0 ? 0 : t == 4 ? 1 : t == 3 ? 2 : 3);  // (l27d_type backwards)
    double* os = out + cube * (int64_t)L27D_CUBE_D + (pg ? (int64_t)n0 * 64 : 0);
    const int kup = c_l27d_kup[pg][q];
    for (int a = 0; a < 4; ++a) {
      const int oi = ci * 8 + 2 * a + ((t >> 2) & 1), gj = cj * 8 + 2 * b + ((t >> 1) & 1), gk = ck * 8 + 2 * c + (t & 1);
      const int gi = oi + G.plo;
      const bool valid = oi < G.m0 && gj < G.m1 && gk < G.m2;
      int li = 0, ni = 1, lj = 0, nj = 1, lk = 0, nk = 1;
      int64_t rp = 0;
      if (valid) {
        l27_range(gi, G.mg, li, ni);
        l27_range(gj, G.m1, lj, nj);
        l27_range(gk, G.m2, lk, nk);
        rp = (int64_t)rowptr[((int64_t)oi * G.m1 + gj) * G.m2 + gk] - base;
      }
      double* orow = os + (a * 16 + rho) * 2;  // the row's lane in the stream
      for (int e = qd; e < kup; e += 4) {
        const int di = c_l27d_lex[pg][q][e][0], dj = (int)c_l27d_lex[pg][q][e][1] - 2, dk = (int)c_l27d_lex[pg][q][e][2] - 2, v = c_l27d_lex[pg][q][e][3];
        const int ci2 = gi + di, cj2 = gj + dj, ck2 = gk + dk;
        double val = 0.0;
        if (valid && ci2 < G.mg && cj2 >= 0 && cj2 < G.m1 && ck2 >= 0 && ck2 < G.m2) {  // (ci2 may be a ghost plane of a slab)
          val = vals[rp + ((int64_t)(di - li) * nj + (dj - lj)) * nk + (dk - lk)];
          double av = fabs(val);
          if (!(av == av)) av = __builtin_huge_val();  // NaN: fmax would drop it
          amax = fmax(amax, av);
        }
        orow[(int64_t)(v >> 1) * 128 + (v & 1)] = val;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amax = fmax(amax, __shfl_down(amax, o, MFEM_WAVE));
  if (lane == 0) atomicMax(stats + 1, (unsigned long long)__double_as_longlong(amax));
}

// pass 2: y[r] = alpha * (sum over the tiles whose block covers r, fixed order) + beta * y[r]; fused dot with dotw.  A thread owns a
// (j, k) position of the tile and its 8 lattice planes: 8 independent loads per covering tile.
// Slab with a lower neighbour (G.plo > 0): the rows of the first owned plane (an even plane: reach 2) also have entries towards the two ghost planes
// below.  No stored entry mirrors onto them (the rows that would belong to the neighbour rank), so they are taken from the caller's CSR values here.
template <typename RP>
__global__ __launch_bounds__(MFEM_BLOCK) void k_lat27_gather(Lat27Geom G, const double* __restrict__ dump, double* __restrict__ y,
                                                               double alpha, double beta, const double* __restrict__ dotw,
                                                               double* __restrict__ partials, const int32_t* __restrict__ done_flag,
                                                               const RP* __restrict__ rowptr, int base, const double* __restrict__ csr_vals,
                                                               const double* __restrict__ x, const double* __restrict__ dsc) {
  __shared__ double red[4];
  if (done_flag && done_flag[0]) return;
  double dot_acc = 0.0;
  const int ntiles = G.nti * G.ntj * G.ntk;
  const int lk = threadIdx.x & (L27_TK - 1), lj = threadIdx.x >> 5;
  const int PC = L27_SJ * L27_SK;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int tk = tile % G.ntk, t2 = tile / G.ntk, tj = t2 % G.ntj, ti = t2 / G.ntj;
    const int gj = tj * L27_TJ + lj, gk = tk * L27_TK + lk, gi0 = ti * L27_TI;
    if (gj >= G.m1 || gk >= G.m2) continue;
    double s[L27_TI];
#pragma unroll
    for (int u = 0; u < L27_TI; ++u) s[u] = 0.0;
    // covering tiles (ti + a, tj + b, tk + c), a in {-1, 0}, b, c in {-1, 0, 1}, in this fixed order: the row's cell must exist in that tile's block
    for (int b = -1; b <= 1; ++b) {
      if ((b < 0 && (lj >= 2 || tj == 0)) || (b > 0 && (lj < L27_TJ - 2 || tj == G.ntj - 1))) continue;
      for (int c = -1; c <= 1; ++c) {
        if ((c < 0 && (lk >= 2 || tk == 0)) || (c > 0 && (lk < L27_TK - 2 || tk == G.ntk - 1))) continue;
        const int cell = (lj - L27_TJ * b + 2) * L27_SK + (lk - L27_TK * c + 2);
        if (ti > 0) {  // the tile below: its planes 8, 9 are this tile's 0, 1
          const double* d = dump + (((int64_t)(ti - 1) * G.ntj + (tj + b)) * G.ntk + (tk + c)) * L27_CELLS + cell;
          s[0] += d[8 * PC];
          s[1] += d[9 * PC];
        }
        const double* d = dump + (((int64_t)ti * G.ntj + (tj + b)) * G.ntk + (tk + c)) * L27_CELLS + cell;
#pragma unroll
        for (int u = 0; u < L27_TI; ++u) s[u] += d[u * PC];
      }
    }
    if (G.plo > 0 && ti == 0) {  // the lower ghost planes (see above): the first two of the row's five i-offsets
      int l1, n1, l2, n2;
      l27_range(gj, G.m1, l1, n1);
      l27_range(gk, G.m2, l2, n2);
      const int64_t rp = (int64_t)rowptr[(int64_t)gj * G.m2 + gk] - base;
      double acc = 0.0;
      for (int a = 0; a < 2; ++a)
        for (int b = 0; b < n1; ++b)
          for (int c = 0; c < n2; ++c) {
            const int64_t xi = l27_xindex(G, G.plo - 2 + a, (int64_t)(gj + l1 + b) * G.m2 + gk + l2 + c);
            acc += csr_vals[rp + ((int64_t)a * n1 + b) * n2 + c] * (dsc ? x[xi] / dsc[xi] : x[xi]);
          }
      s[0] += acc;
    }
#pragma unroll
    for (int u = 0; u < L27_TI; ++u) {
      if (gi0 + u < G.m0) {
        const int64_t r = ((int64_t)(gi0 + u) * G.m1 + gj) * G.m2 + gk;
        double yv = alpha * s[u];
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

// pass 2, staged (round 4): the same sums in the same order, but every global load of a tile is in flight at once.  The kernel above walks the up to 18
// covering blocks of a row in masked code blocks, one memory round trip each (every wave holds lanes on the tile's k rim, so every wave takes at least
// three, the waves on the j rim nine: 3.1 TB/s).  Here the workgroup first copies the 4 320 (row, covering block) values of its tile -- the extended box
// (8 + 2 planes) x (8 + 2 + 2 lines) x (32 + 2 + 2 columns): own cells, the cells the tile below / beside / diagonal to it holds for these rows -- into
// LDS, 17 independent loads per thread, and the row owners then add them from LDS in the order of the kernel above (bitwise the same y).
// Measured (tools/gather_ab.py, C4): 1 % off a 200-iteration solve -- the round trips were not what bounds pass 2 (a 0.18 ms kernel of 0.56 GB).
#define L27_EJ (L27_TJ + 4)
#define L27_EK (L27_TK + 4)
#define L27_ECELLS ((L27_TI + 2) * L27_EJ * L27_EK)  // 4320
#define L27_EU ((L27_ECELLS + MFEM_BLOCK - 1) / MFEM_BLOCK)
template <typename RP>
__global__ __launch_bounds__(MFEM_BLOCK) void k_lat27_gather_st(Lat27Geom G, const double* __restrict__ dump, double* __restrict__ y,
                                                                  double alpha, double beta, const double* __restrict__ dotw,
                                                                  double* __restrict__ partials, const int32_t* __restrict__ done_flag,
                                                                  const RP* __restrict__ rowptr, int base, const double* __restrict__ csr_vals,
                                                                  const double* __restrict__ x, const double* __restrict__ dsc) {
  __shared__ double E[L27_ECELLS];
  __shared__ double red[4];
  if (done_flag && done_flag[0]) return;
  double dot_acc = 0.0;
  const int ntiles = G.nti * G.ntj * G.ntk;
  const int lk = threadIdx.x & (L27_TK - 1), lj = threadIdx.x >> 5;
  const int PC = L27_SJ * L27_SK;
  // extended line / column e -> (neighbour offset, line or column of this tile): 0 .. T - 1 own; T, T + 1: the block below / before holds rows 0, 1;
  // T + 2, T + 3: the block after holds rows T - 2, T - 1
  auto ext = [](int e, int T, int& off, int& l) {
    if (e < T) { off = 0; l = e; }
    else if (e < T + 2) { off = -1; l = e - T; }
    else { off = 1; l = e - 4; }
  };
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {  // (the trip count is the workgroup's: every barrier below is reached by all threads)
    const int tk = tile % G.ntk, t2 = tile / G.ntk, tj = t2 % G.ntj, ti = t2 / G.ntj;
    double t[L27_EU];
#pragma unroll
    for (int u = 0; u < L27_EU; ++u) {
      const int e = threadIdx.x + u * MFEM_BLOCK;
      t[u] = 0.0;
      if (e < L27_ECELLS) {
        const int ei = e / (L27_EJ * L27_EK), r2 = e - ei * (L27_EJ * L27_EK), ej = r2 / L27_EK, ek = r2 - ej * L27_EK;
        int b, c, sj, sk;
        ext(ej, L27_TJ, b, sj);
        ext(ek, L27_TK, c, sk);
        const int a = ei < L27_TI ? 0 : -1;  // (planes 8, 9 of the block below are this tile's planes 0, 1: the block's plane index is ei either way)
        const bool ok = (a == 0 || ti > 0) && (b == 0 || (b < 0 ? tj > 0 : tj < G.ntj - 1)) && (c == 0 || (c < 0 ? tk > 0 : tk < G.ntk - 1));
        if (ok)
          t[u] = __builtin_nontemporal_load(dump + (((int64_t)(ti + a) * G.ntj + (tj + b)) * G.ntk + (tk + c)) * L27_CELLS + ei * PC +
                                            (sj - L27_TJ * b + 2) * L27_SK + (sk - L27_TK * c + 2));
      }
    }
#pragma unroll
    for (int u = 0; u < L27_EU; ++u) {
      const int e = threadIdx.x + u * MFEM_BLOCK;
      if (e < L27_ECELLS) E[e] = t[u];
    }
    __syncthreads();
    const int gj = tj * L27_TJ + lj, gk = tk * L27_TK + lk, gi0 = ti * L27_TI;
    if (gj < G.m1 && gk < G.m2) {
      double s[L27_TI];
#pragma unroll
      for (int u = 0; u < L27_TI; ++u) s[u] = 0.0;
      for (int b = -1; b <= 1; ++b) {
        if ((b < 0 && (lj >= 2 || tj == 0)) || (b > 0 && (lj < L27_TJ - 2 || tj == G.ntj - 1))) continue;
        const int ej = b == 0 ? lj : b < 0 ? L27_TJ + lj : lj + 4;
        for (int c = -1; c <= 1; ++c) {
          if ((c < 0 && (lk >= 2 || tk == 0)) || (c > 0 && (lk < L27_TK - 2 || tk == G.ntk - 1))) continue;
          const int ek = c == 0 ? lk : c < 0 ? L27_TK + lk : lk + 4;
          const double* d = E + ej * L27_EK + ek;
          if (ti > 0) {
            s[0] += d[8 * (L27_EJ * L27_EK)];
            s[1] += d[9 * (L27_EJ * L27_EK)];
          }
#pragma unroll
          for (int u = 0; u < L27_TI; ++u) s[u] += d[u * (L27_EJ * L27_EK)];
        }
      }
      if (G.plo > 0 && ti == 0) {  // the lower ghost planes (see k_lat27_gather): the first two of the row's five i-offsets
        int l1, n1, l2, n2;
        l27_range(gj, G.m1, l1, n1);
        l27_range(gk, G.m2, l2, n2);
        const int64_t rp = (int64_t)rowptr[(int64_t)gj * G.m2 + gk] - base;
        double acc = 0.0;
        for (int a = 0; a < 2; ++a)
          for (int b = 0; b < n1; ++b)
            for (int c = 0; c < n2; ++c) {
              const int64_t xi = l27_xindex(G, G.plo - 2 + a, (int64_t)(gj + l1 + b) * G.m2 + gk + l2 + c);
              acc += csr_vals[rp + ((int64_t)a * n1 + b) * n2 + c] * (dsc ? x[xi] / dsc[xi] : x[xi]);
            }
        s[0] += acc;
      }
#pragma unroll
      for (int u = 0; u < L27_TI; ++u) {
        if (gi0 + u < G.m0) {
          const int64_t r = ((int64_t)(gi0 + u) * G.m1 + gj) * G.m2 + gk;
          double yv = alpha * s[u];
          if (beta != 0.0) yv += beta * y[r];
          y[r] = yv;
          if (dotw) dot_acc += yv * dotw[r];
        }
      }
    }
    __syncthreads();  // the staged values are consumed: the next tile's may land
  }
  if (partials) {
    const double bsum = block_reduce_sum(dot_acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = bsum;
  }
}

// The fused CG iteration on the lattice tiles (one rank): pass 2 and the residual update of Jacobi-CG in one kernel.  q = A p is needed twice in a CG iteration --
// in p . q, which pass 1 now delivers (k_spmv_lat27: dotp), and in r -= alpha q -- so the sums over the covering blocks are formed here, used and never stored:
// no y written and read back, no separate gather launch (3 of the iteration's vector streams and one launch less).  Staging and summation order are
// k_lat27_gather_st's; the update arithmetic is k_cg_update's (krylov.hip), operation for operation.
__global__ __launch_bounds__(MFEM_BLOCK) void k_lat27_gather_cg(Lat27Geom G, const double* __restrict__ dump, LatCgUpdate U) {
  __shared__ double E[L27_ECELLS];
  __shared__ double red[4];
  if (U.flags[F_DONE]) return;
  const double pap = U.np > 0 ? reduce_partials_bcast(U.pap_partials, U.np, red) : U.S[S_PAP];  // (as k_cg_update / k_cg_pupdate: every workgroup folds the partials itself)
  const double alpha = U.S[S_RZ0 + U.cur] / pap;
  const bool exact = U.sw && U.S[S_RR] * U.n_inv <= U.gate2;
  double rz = 0.0, rr = 0.0;
  const int ntiles = G.nti * G.ntj * G.ntk;
  const int lk = threadIdx.x & (L27_TK - 1), lj = threadIdx.x >> 5;
  const int PC = L27_SJ * L27_SK;
  auto ext = [](int e, int T, int& off, int& l) {
    if (e < T) { off = 0; l = e; }
    else if (e < T + 2) { off = -1; l = e - T; }
    else { off = 1; l = e - 4; }
  };
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {  // (the trip count is the workgroup's: every barrier below is reached by all threads)
    const int tk = tile % G.ntk, t2 = tile / G.ntk, tj = t2 % G.ntj, ti = t2 / G.ntj;
    double t[L27_EU];
#pragma unroll
    for (int u = 0; u < L27_EU; ++u) {
      const int e = threadIdx.x + u * MFEM_BLOCK;
      t[u] = 0.0;
      if (e < L27_ECELLS) {
        const int ei = e / (L27_EJ * L27_EK), r2 = e - ei * (L27_EJ * L27_EK), ej = r2 / L27_EK, ek = r2 - ej * L27_EK;
        int b, c, sj, sk;
        ext(ej, L27_TJ, b, sj);
        ext(ek, L27_TK, c, sk);
        const int a = ei < L27_TI ? 0 : -1;
        const bool ok = (a == 0 || ti > 0) && (b == 0 || (b < 0 ? tj > 0 : tj < G.ntj - 1)) && (c == 0 || (c < 0 ? tk > 0 : tk < G.ntk - 1));
        if (ok)
          t[u] = __builtin_nontemporal_load(dump + (((int64_t)(ti + a) * G.ntj + (tj + b)) * G.ntk + (tk + c)) * L27_CELLS + ei * PC +
                                            (sj - L27_TJ * b + 2) * L27_SK + (sk - L27_TK * c + 2));
      }
    }
#pragma unroll
    for (int u = 0; u < L27_EU; ++u) {
      const int e = threadIdx.x + u * MFEM_BLOCK;
      if (e < L27_ECELLS) E[e] = t[u];
    }
    __syncthreads();
    const int gj = tj * L27_TJ + lj, gk = tk * L27_TK + lk, gi0 = ti * L27_TI;
    if (gj < G.m1 && gk < G.m2) {
      double s[L27_TI];
#pragma unroll
      for (int u = 0; u < L27_TI; ++u) s[u] = 0.0;
      for (int b = -1; b <= 1; ++b) {
        if ((b < 0 && (lj >= 2 || tj == 0)) || (b > 0 && (lj < L27_TJ - 2 || tj == G.ntj - 1))) continue;
        const int ej = b == 0 ? lj : b < 0 ? L27_TJ + lj : lj + 4;
        for (int c = -1; c <= 1; ++c) {
          if ((c < 0 && (lk >= 2 || tk == 0)) || (c > 0 && (lk < L27_TK - 2 || tk == G.ntk - 1))) continue;
          const int ek = c == 0 ? lk : c < 0 ? L27_TK + lk : lk + 4;
          const double* d = E + ej * L27_EK + ek;
          if (ti > 0) {
            s[0] += d[8 * (L27_EJ * L27_EK)];
            s[1] += d[9 * (L27_EJ * L27_EK)];
          }
#pragma unroll
          for (int u = 0; u < L27_TI; ++u) s[u] += d[u * (L27_EJ * L27_EK)];
        }
      }
#pragma unroll
      for (int u = 0; u < L27_TI; ++u) {
        if (gi0 + u < G.m0) {
          const int64_t i = ((int64_t)(gi0 + u) * G.m1 + gj) * G.m2 + gk;
          const double av = s[u];  // (A p)[i]
          double rv, z;
          if (U.zrec && U.dinv) {  // the array holds z: z -= alpha dinv .* Ap ; r = z ./ dinv for the two dot products only
            const double dv = U.dinv[i];
            z = U.r[i] - alpha * (av * dv);
            U.r[i] = z;
            rv = dv != 0.0 ? z * mfem_recip_nr(dv) : 0.0;
          } else {
            rv = U.r[i] - alpha * av;
            U.r[i] = rv;
            z = U.dinv ? rv * U.dinv[i] : rv;
          }
          rz += rv * z;
          if (exact) {
            const double tt = U.sw[i] * rv;
            rr += tt * tt;
          } else {
            rr += rv * rv;
          }
        }
      }
    }
    __syncthreads();  // the staged values are consumed: the next tile's may land
  }
  const double s0 = block_reduce_sum(rz, red);
  const double s1 = block_reduce_sum(rr, red);
  if (threadIdx.x == 0) {
    U.partials2[blockIdx.x] = s0;
    U.partials2[gridDim.x + blockIdx.x] = (U.sw && !exact) ? s1 * U.smax2 : s1;
  }
}

static Lat27Geom lat27_geom(const mfem_csr_s* A) {
  Lat27Geom G{};
  G.m1 = A->lat_m1;
  G.m2 = A->lat_m2;
  G.n = A->n;
  G.m0 = (int)(A->n / ((int64_t)A->lat_m1 * A->lat_m2));
  G.plo = A->lat_plo;
  G.mg = A->lat_m0 > 0 ? A->lat_m0 : G.m0;
  G.gw = 2;
  G.nui = (G.m0 + 3) / 4;
  G.nuj = (G.m1 + 3) / 4;
  G.nuk = (G.m2 + 7) / 8;
  G.nti = (G.m0 + L27_TI - 1) / L27_TI;
  G.ntj = (G.m1 + L27_TJ - 1) / L27_TJ;
  G.ntk = (G.m2 + L27_TK - 1) / L27_TK;
  return G;
}

// A pattern without a lattice hint (lat_fields == 0, no ghost columns): propose one from the columns of row 0 -- the corner node of a lattice
// numbered plane by plane, line by line -- for the two stencils the lattice-tile layouts know.  Only a proposal: the plans check every entry.
//   hex-27, one field:      row 0 = 27 columns {a PL + b m2 + c : a, b, c in 0..2}  ->  m2 = col[3], PL = col[9]
//   27-point, F = 1..3 fields: row 0 = F x 8 columns {g N + a PL + b m2 + c : a, b, c in 0..1}  ->  m2 = col[2], PL = col[4], N = col[8] (F > 1)
// lat_fields = -1 afterwards if neither fits (so that the question is asked once per pattern).
int mfem_lattice_from_first_row(mfem_context_s* ctx, mfem_csr_s* A) {
  if (A->lat_fields != 0) return MFEM_OK;
  A->lat_fields = -1;
  A->lat_inferred = 1;
  if (A->n < 8 || (A->ncols > A->n)) return MFEM_OK;
  int64_t rp[2] = {0, 0};
  if (A->rowptr_bits == 64) {
    MFEM_CHECK_HIP(hipMemcpyAsync(rp, A->rowptr, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    MFEM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  } else {
    int32_t r32[2];
    MFEM_CHECK_HIP(hipMemcpyAsync(r32, A->rowptr, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    MFEM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    rp[0] = r32[0];
    rp[1] = r32[1];
  }
  const int64_t len = rp[1] - rp[0];
  if (len != 27 && len != 24 && len != 16 && len != 8) return MFEM_OK;
  int32_t c[27];
  MFEM_CHECK_HIP(hipMemcpyAsync(c, A->colidx + (rp[0] - A->index_base), (size_t)len * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  MFEM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < len; ++i) c[i] -= A->index_base;
  if (c[0] != 0) return MFEM_OK;
  if (len == 27) {
    const int64_t m2 = c[3], PL = c[9];
    if (m2 < 3 || PL < 3 * m2 || PL % m2 != 0 || A->n % PL != 0) return MFEM_OK;
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b)
        for (int k = 0; k < 3; ++k)
          if (c[(a * 3 + b) * 3 + k] != a * PL + b * m2 + k) return MFEM_OK;
    A->lat_fields = 1;
    A->lat_m2 = (int32_t)m2;
    A->lat_m1 = (int32_t)(PL / m2);
    A->lat_m0 = (int32_t)(A->n / PL);
    A->lat_plo = 0;
    A->lat_gw = 2;
  } else {
    const int F = (int)(len / 8);
    if (A->n % F != 0) return MFEM_OK;
    const int64_t m2 = c[2], PL = c[4], N = F > 1 ? c[8] : A->n;
    if (m2 < 2 || PL < 2 * m2 || PL % m2 != 0 || N < 2 * PL || N % PL != 0 || A->n != F * N) return MFEM_OK;
    for (int g = 0; g < F; ++g)
      for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
          for (int k = 0; k < 2; ++k)
            if (c[g * 8 + (a * 2 + b) * 2 + k] != g * N + a * PL + b * m2 + k) return MFEM_OK;
    A->lat_fields = F;
    A->lat_m2 = (int32_t)m2;
    A->lat_m1 = (int32_t)(PL / m2);
    A->lat_m0 = (int32_t)(N / PL);
    A->lat_plo = 0;
    A->lat_gw = 1;
  }
  return MFEM_OK;
}

// lat27_state: 0 not inspected, -1 not the lattice stencil, 1 structure ok
int mfem_lat27_plan(mfem_context_s* ctx, mfem_csr_s* A) {
  if (A->lat27_state != 0) return MFEM_OK;
  if (A->n < g_layout_min_rows_lat27) return MFEM_OK;  // launch-bound sizes stay on the CSR tile kernel
  A->lat27_state = -1;
  if (A->lat_fields == 0) {  // a caller-supplied pattern (mfem_csr_create: the reference's own K_J_ptr / K_J): read the lattice off row 0
    int rc0 = mfem_lattice_from_first_row(ctx, A);
    if (rc0) return rc0;
  }
  if (A->lat_fields != 1 || A->lat_m1 < 3 || A->lat_m2 < 3 || !(A->lat_m1 & 1) || !(A->lat_m2 & 1)) return MFEM_OK;
  const int64_t PL = (int64_t)A->lat_m1 * A->lat_m2;
  if (A->n % PL != 0) return MFEM_OK;
  const int64_t m0 = A->n / PL;
  if (m0 < 1 || m0 > (1 << 20) || A->max_row_nnz > 125) return MFEM_OK;
  if (A->ncols > A->n) {  // slab pattern (ghost columns): the hint must place the owned planes in the lattice (on element boundaries) and describe the ghost blocks
    if (A->lat_m0 < 3 || !(A->lat_m0 & 1) || A->lat_gw != 2 || A->lat_plo < 0 || (A->lat_plo & 1) || A->lat_plo + m0 > A->lat_m0 ||
        A->ncols != A->n + 4 * PL)
      return MFEM_OK;
  } else {
    if (m0 < 3 || !(m0 & 1)) return MFEM_OK;
    if (A->lat_m0 > 0 && (A->lat_m0 != m0 || A->lat_plo != 0)) return MFEM_OK;
  }
  {  // cheap refusal before the entry-by-entry check: the longest row of the stencil is known from the lattice sizes (a hex-8 lattice with odd point
     // counts carries the same hint: 27 against 125)
    const int64_t mg = A->lat_m0 > 0 ? A->lat_m0 : m0;
    auto w = [](int64_t m) { return m >= 5 ? 5 : 3; };
    if (A->max_row_nnz != w(mg) * w(A->lat_m1) * w(A->lat_m2)) return MFEM_OK;
  }
  int rc = lat27_upload_tables();
  if (rc) return rc;
  const Lat27Geom G = lat27_geom(A);
  if ((int64_t)G.nti * G.ntj * G.ntk >= ((int64_t)1 << 28)) return MFEM_OK;
  int32_t* d_bad = ctx->d_flags + 12;
  MFEM_CHECK_HIP(hipMemsetAsync(d_bad, 0, sizeof(int32_t), ctx->stream));
  const int grid = mfem_grid_for(A->n, MFEM_BLOCK, ctx->num_cus * 16);
  if (A->rowptr_bits == 64)
    hipLaunchKernelGGL(k_l27_verify<int64_t>, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, G, (const int64_t*)A->rowptr, A->colidx,
                       A->index_base, d_bad);
  else
    hipLaunchKernelGGL(k_l27_verify<int32_t>, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, G, (const int32_t*)A->rowptr, A->colidx,
                       A->index_base, d_bad);
  MFEM_CHECK_LAUNCH();
  MFEM_CHECK_HIP(hipMemcpyAsync(ctx->h_flags + 12, d_bad, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  MFEM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  if (ctx->h_flags[12] == 0) A->lat27_state = 1;
  return MFEM_OK;
}

static size_t lat27d_vals_doubles(const Lat27Geom& G) { return (size_t)G.nti * G.ntj * ((G.m2 + 7) / 8) * L27D_CUBE_D; }
static size_t lat27q_vals_doubles(const Lat27Geom& G) { return (size_t)G.nui * G.nuj * G.nuk * L27_UNIT_D; }
// (the workspace is sized for whichever form needs more: the knob may change between the plan and a bind)
static size_t lat27_vals_doubles(const Lat27Geom& G) { const size_t a = lat27d_vals_doubles(G), b = lat27q_vals_doubles(G); return a > b ? a : b; }
static size_t lat27_read_doubles(const Lat27Geom& G) { return g_lat27_det ? lat27d_vals_doubles(G) : lat27q_vals_doubles(G); }  // what pass 1 streams
static size_t lat27_dump_doubles(const Lat27Geom& G) { return (size_t)G.nti * G.ntj * G.ntk * L27_CELLS; }

// workspace of the layout: the stored entries, then the per-tile y blocks
size_t mfem_lat27_bytes(const mfem_csr_s* A) {
  if (A->lat27_state != 1 || !g_lat27_enable || A->n < g_layout_min_rows_lat27) return 0;
  const Lat27Geom G = lat27_geom(A);
  return sizeof(double) * (lat27_vals_doubles(G) + lat27_dump_doubles(G) + (size_t)G.nti * G.ntj * G.ntk);  // (+ one dot-product partial per tile: the fused CG iteration)
}

struct Lat27Bind { double *vals, *dump; const double* src; };
static void lat27_probe_unbind(mfem_csr_s* A) { mfem_lat27_unbind(A); }
static void lat27_probe_rebind(mfem_csr_s* A, void* c) {
  const Lat27Bind* b = (const Lat27Bind*)c;
  A->lat27_vals = b->vals;
  A->lat27_dump = b->dump;
  A->lat27_src = b->src;
}

// Makes the layout copy of `vals` in buf and binds it if the values are symmetric (mfem_sym_probe; else leaves the pattern unbound: the caller
// binds the sliced layout instead).  scratch: 3 n doubles, left dirty.  Two stream synchronisations (max |a|, the verdict).
int mfem_lat27_bind(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, double* buf, const double* dsc, double* scratch, bool allow_rem) {
  mfem_lat27_unbind(A);
  if (A->lat27_state != 1 || !g_lat27_enable || !buf || !scratch) return MFEM_OK;
  const Lat27Geom G = lat27_geom(A);
  unsigned long long* d_stats = (unsigned long long*)(ctx->d_flags + 12);
  MFEM_CHECK_HIP(hipMemsetAsync(d_stats, 0, 2 * sizeof(unsigned long long), ctx->stream));
  A->lat27_det = g_lat27_det ? 1 : 0;  // (the form THIS copy is made in: the launches follow the copy, not the knob)
  if (A->lat27_det) {
    const int rt = lat27d_upload_tables();
    if (rt) return rt;
    const int64_t nwork = 8 * (int64_t)G.nti * G.ntj * ((G.m2 + 7) / 8);  // a wave per (cube, node type)
    const int grid = mfem_grid_for(nwork * 64, MFEM_BLOCK, ctx->num_cus * 16);
    if (A->rowptr_bits == 64)
      hipLaunchKernelGGL(k_l27d_fill<int64_t>, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, G, (const int64_t*)A->rowptr, A->index_base, vals, buf, d_stats);
    else
      hipLaunchKernelGGL(k_l27d_fill<int32_t>, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, G, (const int32_t*)A->rowptr, A->index_base, vals, buf, d_stats);
  } else {
  const int64_t nunits = (int64_t)G.nui * G.nuj * G.nuk;
  const int grid = mfem_grid_for(nunits * 64, MFEM_BLOCK, ctx->num_cus * 16);
  if (A->rowptr_bits == 64)
    hipLaunchKernelGGL(k_l27_fill<int64_t>, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, G, (const int64_t*)A->rowptr, A->index_base,
                       vals, buf, d_stats);
  else
    hipLaunchKernelGGL(k_l27_fill<int32_t>, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, G, (const int32_t*)A->rowptr, A->index_base,
                       vals, buf, d_stats);
  }
  MFEM_CHECK_LAUNCH();
  MFEM_CHECK_HIP(hipMemcpyAsync(ctx->h_flags + 12, d_stats, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
  MFEM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  double amax;
  memcpy(&amax, ctx->h_flags + 14, sizeof(double));
  Lat27Bind B{buf, buf + lat27_vals_doubles(G), vals};
  lat27_probe_rebind(A, &B);
  double asym = 1.0;
  int rc = mfem_sym_probe(ctx, A, vals, scratch, amax, lat27_probe_unbind, lat27_probe_rebind, &B, &asym, allow_rem ? 1 : 0);
  A->lat27_asym = asym;
  if (rc || !(asym <= 4e-13)) {  // not symmetric (or NaN): the sliced layout serves this solve
    mfem_lat27_unbind(A);
    return rc;
  }
  A->lat27_dsc = dsc;
  A->lat27_scaled = dsc ? 1 : 0;
  return MFEM_OK;
}

bool mfem_lat27_bound(const mfem_csr_s* A, const double* vals) { return A->lat27_vals && vals == A->lat27_src; }

void mfem_lat27_unbind(mfem_csr_s* A) {
  if (A->lat27_vals) A->rem_active = 0;  // (the remainder belongs to the bind)
  A->lat27_vals = nullptr;
  A->lat27_dump = nullptr;
  A->lat27_src = nullptr;
  A->lat27_dsc = nullptr;
}

// returns 1 if launched, 0 if another kernel should be used, <0 on error
int mfem_spmv_lat27_launch(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, const double* x, double* y, double alpha,
                           double beta, const double* dotw, double* partials, int* n_partials, const int32_t* done_flag, int part) {
  if (!A->lat27_vals || vals != A->lat27_src) return 0;
  if (n_partials) *n_partials = 0;
  const Lat27Geom G = lat27_geom(A);
  const int ntiles = G.nti * G.ntj * G.ntk;
  // Split SpMV of a slab (mfem_spmv_halo): part 1 = the tiles that stage no ghost plane of the upper neighbour (the i-layers below
  // mfem_lat_first_ghost_layer: a contiguous prefix of the i-major tile list), launched beside the halo exchange; part 2 = the remaining layers and
  // the gather pass (which reads the lower ghost planes for the first owned rows), launched after it.  part 0 = everything.
  const int tb = mfem_lat_first_ghost_layer(G.m0, G.gw, G.nti, G.plo + G.m0 < G.mg) * G.ntj * G.ntk;
  const int tile0 = part == 2 ? tb : 0;
  const int tcount = part == 1 ? tb : ntiles - tile0;
  const int chunk = (tcount + 7) / 8;
  if (!y) {
    // the fused CG iteration (mfem_lat27_cg_fused): pass 1 alone, the tiles' blocks stay in the dump for mfem_lat27_gather_cg_update, x . A x comes out
    // as one partial per tile behind the dump
    MFEM_REQUIRE(part == 0 && dotw == x && alpha == 1.0 && beta == 0.0 && !A->lat27_dsc, "lattice tiles: pass 1 alone serves only the fused CG iteration");
    double* dotp = A->lat27_dump + lat27_dump_doubles(G);
    if (A->lat27_det)
      hipLaunchKernelGGL(k_spmv_lat27d, dim3(8 * chunk), dim3(512), 0, ctx->stream, G, A->lat27_vals, x, (const double*)nullptr, A->lat27_dump, done_flag, 0,
                         ntiles, dotp);
    else
      hipLaunchKernelGGL(k_spmv_lat27, dim3(8 * chunk), dim3(512), 0, ctx->stream, G, A->lat27_vals, x, (const double*)nullptr, A->lat27_dump, done_flag, 0,
                         ntiles, dotp);
    MFEM_CHECK_LAUNCH();
    (void)partials;  // (the caller reads the partials where mfem_lat27_dot_partials says)
    if (n_partials) *n_partials = ntiles;
    if (!ctx->probe_active) ++g_lat27_count;
    return 1;
  }
  if (tcount > 0) {
    if (A->lat27_det)
      hipLaunchKernelGGL(k_spmv_lat27d, dim3(8 * chunk), dim3(512), 0, ctx->stream, G, A->lat27_vals, x, A->lat27_dsc, A->lat27_dump, done_flag, tile0, tcount,
                         (double*)nullptr);
    else
      hipLaunchKernelGGL(k_spmv_lat27, dim3(8 * chunk), dim3(512), 0, ctx->stream, G, A->lat27_vals, x, A->lat27_dsc, A->lat27_dump, done_flag, tile0, tcount,
                         (double*)nullptr);
    MFEM_CHECK_LAUNCH();
  }
  if (part == 1) return 1;  // (the gather pass belongs to part 2)
  // persistent grid = what is resident (mfem_resident_per_cu; the staged gather holds 3 workgroups per CU at 145 VGPRs, the plain one 6: both were launched with 8)
  int grid = 1;
#define L27_GATHER(KERNEL, RP)                                                                                                              \
  do {                                                                                                                                      \
    int cap = ctx->num_cus * mfem_resident_per_cu(reinterpret_cast<const void*>(&KERNEL<RP>), MFEM_BLOCK, 0, 3);                            \
    if (cap > MFEM_MAX_PARTIALS) cap = MFEM_MAX_PARTIALS;                                                                                   \
    grid = ntiles < cap ? ntiles : cap;                                                                                                     \
    hipLaunchKernelGGL(KERNEL<RP>, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, G, A->lat27_dump, y, alpha, beta, dotw, partials, done_flag, \
                       (const RP*)A->rowptr, A->index_base, A->lat27_src, x, A->lat27_dsc);                                                 \
  } while (0)
  if (g_lat27_gather_staged) {
    if (A->rowptr_bits == 64) L27_GATHER(k_lat27_gather_st, int64_t); else L27_GATHER(k_lat27_gather_st, int32_t);
  } else {
    if (A->rowptr_bits == 64) L27_GATHER(k_lat27_gather, int64_t); else L27_GATHER(k_lat27_gather, int32_t);
  }
#undef L27_GATHER
  MFEM_CHECK_LAUNCH();
  if (n_partials && partials) *n_partials = grid;
  if (A->rem_active) {  // A = S + N: the skew remainder of the few nonsymmetric rows (spmv_rem.hip)
    const int rcr = mfem_rem_apply(ctx, A, x, A->lat27_dsc, y, alpha, dotw, partials, n_partials, done_flag);
    if (rcr) return rcr;
  }
  if (!ctx->probe_active) ++g_lat27_count;
  return 1;
}

// bytes one SpMV of the layout moves by design: the stored entries, x as the tiles stage it, the y blocks written and read again, y
int64_t mfem_lat27_design_bytes(const mfem_csr_s* A) {
  const Lat27Geom G = lat27_geom(A);
  const int64_t tiles = (int64_t)G.nti * G.ntj * G.ntk;
  return (int64_t)lat27_read_doubles(G) * 8 + tiles * L27_CELLS * 8 * (A->lat27_scaled ? 4 : 3) + A->n * 8 + mfem_rem_design_bytes(A);
}
int64_t mfem_lat27_entries(const mfem_csr_s* A) { return (int64_t)lat27_read_doubles(lat27_geom(A)); }

// ---- the fused CG iteration (krylov.hip, cg_solve_pass): pass 1 alone (mfem_spmv_halo with y = nullptr), the dot-product partials, pass 2 + residual update
bool mfem_lat27_cg_fused(const mfem_context_s* ctx, const mfem_csr_s* A, const double* vals) {
  return g_lat27_cg_fused && mfem_lat27_bound(A, vals) && !A->lat27_dsc && !ctx->comm && !A->rem_active;
}
const double* mfem_lat27_dot_partials(const mfem_csr_s* A, int* np) {
  const Lat27Geom G = lat27_geom(A);
  *np = G.nti * G.ntj * G.ntk;
  return A->lat27_dump + lat27_dump_doubles(G);
}
int mfem_lat27_gather_cg_update(mfem_context_s* ctx, mfem_csr_s* A, const LatCgUpdate& U, int grid) {
  const Lat27Geom G = lat27_geom(A);
  hipLaunchKernelGGL(k_lat27_gather_cg, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, G, (const double*)A->lat27_dump, U);
  MFEM_CHECK_LAUNCH();
  return MFEM_OK;
}
// (accounting for bench.py) is the fused CG iteration on, and what pass 1 alone moves by design: the stored entries, x as the tiles stage it, the y blocks written
extern "C" int mfem_debug_lat27_cg_fused(void) { return g_lat27_cg_fused; }
extern "C" int64_t mfem_debug_lat27_pass1_bytes(mfem_csr A) {
  if (!A || A->lat27_state != 1) return -1;
  const Lat27Geom G = lat27_geom(A);
  return (int64_t)lat27_read_doubles(G) * 8 + (int64_t)G.nti * G.ntj * G.ntk * L27_CELLS * 8 * 2;
}
