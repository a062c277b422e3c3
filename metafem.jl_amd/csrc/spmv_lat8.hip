// Symmetric lattice-tile layout for the Krylov loop on the F-field 27-point lattice matrix, F = 1..3 (hex-8; field-major rows
// row = field * N + node; F = 3: linear elasticity, cantilever/3D_Script.jl's system): solver layout mode 5.  The caller-facing contract stays CSR (mul!,
// misc/04_GPU_Utils.jl:131; iterative_Solve!, linear_solver/02_Preconditioner.jl:32-76).  Companion of spmv_lat27.hip (mode 4).
//
// The diagonal-slotted layout (mode 2) streams all 81 entries of a row.  The stiffness matrix is symmetric (the penalty and traction terms keep
// it so; the right Jacobi scaling A D^-1 of bicgstabl_GS! / idrs! is applied to x while it is staged, so the stored matrix stays A), so per node
// only the 6 upper entries of its own 3 x 3 block and the 3 x 3 blocks towards its 13 upper lattice neighbours are stored: 123 of 243 values.
//
//   * layout (per solve): a unit is 4 x 4 x 4 nodes = the 64 lanes of a wave, lane = node; its 123 steps (+ 1 padding step) are wave-uniform:
//     step s is entry (f, node) -> (g, node + d) for every lane, so the column offsets are immediates and nothing is looked up.  A unit is
//     124 x 512 B, unit-stride.
//   * SpMV pass 1 (k_spmv_lat8): a workgroup owns a tile of 8 x 8 x 16 nodes (16 units, 8 waves), stages x of the tile and of the (+1, +-1, +-1)
//     neighbourhood for the three fields in LDS (3 x 1 620 cells) and accumulates y in a second block of the same shape: for a stored entry
//     a = A[(f, p)][(g, p')] the lane adds a x[g][p'] to its register sum of row (f, p) and a x[f][p] to cell (g, p') (ds_add_f64).  The y
//     block leaves as one contiguous run per tile.
//   * pass 2 (k_lat8_gather) sums the up to 18 tile blocks that cover a row in a fixed order, applies alpha / beta and the fused dot product.
//   * the pattern must be the full stencil (checked entry by entry once), the values of a solve symmetric (a probe product of the layout against
//     the CSR kernel, mfem_sym_probe in spmv_lat27.hip, within 4e-13 of the largest entry; the diagonal-slotted layout serves the solve otherwise).  Results equal the CSR kernel's to round-off, not
//     bitwise, and not bitwise from run to run (order of the LDS adds of different waves).  mfem_debug_set_lat8(0) switches the layout off.
#include "blas1.h"
#include "spmv_lat_tables.h"

#define L8_TI 8
#define L8_TJ 8
#define L8_TK 16
#define L8_SJ (L8_TJ + 2)
#define L8_SK (L8_TK + 2)
#define L8_PI (L8_SJ * L8_SK)            // 180
#define L8_FC ((L8_TI + 1) * L8_PI)      // cells per field: 1620

typedef double m_d2 __attribute__((ext_vector_type(2)));

extern std::atomic<int64_t> g_layout_min_rows_dia;  // spmv_ell.hip
static std::atomic<int> g_lat8_enable{1};
static std::atomic<int> g_lat8_gather_staged{0};  // bit 2 of mfem_debug_set_lat8: 1 = pass 2 by k_lat8_gather_st (measured on C3, tools/gather_ab.py: 1.6 % SLOWER per solve than k_lat8_gather -- its two barriers and the LDS round trip cost more than the round trips it saves on these small tiles; the hex-27 tiles gain 1 %)
static std::atomic<int> g_lat8_one_field_everywhere{0};  // bit 1 of mfem_debug_set_lat8: the query / diagnostic SpMV entry also report and take mode 5 for ONE field
static std::atomic<long long> g_lat8_count{0};
extern "C" long long mfem_debug_lat8_spmv_count(void) { return g_lat8_count; }
extern "C" double mfem_debug_lat8_asymmetry(mfem_csr A) { return A ? A->lat8_asym : -1.0; }
extern "C" int mfem_debug_set_lat8(int enable) try {
  ++mfem_debug_epoch;
  g_lat8_enable = enable & 1;
  g_lat8_one_field_everywhere = (enable >> 1) & 1;
  g_lat8_gather_staged = (enable >> 2) & 1;
  return MFEM_OK;
} MFEM_API_CATCH("mfem_debug_set_lat8")
// One field: cg! keeps the bitwise patch sweep of mode 2 (it moves the same bytes); the solvers that work on A D^-1 (idrs!, bicgstabl_GS!, cgs2!) cannot
// use that sweep -- the scaled copy is not symmetric -- and take the tiles.  The layout query and the diagnostic SpMV entry answer for cg!.
bool mfem_lat8_for_method(const mfem_csr_s* A, bool is_cg) { return A->lat_fields != 1 || !is_cg || g_lat8_one_field_everywhere; }

struct Lat8Geom {
  int m0, m1, m2;     // OWNED nodes per direction (m0 = owned lattice planes of a slab)
  int nui, nuj, nuk;  // units of 4 x 4 x 4 nodes
  int nti, ntj, ntk;  // tiles of 8 x 8 x 16 nodes
  int64_t N;          // m0 * m1 * m2 owned nodes
  // slab: the owned planes are [plo, plo + m0) of a lattice of mg planes; x carries, behind the F N owned entries, per field a low and a high
  // block of gw ghost planes (brick_xindex); plo = 0, mg = m0 for a whole brick
  int plo, mg, gw, F;
};
// local x index of field f at GLOBAL plane gi (owned or ghost), in-plane position ip
__device__ __forceinline__ int64_t l8_xindex(const Lat8Geom& G, int f, int gi, int64_t ip) {
  const int64_t PL = (int64_t)G.m1 * G.m2;
  if (gi >= G.plo && gi < G.plo + G.m0) return f * G.N + (int64_t)(gi - G.plo) * PL + ip;
  const int side = gi < G.plo ? 0 : 1;
  const int off = side ? gi - (G.plo + G.m0) : gi - (G.plo - G.gw);
  return (int64_t)G.F * G.N + ((int64_t)(f * 2 + side) * G.gw + off) * PL + ip;
}

// ---- the step list of a unit (compile-time): spmv_lat_tables.h
__host__ __device__ constexpr int l8_off(int e) { return l8_di(e) * L8_PI + l8_dj(e) * L8_SK + l8_dk(e); }

// offsets a node at lattice coordinate g (of m) has along one direction: [lo, lo + cnt)
__device__ __forceinline__ void l8_range(int g, int m, int& lo, int& cnt) {
  lo = g > 0 ? -1 : 0;
  cnt = (g < m - 1 ? 1 : 0) - lo + 1;
}

// 1 in *bad if some row is not the F-field stencil row: F x (present neighbours), columns field-major then lexicographic
template <typename RP>
__global__ __launch_bounds__(MFEM_BLOCK) void k_l8_verify(Lat8Geom G, const RP* __restrict__ rowptr, const int32_t* __restrict__ col, int base,
                                                            int32_t* __restrict__ bad) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, PL = (int64_t)G.m1 * G.m2;
  int fail = 0;
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < G.F * G.N; r += stride) {
    const int64_t p = r % G.N;
    const int gi = (int)(p / PL) + G.plo;  // global plane
    const int64_t rem = p % PL;
    const int gj = (int)(rem / G.m2), gk = (int)(rem - (int64_t)gj * G.m2);
    int li, ni, lj, nj, lk, nk;
    l8_range(gi, G.mg, li, ni);
    l8_range(gj, G.m1, lj, nj);
    l8_range(gk, G.m2, lk, nk);
    const int64_t lo = (int64_t)rowptr[r] - base, hi = (int64_t)rowptr[r + 1] - base;
    if (hi - lo != (int64_t)G.F * ni * nj * nk) {
      fail = 1;
      continue;
    }
    int64_t j = lo;
    for (int g = 0; g < G.F; ++g)
      for (int a = 0; a < ni; ++a)
        for (int b = 0; b < nj; ++b) {
          const int64_t c0 = l8_xindex(G, g, gi + li + a, (int64_t)(gj + lj + b) * G.m2 + gk + lk);
          for (int c = 0; c < nk; ++c, ++j)
            if ((int64_t)col[j] - base != c0 + c) fail = 1;
        }
  }
  if (fail) bad[0] = 1;
}

// The layout pass: a wave per unit, lane = node.  stats[1] = max |a| over the stored entries (bit pattern).
// The steps of a unit are compile-time constants here too (template recursion): the entries of one row field are requested back to back (a lane's
// F runs of 13 - 14 consecutive CSR entries: each cache line is touched while it is still in flight), then leave as 16-byte stores.
struct L8Row {  // what a lane knows about its node
  bool valid;
  int gi, gj, gk, li, lj, lk, nj, nk, cnt;
  int64_t rp[3];
};
// Round 5: the fill reads the caller's CSR values through LDS.  A lane's entries sit in its own row, 216 F bytes from the next lane's: read straight from memory every
// load instruction touched 64 different lines (3.4 ms for the 5.5 GB of the 256^3 one-field copy = 1.6 TB/s: round 4's weakest per-solve pass).  The rows of four
// k-consecutive nodes are ONE contiguous piece of the CSR value array, so the wave copies the four pieces of an i-layer of its unit (row field f) into LDS with unit-stride
// loads and the 16 lanes of that layer pick their entries there; four layers per field.  `src` = the lane's row in the staged copy (valid in the lane's own phase).
template <int F, int S, int SEND>
__device__ __forceinline__ void l8_fetch(const L8Row& R, const Lat8Geom& G, const double* src, bool active, double* v, double& amax) {
  if constexpr (S < SEND) {
    if constexpr (S < l8_nsteps(F)) {
      constexpr int g = l8_g(F, S), e = l8_e(F, S);
      constexpr int di = l8_di(e), dj = l8_dj(e), dk = l8_dk(e);
      const int ci = R.gi + di, cj = R.gj + dj, ck = R.gk + dk;
      if (active) {
        double val = 0.0;
        if (R.valid && ci < G.mg && cj >= 0 && cj < G.m1 && ck >= 0 && ck < G.m2) {  // (the neighbour may sit in a ghost plane: its x comes from the ghost block)
          val = src[g * R.cnt + ((di - R.li) * R.nj + (dj - R.lj)) * R.nk + (dk - R.lk)];
          double av = fabs(val);
          if (!(av == av)) av = __builtin_huge_val();  // NaN: fmax would drop it
          amax = fmax(amax, av);
        }
        v[0] = val;
      }
    } else {
      if (active) v[0] = 0.0;  // (the padding step)
    }
    l8_fetch<F, S + 1, SEND>(R, G, src, active, v + 1, amax);
  }
}
// steps [S0, S0 + N) from v to their places: the two i-stacked units of a wave of pass 1 are stored as ONE stream of 2 x nsteps steps (l8_stream: phase-major,
// unit 0 before unit 1 inside a phase) -- the wave reads its 2 x 63 KB front to back --, a pair of stream steps per lane side by side (16-byte loads):
// step v of lane l at doubles ((v / 2) * 64 + l) * 2 + (v & 1).  ou = the pair's base + 2 * lane, h = which unit of the pair this is.
template <int F, int S0, int I, int N>
__device__ __forceinline__ void l8_put(double* __restrict__ ou, int h, const double* v) {
  if constexpr (I < N) {
    constexpr L8Order O = l8_order(F);
    constexpr L8Stream T = l8_stream(F);
    constexpr int v0 = T.at[0][O.pos[S0 + I]], v1 = T.at[1][O.pos[S0 + I]];
    const int vv = h ? v1 : v0;
    ou[(int64_t)(vv >> 1) * 128 + (vv & 1)] = v[I];
    l8_put<F, S0, I + 1, N>(ou, h, v);
  }
}
#define L8_SEG (4 * 27)  // entries of the four rows of a staged piece, per field of columns
// the steps of row field f: [first(f), first(f + 1))
template <int F, int f>
__device__ __forceinline__ void l8_fill_field(const L8Row& R, const Lat8Geom& G, const double* __restrict__ vals, double* stage, double* __restrict__ ou,
                                              int h, double& amax) {
  if constexpr (f < F) {
    constexpr int S0 = l8_first(F, f), S1 = l8_first(F, f + 1);
    constexpr int lead = 0;
    constexpr int cnt = S1 - S0;                 // values in v
    constexpr int SEGCAP = F * L8_SEG + 4;       // doubles per staged piece (+ padding: the four pieces start on different banks)
    double v[cnt];
    const int lane = threadIdx.x & 63;
    // the piece a lane's row belongs to: rows of the four lanes (la, lb, 0..3); its start = the first lane's row, its end = the end of the last valid row
    const int64_t rowlen = R.valid ? (int64_t)F * R.cnt : 0;
    const int64_t rstart = R.valid ? R.rp[f] : ((int64_t)1 << 62), rend = R.valid ? R.rp[f] + rowlen : 0;
    int64_t ps = rstart, pe = rend;
#pragma unroll
    for (int o = 1; o <= 2; o <<= 1) {
      const int64_t os = __shfl_xor(ps, o, MFEM_WAVE), oe = __shfl_xor(pe, o, MFEM_WAVE);
      ps = os < ps ? os : ps;
      pe = oe > pe ? oe : pe;
    }
    const int plen = pe > ps ? (int)(pe - ps) : 0;  // (0: no valid row in the piece)
    const int lb = (lane >> 2) & 3;
    const double* src = stage + lb * SEGCAP + (R.valid ? (int)(rstart - ps) : 0);
#pragma unroll 1
    for (int la = 0; la < 4; ++la) {
      // the four pieces of layer la: start / length are those of lanes 16 la + 4 lb (wave-uniform through readlane)
      double t[4][(F * L8_SEG + 63) / 64];
      int len[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int sl = 16 * la + 4 * b;
        const uint32_t lo32 = __builtin_amdgcn_readlane((uint32_t)(uint64_t)ps, sl), hi32 = __builtin_amdgcn_readlane((uint32_t)((uint64_t)ps >> 32), sl);
        const int64_t st = (int64_t)(((uint64_t)hi32 << 32) | lo32);
        len[b] = __builtin_amdgcn_readlane(plen, sl);
#pragma unroll
        for (int u = 0; u < (F * L8_SEG + 63) / 64; ++u) {
          const int i = lane + 64 * u;
          t[b][u] = i < len[b] ? __builtin_nontemporal_load(vals + st + i) : 0.0;
        }
      }
      __builtin_amdgcn_wave_barrier();  // (the previous layer's reads of the staged copy are done)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int u = 0; u < (F * L8_SEG + 63) / 64; ++u) {
          const int i = lane + 64 * u;
          if (i < len[b]) stage[b * SEGCAP + i] = t[b][u];
        }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_wave_barrier();
      l8_fetch<F, S0, S1>(R, G, src, (lane >> 4) == la, v + lead, amax);
    }
    l8_put<F, S0, 0, cnt>(ou, h, v);
    l8_fill_field<F, f + 1>(R, G, vals, stage, ou, h, amax);
  }
}
template <typename RP, int F>
__global__ __launch_bounds__(MFEM_BLOCK) void k_l8_fill(Lat8Geom G, const RP* __restrict__ rowptr, int base, const double* __restrict__ vals,
                                                          double* __restrict__ out, unsigned long long* __restrict__ stats) {
  __shared__ double stage_all[MFEM_BLOCK / 64][4 * (F * L8_SEG + 4)];
  const int lane = threadIdx.x & 63;
  double* stage = stage_all[threadIdx.x >> 6];
  const int la = lane >> 4, lb = (lane >> 2) & 3, lc = lane & 3;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int64_t nunits = (int64_t)G.nui * G.nuj * G.nuk;
  double amax = 0.0;
  for (int64_t u = wave; u < nunits; u += nwaves) {
    const int uk = (int)(u % G.nuk);
    const int64_t u2 = u / G.nuk;
    const int uj = (int)(u2 % G.nuj), ui = (int)(u2 / G.nuj);
    const int oi = ui * 4 + la;  // owned plane
    L8Row R;
    R.gi = oi + G.plo;
    R.gj = uj * 4 + lb;
    R.gk = uk * 4 + lc;
    R.valid = oi < G.m0 && R.gj < G.m1 && R.gk < G.m2;
    const int64_t p = ((int64_t)oi * G.m1 + R.gj) * G.m2 + R.gk;
    int ni = 1;
    R.li = R.lj = R.lk = 0;
    R.nj = R.nk = 1;
    R.rp[0] = R.rp[1] = R.rp[2] = 0;
    if (R.valid) {
      l8_range(R.gi, G.mg, R.li, ni);
      l8_range(R.gj, G.m1, R.lj, R.nj);
      l8_range(R.gk, G.m2, R.lk, R.nk);
#pragma unroll
      for (int f = 0; f < F; ++f) R.rp[f] = (int64_t)rowptr[f * G.N + p] - base;
    }
    R.cnt = ni * R.nj * R.nk;
    const int64_t pair = ((int64_t)(ui >> 1) * G.nuj + uj) * G.nuk + uk;
    l8_fill_field<F, 0>(R, G, vals, stage, out + pair * (int64_t)(2 * l8_nsteps(F) * 64) + lane * 2, ui & 1, amax);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amax = fmax(amax, __shfl_down(amax, o, MFEM_WAVE));
  if (lane == 0) atomicMax(stats + 1, (unsigned long long)__double_as_longlong(amax));
}

#define L8_LDS_ADD(ptr, val) __builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double*)(ptr), (val))

// ---- pass 1, deterministic (round 6; the order tables: spmv_lat_tables.h).  The two units of a wave are ONE stream of 2 x nsteps steps, phase-major (a
// phase = the (dj, dk) of the steps' offset; inside a phase unit 0's steps, then unit 1's), worked through in chunks of 8 steps from two register buffers
// that swap roles: the values of chunk k + 1 are in flight while chunk k goes through LDS, barriers or not.  Between two phases the workgroup meets at an
// LDS-only barrier (mfem_lds_barrier: the loads in flight stay in flight).  Inside a phase a cell receives mirrored products from ONE wave -- the one that
// owns the node column (j - dj, k - dk) --, in that wave's program order, so every LDS sum, and with it y, has ONE order: bitwise the same from run to run.
// Until round 5 the waves added concurrently (ds_add_f64 across waves: ~1e-16 relative, not bitwise -- and IDR(8) / BiCGStab(2) iteration counts on C3
// swung by 30 % from that round-off alone).
template <int F, int CH, int V0, int N>
__device__ __forceinline__ void l8_load(double (&v)[CH], const double* __restrict__ sb) {
  static_assert(V0 % 2 == 0 && N % 2 == 0 && CH % 2 == 0, "stream steps leave in pairs");
#pragma unroll
  for (int i = 0; i < N; i += 2) {
    const m_d2 pr = __builtin_nontemporal_load((const m_d2*)sb + (int64_t)((V0 + i) >> 1) * 64);
    v[i] = pr.x;
    v[i + 1] = pr.y;
  }
}

// steps [V0 + I, V0 + N) of the stream from the register buffer v; xo[h][f] = the lane's own x of unit h, acc[h][f] = its row sums.  Everything about a
// step is a compile-time constant (template recursion, not a loop: the row sums must stay in registers).
template <int F, int CH, int V0, int I, int N>
__device__ __forceinline__ void l8_proc(const double (&v)[CH], const int (&pos)[2], const bool (&act)[2], const double (&xo)[2][3], double (&acc)[2][3],
                                        const double* xs, double* ys) {
  if constexpr (I < N) {
    constexpr L8Stream T = l8_stream(F);
    constexpr L8Order O = l8_order(F);
    constexpr int vv = V0 + I, h = T.unit[vv], s = O.step[T.pos[vv]];
    constexpr int f = l8_row_field(F, s), g = l8_g(F, s), e = l8_e(F, s);
    constexpr int coff = g * L8_FC + l8_off(e);
    // (measured, profiles/r06_lat8_deterministic.txt: neither the eight barriers nor the wave-uniform branches cost time -- what did was reading the two
    //  units as two interleaved far-apart runs of 8-byte loads: +8 % on C3; stored as ONE stream per wave, 16 bytes per lane, the kernel is back at round 5's time)
    if constexpr (vv > 0 && T.phase[vv] != T.phase[vv > 0 ? vv - 1 : 0]) mfem_lds_barrier();  // every wave of the workgroup passes here, whatever it owns
    if (act[h]) {  // (wave-uniform)
      const double a = v[I];
      acc[h][f] += a * xs[pos[h] + coff];
      if constexpr (!(e == 0 && g == f)) L8_LDS_ADD(ys + pos[h] + coff, a * xo[h][f]);  // (the diagonal entry has no mirror)
    }
    l8_proc<F, CH, V0, I + 1, N>(v, pos, act, xo, acc, xs, ys);
  }
}

// chunk C (CH steps; the last one what is left) from one buffer while chunk C + 1 is loaded into the other
template <int F, int CH, int C>
__device__ __forceinline__ void l8_run(double (&A)[CH], double (&B)[CH], const double* __restrict__ ub, const int (&pos)[2], const bool (&act)[2],
                                       const double (&xo)[2][3], double (&acc)[2][3], const double* xs, double* ys) {
  constexpr int NV = 2 * l8_nsteps(F), NCH = (NV + CH - 1) / CH, LAST = NCH - 1;
  constexpr int nthis = (C == LAST) ? NV - CH * LAST : CH;
  if constexpr (C < LAST) {
    constexpr int nnext = (C + 1 == LAST) ? NV - CH * LAST : CH;
    l8_load<F, CH, (C + 1) * CH, nnext>((C & 1) ? A : B, ub);
  }
  __builtin_amdgcn_sched_barrier(0);  // (keeps the scheduler from hoisting the LDS reads of later chunks: spills without)
  l8_proc<F, CH, C * CH, 0, nthis>((C & 1) ? B : A, pos, act, xo, acc, xs, ys);
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int f = 0; f < F; ++f) asm volatile("" : "+v"(acc[h][f]));  // the row sums are due HERE (the compiler otherwise sinks the whole chain of
  __builtin_amdgcn_sched_barrier(0);                                   // multiply-adds to the end and keeps every value and x it needs alive until then)
  if constexpr (C < LAST) l8_run<F, CH, C + 1>(A, B, ub, pos, act, xo, acc, xs, ys);
}

// pass 1: one workgroup per tile; dump[tile][field][cell]
template <int F, int CH>
__global__ __launch_bounds__(512, 4) void k_spmv_lat8(Lat8Geom G, const double* __restrict__ vals, const double* __restrict__ x,
                                                      const double* __restrict__ dsc, double* __restrict__ dump,
                                                      const int32_t* __restrict__ done_flag, int tile0, int tcount) {
  constexpr int NV = 2 * l8_nsteps(F), PAIR_D = NV * 64;
  __shared__ double xs[F * L8_FC];
  __shared__ double ys[F * L8_FC];
  if (done_flag && done_flag[0]) return;
  // (this launch covers the tiles [tile0, tile0 + tcount) of the i-major tile list: all of them, or the interior / boundary part of a slab's SpMV)
  const int chunk = (tcount + 7) >> 3;
  const int tsub = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);  // every XCD walks a contiguous eighth of the tiles
  if ((int)(blockIdx.x >> 3) >= chunk || tsub >= tcount) return;           // (the whole workgroup leaves: no barrier is left waiting)
  const int tile = tile0 + tsub;
  const int tk = tile % G.ntk, t2 = tile / G.ntk, tj = t2 % G.ntj, ti = t2 / G.ntj;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int la = lane >> 4, lb = (lane >> 2) & 3, lc = lane & 3;
  // the wave's two units: (0, b, c) and (1, b, c) of the tile's 2 x 2 x 4
  const int ub_ = wv >> 2, uc = wv & 3;
  const int ui = ti * 2, uj = tj * 2 + ub_, uk = tk * 4 + uc;
  const bool act[2] = {uj < G.nuj && uk < G.nuk, uj < G.nuj && uk < G.nuk && ui + 1 < G.nui};
  // a pair that does not exist (lattice edge) is read from a place that does -- every load on every path -- and not worked on; the second unit of a
  // pair cut by the lattice has its (unwritten) place in the stream: read, not worked on
  const double* ub = vals + (act[0] ? (((int64_t)ti * G.nuj + uj) * G.nuk + uk) * PAIR_D : 0) + lane * 2;
  double A[CH], B[CH];
  l8_load<F, CH, 0, (NV < CH ? NV : CH)>(A, ub);  // in flight while x is staged
  const int i0 = ti * L8_TI, j0 = tj * L8_TJ - 1, k0 = tk * L8_TK - 1;
  for (int e = tid; e < F * L8_FC; e += 512) {
    const int f = e / L8_FC, c = e - f * L8_FC;
    const int li = c / L8_PI, r2 = c - li * L8_PI, lj = r2 / L8_SK, lk = r2 - lj * L8_SK;
    const int gi = G.plo + i0 + li, gj = j0 + lj, gk = k0 + lk;  // global plane: the plane behind the last owned one is a ghost plane
    double xv = 0.0;
    if (gi < G.mg && gi <= G.plo + G.m0 && gj >= 0 && gj < G.m1 && gk >= 0 && gk < G.m2) {
      const int64_t r = l8_xindex(G, f, gi, (int64_t)gj * G.m2 + gk);
      xv = dsc ? x[r] / dsc[r] : x[r];
    }
    xs[e] = xv;
    ys[e] = 0.0;
  }
  __syncthreads();
  {
    const int p0 = la * L8_PI + (ub_ * 4 + lb + 1) * L8_SK + (uc * 4 + lc + 1);
    const int pos[2] = {p0, p0 + 4 * L8_PI};
    double xo[2][3], acc[2][3];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int f = 0; f < 3; ++f) {
        xo[h][f] = f < F ? xs[f * L8_FC + pos[h]] : 0.0;
        acc[h][f] = 0.0;
      }
    l8_run<F, CH, 0>(A, B, ub, pos, act, xo, acc, xs, ys);
    // the row sums: still phase 8 (own cells; the other adds into them in this phase come from this wave)
#pragma unroll
    for (int h = 0; h < 2; ++h)
      if (act[h]) {
#pragma unroll
        for (int f = 0; f < F; ++f) L8_LDS_ADD(ys + pos[h] + f * L8_FC, acc[h][f]);
      }
  }
  __syncthreads();
  double* dt = dump + (int64_t)tile * (F * L8_FC);
  for (int e = tid; e < F * L8_FC; e += 512) dt[e] = ys[e];
}

// pass 2: a thread owns a (j, k) position of the tile and four of its planes, for the F fields
// Slab with a lower neighbour (G.plo > 0): the rows of the first owned plane also have entries towards the ghost plane below.  No stored entry mirrors
// onto them (the rows that would belong to the neighbour rank), so they are taken from the caller's CSR values here: 9 F products per row of that plane.
template <typename RP, int F>
__global__ __launch_bounds__(MFEM_BLOCK) void k_lat8_gather(Lat8Geom G, const double* __restrict__ dump, double* __restrict__ y, double alpha,
                                                              double beta, const double* __restrict__ dotw, double* __restrict__ partials,
                                                              const int32_t* __restrict__ done_flag, const RP* __restrict__ rowptr, int base,
                                                              const double* __restrict__ csr_vals, const double* __restrict__ x,
                                                              const double* __restrict__ dsc) {
  __shared__ double red[4];
  if (done_flag && done_flag[0]) return;
  double dot_acc = 0.0;
  const int ntiles = G.nti * G.ntj * G.ntk;
  const int lk = threadIdx.x & (L8_TK - 1), lj = (threadIdx.x >> 4) & (L8_TJ - 1), lh = threadIdx.x >> 7;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int tk = tile % G.ntk, t2 = tile / G.ntk, tj = t2 % G.ntj, ti = t2 / G.ntj;
    const int gj = tj * L8_TJ + lj, gk = tk * L8_TK + lk, gi0 = ti * L8_TI + 4 * lh;
    if (gj >= G.m1 || gk >= G.m2) continue;
    double s[F][4];
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
      for (int u = 0; u < 4; ++u) s[f][u] = 0.0;
    for (int b = -1; b <= 1; ++b) {
      if ((b < 0 && (lj >= 1 || tj == 0)) || (b > 0 && (lj < L8_TJ - 1 || tj == G.ntj - 1))) continue;
      for (int c = -1; c <= 1; ++c) {
        if ((c < 0 && (lk >= 1 || tk == 0)) || (c > 0 && (lk < L8_TK - 1 || tk == G.ntk - 1))) continue;
        const int cell = (lj - L8_TJ * b + 1) * L8_SK + (lk - L8_TK * c + 1);
        if (ti > 0 && lh == 0) {  // the tile below: its plane 8 is this tile's plane 0
          const double* d = dump + (((int64_t)(ti - 1) * G.ntj + (tj + b)) * G.ntk + (tk + c)) * (F * L8_FC) + cell + 8 * L8_PI;
#pragma unroll
          for (int f = 0; f < F; ++f) s[f][0] += d[f * L8_FC];
        }
        const double* d = dump + (((int64_t)ti * G.ntj + (tj + b)) * G.ntk + (tk + c)) * (F * L8_FC) + cell + 4 * lh * L8_PI;
#pragma unroll
        for (int f = 0; f < F; ++f)
#pragma unroll
          for (int u = 0; u < 4; ++u) s[f][u] += d[f * L8_FC + u * L8_PI];
      }
    }
    if (G.plo > 0 && ti == 0 && lh == 0) {  // the lower ghost plane (see above)
      int l1, n1, l2, n2, l0, n0;
      l8_range(G.plo, G.mg, l0, n0);
      l8_range(gj, G.m1, l1, n1);
      l8_range(gk, G.m2, l2, n2);
      const int cnt = n0 * n1 * n2;
#pragma unroll
      for (int f = 0; f < F; ++f) {
        const int64_t rp = (int64_t)rowptr[f * G.N + (int64_t)gj * G.m2 + gk] - base;
        double acc = 0.0;
        for (int g = 0; g < F; ++g)
          for (int b = 0; b < n1; ++b)
            for (int c = 0; c < n2; ++c) {
              const int64_t xi = l8_xindex(G, g, G.plo - 1, (int64_t)(gj + l1 + b) * G.m2 + gk + l2 + c);
              acc += csr_vals[rp + (int64_t)g * cnt + b * n2 + c] * (dsc ? x[xi] / dsc[xi] : x[xi]);
            }
        s[f][0] += acc;
      }
    }
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (gi0 + u < G.m0) {
          const int64_t r = f * G.N + ((int64_t)(gi0 + u) * G.m1 + gj) * G.m2 + gk;
          double yv = alpha * s[f][u];
          if (beta != 0.0) yv += beta * y[r];
          y[r] = yv;
          if (dotw) dot_acc += yv * dotw[r];
        }
  }
  if (partials) {
    const double bsum = block_reduce_sum(dot_acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = bsum;
  }
}

// pass 2, staged (round 4; see k_lat27_gather_st; NOT the default here, see g_lat8_gather_staged): the F x 1 620 (row, covering block) values of a tile -- the extended box (8 + 1 planes) x (8 + 1 + 1 lines)
// x (16 + 1 + 1 columns) per field -- are copied to LDS with all loads in flight at once (19 per thread for three fields), the row owners then add them in the
// order of k_lat8_gather (bitwise the same y).
#define L8_EJ (L8_TJ + 2)
#define L8_EK (L8_TK + 2)
#define L8_EF ((L8_TI + 1) * L8_EJ * L8_EK)  // 1620 per field
template <typename RP, int F>
__global__ __launch_bounds__(MFEM_BLOCK) void k_lat8_gather_st(Lat8Geom G, const double* __restrict__ dump, double* __restrict__ y, double alpha,
                                                                 double beta, const double* __restrict__ dotw, double* __restrict__ partials,
                                                                 const int32_t* __restrict__ done_flag, const RP* __restrict__ rowptr, int base,
                                                                 const double* __restrict__ csr_vals, const double* __restrict__ x,
                                                                 const double* __restrict__ dsc) {
  constexpr int EC = F * L8_EF, EU = (EC + MFEM_BLOCK - 1) / MFEM_BLOCK;
  __shared__ double E[EC];
  __shared__ double red[4];
  if (done_flag && done_flag[0]) return;
  double dot_acc = 0.0;
  const int ntiles = G.nti * G.ntj * G.ntk;
  const int lk = threadIdx.x & (L8_TK - 1), lj = (threadIdx.x >> 4) & (L8_TJ - 1), lh = threadIdx.x >> 7;
  // extended line / column e -> (neighbour offset, line or column of this tile): 0 .. T - 1 own; T: the block before holds row 0; T + 1: the block after holds row T - 1
  auto ext = [](int e, int T, int& off, int& l) {
    if (e < T) { off = 0; l = e; }
    else if (e == T) { off = -1; l = 0; }
    else { off = 1; l = T - 1; }
  };
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {  // (the trip count is the workgroup's: every barrier below is reached by all threads)
    const int tk = tile % G.ntk, t2 = tile / G.ntk, tj = t2 % G.ntj, ti = t2 / G.ntj;
    double t[EU];
#pragma unroll
    for (int u = 0; u < EU; ++u) {
      const int e = threadIdx.x + u * MFEM_BLOCK;
      t[u] = 0.0;
      if (e < EC) {
        const int f = e / L8_EF, r1 = e - f * L8_EF, ei = r1 / (L8_EJ * L8_EK), r2 = r1 - ei * (L8_EJ * L8_EK), ej = r2 / L8_EK, ek = r2 - ej * L8_EK;
        int b, c, sj, sk;
        ext(ej, L8_TJ, b, sj);
        ext(ek, L8_TK, c, sk);
        const int a = ei < L8_TI ? 0 : -1;  // (plane 8 of the block below is this tile's plane 0: the block's plane index is ei either way)
        const bool ok = (a == 0 || ti > 0) && (b == 0 || (b < 0 ? tj > 0 : tj < G.ntj - 1)) && (c == 0 || (c < 0 ? tk > 0 : tk < G.ntk - 1));
        if (ok)
          t[u] = __builtin_nontemporal_load(dump + (((int64_t)(ti + a) * G.ntj + (tj + b)) * G.ntk + (tk + c)) * (F * L8_FC) + f * L8_FC + ei * L8_PI +
                                            (sj - L8_TJ * b + 1) * L8_SK + (sk - L8_TK * c + 1));
      }
    }
#pragma unroll
    for (int u = 0; u < EU; ++u) {
      const int e = threadIdx.x + u * MFEM_BLOCK;
      if (e < EC) E[e] = t[u];
    }
    __syncthreads();
    const int gj = tj * L8_TJ + lj, gk = tk * L8_TK + lk, gi0 = ti * L8_TI + 4 * lh;
    if (gj < G.m1 && gk < G.m2) {
      double s[F][4];
#pragma unroll
      for (int f = 0; f < F; ++f)
#pragma unroll
        for (int u = 0; u < 4; ++u) s[f][u] = 0.0;
      for (int b = -1; b <= 1; ++b) {
        if ((b < 0 && (lj >= 1 || tj == 0)) || (b > 0 && (lj < L8_TJ - 1 || tj == G.ntj - 1))) continue;
        const int ej = b == 0 ? lj : b < 0 ? L8_TJ : L8_TJ + 1;
        for (int c = -1; c <= 1; ++c) {
          if ((c < 0 && (lk >= 1 || tk == 0)) || (c > 0 && (lk < L8_TK - 1 || tk == G.ntk - 1))) continue;
          const int ek = c == 0 ? lk : c < 0 ? L8_TK : L8_TK + 1;
          const double* d = E + ej * L8_EK + ek;
          if (ti > 0 && lh == 0) {
#pragma unroll
            for (int f = 0; f < F; ++f) s[f][0] += d[f * L8_EF + 8 * (L8_EJ * L8_EK)];
          }
#pragma unroll
          for (int f = 0; f < F; ++f)
#pragma unroll
            for (int u = 0; u < 4; ++u) s[f][u] += d[f * L8_EF + (4 * lh + u) * (L8_EJ * L8_EK)];
        }
      }
      if (G.plo > 0 && ti == 0 && lh == 0) {  // the lower ghost plane (see k_lat8_gather)
        int l1, n1, l2, n2, l0, n0;
        l8_range(G.plo, G.mg, l0, n0);
        l8_range(gj, G.m1, l1, n1);
        l8_range(gk, G.m2, l2, n2);
        const int cnt = n0 * n1 * n2;
#pragma unroll
        for (int f = 0; f < F; ++f) {
          const int64_t rp = (int64_t)rowptr[f * G.N + (int64_t)gj * G.m2 + gk] - base;
          double acc = 0.0;
          for (int g = 0; g < F; ++g)
            for (int b = 0; b < n1; ++b)
              for (int c = 0; c < n2; ++c) {
                const int64_t xi = l8_xindex(G, g, G.plo - 1, (int64_t)(gj + l1 + b) * G.m2 + gk + l2 + c);
                acc += csr_vals[rp + (int64_t)g * cnt + b * n2 + c] * (dsc ? x[xi] / dsc[xi] : x[xi]);
              }
          s[f][0] += acc;
        }
      }
#pragma unroll
      for (int f = 0; f < F; ++f)
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (gi0 + u < G.m0) {
            const int64_t r = f * G.N + ((int64_t)(gi0 + u) * G.m1 + gj) * G.m2 + gk;
            double yv = alpha * s[f][u];
            if (beta != 0.0) yv += beta * y[r];
            y[r] = yv;
            if (dotw) dot_acc += yv * dotw[r];
          }
    }
    __syncthreads();  // the staged values are consumed: the next tile's may land
  }
  if (partials) {
    const double bsum = block_reduce_sum(dot_acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = bsum;
  }
}

static Lat8Geom lat8_geom(const mfem_csr_s* A) {
  Lat8Geom G{};
  G.F = A->lat_fields;
  G.m1 = A->lat_m1;
  G.m2 = A->lat_m2;
  G.N = A->n / G.F;
  G.m0 = (int)(G.N / ((int64_t)A->lat_m1 * A->lat_m2));
  G.plo = A->lat_plo;
  G.mg = A->lat_m0 > 0 ? A->lat_m0 : G.m0;
  G.gw = A->lat_gw > 0 ? A->lat_gw : 1;
  G.nui = (G.m0 + 3) / 4;
  G.nuj = (G.m1 + 3) / 4;
  G.nuk = (G.m2 + 3) / 4;
  G.nti = (G.m0 + L8_TI - 1) / L8_TI;
  G.ntj = (G.m1 + L8_TJ - 1) / L8_TJ;
  G.ntk = (G.m2 + L8_TK - 1) / L8_TK;
  return G;
}
// doubles of one stored pair of i-stacked units (the stream of a wave of pass 1: 2 x nsteps steps of 64 lanes)
static int lat8_pair_doubles(int F) { return 2 * (F == 1 ? l8_nsteps(1) : F == 2 ? l8_nsteps(2) : l8_nsteps(3)) * 64; }

// lat8_state: 0 not inspected, -1 not the F-field stencil, 1 structure ok
int mfem_lat8_plan(mfem_context_s* ctx, mfem_csr_s* A) {
  if (A->lat8_state != 0) return MFEM_OK;
  if (A->n < g_layout_min_rows_dia) return MFEM_OK;  // launch-bound sizes stay on the CSR tile kernel
  A->lat8_state = -1;
  if (A->lat_fields == 0) {  // a caller-supplied pattern: read the lattice off row 0 (spmv_lat27.hip)
    int rc0 = mfem_lattice_from_first_row(ctx, A);
    if (rc0) return rc0;
  }
  const int F = A->lat_fields;
  if (F < 1 || F > 3 || A->lat_m1 < 2 || A->lat_m2 < 2 || A->n % F != 0) return MFEM_OK;
  const int64_t PL = (int64_t)A->lat_m1 * A->lat_m2, N = A->n / F;
  if (N % PL != 0) return MFEM_OK;
  const int64_t m0 = N / PL;
  if (m0 < 1 || m0 > (1 << 20) || A->max_row_nnz > 27 * F) return MFEM_OK;
  if (A->ncols > A->n) {  // slab pattern (ghost columns): the hint must say where the owned planes sit in the lattice and how the ghost blocks are laid out
    if (A->lat_m0 < m0 || A->lat_gw != 1 || A->lat_plo < 0 || A->lat_plo + m0 > A->lat_m0 || A->ncols != A->n + 2 * F * PL) return MFEM_OK;
  } else if (A->lat_m0 > 0 && (A->lat_m0 != m0 || A->lat_plo != 0)) {
    return MFEM_OK;
  }
  {  // cheap refusal before the entry-by-entry check: the longest row of the stencil is known from the lattice sizes
    const int64_t mg = A->lat_m0 > 0 ? A->lat_m0 : m0;
    auto w = [](int64_t m) { return m >= 3 ? 3 : (int)m; };
    if (A->max_row_nnz != F * w(mg) * w(A->lat_m1) * w(A->lat_m2)) return MFEM_OK;
  }
  const Lat8Geom G = lat8_geom(A);
  if ((int64_t)G.nti * G.ntj * G.ntk >= ((int64_t)1 << 28)) return MFEM_OK;
  int32_t* d_bad = ctx->d_flags + 12;
  MFEM_CHECK_HIP(hipMemsetAsync(d_bad, 0, sizeof(int32_t), ctx->stream));
  const int grid = mfem_grid_for(A->n, MFEM_BLOCK, ctx->num_cus * 16);
  if (A->rowptr_bits == 64)
    hipLaunchKernelGGL(k_l8_verify<int64_t>, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, G, (const int64_t*)A->rowptr, A->colidx, A->index_base,
                       d_bad);
  else
    hipLaunchKernelGGL(k_l8_verify<int32_t>, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, G, (const int32_t*)A->rowptr, A->colidx, A->index_base,
                       d_bad);
  MFEM_CHECK_LAUNCH();
  MFEM_CHECK_HIP(hipMemcpyAsync(ctx->h_flags + 12, d_bad, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  MFEM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  if (ctx->h_flags[12] == 0) A->lat8_state = 1;
  return MFEM_OK;
}

static size_t lat8_vals_doubles(const Lat8Geom& G) { return (size_t)((G.nui + 1) / 2) * G.nuj * G.nuk * lat8_pair_doubles(G.F); }
static size_t lat8_dump_doubles(const Lat8Geom& G) { return (size_t)G.nti * G.ntj * G.ntk * G.F * L8_FC; }

size_t mfem_lat8_bytes(const mfem_csr_s* A) {
  if (A->lat8_state != 1 || !g_lat8_enable || A->n < g_layout_min_rows_dia) return 0;
  const Lat8Geom G = lat8_geom(A);
  return sizeof(double) * (lat8_vals_doubles(G) + lat8_dump_doubles(G));
}

struct Lat8Bind { double *vals, *dump; const double* src; };
static void lat8_probe_unbind(mfem_csr_s* A) { mfem_lat8_unbind(A); }
static void lat8_probe_rebind(mfem_csr_s* A, void* c) {
  const Lat8Bind* b = (const Lat8Bind*)c;
  A->lat8_vals = b->vals;
  A->lat8_dump = b->dump;
  A->lat8_src = b->src;
}

#define L8_DISPATCH_F(F_, CALL) \
  do {                          \
    if ((F_) == 1) { CALL(1); } \
    else if ((F_) == 2) { CALL(2); } \
    else { CALL(3); }           \
  } while (0)

// Makes the layout copy of `vals` in buf and binds it if the values are symmetric (mfem_sym_probe, spmv_lat27.hip).  dsc: right Jacobi scaling the
// SpMV applies to x (nullptr: none); only the pointer is kept, it may be filled after the bind.  scratch: ncols + 2 n doubles, left dirty.
int mfem_lat8_bind(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, double* buf, const double* dsc, double* scratch, bool allow_rem) {
  mfem_lat8_unbind(A);
  if (A->lat8_state != 1 || !g_lat8_enable || !buf || !scratch) return MFEM_OK;
  const Lat8Geom G = lat8_geom(A);
  unsigned long long* d_stats = (unsigned long long*)(ctx->d_flags + 12);
  MFEM_CHECK_HIP(hipMemsetAsync(d_stats, 0, 2 * sizeof(unsigned long long), ctx->stream));
  const int64_t nunits = (int64_t)G.nui * G.nuj * G.nuk;
  const int grid = mfem_grid_for(nunits * 64, MFEM_BLOCK, ctx->num_cus * 16);
#define L8_FILL(FF)                                                                                                                             \
  if (A->rowptr_bits == 64)                                                                                                                     \
    hipLaunchKernelGGL((k_l8_fill<int64_t, FF>), dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, G, (const int64_t*)A->rowptr, A->index_base, vals, \
                       buf, d_stats);                                                                                                           \
  else                                                                                                                                          \
    hipLaunchKernelGGL((k_l8_fill<int32_t, FF>), dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, G, (const int32_t*)A->rowptr, A->index_base, vals, \
                       buf, d_stats)
  L8_DISPATCH_F(G.F, L8_FILL);
#undef L8_FILL
  MFEM_CHECK_LAUNCH();
  MFEM_CHECK_HIP(hipMemcpyAsync(ctx->h_flags + 12, d_stats, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
  MFEM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  double amax;
  memcpy(&amax, ctx->h_flags + 14, sizeof(double));
  Lat8Bind B{buf, buf + lat8_vals_doubles(G), vals};
  lat8_probe_rebind(A, &B);
  double asym = 1.0;
  int rc = mfem_sym_probe(ctx, A, vals, scratch, amax, lat8_probe_unbind, lat8_probe_rebind, &B, &asym, allow_rem ? G.F : 0);
  A->lat8_asym = asym;
  if (rc || !(asym <= 4e-13)) {  // not symmetric (or NaN): the other layouts serve this solve
    mfem_lat8_unbind(A);
    return rc;
  }
  A->lat8_dsc = dsc;
  A->lat8_scaled = dsc ? 1 : 0;
  return MFEM_OK;
}

bool mfem_lat8_bound(const mfem_csr_s* A, const double* vals) { return A->lat8_vals && vals == A->lat8_src; }

void mfem_lat8_unbind(mfem_csr_s* A) {
  if (A->lat8_vals) A->rem_active = 0;  // (the remainder belongs to the bind)
  A->lat8_vals = nullptr;
  A->lat8_dump = nullptr;
  A->lat8_src = nullptr;
  A->lat8_dsc = nullptr;
}

// returns 1 if launched, 0 if another kernel should be used, <0 on error
int mfem_spmv_lat8_launch(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, const double* x, double* y, double alpha, double beta,
                          const double* dotw, double* partials, int* n_partials, const int32_t* done_flag, int part) {
  if (!A->lat8_vals || vals != A->lat8_src) return 0;
  if (n_partials) *n_partials = 0;
  const Lat8Geom G = lat8_geom(A);
  const int ntiles = G.nti * G.ntj * G.ntk;
  // split SpMV of a slab: see mfem_spmv_lat27_launch -- part 1 = the i-layers of tiles that stage no ghost plane, part 2 = the rest + the gather pass
  const int tb = mfem_lat_first_ghost_layer(G.m0, G.gw, G.nti, G.plo + G.m0 < G.mg) * G.ntj * G.ntk;
  const int tile0 = part == 2 ? tb : 0;
  const int tcount = part == 1 ? tb : ntiles - tile0;
  const int chunk = (tcount + 7) / 8;
#define L8_PASS1_CH(FF, CH_)                                                                                                                         \
  hipLaunchKernelGGL((k_spmv_lat8<FF, CH_>), dim3(8 * chunk), dim3(512), 0, ctx->stream, G, A->lat8_vals, x, A->lat8_dsc, A->lat8_dump, done_flag, tile0, \
                     tcount)
#define L8_PASS1(FF) L8_PASS1_CH(FF, 8)  // (12 and 16 steps per buffer were measured: the same time to 0.3 %)
  if (tcount > 0) {
    L8_DISPATCH_F(G.F, L8_PASS1);
    MFEM_CHECK_LAUNCH();
  }
#undef L8_PASS1
  if (part == 1) return 1;  // (the gather pass belongs to part 2)
  // persistent grid = what is resident (mfem_resident_per_cu): the F = 1 kernel holds 6 workgroups per CU, F = 2 five, F = 3 four -- launched with 8 per CU
  // (until round 5) the one- and two-field gathers ran two rounds for the work of 1.33 / 1.6
  int grid = 1;
#define L8_GATHER(KERNEL, RP, FF)                                                                                                             \
  do {                                                                                                                                        \
    int cap = ctx->num_cus * mfem_resident_per_cu(reinterpret_cast<const void*>(&KERNEL<RP, FF>), MFEM_BLOCK, 0, 4);                          \
    if (cap > MFEM_MAX_PARTIALS) cap = MFEM_MAX_PARTIALS;                                                                                     \
    grid = ntiles < cap ? ntiles : cap;                                                                                                       \
    hipLaunchKernelGGL((KERNEL<RP, FF>), dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, G, A->lat8_dump, y, alpha, beta, dotw, partials, done_flag, \
                       (const RP*)A->rowptr, A->index_base, A->lat8_src, x, A->lat8_dsc);                                                     \
  } while (0)
#define L8_PASS2(FF)                                                                                                             \
  if (g_lat8_gather_staged) {                                                                                                   \
    if (A->rowptr_bits == 64) L8_GATHER(k_lat8_gather_st, int64_t, FF); else L8_GATHER(k_lat8_gather_st, int32_t, FF);          \
  } else {                                                                                                                      \
    if (A->rowptr_bits == 64) L8_GATHER(k_lat8_gather, int64_t, FF); else L8_GATHER(k_lat8_gather, int32_t, FF);                \
  }
  L8_DISPATCH_F(G.F, L8_PASS2);
#undef L8_PASS2
#undef L8_GATHER
  MFEM_CHECK_LAUNCH();
  if (n_partials && partials) *n_partials = grid;
  if (A->rem_active) {  // A = S + N: the skew remainder of the few nonsymmetric rows (spmv_rem.hip)
    const int rcr = mfem_rem_apply(ctx, A, x, A->lat8_dsc, y, alpha, dotw, partials, n_partials, done_flag);
    if (rcr) return rcr;
  }
  if (!ctx->probe_active) ++g_lat8_count;
  return 1;
}

// bytes one SpMV of the layout moves by design: the stored entries, x (and d) as the tiles stage it, the y blocks written and read again, y
int64_t mfem_lat8_design_bytes(const mfem_csr_s* A) {
  const Lat8Geom G = lat8_geom(A);
  const int64_t tiles = (int64_t)G.nti * G.ntj * G.ntk;
  return (int64_t)lat8_vals_doubles(G) * 8 + tiles * G.F * L8_FC * 8 * (A->lat8_scaled ? 4 : 3) + A->n * 8 + mfem_rem_design_bytes(A);
}
int64_t mfem_lat8_entries(const mfem_csr_s* A) { return (int64_t)lat8_vals_doubles(lat8_geom(A)); }
