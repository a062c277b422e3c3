// Geometry of the wave-private patch sweep (spmv_ell.hip: k_spmv_symp): patch shape, LDS mirror tables and the per-step edge block.
// Host + device: tools/host_check_symp.cpp replays the table bookkeeping on the CPU (tests/test_host_checks.py).
#pragma once
#ifndef __HIPCC__
#define __host__
#define __device__
#endif
#define SP_L 4            // lattice lines per patch (8 x 16: 5 % fewer bytes -- rim and x halo -- but no faster: 0.885 against 0.878 ms per CG iteration at 256^3)
#define SP_W 32           // lattice points per line segment; a lane owns two neighbouring points: SP_L * SP_W / 2 = 64 lanes
#define SP_PW (SP_W / 2)  // lane pairs per line
#define SP_ROWS (SP_L * SP_W)
#define SP_XL (SP_L + 2)
#define SP_XC (SP_W + 2)  // staged x points per line: k0 - 1 .. k0 + SP_W
#define SP_XW (SP_W + 4)  // line stride of the x tile: lines stay 16-byte aligned
#define SP_XN (SP_XL * SP_XC)
#define SP_XU ((SP_XN + 63) / 64)  // x entries per lane and plane
#define SP_WG_PER_CU 7
// Mirror tables: one per lower slot s = (di, dj, dk), holding slot 26 - s of the SOURCE rows (row + offset), indexed by source
// cell (line lj + dj, column 2 pk + dk): lines of SP_W + 4 doubles, column c at index c + 2 (column -1 and column SP_W are halo cells), a
// halo line on the side the slot points to.  A halo cell cannot be mirrored (its source row belongs to another patch): it
// receives the referencing row's OWN slot-s entry from the step's edge block, so that the reads are the same two LDS loads for
// every lane and slot.
#define SP_LS (SP_W + 4)
__host__ __device__ constexpr int sp_dj(int s) { return (s / 3) % 3 - 1; }
__host__ __device__ constexpr int sp_dk(int s) { return s % 3 - 1; }
__host__ __device__ constexpr int sp_tsize(int s) { return (sp_dj(s) == 0 ? SP_L : SP_L + 1) * SP_LS; }
__host__ __device__ constexpr int sp_tbase(int s) { return s == 0 ? 0 : sp_tbase(s - 1) + sp_tsize(s - 1); }
__host__ __device__ constexpr int sp_adj(int s) { return sp_dj(s) == -1 ? 1 : 0; }  // table line of source line 0
#define SP_TAB (sp_tbase(12) + sp_tsize(12))
// edge block of a step: for s = 0..12 the halo cells of table s -- the halo line (SP_W cells) if dj != 0, then the halo column
// (lines in ascending order) if dk != 0
__host__ __device__ constexpr int sp_ecnt(int s) { return (sp_dj(s) != 0 ? SP_W : 0) + (sp_dk(s) != 0 ? (sp_dj(s) != 0 ? SP_L - 1 : SP_L) : 0); }
__host__ __device__ constexpr int sp_ebase(int s) { return s == 0 ? 0 : sp_ebase(s - 1) + sp_ecnt(s - 1); }
#define SP_NE (sp_ebase(12) + sp_ecnt(12))  // 318 for 4 x 32 patches (210 for 8 x 16)
#define SP_EU ((SP_NE + 63) / 64)            // edge entries per lane
#define SP_EPAD (64 * SP_EU)
#define SP_STEP (27 * SP_ROWS + SP_EPAD)    // doubles per (plane, patch) in the patch-major copy, in two parts:
#define SP_MAIN (14 * SP_ROWS + SP_EPAD)    //   what every step reads -- slots 13..26 and the edge block -- contiguous per step, steps [plane][patch]
#define SP_LOW (13 * SP_ROWS)               //   the lower slots 0..12 (read where a run starts and by the symmetry check), behind all main parts
// entry e of the edge block: lower slot s, the referencing row's (line, column) in the patch, the LDS cell of table s it fills
__host__ __device__ inline bool sp_edge(int e, int& s, int& line, int& col, int& cell) {
  if (e >= SP_NE) return false;
  s = 0;
  while (e >= sp_ebase(s) + sp_ecnt(s)) ++s;
  int q = e - sp_ebase(s);
  const int dj = sp_dj(s), dk = sp_dk(s);
  int sl, sc;
  if (dj != 0 && q < SP_W) {
    sl = dj < 0 ? -1 : SP_L;
    sc = dk + q;
  } else {
    if (dj != 0) q -= SP_W;
    sl = dj > 0 ? q + 1 : q;
    sc = dk < 0 ? -1 : SP_W;
  }
  line = sl - dj;
  col = sc - dk;
  cell = sp_tbase(s) + (sl + sp_adj(s)) * SP_LS + sc + 2;
  return true;
}
// inverse of sp_edge for the row at (line, col) of its patch: the edge block entry that holds the row's own slot-s entry (s = 0..12), or
// -1 when the row's slot-s neighbour lies inside the patch.  A row owns at most one entry per slot: the halo line takes the corner cells.
__host__ __device__ constexpr int sp_edge_of(int s, int line, int col) {
  const int dj = sp_dj(s), dk = sp_dk(s);
  if (dj != 0 && line == (dj < 0 ? 0 : SP_L - 1)) return sp_ebase(s) + col;
  if (dk != 0 && col == (dk < 0 ? 0 : SP_W - 1)) {
    const int q = dj < 0 ? line - 1 : line;  // dj < 0: lines 1 .. SP_L - 1; dj > 0: lines 0 .. SP_L - 2; dj = 0: all lines
    return sp_ebase(s) + (dj != 0 ? SP_W : 0) + q;
  }
  return -1;
}
