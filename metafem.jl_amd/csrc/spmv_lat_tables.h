// Step / slot tables of the symmetric lattice-tile layouts (spmv_lat27.hip: mode 4, spmv_lat8.hip: mode 5).  Host + device:
// tools/host_check_lat.cpp replays a product with them on the CPU (tests/test_host_checks.py).
#pragma once
#ifndef __HIPCC__
#define __host__
#define __device__
#endif
#include <stdint.h>

// ---- mode 5: F = 1..3 fields on the 27-point stencil (field-major rows), lane = node.  The steps of a unit, wave-uniform:
// row field f: for g = 0..F-1: [the node's own block entry (f, g) if g >= f], then the 13 upper neighbours e = 1..13 (d = (0,0,1) .. (1,1,1))
__host__ __device__ constexpr int l8_field_steps(int F, int f) { return 13 * F + (F - f); }
__host__ __device__ constexpr int l8_first(int F, int f) { return f == 0 ? 0 : l8_first(F, f - 1) + l8_field_steps(F, f - 1); }
__host__ __device__ constexpr int l8_nsteps(int F) { return l8_first(F, F); }           // 14, 55, 123
__host__ __device__ constexpr int l8_padded(int F) { return (l8_nsteps(F) + 1) & ~1; }  // 14, 56, 124: steps leave in pairs (16-byte loads)
__host__ __device__ constexpr int l8_row_field(int F, int s) {
  int f = 0;
  while (f + 1 < F && s >= l8_first(F, f + 1)) ++f;
  return f;
}
// (g, e) of step s; e = 0: the node itself
__host__ __device__ constexpr int l8_g(int F, int s) {
  const int f = l8_row_field(F, s);
  int t = s - l8_first(F, f);
  for (int g = 0; g < F; ++g) {
    const int len = 13 + (g >= f ? 1 : 0);
    if (t < len) return g;
    t -= len;
  }
  return 0;
}
__host__ __device__ constexpr int l8_e(int F, int s) {
  const int f = l8_row_field(F, s);
  int t = s - l8_first(F, f);
  for (int g = 0; g < F; ++g) {
    const int own = g >= f ? 1 : 0;
    const int len = 13 + own;
    if (t < len) return own ? t : t + 1;
    t -= len;
  }
  return 0;
}
__host__ __device__ constexpr int l8_di(int e) { return (e + 13) / 9 - 1; }
__host__ __device__ constexpr int l8_dj(int e) { return ((e + 13) / 3) % 3 - 1; }
__host__ __device__ constexpr int l8_dk(int e) { return (e + 13) % 3 - 1; }

// ---- mode 5, deterministic order (round 6).  The mirrored product of a step goes to the LDS cell of the neighbour at (di, dj, dk); two WAVES of a workgroup
// add into one cell only across the (j, k) edges of their 4 x 4 columns of nodes.  Steps are therefore stored and run PHASE-MAJOR, a phase = one (dj, dk):
// within a phase every cell receives products from ONE node column -- one wave -- in that wave's program order; workgroup barriers separate the phases,
// so the order of the additions into a cell is fixed: y is bitwise reproducible.  Phases 0-7: (dj, dk) != (0, 0) in lexicographic order, phase 8:
// (0, 0) -- the node's own block entries and the (1, 0, 0) neighbour --, which also takes the row sums.
#define L8_NPHASE 9
__host__ __device__ constexpr int l8_phase_of(int e) {
  const int c = (l8_dj(e) + 1) * 3 + (l8_dk(e) + 1);  // 4 = (0, 0)
  return c < 4 ? c : c == 4 ? 8 : c - 1;
}
struct L8Order {
  int pos[128];     // position (in steps of 64 lanes x 8 bytes) of original step s inside a unit
  int step[128];    // ... and back
  int first[L8_NPHASE + 1];  // first position of a phase
};
__host__ __device__ constexpr L8Order l8_order(int F) {
  L8Order o{};
  int p = 0;
  for (int ph = 0; ph < L8_NPHASE; ++ph) {
    o.first[ph] = p;
    for (int s = 0; s < l8_nsteps(F); ++s)
      if (l8_phase_of(l8_e(F, s)) == ph) {
        o.pos[s] = p;
        o.step[p] = s;
        ++p;
      }
  }
  o.first[L8_NPHASE] = p;
  return o;
}
// the two units of a wave as ONE stream of 2 * nsteps steps, phase-major, inside a phase unit 0's steps, then unit 1's: v -> (phase, unit, position)
struct L8Stream {
  int phase[256], unit[256], pos[256];
  int at[2][128];  // ... and back: stream index of (unit, position)
};
__host__ __device__ constexpr L8Stream l8_stream(int F) {
  const L8Order o = l8_order(F);
  L8Stream t{};
  int v = 0;
  for (int ph = 0; ph < L8_NPHASE; ++ph)
    for (int h = 0; h < 2; ++h)
      for (int p = o.first[ph]; p < o.first[ph + 1]; ++p) {
        t.phase[v] = ph;
        t.unit[v] = h;
        t.pos[v] = p;
        t.at[h][p] = v;
        ++v;
      }
  return t;
}

// ---- mode 4: the hex-27 lattice, 8 node types t = 4 (i odd) + 2 (j odd) + (k odd), four lanes per row, slot s = 4 it + q of lane q at step it.
// Slot 0 is the diagonal, then the offsets with (di, dj, dk) > 0 in lexicographic order; reach 2 in an even direction, 1 in an odd one.
#define L27_TAB 272  // table entries: 4 x (16 + 3 x 10 + 3 x 6 + 4)
#define L27_PAD 127
static const int l27_K4[8] = {16, 10, 10, 6, 10, 6, 6, 4};             // steps per lane
static const int l27_gb[8] = {0, 1024, 1664, 2304, 2688, 3328, 3712, 4096};  // group base inside a unit (doubles)
static const int l27_tb[8] = {0, 64, 104, 144, 168, 208, 232, 256};    // table base (entries); entry of (t, q, it) at tb[t] + q * K4[t] + it
// d[idx] = (di, dj, dk, 0) of the slot, di = L27_PAD for padding; Kup[t] = stored slots of type t.  Returns false if a type does not fit its steps.
inline bool l27_build_tables(int8_t (*d)[4], int* Kup) {
  for (int i = 0; i < L27_TAB; ++i) {
    d[i][0] = L27_PAD;
    d[i][1] = d[i][2] = d[i][3] = 0;
  }
  for (int t = 0; t < 8; ++t) {
    const int R[3] = {(t & 4) ? 1 : 2, (t & 2) ? 1 : 2, (t & 1) ? 1 : 2};
    const int K4 = l27_K4[t];
    int s = 0;
    auto put = [&](int di, int dj, int dk) {
      if (s < 4 * K4) {
        const int it = s / 4, q = s % 4, idx = l27_tb[t] + q * K4 + it;
        d[idx][0] = (int8_t)di;
        d[idx][1] = (int8_t)dj;
        d[idx][2] = (int8_t)dk;
      }
      ++s;
    };
    put(0, 0, 0);  // slot 0: the diagonal
    for (int di = 0; di <= R[0]; ++di)
      for (int dj = -R[1]; dj <= R[1]; ++dj)
        for (int dk = -R[2]; dk <= R[2]; ++dk)
          if (di > 0 || dj > 0 || (dj == 0 && dk > 0)) put(di, dj, dk);
    Kup[t] = s;
    if (s > 4 * K4) return false;
  }
  return true;
}

// ---- mode 4, deterministic order (round 6): LANE = ROW.  A wave owns the rows of FOUR node types in a cube of 8 x 8 x 8 lattice points -- 4 x 4 x 4 = 64
// rows per type, lane = (a, b, c), row = (2 a + pi, 2 b + pj, 2 c + pk) --, the two waves of a cube split the types by the parity of their (j, k) column:
// pg = (pj + pk) & 1, so a lattice column belongs to ONE wave.  Every stored slot of a type is one wave-wide step with a compile-time offset (no padding
// steps: 63 + 38 + 23 + 14 = 138 and 38 + 23 + 38 + 23 = 122 steps for 256 rows each, 32.5 entries per row against 34 with four lanes per row).  Steps run
// phase-major (phases: below; the one of (0, 0) last, with the row sums), workgroup barriers between phases: inside a phase a cell receives products
// from columns of ONE wave, in program order.
// Offsets that differ by 2 in dj alone share a phase: their source columns (j - dj, k - dk) have the same k -- the same cube -- and the same parity of
// j + k -- the same wave.  So a phase = (dk, parity of dj): 10 phases, 9 barriers per tile; the phase of (0, 0) -- (dk = 0, dj even) -- is the last.
#define L27D_NPHASE 10
#define L27D_MAXS 140
__host__ __device__ constexpr int l27d_phase_of(int dj, int dk) {
  const int c = (dk + 2) * 2 + (dj & 1);  // 4 = (dk = 0, dj even)
  return c < 4 ? c : c == 4 ? 9 : c - 1;
}
__host__ __device__ constexpr int l27d_type(int pg, int q) { return pg == 0 ? (q == 0 ? 0 : q == 1 ? 4 : q == 2 ? 3 : 7) : (q == 0 ? 1 : q == 1 ? 5 : q == 2 ? 2 : 6); }
struct L27DStream {
  int n;                  // steps
  int q[L27D_MAXS];       // which of the wave's four types
  int di[L27D_MAXS], dj[L27D_MAXS], dk[L27D_MAXS];
  int phase[L27D_MAXS];
};
__host__ __device__ constexpr L27DStream l27d_stream(int pg) {
  L27DStream S{};
  int v = 0;
  for (int ph = 0; ph < L27D_NPHASE; ++ph)
    for (int q = 0; q < 4; ++q) {
      const int t = l27d_type(pg, q);
      const int R0 = (t & 4) ? 1 : 2, R1 = (t & 2) ? 1 : 2, R2 = (t & 1) ? 1 : 2;
      if (ph == L27D_NPHASE - 1) {  // the diagonal: slot 0
        S.q[v] = q; S.di[v] = 0; S.dj[v] = 0; S.dk[v] = 0; S.phase[v] = ph;
        ++v;
      }
      for (int di = 0; di <= R0; ++di)
        for (int dj = -R1; dj <= R1; ++dj)
          for (int dk = -R2; dk <= R2; ++dk)
            if ((di > 0 || dj > 0 || (dj == 0 && dk > 0)) && l27d_phase_of(dj, dk) == ph) {
              S.q[v] = q; S.di[v] = di; S.dj[v] = dj; S.dk[v] = dk; S.phase[v] = ph;
              ++v;
            }
    }
  S.n = v;
  return S;
}
#define L27D_CUBE_STEPS 260  // 138 + 122
