// CPU replay of the mirror-table bookkeeping of the wave-private patch sweep (csrc/spmv_symp.h, k_spmv_symp in spmv_ell.hip) on a small
// lattice with a symmetric 27-point operator: for every patch, plane, lane, row and lower slot the value the kernel would read from its
// LDS tables (interior cells written from the upper slots of the source rows, halo cells from the edge block, run starts from the row's
// own slots) must be the row's own entry; edge-block cells must be distinct halo cells, interior cells distinct non-halo cells.
//   g++ -O2 -std=c++17 -I metafem.jl_amd/csrc tools/host_check_symp.cpp -o tools/bin/host_check_symp && tools/bin/host_check_symp
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <set>
#include <vector>
#include "spmv_symp.h"

static int m1, m2;
static int64_t PL;
// symmetric operator on the lattice: entry (r, c) for lattice neighbours, 0 where the neighbour does not exist
static bool coords(int64_t r, int& i, int& j, int& k) {
  i = (int)(r / PL);
  j = (int)((r % PL) / m2);
  k = (int)(r % m2);
  return true;
}
static double entry(int64_t r, int s) {  // slot s = (di, dj, dk) of row r
  int i, j, k;
  coords(r, i, j, k);
  const int di = s / 9 - 1, dj = (s / 3) % 3 - 1, dk = s % 3 - 1;
  const int jj = j + dj, kk = k + dk;
  if (jj < 0 || jj >= m1 || kk < 0 || kk >= m2) return 0.0;  // structurally absent
  const int64_t c = r + di * PL + dj * m2 + dk;
  const int64_t a = r < c ? r : c, b = r < c ? c : r;
  return 1.0 + (double)((a * 1315423911LL + b * 2654435761LL) % 1000003) / 7.0;  // symmetric in (r, c)
}

int main() {
  int bad = 0;
  // ---- static checks
  std::set<int> halo;
  for (int e = 0; e < SP_EPAD; ++e) {
    int s, line, col, cell;
    if (!sp_edge(e, s, line, col, cell)) {
      if (e < SP_NE) { printf("edge %d not decoded\n", e); ++bad; }
      continue;
    }
    if (cell < sp_tbase(s) || cell >= sp_tbase(s) + sp_tsize(s)) { printf("edge %d: cell outside table %d\n", e, s); ++bad; }
    if (!halo.insert(cell).second) { printf("edge %d: cell %d filled twice\n", e, cell); ++bad; }
    if (line < 0 || line >= SP_L || col < 0 || col >= SP_W) { printf("edge %d: referencing row outside the patch\n", e); ++bad; }
  }
  {  // sp_edge_of is the inverse of sp_edge: every (slot, line, col) owns at most one entry, all SP_NE entries are owned
    int owned = 0;
    for (int s = 0; s < 13; ++s)
      for (int line = 0; line < SP_L; ++line)
        for (int col = 0; col < SP_W; ++col) {
          const int e = sp_edge_of(s, line, col);
          if (e < 0) continue;
          ++owned;
          int s2, l2, c2, cell2;
          if (e >= SP_NE || !sp_edge(e, s2, l2, c2, cell2) || s2 != s || l2 != line || c2 != col) {
            printf("sp_edge_of(%d, %d, %d) = %d does not decode back\n", s, line, col, e);
            ++bad;
          }
        }
    if (owned != SP_NE) { printf("sp_edge_of owns %d entries, SP_NE = %d\n", owned, SP_NE); ++bad; }
  }
  if ((int)halo.size() != SP_NE) { printf("edge block has %zu cells, SP_NE = %d\n", halo.size(), SP_NE); ++bad; }
  for (int s = 0; s < 13; ++s)
    for (int lane = 0; lane < 64; ++lane) {
      const int lj = lane / SP_PW, pk = lane % SP_PW, lb = lj * SP_LS + 2 * pk;
      const int w = sp_tbase(s) + sp_adj(s) * SP_LS + 2 + lb;  // the two interior cells the lane writes
      if (halo.count(w) || halo.count(w + 1)) { printf("slot %d lane %d: interior cell is a halo cell\n", s, lane); ++bad; }
      if (w + 1 >= sp_tbase(s) + sp_tsize(s)) { printf("slot %d lane %d: interior cell outside its table\n", s, lane); ++bad; }
    }
  // ---- replay on lattices with partial patches in both directions
  const int cases[3][3] = {{5, 9, 70}, {4, 4, 33}, {6, 13, 32}};
  for (auto& cs : cases) {
    const int m0 = cs[0];
    m1 = cs[1];
    m2 = cs[2];
    PL = (int64_t)m1 * m2;
    const int NS = (m1 + SP_L - 1) / SP_L, NPk = (m2 + SP_W - 1) / SP_W;
    for (int patch = 0; patch < NS * NPk; ++patch) {
      const int j0 = (patch / NPk) * SP_L, k0 = (patch % NPk) * SP_W;
      std::vector<double> tab(SP_TAB + 2, NAN);
      for (int start = 1; start <= 2; ++start) {  // run starts at plane 1 and at plane 2
        std::fill(tab.begin(), tab.end(), NAN);
        bool have_hist = false;
        for (int p = start; p < m0 - 1; ++p) {
          auto rowof = [&](int lane, int h, bool& valid) -> int64_t {
            const int j = j0 + (lane / SP_PW), k = k0 + 2 * (lane % SP_PW) + h;
            valid = j < m1 && k < m2;
            return (int64_t)p * PL + (int64_t)j * m2 + k;
          };
          if (!have_hist)  // run start: own previous-plane slots to where the mirror reads look
            for (int lane = 0; lane < 64; ++lane)
              for (int s = 0; s < 9; ++s)
                for (int h = 0; h < 2; ++h) {
                  bool v;
                  const int64_t r = rowof(lane, h, v);
                  const int lb = (lane / SP_PW) * SP_LS + 2 * (lane % SP_PW);
                  tab[sp_tbase(s) + (sp_dj(s) + sp_adj(s)) * SP_LS + sp_dk(s) + 2 + lb + h] = v ? entry(r, s) : 0.0;
                }
          // phase B: +z / +y slots of this plane's rows, edge block
          for (int lane = 0; lane < 64; ++lane)
            for (int s = 9; s < 13; ++s)
              for (int h = 0; h < 2; ++h) {
                bool v;
                const int64_t r = rowof(lane, h, v);
                const int lb = (lane / SP_PW) * SP_LS + 2 * (lane % SP_PW);
                tab[sp_tbase(s) + sp_adj(s) * SP_LS + 2 + lb + h] = v ? entry(r, 26 - s) : 0.0;
              }
          for (int e = 0; e < SP_NE; ++e) {
            int s, line, col, cell;
            sp_edge(e, s, line, col, cell);
            const bool v = j0 + line < m1 && k0 + col < m2;
            tab[cell] = v ? entry((int64_t)p * PL + (int64_t)(j0 + line) * m2 + k0 + col, s) : 0.0;
          }
          // phase C: every lower slot of every valid row read through the tables
          for (int lane = 0; lane < 64; ++lane)
            for (int s = 0; s < 13; ++s)
              for (int h = 0; h < 2; ++h) {
                bool v;
                const int64_t r = rowof(lane, h, v);
                if (!v) continue;
                const int lb = (lane / SP_PW) * SP_LS + 2 * (lane % SP_PW);
                const double got = tab[sp_tbase(s) + (sp_dj(s) + sp_adj(s)) * SP_LS + sp_dk(s) + 2 + lb + h];
                const double want = entry(r, s);
                if (!(got == want)) {
                  if (bad < 10) printf("lattice %dx%dx%d patch %d plane %d lane %d row %d slot %d: table %.6f, entry %.6f\n", m0, m1, m2, patch, p, lane, h, s, got, want);
                  ++bad;
                }
              }
          // phase D: next-plane slots into the previous-plane tables
          for (int lane = 0; lane < 64; ++lane)
            for (int s = 0; s < 9; ++s)
              for (int h = 0; h < 2; ++h) {
                bool v;
                const int64_t r = rowof(lane, h, v);
                const int lb = (lane / SP_PW) * SP_LS + 2 * (lane % SP_PW);
                tab[sp_tbase(s) + sp_adj(s) * SP_LS + 2 + lb + h] = v ? entry(r, 26 - s) : 0.0;
              }
          have_hist = true;
        }
      }
    }
  }
  printf(bad ? "FAIL (%d)\n" : "OK\n", bad);
  return bad ? 1 : 0;
}
