// CPU check of the step / slot tables of the symmetric lattice-tile layouts (csrc/spmv_lat_tables.h; tests/test_host_checks.py).
// A random symmetric matrix with the lattice stencil is multiplied twice: entry by entry (every stored entry of the full stencil), and the way the
// kernels do it -- only the slots the tables list, each used for row r (a x[c]) and, unless it is the diagonal, mirrored onto row c (a x[r]).
// If the tables list every unordered pair exactly once, the two products agree to round-off on every lattice size.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string.h>
#include <vector>

#include "spmv_lat_tables.h"

static double rnd() { return (double)rand() / RAND_MAX - 0.5; }

// ---- mode 4: one field, reach 2 in an even direction and 1 in an odd one
static int check27(int m0, int m1, int m2) {
  int8_t d[L27_TAB][4];
  int Kup[8];
  if (!l27_build_tables(d, Kup)) return printf("table build failed\n"), 1;
  const int want[8] = {63, 38, 38, 23, 38, 23, 23, 14};
  for (int t = 0; t < 8; ++t)
    if (Kup[t] != want[t]) return printf("type %d: %d stored slots, expected %d\n", t, Kup[t], want[t]), 1;
  // the deterministic form (round 6, l27d_stream): the two parity groups' streams hold every stored slot of every type exactly once, phase-major, a phase
  // = one (dj, dk), (0, 0) last; a (j, k) column belongs to one parity group
  {
    int seen[8][3][5][5];
    memset(seen, 0, sizeof(seen));
    int per_type[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int pg = 0; pg < 2; ++pg) {
      const L27DStream S = l27d_stream(pg);
      if (S.n != (pg ? 122 : 138) || (S.n & 1)) return printf("stream %d has %d steps\n", pg, S.n), 1;
      for (int v = 0; v < S.n; ++v) {
        const int t = l27d_type(pg, S.q[v]);
        if ((((t >> 1) & 1) + (t & 1)) % 2 != pg) return printf("type %d in the wrong parity group\n", t), 1;
        const int R[3] = {(t & 4) ? 1 : 2, (t & 2) ? 1 : 2, (t & 1) ? 1 : 2};
        const int di = S.di[v], dj = S.dj[v], dk = S.dk[v];
        if (di < 0 || di > R[0] || dj < -R[1] || dj > R[1] || dk < -R[2] || dk > R[2]) return printf("offset out of reach\n"), 1;
        if (!(di > 0 || dj > 0 || (dj == 0 && dk >= 0))) return printf("lower entry stored\n"), 1;
        if (S.phase[v] != l27d_phase_of(dj, dk) || (v > 0 && S.phase[v] < S.phase[v - 1])) return printf("phase order\n"), 1;
        if (dj == 0 && dk == 0 && S.phase[v] != L27D_NPHASE - 1) return printf("(0, 0) is not in the last phase\n"), 1;
        // two steps of one phase have the same dk and dj of the same parity: their source columns lie in one cube and have one parity of j + k -- one wave
        for (int u = 0; u < v; ++u)
          if (S.phase[u] == S.phase[v] && (S.dk[u] != dk || ((S.dj[u] - dj) & 1))) return printf("phase mixes waves\n"), 1;
        if (seen[t][di][dj + 2][dk + 2]++) return printf("slot twice\n"), 1;
        ++per_type[t];
      }
    }
    for (int t = 0; t < 8; ++t)
      if (per_type[t] != want[t]) return printf("type %d: %d steps in the streams, %d stored slots\n", t, per_type[t], want[t]), 1;
  }
  const long n = (long)m0 * m1 * m2;
  auto id = [&](int i, int j, int k) { return ((long)i * m1 + j) * m2 + k; };
  auto reach = [](int g) { return (g & 1) ? 1 : 2; };
  std::map<std::pair<long, long>, double> A;  // symmetric values on the stencil pairs
  for (int i = 0; i < m0; ++i)
    for (int j = 0; j < m1; ++j)
      for (int k = 0; k < m2; ++k)
        for (int di = -reach(i); di <= reach(i); ++di)
          for (int dj = -reach(j); dj <= reach(j); ++dj)
            for (int dk = -reach(k); dk <= reach(k); ++dk) {
              const int ci = i + di, cj = j + dj, ck = k + dk;
              if (ci < 0 || ci >= m0 || cj < 0 || cj >= m1 || ck < 0 || ck >= m2) continue;
              const long r = id(i, j, k), c = id(ci, cj, ck);
              // (the stencil must be symmetric as a pattern: c reaches r as well)
              if (std::abs(di) > reach(ci) || std::abs(dj) > reach(cj) || std::abs(dk) > reach(ck)) return printf("pattern not symmetric\n"), 1;
              if (r <= c) A[{r, c}] = rnd();
            }
  std::vector<double> x(n), y0(n, 0.0), y1(n, 0.0);
  for (auto& v : x) v = rnd();
  for (auto& e : A) {
    y0[e.first.first] += e.second * x[e.first.second];
    if (e.first.first != e.first.second) y0[e.first.second] += e.second * x[e.first.first];
  }
  long used = 0;
  for (int i = 0; i < m0; ++i)
    for (int j = 0; j < m1; ++j)
      for (int k = 0; k < m2; ++k) {
        const int t = 4 * (i & 1) + 2 * (j & 1) + (k & 1), K4 = l27_K4[t];
        const long r = id(i, j, k);
        for (int q = 0; q < 4; ++q)
          for (int it = 0; it < K4; ++it) {
            const int8_t* o = d[l27_tb[t] + q * K4 + it];
            if (o[0] == L27_PAD) continue;
            const int ci = i + o[0], cj = j + o[1], ck = k + o[2];
            if (ci >= m0 || cj < 0 || cj >= m1 || ck < 0 || ck >= m2) continue;  // outside the lattice: the kernels store 0 there
            const long c = id(ci, cj, ck);
            if (c < r) return printf("slot below the diagonal\n"), 1;
            const auto f = A.find({r, c});
            if (f == A.end()) return printf("slot off the stencil\n"), 1;
            y1[r] += f->second * x[c];
            if (!(it == 0 && q == 0)) y1[c] += f->second * x[r];
            else if (c != r) return printf("slot 0 is not the diagonal\n"), 1;
            ++used;
          }
      }
  if (used != (long)A.size()) return printf("%ld slots used, %zu pairs\n", used, A.size()), 1;
  double err = 0.0, scale = 0.0;
  for (long r = 0; r < n; ++r) {
    err = std::fmax(err, std::fabs(y0[r] - y1[r]));
    scale = std::fmax(scale, std::fabs(y0[r]));
  }
  if (!(err <= 1e-13 * scale)) return printf("mode 4 on %d x %d x %d: error %.3e\n", m0, m1, m2, err / scale), 1;
  return 0;
}

// ---- mode 5: F fields on the 27-point stencil, field-major
static int check8(int F, int m0, int m1, int m2) {
  const long N = (long)m0 * m1 * m2, n = F * N;
  auto id = [&](int f, int i, int j, int k) { return f * N + ((long)i * m1 + j) * m2 + k; };
  std::map<std::pair<long, long>, double> A;
  for (int f = 0; f < F; ++f)
    for (int i = 0; i < m0; ++i)
      for (int j = 0; j < m1; ++j)
        for (int k = 0; k < m2; ++k)
          for (int g = 0; g < F; ++g)
            for (int di = -1; di <= 1; ++di)
              for (int dj = -1; dj <= 1; ++dj)
                for (int dk = -1; dk <= 1; ++dk) {
                  const int ci = i + di, cj = j + dj, ck = k + dk;
                  if (ci < 0 || ci >= m0 || cj < 0 || cj >= m1 || ck < 0 || ck >= m2) continue;
                  const long r = id(f, i, j, k), c = id(g, ci, cj, ck);
                  A[{r < c ? r : c, r < c ? c : r}] = 0.0;
                }
  for (auto& e : A) e.second = rnd();
  std::vector<double> x(n), y0(n, 0.0), y1(n, 0.0);
  for (auto& v : x) v = rnd();
  for (auto& e : A) {
    y0[e.first.first] += e.second * x[e.first.second];
    if (e.first.first != e.first.second) y0[e.first.second] += e.second * x[e.first.first];
  }
  long used = 0;
  const int NS = l8_nsteps(F);
  if (NS != 13 * F * F + F * (F + 1) / 2 || l8_padded(F) != ((NS + 1) & ~1)) return printf("step count\n"), 1;
  for (int s = 0; s < NS; ++s) {
    const int f = l8_row_field(F, s);
    if (f < 0 || f >= F || s < l8_first(F, f) || s >= l8_first(F, f + 1) || l8_g(F, s) < 0 || l8_g(F, s) >= F || l8_e(F, s) < 0 || l8_e(F, s) > 13)
      return printf("step %d malformed\n", s), 1;
  }
  // the phase-major order of round 6 (l8_order / l8_stream): a permutation of the steps, phases = the (dj, dk) of the offset, (0, 0) last; the stream of a
  // wave's two units visits every (unit, position) once, phase-major, unit 0 before unit 1 inside a phase
  {
    const L8Order O = l8_order(F);
    const L8Stream T = l8_stream(F);
    std::vector<int> seen(NS, 0);
    if (O.first[0] != 0 || O.first[L8_NPHASE] != NS) return printf("phase table\n"), 1;
    for (int ph = 0; ph < L8_NPHASE; ++ph)
      for (int p = O.first[ph]; p < O.first[ph + 1]; ++p) {
        const int s = O.step[p], e = l8_e(F, s);
        if (s < 0 || s >= NS || O.pos[s] != p || seen[s]++) return printf("order is not a permutation\n"), 1;
        if (l8_phase_of(e) != ph || ((l8_dj(e) == 0 && l8_dk(e) == 0) != (ph == L8_NPHASE - 1))) return printf("step in the wrong phase\n"), 1;
        if (p > O.first[ph] && l8_dj(l8_e(F, O.step[p - 1])) * 3 + l8_dk(l8_e(F, O.step[p - 1])) != l8_dj(e) * 3 + l8_dk(e)) return printf("phase mixes offsets\n"), 1;
      }
    std::vector<int> visits(2 * NS, 0);
    for (int v = 0; v < 2 * NS; ++v) {
      if (T.unit[v] < 0 || T.unit[v] > 1 || T.pos[v] < 0 || T.pos[v] >= NS) return printf("stream entry\n"), 1;
      if (T.pos[v] < O.first[T.phase[v]] || T.pos[v] >= O.first[T.phase[v] + 1]) return printf("stream phase\n"), 1;
      if (v > 0 && (T.phase[v] < T.phase[v - 1] || (T.phase[v] == T.phase[v - 1] && T.unit[v] < T.unit[v - 1]))) return printf("stream order\n"), 1;
      if (visits[T.unit[v] * NS + T.pos[v]]++) return printf("stream visits twice\n"), 1;
    }
  }
  for (int i = 0; i < m0; ++i)
    for (int j = 0; j < m1; ++j)
      for (int k = 0; k < m2; ++k)
        for (int s = 0; s < NS; ++s) {
          const int f = l8_row_field(F, s), g = l8_g(F, s), e = l8_e(F, s);
          const int ci = i + l8_di(e), cj = j + l8_dj(e), ck = k + l8_dk(e);
          if (e == 0 && g < f) return printf("own-block slot below the diagonal\n"), 1;
          if (ci >= m0 || cj < 0 || cj >= m1 || ck < 0 || ck >= m2) continue;
          const long r = id(f, i, j, k), c = id(g, ci, cj, ck);
          const auto it = A.find({r < c ? r : c, r < c ? c : r});
          if (it == A.end()) return printf("step off the stencil\n"), 1;
          y1[r] += it->second * x[c];
          if (!(e == 0 && g == f)) y1[c] += it->second * x[r];
          ++used;
        }
  if (used != (long)A.size()) return printf("%ld steps used, %zu pairs\n", used, A.size()), 1;
  double err = 0.0, scale = 0.0;
  for (long r = 0; r < n; ++r) {
    err = std::fmax(err, std::fabs(y0[r] - y1[r]));
    scale = std::fmax(scale, std::fabs(y0[r]));
  }
  if (!(err <= 1e-13 * scale)) return printf("mode 5, %d fields, on %d x %d x %d: error %.3e\n", F, m0, m1, m2, err / scale), 1;
  return 0;
}

int main() {
  srand(12345);
  const int s27[][3] = {{3, 3, 3}, {5, 3, 7}, {9, 5, 5}, {7, 11, 3}};
  for (auto& s : s27)
    if (check27(s[0], s[1], s[2])) return 1;
  const int s8[][3] = {{1, 1, 1}, {2, 2, 2}, {3, 4, 5}, {6, 2, 7}, {1, 5, 4}};
  for (int F = 1; F <= 3; ++F)
    for (auto& s : s8)
      if (check8(F, s[0], s[1], s[2])) return 1;
  printf("lattice-tile tables: every stencil pair listed exactly once, products agree\nOK\n");
  return 0;
}
