// CPU check of csrc/hex8_sumfac.h against the stored-table form (J, J^-1, grad N_a per Gauss point) on distorted elements.
//   g++ -O2 -std=c++17 -I metafem.jl_amd/csrc tools/host_check_hex8.cpp -o tools/bin/host_check_hex8 && tools/bin/host_check_hex8
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include "hex8_sumfac.h"

template <int NG>
static double check(unsigned seed) {
  srand(seed);
  auto rnd = [] { return rand() / (double)RAND_MAX - 0.5; };
  double X[3][2][4], T[2][4], S[2][4];
  for (int b = 0; b < 8; ++b) {
    const int bx = b & 1, by = (b >> 1) & 1, bz = b >> 2;
    X[0][bx][by + 2 * bz] = 0.7 * bx + 0.25 * rnd();
    X[1][bx][by + 2 * bz] = 1.1 * by + 0.25 * rnd();
    X[2][bx][by + 2 * bz] = 0.9 * bz + 0.25 * rnd();
    T[bx][by + 2 * bz] = 300 + 50 * rnd();
    S[bx][by + 2 * bz] = 1600 + 100 * rnd();
  }
  const double kc = 0.6;
  double fe[2][4], ke[36];
  sf_thermal_fe<NG>(X, T, S, true, kc, fe);
  sf_thermal_ke<NG>(X, kc, ke);
  // elasticity residual: nodal displacements, lam / mu of E = 1, nu = 0.3
  const double lam = 0.5769230769230769, mu = 0.38461538461538464;
  double U[3][2][4], fel[3][2][4], felr[3][8] = {{0}};
  for (int i = 0; i < 3; ++i)
    for (int b = 0; b < 8; ++b) U[i][b & 1][b >> 1] = 0.1 * rnd();
  sf_elasticity_fe<NG>(X, U, lam, mu, fel);
  // table form
  double fr[8] = {0}, kr[8][8] = {{0}};
  for (int qz = 0; qz < NG; ++qz)
    for (int qy = 0; qy < NG; ++qy)
      for (int qx = 0; qx < NG; ++qx) {
        const double xi[3] = {sf_xi<NG>(qx), sf_xi<NG>(qy), sf_xi<NG>(qz)};
        const double w = sf_w<NG>(qx) * sf_w<NG>(qy) * sf_w<NG>(qz);
        double N[8], dN[8][3];
        for (int b = 0; b < 8; ++b) {
          double f[3], df[3];
          for (int d = 0; d < 3; ++d) {
            const int bd = (b >> d) & 1;
            f[d] = bd ? xi[d] : 1 - xi[d];
            df[d] = bd ? 1 : -1;
          }
          N[b] = f[0] * f[1] * f[2];
          dN[b][0] = df[0] * f[1] * f[2];
          dN[b][1] = f[0] * df[1] * f[2];
          dN[b][2] = f[0] * f[1] * df[2];
        }
        double J[3][3] = {{0}};
        for (int i = 0; i < 3; ++i)
          for (int m = 0; m < 3; ++m)
            for (int b = 0; b < 8; ++b) J[i][m] += dN[b][m] * X[i][b & 1][(b >> 1)];
        const double det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) +
                           J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
        double I[3][3];
        I[0][0] = (J[1][1] * J[2][2] - J[1][2] * J[2][1]) / det;
        I[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) / det;
        I[0][2] = (J[0][1] * J[1][2] - J[1][1] * J[0][2]) / det;
        I[1][0] = (J[1][2] * J[2][0] - J[2][2] * J[1][0]) / det;
        I[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) / det;
        I[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) / det;
        I[2][0] = (J[1][0] * J[2][1] - J[1][1] * J[2][0]) / det;
        I[2][1] = (J[0][1] * J[2][0] - J[2][1] * J[0][0]) / det;
        I[2][2] = (J[0][0] * J[1][1] - J[1][0] * J[0][1]) / det;
        double g[8][3], gT[3] = {0}, sq = 0;
        for (int b = 0; b < 8; ++b) {
          for (int s = 0; s < 3; ++s) g[b][s] = dN[b][0] * I[0][s] + dN[b][1] * I[1][s] + dN[b][2] * I[2][s];
          for (int s = 0; s < 3; ++s) gT[s] += g[b][s] * T[b & 1][b >> 1];
          sq += N[b] * S[b & 1][b >> 1];
        }
        double du[3][3] = {{0}};
        for (int b = 0; b < 8; ++b)
          for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) du[i][j] += U[i][b & 1][b >> 1] * g[b][j];
        const double tr = du[0][0] + du[1][1] + du[2][2];
        for (int a = 0; a < 8; ++a)
          for (int i = 0; i < 3; ++i) {
            double acc = 0;
            for (int j = 0; j < 3; ++j) acc += (mu * (du[i][j] + du[j][i]) + (i == j ? lam * tr : 0.0)) * g[a][j];
            felr[i][a] -= w * det * acc;  // k_elasticity_residual: r[fi] -= wd * acc
          }
        for (int a = 0; a < 8; ++a) {
          fr[a] += w * det * (-kc * (g[a][0] * gT[0] + g[a][1] * gT[1] + g[a][2] * gT[2]) + N[a] * sq);
          for (int b = 0; b < 8; ++b) kr[a][b] += -kc * w * det * (g[a][0] * g[b][0] + g[a][1] * g[b][1] + g[a][2] * g[b][2]);
        }
      }
  double err = 0, nf = 0, nk = 0, ek = 0;
  for (int a = 0; a < 8; ++a) {
    err = fmax(err, fabs(fr[a] - fe[a & 1][a >> 1]));
    nf = fmax(nf, fabs(fr[a]));
    for (int b = 0; b < 8; ++b) {
      ek = fmax(ek, fabs(kr[a][b] - ke[sf_sym36(a, b)]));
      nk = fmax(nk, fabs(kr[a][b]));
    }
  }
  double ee = 0, ne = 0;
  for (int i = 0; i < 3; ++i)
    for (int a = 0; a < 8; ++a) {
      ee = fmax(ee, fabs(felr[i][a] - fel[i][a & 1][a >> 1]));
      ne = fmax(ne, fabs(felr[i][a]));
    }
  printf("NG=%d seed=%u  fe rel err %.2e   ke rel err %.2e   elasticity fe rel err %.2e\n", NG, seed, err / nf, ek / nk, ee / ne);
  return fmax(fmax(err / nf, ek / nk), ee / ne);
}
int main() {
  double e = 0;
  for (unsigned s = 1; s <= 3; ++s) {
    e = fmax(e, check<1>(s));
    e = fmax(e, check<2>(s));
    e = fmax(e, check<3>(s));
    e = fmax(e, check<4>(s));
  }
  printf(e < 1e-12 ? "OK\n" : "FAIL\n");
  return e < 1e-12 ? 0 : 1;
}
