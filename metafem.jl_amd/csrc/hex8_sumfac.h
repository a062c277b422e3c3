// Sum-factorised element routines of the trilinear hex-8 (tensor basis on [0,1]^3, NG-point Gauss rule per direction).
// Same quantities as update_BasicElements_3D + inv_Jac_3D + update_Basic_itgval_1_3D (mesh/unstructured_mesh/
// 4_Update_Integrator.jl:2-33,90-154) followed by the _Kval_Basic / _Res_Basic contractions of the thermal weak form
// (solver/06_FEM_Kernel.jl:28-45,65-79), evaluated through the tensor structure of the basis instead of the stored
// [q][a][s] table:
//   * a column dx/dxi_m of the Jacobian does not depend on xi_m: 3 * NG^2 distinct columns per element instead of 3 * NG^3,
//     each a bilinear interpolation of the four edge vectors along direction m;
//   * J^-1 det = (c1 x c2, c2 x c0, c0 x c1): the physical gradients grad N_a = J^-T dN_a are never formed -- the weak form
//     is contracted in reference coordinates with G_q = (coef w_q / det_q) adj adj^T (symmetric 3 x 3) and the products
//     dN_a/dxi_m (q) dN_b/dxi_n (q) are summed over the Gauss points one direction at a time.
// ~870 FP64 operations per element for the residual and ~1250 for the 36 unique entries of Ke at NG = 2 (2400 / 2700 with the
// table form).  Host + device so that tools/host_check_hex8.cpp can compare them with the table form on the CPU.
#pragma once

#ifndef __HIPCC__
#define __host__
#define __device__
#define __forceinline__ inline
#endif
#define SF_HD __host__ __device__ __forceinline__

// ---- 1-D Gauss rule on [0,1] (spatial_discretization/103_Integrations.jl:1-12) and the linear basis at its points
SF_HD constexpr double sf_gp(int ng, int q) {
  return ng == 1 ? 0.0
         : ng == 2 ? (q == 0 ? -0.57735026918962576451 : 0.57735026918962576451)
         : ng == 3 ? (q == 0 ? -0.77459666924148337704 : q == 1 ? 0.0 : 0.77459666924148337704)
                   : (q == 0 ? -0.86113631159405257522 : q == 1 ? -0.33998104358485626480 : q == 2 ? 0.33998104358485626480 : 0.86113631159405257522);
}
SF_HD constexpr double sf_gw(int ng, int q) {
  return ng == 1 ? 2.0
         : ng == 2 ? 1.0
         : ng == 3 ? (q == 1 ? 8.0 / 9.0 : 5.0 / 9.0)
                   : ((q == 0 || q == 3) ? 0.34785484513745385737 : 0.65214515486254614263);
}
template <int NG> SF_HD constexpr double sf_xi(int q) { return sf_gp(NG, q) / 2.0 + 0.5; }      // shift_gauss_point  :1
template <int NG> SF_HD constexpr double sf_w(int q) { return sf_gw(NG, q) / 2.0; }              // shift_gauss_weight :2
template <int NG> SF_HD constexpr double sf_phi(int b, int q) { return b ? sf_xi<NG>(q) : 1.0 - sf_xi<NG>(q); }
// products phi_a phi_b at a Gauss point for the unordered pair p = a + b (0: (0,0), 1: (0,1), 2: (1,1))
template <int NG> SF_HD constexpr double sf_pp(int p, int q) {
  return p == 0 ? sf_phi<NG>(0, q) * sf_phi<NG>(0, q) : p == 1 ? sf_phi<NG>(0, q) * sf_phi<NG>(1, q) : sf_phi<NG>(1, q) * sf_phi<NG>(1, q);
}

// Nodal values of one element: v[bx][byz], byz = by + 2 bz (bx is the sweep direction of the assembly kernels).
// Reference-coordinate derivatives at the Gauss points:  d0[qy][qz] = dv/dxi_0,  d1[qx][qz] = dv/dxi_1,  d2[qx][qy] = dv/dxi_2.
template <int NG>
SF_HD void sf_ref_grads(const double (&v)[2][4], double (&d0)[NG][NG], double (&d1)[NG][NG], double (&d2)[NG][NG]) {
  {
    double e[2][2], a[NG][2];
#pragma unroll
    for (int bz = 0; bz < 2; ++bz)
#pragma unroll
      for (int by = 0; by < 2; ++by) e[by][bz] = v[1][by + 2 * bz] - v[0][by + 2 * bz];
#pragma unroll
    for (int q = 0; q < NG; ++q)
#pragma unroll
      for (int bz = 0; bz < 2; ++bz) a[q][bz] = sf_phi<NG>(0, q) * e[0][bz] + sf_phi<NG>(1, q) * e[1][bz];
#pragma unroll
    for (int qy = 0; qy < NG; ++qy)
#pragma unroll
      for (int qz = 0; qz < NG; ++qz) d0[qy][qz] = sf_phi<NG>(0, qz) * a[qy][0] + sf_phi<NG>(1, qz) * a[qy][1];
  }
  {
    double e[2][2], a[NG][2];
#pragma unroll
    for (int bz = 0; bz < 2; ++bz)
#pragma unroll
      for (int bx = 0; bx < 2; ++bx) e[bx][bz] = v[bx][1 + 2 * bz] - v[bx][2 * bz];
#pragma unroll
    for (int q = 0; q < NG; ++q)
#pragma unroll
      for (int bz = 0; bz < 2; ++bz) a[q][bz] = sf_phi<NG>(0, q) * e[0][bz] + sf_phi<NG>(1, q) * e[1][bz];
#pragma unroll
    for (int qx = 0; qx < NG; ++qx)
#pragma unroll
      for (int qz = 0; qz < NG; ++qz) d1[qx][qz] = sf_phi<NG>(0, qz) * a[qx][0] + sf_phi<NG>(1, qz) * a[qx][1];
  }
  {
    double e[2][2], a[NG][2];
#pragma unroll
    for (int by = 0; by < 2; ++by)
#pragma unroll
      for (int bx = 0; bx < 2; ++bx) e[bx][by] = v[bx][by + 2] - v[bx][by];
#pragma unroll
    for (int q = 0; q < NG; ++q)
#pragma unroll
      for (int by = 0; by < 2; ++by) a[q][by] = sf_phi<NG>(0, q) * e[0][by] + sf_phi<NG>(1, q) * e[1][by];
#pragma unroll
    for (int qx = 0; qx < NG; ++qx)
#pragma unroll
      for (int qy = 0; qy < NG; ++qy) d2[qx][qy] = sf_phi<NG>(0, qy) * a[qx][0] + sf_phi<NG>(1, qy) * a[qx][1];
  }
}

// values at the Gauss points: out[qx][qy][qz] = sum_b N_b(q) v_b
template <int NG>
SF_HD void sf_ref_interp(const double (&v)[2][4], double (&out)[NG][NG][NG]) {
  double a[NG][4];
#pragma unroll
  for (int qx = 0; qx < NG; ++qx)
#pragma unroll
    for (int c = 0; c < 4; ++c) a[qx][c] = sf_phi<NG>(0, qx) * v[0][c] + sf_phi<NG>(1, qx) * v[1][c];
#pragma unroll
  for (int qx = 0; qx < NG; ++qx)
#pragma unroll
    for (int qy = 0; qy < NG; ++qy) {
      const double b0 = sf_phi<NG>(0, qy) * a[qx][0] + sf_phi<NG>(1, qy) * a[qx][1];
      const double b1 = sf_phi<NG>(0, qy) * a[qx][2] + sf_phi<NG>(1, qy) * a[qx][3];
#pragma unroll
      for (int qz = 0; qz < NG; ++qz) out[qx][qy][qz] = sf_phi<NG>(0, qz) * b0 + sf_phi<NG>(1, qz) * b1;
    }
}

// transposes: fe[bx][byz] += sum_q ( sum_m dN_b/dxi_m(q) F[m][q] + N_b(q) S[q] )
template <int NG>
SF_HD void sf_ref_grads_T(const double (&F)[3][NG][NG][NG], double (&fe)[2][4]) {
  {  // m = 0: dN_b/dxi_0 = sigma_bx phi_by(qy) phi_bz(qz)
    double g[NG][NG];
#pragma unroll
    for (int qy = 0; qy < NG; ++qy)
#pragma unroll
      for (int qz = 0; qz < NG; ++qz) {
        double s = F[0][0][qy][qz];
#pragma unroll
        for (int qx = 1; qx < NG; ++qx) s += F[0][qx][qy][qz];
        g[qy][qz] = s;
      }
#pragma unroll
    for (int by = 0; by < 2; ++by)
#pragma unroll
      for (int bz = 0; bz < 2; ++bz) {
        double l = 0.0;
#pragma unroll
        for (int qz = 0; qz < NG; ++qz) {
          double h = sf_phi<NG>(by, 0) * g[0][qz];
#pragma unroll
          for (int qy = 1; qy < NG; ++qy) h += sf_phi<NG>(by, qy) * g[qy][qz];
          l += sf_phi<NG>(bz, qz) * h;
        }
        fe[0][by + 2 * bz] -= l;
        fe[1][by + 2 * bz] += l;
      }
  }
  {  // m = 1: phi_bx(qx) sigma_by phi_bz(qz)
    double g[NG][NG];
#pragma unroll
    for (int qx = 0; qx < NG; ++qx)
#pragma unroll
      for (int qz = 0; qz < NG; ++qz) {
        double s = F[1][qx][0][qz];
#pragma unroll
        for (int qy = 1; qy < NG; ++qy) s += F[1][qx][qy][qz];
        g[qx][qz] = s;
      }
#pragma unroll
    for (int bx = 0; bx < 2; ++bx)
#pragma unroll
      for (int bz = 0; bz < 2; ++bz) {
        double l = 0.0;
#pragma unroll
        for (int qz = 0; qz < NG; ++qz) {
          double h = sf_phi<NG>(bx, 0) * g[0][qz];
#pragma unroll
          for (int qx = 1; qx < NG; ++qx) h += sf_phi<NG>(bx, qx) * g[qx][qz];
          l += sf_phi<NG>(bz, qz) * h;
        }
        fe[bx][2 * bz] -= l;
        fe[bx][1 + 2 * bz] += l;
      }
  }
  {  // m = 2: phi_bx(qx) phi_by(qy) sigma_bz
    double g[NG][NG];
#pragma unroll
    for (int qx = 0; qx < NG; ++qx)
#pragma unroll
      for (int qy = 0; qy < NG; ++qy) {
        double s = F[2][qx][qy][0];
#pragma unroll
        for (int qz = 1; qz < NG; ++qz) s += F[2][qx][qy][qz];
        g[qx][qy] = s;
      }
#pragma unroll
    for (int bx = 0; bx < 2; ++bx)
#pragma unroll
      for (int by = 0; by < 2; ++by) {
        double l = 0.0;
#pragma unroll
        for (int qy = 0; qy < NG; ++qy) {
          double h = sf_phi<NG>(bx, 0) * g[0][qy];
#pragma unroll
          for (int qx = 1; qx < NG; ++qx) h += sf_phi<NG>(bx, qx) * g[qx][qy];
          l += sf_phi<NG>(by, qy) * h;
        }
        fe[bx][by] -= l;
        fe[bx][by + 2] += l;
      }
  }
}
template <int NG>
SF_HD void sf_ref_interp_T(const double (&S)[NG][NG][NG], double (&fe)[2][4]) {
  double a[NG][NG][2];  // [qx][qy][bz]
#pragma unroll
  for (int qx = 0; qx < NG; ++qx)
#pragma unroll
    for (int qy = 0; qy < NG; ++qy)
#pragma unroll
      for (int bz = 0; bz < 2; ++bz) {
        double s = sf_phi<NG>(bz, 0) * S[qx][qy][0];
#pragma unroll
        for (int qz = 1; qz < NG; ++qz) s += sf_phi<NG>(bz, qz) * S[qx][qy][qz];
        a[qx][qy][bz] = s;
      }
#pragma unroll
  for (int qx = 0; qx < NG; ++qx) {
    double b[4];
#pragma unroll
    for (int bz = 0; bz < 2; ++bz)
#pragma unroll
      for (int by = 0; by < 2; ++by) {
        double s = sf_phi<NG>(by, 0) * a[qx][0][bz];
#pragma unroll
        for (int qy = 1; qy < NG; ++qy) s += sf_phi<NG>(by, qy) * a[qx][qy][bz];
        b[by + 2 * bz] = s;
      }
#pragma unroll
    for (int bx = 0; bx < 2; ++bx)
#pragma unroll
      for (int c = 0; c < 4; ++c) fe[bx][c] += sf_phi<NG>(bx, qx) * b[c];
  }
}

// Geometry at the Gauss points from the nodal coordinates X[i][bx][byz]:  columns of J (c0, c1, c2 = dx/dxi_0,1,2).
template <int NG>
struct SfCols {
  double c0[3][NG][NG], c1[3][NG][NG], c2[3][NG][NG];  // [component][..][..] indexed as sf_ref_grads
};
template <int NG>
SF_HD void sf_columns(const double (&X)[3][2][4], SfCols<NG>& C) {
#pragma unroll
  for (int i = 0; i < 3; ++i) sf_ref_grads<NG>(X[i], C.c0[i], C.c1[i], C.c2[i]);
}
// rows of adj J = det * J^-1 (inv_Jac_3D :90-121) and det at Gauss point (qx, qy, qz)
template <int NG>
SF_HD double sf_adjugate(const SfCols<NG>& C, int qx, int qy, int qz, double (&R)[3][3]) {
  const double a0 = C.c0[0][qy][qz], a1 = C.c0[1][qy][qz], a2 = C.c0[2][qy][qz];
  const double b0 = C.c1[0][qx][qz], b1 = C.c1[1][qx][qz], b2 = C.c1[2][qx][qz];
  const double d0 = C.c2[0][qx][qy], d1 = C.c2[1][qx][qy], d2 = C.c2[2][qx][qy];
  R[0][0] = b1 * d2 - b2 * d1; R[0][1] = b2 * d0 - b0 * d2; R[0][2] = b0 * d1 - b1 * d0;  // c1 x c2
  R[1][0] = d1 * a2 - d2 * a1; R[1][1] = d2 * a0 - d0 * a2; R[1][2] = d0 * a1 - d1 * a0;  // c2 x c0
  R[2][0] = a1 * b2 - a2 * b1; R[2][1] = a2 * b0 - a0 * b2; R[2][2] = a0 * b1 - a1 * b0;  // c0 x c1
  return a0 * R[0][0] + a1 * R[0][1] + a2 * R[0][2];
}

// ---- thermal residual of one element (3D_Script.jl:30-31, domain part):
//   fe[a] = sum_q w det ( -k grad N_a . grad T + N_a s )
template <int NG, bool AFFINE = false>  // AFFINE: see sf_thermal_ke (one adjugate for the element)
SF_HD void sf_thermal_fe(const double (&X)[3][2][4], const double (&T)[2][4], const double (&Sn)[2][4], bool has_src, double kcond,
                         double (&fe)[2][4]) {
  SfCols<NG> C;
  sf_columns<NG>(X, C);
  double t0[NG][NG], t1[NG][NG], t2[NG][NG];
  sf_ref_grads<NG>(T, t0, t1, t2);
  double F[3][NG][NG][NG], S[NG][NG][NG];
  if (has_src) sf_ref_interp<NG>(Sn, S);
  double R0[3][3];
  double det0 = 1.0;
  if (AFFINE) det0 = sf_adjugate<NG>(C, 0, 0, 0, R0);
#pragma unroll
  for (int qx = 0; qx < NG; ++qx)
#pragma unroll
    for (int qy = 0; qy < NG; ++qy)
#pragma unroll
      for (int qz = 0; qz < NG; ++qz) {
        double R[3][3];
        double det = det0;
        if (AFFINE) {
#pragma unroll
          for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int n = 0; n < 3; ++n) R[m][n] = R0[m][n];
        } else {
          det = sf_adjugate<NG>(C, qx, qy, qz, R);
        }
        const double w = sf_w<NG>(qx) * sf_w<NG>(qy) * sf_w<NG>(qz);
        const double g0 = t0[qy][qz], g1 = t1[qx][qz], g2 = t2[qx][qy];
        double v[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) v[s] = R[0][s] * g0 + R[1][s] * g1 + R[2][s] * g2;  // det * grad T
        const double sc = (-kcond * w) / det;
#pragma unroll
        for (int m = 0; m < 3; ++m) F[m][qx][qy][qz] = sc * (R[m][0] * v[0] + R[m][1] * v[1] + R[m][2] * v[2]);
        if (has_src) S[qx][qy][qz] *= w * det;
      }
#pragma unroll
  for (int bx = 0; bx < 2; ++bx)
#pragma unroll
    for (int c = 0; c < 4; ++c) fe[bx][c] = 0.0;
  sf_ref_grads_T<NG>(F, fe);
  if (has_src) sf_ref_interp_T<NG>(S, fe);
}

// ---- elasticity residual of one element (examples/linear_elasticity/cantilever/3D_Script.jl:52-58, domain part):
//   fe[i][a] = -sum_q w det sigma_ij(u) d_jN_a,   sigma = lam tr(eps) I + 2 mu eps,  eps = sym(grad u)
// U[i] = nodal values of displacement component i.  Contracted in reference coordinates like the thermal residual: with R = adj J
// (rows of det J^-1), grad u_i = R^T d(u_i)/dxi / det and fe[i][a] = sum_q sum_m dN_a/dxi_m(q) F_i[m][q] with F_i[m] = -w sum_j R[m][j] sigma_ij.
template <int NG>
SF_HD void sf_elasticity_fe(const double (&X)[3][2][4], const double (&U)[3][2][4], double lam, double mu, double (&fe)[3][2][4]) {
  SfCols<NG> C;
  sf_columns<NG>(X, C);
  double u0[3][NG][NG], u1[3][NG][NG], u2[3][NG][NG];
#pragma unroll
  for (int i = 0; i < 3; ++i) sf_ref_grads<NG>(U[i], u0[i], u1[i], u2[i]);
  // NG <= 2: all three fields in one pass over the Gauss points (F = 9 NG^3 doubles); larger rules: one field per pass, so that only
  // 3 NG^3 doubles are live (grad u is recomputed per pass; the products are the same, so are the results)
  constexpr int NP = NG <= 2 ? 1 : 3, NF = NG <= 2 ? 3 : 1;
#pragma unroll
  for (int pass = 0; pass < NP; ++pass) {
    double F[NF][3][NG][NG][NG];  // [field][m]
#pragma unroll
    for (int qx = 0; qx < NG; ++qx)
#pragma unroll
      for (int qy = 0; qy < NG; ++qy)
#pragma unroll
        for (int qz = 0; qz < NG; ++qz) {
          double R[3][3];
          const double det = sf_adjugate<NG>(C, qx, qy, qz, R);
          const double w = sf_w<NG>(qx) * sf_w<NG>(qy) * sf_w<NG>(qz);
          const double idet = 1.0 / det;
          double du[3][3];  // du[i][j] = d u_i / d x_j
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) du[i][j] = (R[0][j] * u0[i][qy][qz] + R[1][j] * u1[i][qx][qz] + R[2][j] * u2[i][qx][qy]) * idet;
          const double tr = du[0][0] + du[1][1] + du[2][2];
          // fe[i][a] = -sum_q w det sum_j sigma_ij (1/det) sum_m R[m][j] dN_a/dxi_m = sum_q sum_m dN_a/dxi_m ( -w sum_j R[m][j] sigma_ij )
#pragma unroll
          for (int fi = 0; fi < NF; ++fi) {
            const int i = NP == 1 ? fi : pass;
            double sg[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) sg[j] = mu * (du[i][j] + du[j][i]);
            sg[i] += lam * tr;
#pragma unroll
            for (int m = 0; m < 3; ++m) F[fi][m][qx][qy][qz] = -w * (R[m][0] * sg[0] + R[m][1] * sg[1] + R[m][2] * sg[2]);
          }
        }
#pragma unroll
    for (int fi = 0; fi < NF; ++fi) {
      const int i = NP == 1 ? fi : pass;
#pragma unroll
      for (int bx = 0; bx < 2; ++bx)
#pragma unroll
        for (int c = 0; c < 4; ++c) fe[i][bx][c] = 0.0;
      sf_ref_grads_T<NG>(F[fi], fe[i]);
    }
  }
}

// All eight nodes of X[i][bx][c] (c = by + 2 bz) on the affine image of the reference cube, to 16 ulp of the coordinates' magnitude (what the Jacobian's own
// cancellation error is made of): every element of make_Brick until a caller moves coordinates.
SF_HD bool sf_is_affine(const double (&X)[3][2][4]) {
  bool ok = true;
  for (int i = 0; i < 3; ++i) {
    const double x0 = X[i][0][0];
    const double e0 = X[i][1][0] - x0, e1 = X[i][0][1] - x0, e2 = X[i][0][2] - x0;
    const double a0 = e0 < 0 ? -e0 : e0, a1 = e1 < 0 ? -e1 : e1, a2 = e2 < 0 ? -e2 : e2, ax = x0 < 0 ? -x0 : x0;
    const double tol = 3.6e-15 * (ax + a0 + a1 + a2);
    const double d1 = X[i][1][1] - (x0 + e0 + e1), d2 = X[i][1][2] - (x0 + e0 + e2), d3 = X[i][0][3] - (x0 + e1 + e2), d4 = X[i][1][3] - (x0 + e0 + e1 + e2);
    ok = ok && (d1 < 0 ? -d1 : d1) <= tol && (d2 < 0 ? -d2 : d2) <= tol && (d3 < 0 ? -d3 : d3) <= tol && (d4 < 0 ? -d4 : d4) <= tol;
  }
  return ok;
}

// ---- thermal element matrix, the 36 unique entries of the symmetric 8 x 8 Ke = sum_q w det (-k) grad N_a . grad N_b.
// Node a = ax + 2 ay + 4 az (tensor order, x fastest = c_dN's order); packing ke36[sym36(a, b)] of assemble_hex8.hip.
SF_HD constexpr int sf_sym36(int a, int b) {
  return a <= b ? a * 8 - (a * (a - 1)) / 2 + (b - a) : b * 8 - (b * (b - 1)) / 2 + (a - b);
}
// AFFINE = true (round 4): the caller has found the element to be a parallelepiped (sf_is_affine below) -- the adjugate and det are those of ONE Gauss point
// (they are the same at all of them up to round-off), 1 instead of NG^3 evaluations; everything else is the same code.
template <int NG, bool AFFINE = false>
SF_HD void sf_thermal_ke(const double (&X)[3][2][4], double kcond, double (&ke)[36]) {
  SfCols<NG> C;
  sf_columns<NG>(X, C);
  double G[6][NG][NG][NG];  // 00 01 02 11 12 22
  double R0[3][3];
  double det0 = 1.0;
  if (AFFINE) det0 = sf_adjugate<NG>(C, 0, 0, 0, R0);
#pragma unroll
  for (int qx = 0; qx < NG; ++qx)
#pragma unroll
    for (int qy = 0; qy < NG; ++qy)
#pragma unroll
      for (int qz = 0; qz < NG; ++qz) {
        double R[3][3];
        double det = det0;
        if (AFFINE) {
#pragma unroll
          for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int n = 0; n < 3; ++n) R[m][n] = R0[m][n];
        } else {
          det = sf_adjugate<NG>(C, qx, qy, qz, R);
        }
        const double sc = (-kcond * (sf_w<NG>(qx) * sf_w<NG>(qy) * sf_w<NG>(qz))) / det;
        int t = 0;
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
          for (int n = m; n < 3; ++n) G[t++][qx][qy][qz] = sc * (R[m][0] * R[n][0] + R[m][1] * R[n][1] + R[m][2] * R[n][2]);
      }
  // diagonal pairs (m, m): the direction m itself only enters through the sign, the other two through pair products
  double T00[3][3], T11[3][3], T22[3][3];  // [pair in the first remaining direction][pair in the second]
  {
    double g[NG][NG], u[3][NG];
#pragma unroll
    for (int qy = 0; qy < NG; ++qy)
#pragma unroll
      for (int qz = 0; qz < NG; ++qz) {
        double s = G[0][0][qy][qz];
#pragma unroll
        for (int qx = 1; qx < NG; ++qx) s += G[0][qx][qy][qz];
        g[qy][qz] = s;
      }
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int qz = 0; qz < NG; ++qz) {
        double s = sf_pp<NG>(p, 0) * g[0][qz];
#pragma unroll
        for (int qy = 1; qy < NG; ++qy) s += sf_pp<NG>(p, qy) * g[qy][qz];
        u[p][qz] = s;
      }
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        double s = sf_pp<NG>(r, 0) * u[p][0];
#pragma unroll
        for (int qz = 1; qz < NG; ++qz) s += sf_pp<NG>(r, qz) * u[p][qz];
        T00[p][r] = s;  // p = (ay, by), r = (az, bz)
      }
  }
  {
    double g[NG][NG], u[3][NG];
#pragma unroll
    for (int qx = 0; qx < NG; ++qx)
#pragma unroll
      for (int qz = 0; qz < NG; ++qz) {
        double s = G[3][qx][0][qz];
#pragma unroll
        for (int qy = 1; qy < NG; ++qy) s += G[3][qx][qy][qz];
        g[qx][qz] = s;
      }
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int qz = 0; qz < NG; ++qz) {
        double s = sf_pp<NG>(p, 0) * g[0][qz];
#pragma unroll
        for (int qx = 1; qx < NG; ++qx) s += sf_pp<NG>(p, qx) * g[qx][qz];
        u[p][qz] = s;
      }
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        double s = sf_pp<NG>(r, 0) * u[p][0];
#pragma unroll
        for (int qz = 1; qz < NG; ++qz) s += sf_pp<NG>(r, qz) * u[p][qz];
        T11[p][r] = s;  // p = (ax, bx), r = (az, bz)
      }
  }
  {
    double g[NG][NG], u[3][NG];
#pragma unroll
    for (int qx = 0; qx < NG; ++qx)
#pragma unroll
      for (int qy = 0; qy < NG; ++qy) {
        double s = G[5][qx][qy][0];
#pragma unroll
        for (int qz = 1; qz < NG; ++qz) s += G[5][qx][qy][qz];
        g[qx][qy] = s;
      }
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int qy = 0; qy < NG; ++qy) {
        double s = sf_pp<NG>(p, 0) * g[0][qy];
#pragma unroll
        for (int qx = 1; qx < NG; ++qx) s += sf_pp<NG>(p, qx) * g[qx][qy];
        u[p][qy] = s;
      }
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        double s = sf_pp<NG>(r, 0) * u[p][0];
#pragma unroll
        for (int qy = 1; qy < NG; ++qy) s += sf_pp<NG>(r, qy) * u[p][qy];
        T22[p][r] = s;  // p = (ax, bx), r = (ay, by)
      }
  }
  // mixed pairs: term_mn[a][b] = sum_q dN_a/dxi_m G_mn dN_b/dxi_n
  //   T01[bx][ay][r = (az, bz)] = sum_q phi_bx(qx) phi_ay(qy) pp_r(qz) G01
  //   T02[bx][az][r = (ay, by)] = sum_q phi_bx(qx) phi_az(qz) pp_r(qy) G02
  //   T12[by][az][r = (ax, bx)] = sum_q phi_by(qy) phi_az(qz) pp_r(qx) G12
  double T01[2][2][3], T02[2][2][3], T12[2][2][3];
  {
    double v[2][NG][NG];
#pragma unroll
    for (int bx = 0; bx < 2; ++bx)
#pragma unroll
      for (int qy = 0; qy < NG; ++qy)
#pragma unroll
        for (int qz = 0; qz < NG; ++qz) {
          double s = sf_phi<NG>(bx, 0) * G[1][0][qy][qz];
#pragma unroll
          for (int qx = 1; qx < NG; ++qx) s += sf_phi<NG>(bx, qx) * G[1][qx][qy][qz];
          v[bx][qy][qz] = s;
        }
#pragma unroll
    for (int bx = 0; bx < 2; ++bx)
#pragma unroll
      for (int ay = 0; ay < 2; ++ay) {
        double wq[NG];
#pragma unroll
        for (int qz = 0; qz < NG; ++qz) {
          double s = sf_phi<NG>(ay, 0) * v[bx][0][qz];
#pragma unroll
          for (int qy = 1; qy < NG; ++qy) s += sf_phi<NG>(ay, qy) * v[bx][qy][qz];
          wq[qz] = s;
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          double s = sf_pp<NG>(r, 0) * wq[0];
#pragma unroll
          for (int qz = 1; qz < NG; ++qz) s += sf_pp<NG>(r, qz) * wq[qz];
          T01[bx][ay][r] = s;
        }
      }
  }
  {
    double v[2][NG][NG];  // [bx][qy][qz]
#pragma unroll
    for (int bx = 0; bx < 2; ++bx)
#pragma unroll
      for (int qy = 0; qy < NG; ++qy)
#pragma unroll
        for (int qz = 0; qz < NG; ++qz) {
          double s = sf_phi<NG>(bx, 0) * G[2][0][qy][qz];
#pragma unroll
          for (int qx = 1; qx < NG; ++qx) s += sf_phi<NG>(bx, qx) * G[2][qx][qy][qz];
          v[bx][qy][qz] = s;
        }
#pragma unroll
    for (int bx = 0; bx < 2; ++bx)
#pragma unroll
      for (int az = 0; az < 2; ++az) {
        double wq[NG];
#pragma unroll
        for (int qy = 0; qy < NG; ++qy) {
          double s = sf_phi<NG>(az, 0) * v[bx][qy][0];
#pragma unroll
          for (int qz = 1; qz < NG; ++qz) s += sf_phi<NG>(az, qz) * v[bx][qy][qz];
          wq[qy] = s;
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          double s = sf_pp<NG>(r, 0) * wq[0];
#pragma unroll
          for (int qy = 1; qy < NG; ++qy) s += sf_pp<NG>(r, qy) * wq[qy];
          T02[bx][az][r] = s;
        }
      }
  }
  {
    double v[2][NG][NG];  // [by][qx][qz]
#pragma unroll
    for (int by = 0; by < 2; ++by)
#pragma unroll
      for (int qx = 0; qx < NG; ++qx)
#pragma unroll
        for (int qz = 0; qz < NG; ++qz) {
          double s = sf_phi<NG>(by, 0) * G[4][qx][0][qz];
#pragma unroll
          for (int qy = 1; qy < NG; ++qy) s += sf_phi<NG>(by, qy) * G[4][qx][qy][qz];
          v[by][qx][qz] = s;
        }
#pragma unroll
    for (int by = 0; by < 2; ++by)
#pragma unroll
      for (int az = 0; az < 2; ++az) {
        double wq[NG];
#pragma unroll
        for (int qx = 0; qx < NG; ++qx) {
          double s = sf_phi<NG>(az, 0) * v[by][qx][0];
#pragma unroll
          for (int qz = 1; qz < NG; ++qz) s += sf_phi<NG>(az, qz) * v[by][qx][qz];
          wq[qx] = s;
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          double s = sf_pp<NG>(r, 0) * wq[0];
#pragma unroll
          for (int qx = 1; qx < NG; ++qx) s += sf_pp<NG>(r, qx) * wq[qx];
          T12[by][az][r] = s;
        }
      }
  }
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = a; b < 8; ++b) {
      const int ax = a & 1, ay = (a >> 1) & 1, az = a >> 2, bx = b & 1, by = (b >> 1) & 1, bz = b >> 2;
      const double sxx = (ax == bx) ? 1.0 : -1.0, syy = (ay == by) ? 1.0 : -1.0, szz = (az == bz) ? 1.0 : -1.0;
      const double sxy = (ax == by) ? 1.0 : -1.0, syx = (bx == ay) ? 1.0 : -1.0;
      const double sxz = (ax == bz) ? 1.0 : -1.0, szx = (bx == az) ? 1.0 : -1.0;
      const double syz = (ay == bz) ? 1.0 : -1.0, szy = (by == az) ? 1.0 : -1.0;
      const int px = ax + bx, py = ay + by, pz = az + bz;
      double s = sxx * T00[py][pz];
      s += syy * T11[px][pz];
      s += szz * T22[px][py];
      s += sxy * T01[bx][ay][pz];
      s += syx * T01[ax][by][pz];
      s += sxz * T02[bx][az][py];
      s += szx * T02[ax][bz][py];
      s += syz * T12[by][az][px];
      s += szy * T12[ay][bz][px];
      ke[sf_sym36(a, b)] = s;
    }
}
