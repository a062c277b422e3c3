// Fused hex-8 assembly: geometry update + element operators + scatter in ONE pass per quantity.
// Replaces, for the constant-coefficient thermal / elasticity weak forms:
//   update_BasicElements_3D + inv_Jac_3D + update_Basic_itgval_1_3D   mesh/unstructured_mesh/4_Update_Integrator.jl:2-33,90-154
//   update_BasicBoundary_3D + tangents/normals                        :35-75,173-226
//   _Var_Basic / _Kval_Basic / _Res_Basic                             solver/06_FEM_Kernel.jl:1-13,28-45,65-79
//   the generated K_linear / K_nonlinear bodies                        solver/05_CodeGenerator.jl:52-154
// The reference stores 4 KB of physical-space basis tables per element and launches one
// thread-per-element kernel PER bilinear term with FP64 atomics into hash-ordered slots.  Here nothing
// per-element is stored: a row-owner thread (one per control point) recomputes J, det, J^-1 and the
// pushed-forward gradients of its <= 8 adjacent elements from nodal coordinates, accumulates its own
// matrix row and writes it once -- race-free without atomics or colours, CSR order, nnz*8 B written
// exactly once (the algorithmic minimum).  Thermal rows are staged in LDS and leave the CU as one
// contiguous, fully coalesced stream per workgroup.
#include "brick.h"
#include "hex8_sumfac.h"

// reference-cell tables for the tensor hex-8 on [0,1]^3 (spatial_discretization/102_Interpolations.jl:30-39,
// 103_Integrations.jl:1-19): index [q][b], q = qx + ng*(qy + ng*qz) (x fastest), b = bx + 2*by + 4*bz.
__constant__ double c_w[BRICK_MAX_Q];
__constant__ double c_N[BRICK_MAX_Q][8];
__constant__ double c_dN[BRICK_MAX_Q][8][3];
// face tables: 2-D Gauss on [0,1]^2, bilinear face basis [q][c], c = c1 + 2*c2 (first tangential coord fastest)
__constant__ double c_xi[BRICK_MAX_Q][6];  // xi0, xi1, xi2, xi1 xi2, xi0 xi2, xi0 xi1 at Gauss point q (trilinear-map form of the Jacobian)
__constant__ double c_fw[BRICK_MAX_NG * BRICK_MAX_NG];
__constant__ double c_M[8][8][3][3];   // reference integrals M[a][b][m][n] = sum_q w_q dN_a,m dN_b,n of THIS quadrature (affine-element shortcut of the elasticity matrix)
__constant__ double c_fN[BRICK_MAX_NG * BRICK_MAX_NG][4];
__constant__ double c_fdN[BRICK_MAX_NG * BRICK_MAX_NG][4][2];
static std::atomic<int> g_tables_ng{0};

static const double GP[4][4] = {{0.0, 0, 0, 0},
                                {-0.57735026918962576451, 0.57735026918962576451, 0, 0},
                                {-0.77459666924148337704, 0.0, 0.77459666924148337704, 0},
                                {-0.86113631159405257522, -0.33998104358485626480, 0.33998104358485626480, 0.86113631159405257522}};
static const double GW[4][4] = {{2.0, 0, 0, 0},
                                {1.0, 1.0, 0, 0},
                                {5.0 / 9.0, 8.0 / 9.0, 5.0 / 9.0, 0},
                                {0.34785484513745385737, 0.65214515486254614263, 0.65214515486254614263, 0.34785484513745385737}};

int mfem_hex8_upload_tables(int ng) {
  static std::mutex mu;  // uploads from two host threads must not interleave (the tables themselves are process-wide: see the threading note in include/metafem_mi355x.h)
  std::lock_guard<std::mutex> lk(mu);
  if (g_tables_ng == ng) return MFEM_OK;
  double w[BRICK_MAX_Q], N[BRICK_MAX_Q][8], dN[BRICK_MAX_Q][8][3], xiq[BRICK_MAX_Q][6];
  double fw[16], fN[16][4], fdN[16][4][2];
  double gp[4], gw[4];
  for (int i = 0; i < ng; ++i) {
    gp[i] = GP[ng - 1][i] / 2.0 + 0.5;  // shift_gauss_point  103_Integrations.jl:1
    gw[i] = GW[ng - 1][i] / 2.0;        // shift_gauss_weight :2
  }
  memset(w, 0, sizeof(w)); memset(N, 0, sizeof(N)); memset(dN, 0, sizeof(dN)); memset(xiq, 0, sizeof(xiq));
  memset(fw, 0, sizeof(fw)); memset(fN, 0, sizeof(fN)); memset(fdN, 0, sizeof(fdN));
  for (int qz = 0; qz < ng; ++qz)
    for (int qy = 0; qy < ng; ++qy)
      for (int qx = 0; qx < ng; ++qx) {
        const int q = qx + ng * (qy + ng * qz);
        const double xi[3] = {gp[qx], gp[qy], gp[qz]};
        w[q] = gw[qx] * gw[qy] * gw[qz];
        xiq[q][0] = xi[0]; xiq[q][1] = xi[1]; xiq[q][2] = xi[2];
        xiq[q][3] = xi[1] * xi[2]; xiq[q][4] = xi[0] * xi[2]; xiq[q][5] = xi[0] * xi[1];
        for (int b = 0; b < 8; ++b) {
          double f[3], df[3];
          for (int d = 0; d < 3; ++d) {
            const int bd = (b >> d) & 1;
            f[d] = bd ? xi[d] : 1.0 - xi[d];
            df[d] = bd ? 1.0 : -1.0;
          }
          N[q][b] = f[0] * f[1] * f[2];
          dN[q][b][0] = df[0] * f[1] * f[2];
          dN[q][b][1] = f[0] * df[1] * f[2];
          dN[q][b][2] = f[0] * f[1] * df[2];
        }
      }
  for (int q2 = 0; q2 < ng; ++q2)
    for (int q1 = 0; q1 < ng; ++q1) {
      const int q = q1 + ng * q2;
      fw[q] = gw[q1] * gw[q2];
      const double xi[2] = {gp[q1], gp[q2]};
      for (int c = 0; c < 4; ++c) {
        const int c1 = c & 1, c2 = c >> 1;
        const double f1 = c1 ? xi[0] : 1.0 - xi[0], f2 = c2 ? xi[1] : 1.0 - xi[1];
        fN[q][c] = f1 * f2;
        fdN[q][c][0] = (c1 ? 1.0 : -1.0) * f2;
        fdN[q][c][1] = f1 * (c2 ? 1.0 : -1.0);
      }
    }
  {
    static double M[8][8][3][3];
    const int nq3 = ng * ng * ng;
    for (int a = 0; a < 8; ++a)
      for (int b = 0; b < 8; ++b)
        for (int m = 0; m < 3; ++m)
          for (int n = 0; n < 3; ++n) {
            double acc = 0.0;
            for (int q = 0; q < nq3; ++q) acc += w[q] * (dN[q][a][m] * dN[q][b][n]);  // (w * (x * y): M[a][b][m][n] and M[b][a][n][m] are the same bits)
            M[a][b][m][n] = acc;
          }
    MFEM_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_M), M, sizeof(M)));
  }
  MFEM_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_w), w, sizeof(w)));
  MFEM_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_N), N, sizeof(N)));
  MFEM_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_dN), dN, sizeof(dN)));
  MFEM_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_xi), xiq, sizeof(xiq)));
  MFEM_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_fw), fw, sizeof(fw)));
  MFEM_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_fN), fN, sizeof(fN)));
  MFEM_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_fdN), fdN, sizeof(fdN)));
  g_tables_ng = ng;
  return MFEM_OK;
}

// ---- geometry at one Gauss point: J = dx/dxi, det, J^-1 (adjugate, inv_Jac_3D :90-121),
//      physical gradients g[b][s] = sum_m dN_b/dxi_m Jinv[m][s] (:133-142); returns w_q * det (:30)
__device__ __forceinline__ double hex8_geom(const double (&X)[8][3], int q, double (&g)[8][3]) {
  double J[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      double s = 0.0;
#pragma unroll
      for (int b = 0; b < 8; ++b) s += c_dN[q][b][m] * X[b][i];
      J[i][m] = s;
    }
  const double det = J[0][0] * J[1][1] * J[2][2] - J[0][0] * J[1][2] * J[2][1] - J[0][1] * J[1][0] * J[2][2] +
                     J[0][1] * J[1][2] * J[2][0] + J[0][2] * J[1][0] * J[2][1] - J[0][2] * J[1][1] * J[2][0];
  const double id = 1.0 / det;
  double I[3][3];
  I[0][0] = (J[1][1] * J[2][2] - J[1][2] * J[2][1]) * id;
  I[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * id;
  I[0][2] = (J[0][1] * J[1][2] - J[1][1] * J[0][2]) * id;
  I[1][0] = (J[1][2] * J[2][0] - J[2][2] * J[1][0]) * id;
  I[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id;
  I[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * id;
  I[2][0] = (J[1][0] * J[2][1] - J[1][1] * J[2][0]) * id;
  I[2][1] = (J[0][1] * J[2][0] - J[2][1] * J[0][0]) * id;
  I[2][2] = (J[0][0] * J[1][1] - J[1][0] * J[0][1]) * id;
#pragma unroll
  for (int b = 0; b < 8; ++b)
#pragma unroll
    for (int s = 0; s < 3; ++s)
      g[b][s] = c_dN[q][b][0] * I[0][s] + c_dN[q][b][1] * I[1][s] + c_dN[q][b][2] * I[2][s];
  return c_w[q] * det;
}

__device__ __forceinline__ void hex8_load_coords(const BrickView& B, int I, int J, int K, double (&X)[8][3]) {
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const int64_t c = brick_cindex(B, I + (b & 1), J + ((b >> 1) & 1), K + (b >> 2));
    X[b][0] = B.X0[c];
    X[b][1] = B.X1[c];
    X[b][2] = B.X2[c];
  }
}

// the same, nodes b and b + 4 (neighbours along k: adjacent in memory) by one 16-byte load: 12 load instructions per element instead of 24
typedef double h8_d2 __attribute__((ext_vector_type(2), aligned(8)));
__device__ __forceinline__ void hex8_load_coords_pairs(const BrickView& B, int I, int J, int K, double (&X)[8][3]) {
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int64_t c = brick_cindex(B, I + (b & 1), J + ((b >> 1) & 1), K);
    const h8_d2 v0 = *reinterpret_cast<const h8_d2*>(B.X0 + c), v1 = *reinterpret_cast<const h8_d2*>(B.X1 + c), v2 = *reinterpret_cast<const h8_d2*>(B.X2 + c);
    X[b][0] = v0.x; X[b + 4][0] = v0.y;
    X[b][1] = v1.x; X[b + 4][1] = v1.y;
    X[b][2] = v2.x; X[b + 4][2] = v2.y;
  }
}

// The same geometry from the element's trilinear map x(xi) = X_0 + sum_{b > 0} C[b - 1] prod_{d in b} xi_d (C: differences of the nodal
// coordinates along the set bits of b, one butterfly per element): a column of J is 3 multiply-adds instead of the 8 of the table form.
__device__ __forceinline__ void hex8_trilinear(const double (&X)[8][3], double (&C)[7][3]) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    double v[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) v[b] = X[b][i];
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
      for (int b = 0; b < 8; ++b)
        if (b & (1 << d)) v[b] -= v[b ^ (1 << d)];
#pragma unroll
    for (int b = 1; b < 8; ++b) C[b - 1][i] = v[b];
  }
}
__device__ __forceinline__ double hex8_geom_trilinear(const double (&C)[7][3], int q, double (&g)[8][3]) {
  const double x0 = c_xi[q][0], x1 = c_xi[q][1], x2 = c_xi[q][2], x12 = c_xi[q][3], x02 = c_xi[q][4], x01 = c_xi[q][5];
  double J[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    J[i][0] = C[0][i] + x1 * C[2][i] + x2 * C[4][i] + x12 * C[6][i];
    J[i][1] = C[1][i] + x0 * C[2][i] + x2 * C[5][i] + x02 * C[6][i];
    J[i][2] = C[3][i] + x0 * C[4][i] + x1 * C[5][i] + x01 * C[6][i];
  }
  const double det = J[0][0] * J[1][1] * J[2][2] - J[0][0] * J[1][2] * J[2][1] - J[0][1] * J[1][0] * J[2][2] +
                     J[0][1] * J[1][2] * J[2][0] + J[0][2] * J[1][0] * J[2][1] - J[0][2] * J[1][1] * J[2][0];
  const double id = 1.0 / det;
  double I[3][3];
  I[0][0] = (J[1][1] * J[2][2] - J[1][2] * J[2][1]) * id;
  I[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * id;
  I[0][2] = (J[0][1] * J[1][2] - J[1][1] * J[0][2]) * id;
  I[1][0] = (J[1][2] * J[2][0] - J[2][2] * J[1][0]) * id;
  I[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id;
  I[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * id;
  I[2][0] = (J[1][0] * J[2][1] - J[1][1] * J[2][0]) * id;
  I[2][1] = (J[0][1] * J[2][0] - J[2][1] * J[0][0]) * id;
  I[2][2] = (J[0][0] * J[1][1] - J[1][0] * J[0][1]) * id;
#pragma unroll
  for (int b = 0; b < 8; ++b)
#pragma unroll
    for (int s = 0; s < 3; ++s)
      g[b][s] = c_dN[q][b][0] * I[0][s] + c_dN[q][b][1] * I[1][s] + c_dN[q][b][2] * I[2][s];
  return c_w[q] * det;
}

// Face quadrature on the brick face with normal dim nd (0,1,2) -- tangential dims follow the reference
// (103_Integrations.jl:37): nd=0 -> (1,2), nd=1 -> (2,0), nd=2 -> (0,1).  Xf[c][3] are the 4 face nodes,
// c = c1 + 2*c2.  Returns w^s = w_q * |t1 x t2| (4_Update_Integrator.jl:71,210-226); `nrm` (optional) is
// the OUTWARD unit normal: the first tangent is negated on the low face (103_Integrations.jl:45).
__device__ __forceinline__ double face_geom(const double (&Xf)[4][3], int q, bool low_face, double* nrm) {
  double t1[3], t2[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      a += c_fdN[q][c][0] * Xf[c][i];
      b += c_fdN[q][c][1] * Xf[c][i];
    }
    t1[i] = low_face ? -a : a;
    t2[i] = b;
  }
  const double r0 = t1[1] * t2[2] - t1[2] * t2[1];
  const double r1 = -t1[0] * t2[2] + t1[2] * t2[0];
  const double r2 = t1[0] * t2[1] - t1[1] * t2[0];
  const double ld = sqrt(r0 * r0 + r1 * r1 + r2 * r2);
  if (nrm) {
    nrm[0] = r0 / ld;
    nrm[1] = r1 / ld;
    nrm[2] = r2 / ld;
  }
  return c_fw[q] * ld;
}

__device__ __forceinline__ int face_bit(int nd, int high) {
  // reference local face ids (002_Initialization.jl:8): 1 z=0, 2 y=0, 3 x=L, 4 y=L, 5 x=0, 6 z=L
  const int id = (nd == 0) ? (high ? 3 : 5) : (nd == 1) ? (high ? 4 : 2) : (high ? 6 : 1);
  return 1 << (id - 1);
}

__device__ __forceinline__ void node_ijk(const BrickView& B, int64_t node, int& i, int& j, int& k) {
  if (B.n_owned < ((int64_t)1 << 31)) {  // (uniform) two 32-bit divisions: ~60 instructions against ~300 for the 64-bit pair
    const uint32_t nd = (uint32_t)node, pl = (uint32_t)B.plane_len, m2 = (uint32_t)B.m2;
    const uint32_t ip = nd / pl, rem = nd - ip * pl, jj = rem / m2;
    i = (int)ip + B.plo;
    j = (int)jj;
    k = (int)(rem - jj * m2);
    return;
  }
  i = (int)(node / B.plane_len) + B.plo;
  const int64_t rem = node % B.plane_len;
  j = (int)(rem / B.m2);
  k = (int)(rem % B.m2);
}

// =================================================================================================
// Boundary-face visitor: for an owned node (i,j,k), every element face that (a) lies on a brick face
// selected in `mask` and (b) contains the node.  f(nd, side, ca, fn, Xf): nd normal dim, side 0 low /
// 1 high, ca = the node's id among the 4 face nodes, fn[c][3] lattice ids and Xf[c][3] coordinates.
// =================================================================================================
template <typename Fn>
__device__ __forceinline__ void visit_boundary_faces(const BrickView& B, int i, int j, int k, uint32_t mask, Fn f) {
  if (mask == 0u) return;
  const int idx[3] = {i, j, k};
  const int ne[3] = {B.ne0, B.ne1, B.ne2};
  for (int nd = 0; nd < 3; ++nd) {
    for (int side = 0; side < 2; ++side) {
      const bool on = side == 0 ? (idx[nd] == 0) : (idx[nd] == ne[nd]);
      if (!on || !(mask & face_bit(nd, side))) continue;
      const int t1 = (nd + 1) % 3, t2 = (nd + 2) % 3;
      for (int q4 = 0; q4 < 4; ++q4) {
        const int f1 = q4 & 1, f2 = q4 >> 1;
        const int E1 = idx[t1] - 1 + f1, E2 = idx[t2] - 1 + f2;
        if (E1 < 0 || E1 >= ne[t1] || E2 < 0 || E2 >= ne[t2]) continue;
        const int ca = (1 - f1) + 2 * (1 - f2);
        int fn[4][3];
        double Xf[4][3];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          fn[c][nd] = idx[nd];
          fn[c][t1] = E1 + (c & 1);
          fn[c][t2] = E2 + (c >> 1);
          const int64_t ci = brick_cindex(B, fn[c][0], fn[c][1], fn[c][2]);
          Xf[c][0] = B.X0[ci];
          Xf[c][1] = B.X1[ci];
          Xf[c][2] = B.X2[ci];
        }
        f(nd, side, ca, fn, Xf);
      }
    }
  }
}

// =================================================================================================
// Thermal K:  vals = sum_el sum_q w (-k) grad N_a . grad N_b  +  sum_facets sum_q w^s (-h) N_a N_b
// Row-owner with element sharing: a workgroup owns a 4 x 4 x 8 tile of control points (128 threads).
//   phase A: the 5 x 5 x 9 elements touching the tile are integrated ONCE each (one thread per element: J, det,
//            J^-1, gradients at every Gauss point -> the 36 unique entries of the symmetric 8 x 8 Ke) into LDS;
//   phase B: every control point gathers its row from the <= 8 adjacent Ke (64 statically indexed adds) and
//            writes its <= 27 CSR values once -- no atomics, no colours, nnz*8 B written exactly once.
// Geometry is evaluated 225/128 = 1.76x per element instead of 8x in the plain row-owner form.
// =================================================================================================
#define TT_NI 4
#define TT_NJ 4
#define TT_NK 8
#define TT_THREADS (TT_NI * TT_NJ * TT_NK)
#define TT_EI (TT_NI + 1)
#define TT_EJ (TT_NJ + 1)
#define TT_EK (TT_NK + 1)
#define TT_NEL (TT_EI * TT_EJ * TT_EK)
#define TT_KSTRIDE 37  // 36 unique entries + 1 pad (LDS bank spread)

__device__ __forceinline__ constexpr int sym36(int a, int b) {  // upper-triangular packing of a symmetric 8 x 8
  return a <= b ? a * 8 - (a * (a - 1)) / 2 + (b - a) : b * 8 - (b * (b - 1)) / 2 + (a - b);
}

struct TileGeom {
  int ti, tj, tk;  // first control point of the tile (global lattice ids)
};
__device__ __forceinline__ TileGeom tile_origin(const BrickView& B, int64_t tile) {
  const int ntk = (B.m2 + TT_NK - 1) / TT_NK, ntj = (B.m1 + TT_NJ - 1) / TT_NJ;
  TileGeom t;
  t.tk = (int)(tile % ntk) * TT_NK;
  t.tj = (int)((tile / ntk) % ntj) * TT_NJ;
  t.ti = (int)(tile / ((int64_t)ntk * ntj)) * TT_NI + B.plo;
  return t;
}

__global__ __launch_bounds__(TT_THREADS) void k_thermal_matrix(BrickView B, double kcond, double* __restrict__ vals) {
  __shared__ double Ke[TT_NEL * TT_KSTRIDE];
  const int tid = threadIdx.x;
  const TileGeom T = tile_origin(B, blockIdx.x);
  // ---- phase A: one thread per element of the halo'd tile
  for (int e = tid; e < TT_NEL; e += TT_THREADS) {
    const int ek = e % TT_EK, ej = (e / TT_EK) % TT_EJ, ei = e / (TT_EK * TT_EJ);
    const int I = T.ti - 1 + ei, J = T.tj - 1 + ej, K = T.tk - 1 + ek;
    // elements needed by owned rows only: I in [plo-1, phi-1]
    if (I < 0 || I >= B.ne0 || J < 0 || J >= B.ne1 || K < 0 || K >= B.ne2 || I < B.plo - 1 || I > B.phi - 1) continue;
    double X[8][3];
    hex8_load_coords(B, I, J, K, X);
    double k36[36];
#pragma unroll
    for (int t = 0; t < 36; ++t) k36[t] = 0.0;
    const int nq = B.ng * B.ng * B.ng;
    for (int q = 0; q < nq; ++q) {
      double g[8][3];
      const double c = -kcond * hex8_geom(X, q, g);
#pragma unroll
      for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = a; b < 8; ++b) k36[sym36(a, b)] += c * (g[a][0] * g[b][0] + g[a][1] * g[b][1] + g[a][2] * g[b][2]);
    }
#pragma unroll
    for (int t = 0; t < 36; ++t) Ke[e * TT_KSTRIDE + t] = k36[t];
  }
  __syncthreads();
  // ---- phase B: one thread per control point
  const int lk = tid % TT_NK, lj = (tid / TT_NK) % TT_NJ, li = tid / (TT_NK * TT_NJ);
  const int i = T.ti + li, j = T.tj + lj, k = T.tk + lk;
  if (i >= B.phi || j >= B.m1 || k >= B.m2) return;
  double acc[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) acc[t] = 0.0;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ex = e & 1, ey = (e >> 1) & 1, ez = e >> 2;
    const int I = i - 1 + ex, J = j - 1 + ey, K = k - 1 + ez;
    if (I < 0 || I >= B.ne0 || J < 0 || J >= B.ne1 || K < 0 || K >= B.ne2) continue;
    const int a = (1 - ex) + 2 * (1 - ey) + 4 * (1 - ez);
    const double* ke = Ke + ((li + ex) * (TT_EJ * TT_EK) + (lj + ey) * TT_EK + (lk + ez)) * TT_KSTRIDE;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const int di = ex + (b & 1), dj = ey + ((b >> 1) & 1), dk = ez + (b >> 2);  // 0..2 = neighbour offset + 1
      acc[(di * 3 + dj) * 3 + dk] += ke[sym36(a, b)];
    }
  }
  const int li0 = B.lo0[i], lj0 = B.lo1[j], lk0 = B.lo2[k];
  const int cj = B.c1[j], ck = B.c2[k];
  double* row = vals + brick_prefix(B, i, j, k);
#pragma unroll
  for (int di = 0; di < 3; ++di)
#pragma unroll
    for (int dj = 0; dj < 3; ++dj)
#pragma unroll
      for (int dk = 0; dk < 3; ++dk) {
        const int ni = i - 1 + di, nj = j - 1 + dj, nk = k - 1 + dk;
        if (ni < 0 || ni >= B.m0 || nj < 0 || nj >= B.m1 || nk < 0 || nk >= B.m2) continue;
        row[((ni - li0) * cj + (nj - lj0)) * ck + (nk - lk0)] = acc[(di * 3 + dj) * 3 + dk];
      }
}

// Boundary control points of the owned planes, enumerated compactly by one 1-D grid: the whole first lattice plane (if owned), the whole
// last one (if owned), then the rim (2 m2 + 2 (m1 - 2) points) of every other owned plane.
__host__ __device__ __forceinline__ int64_t boundary_count(int plo, int phi, int ne0, int64_t plane_len, int m1, int m2) {
  const int64_t rim = 2 * (int64_t)m2 + 2 * (int64_t)(m1 - 2);
  const int f0 = plo == 0 ? 1 : 0, f1 = phi - 1 == ne0 ? 1 : 0;
  return (f0 + f1) * plane_len + (int64_t)(phi - plo - f0 - f1) * rim;
}
__device__ __forceinline__ bool boundary_point(const BrickView& B, int& i, int& j, int& k) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int f0 = B.plo == 0 ? 1 : 0, f1 = B.phi - 1 == B.ne0 ? 1 : 0;
  if (t < (f0 + f1) * B.plane_len) {
    i = (f0 && t < B.plane_len) ? 0 : B.ne0;
    if (f0 && t >= B.plane_len) t -= B.plane_len;
    j = (int)(t / B.m2);
    k = (int)(t % B.m2);
    return true;
  }
  t -= (f0 + f1) * B.plane_len;
  const int64_t rim = 2 * (int64_t)B.m2 + 2 * (int64_t)(B.m1 - 2);
  const int64_t pl = t / rim;
  if (pl >= B.phi - B.plo - f0 - f1) return false;
  i = B.plo + f0 + (int)pl;
  t -= pl * rim;
  if (t < 2 * (int64_t)B.m2) {
    j = t < B.m2 ? 0 : B.ne1;
    k = (int)(t < B.m2 ? t : t - B.m2);
  } else {
    const int u = (int)(t - 2 * (int64_t)B.m2);
    j = 1 + (u >> 1);
    k = (u & 1) ? B.ne2 : 0;
  }
  return true;
}

// brick_prefix of an order-1 lattice from closed forms (what upload_dim_tables of brick.hip tabulates: a point couples to itself and its neighbours -- 2 at the two
// ends of a direction, 3 in between -- so the entries of the points in front of point g are 3 g - 1 from the second point on): the write-out below asked the
// tables per tile line and plane, five dependent loads from memory in front of the stores of every wave (round 5: the same finding as PX[] in k_hex27_rows_gq)
__device__ __forceinline__ int sw1_cnt(int g, int m) { return (g > 0 ? 1 : 0) + 1 + (g < m - 1 ? 1 : 0); }
__device__ __forceinline__ int64_t sw1_pre(int g) { return g > 0 ? 3 * (int64_t)g - 1 : 0; }
__device__ __forceinline__ int64_t sw1_prefix(const BrickView& B, int i, int j, int k) {
  return (sw1_pre(i) - B.Pplo) * B.S1 * B.S2 + (int64_t)sw1_cnt(i, B.m0) * (sw1_pre(j) * B.S2 + (int64_t)sw1_cnt(j, B.m1) * sw1_pre(k));
}
// Robin faces: h*Bilinear(T, Tenv - T) contributes -h N_a N_b (3D_Script.jl:31).  One thread per BOUNDARY control point, which
// read-modify-writes its OWN row (row owner => race-free), after the matrix kernel.
__global__ __launch_bounds__(MFEM_BLOCK) void k_thermal_matrix_robin(BrickView B, double h, uint32_t robin,
                                                                       double* __restrict__ vals) {
  int i, j, k;
  if (!boundary_point(B, i, j, k)) return;
  const int li = i > 0 ? i - 1 : 0, lj = j > 0 ? j - 1 : 0, lk = k > 0 ? k - 1 : 0;  // (the row box from closed forms: no table loads in front of the face loads)
  const int cj = sw1_cnt(j, B.m1), ck = sw1_cnt(k, B.m2);
  double* row = vals + sw1_prefix(B, i, j, k);
  visit_boundary_faces(B, i, j, k, robin, [&](int nd, int side, int ca, const int (&fn)[4][3], const double (&Xf)[4][3]) {
    double mab[4] = {0.0, 0.0, 0.0, 0.0};
    for (int q = 0; q < B.ng * B.ng; ++q) {
      const double ws = face_geom(Xf, q, side == 0, nullptr);
      double na = 0.0;
#pragma unroll
      for (int c = 0; c < 4; ++c) na = (c == ca) ? c_fN[q][c] : na;
#pragma unroll
      for (int c = 0; c < 4; ++c) mab[c] += (-h * ws) * (na * c_fN[q][c]);  // N_a N_c first: entry (a, c) == entry (c, a) bitwise
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) row[((fn[c][0] - li) * cj + (fn[c][1] - lj)) * ck + (fn[c][2] - lk)] += mab[c];
  });
}

// Robin face terms of the residual at an owned control point: sum_q w^s N_a h (Tenv - T)
__device__ __forceinline__ double thermal_robin_residual(const BrickView& B, int i, int j, int k, double h, double Tenv,
                                                         uint32_t robin, const double* __restrict__ x) {
  double r = 0.0;
  const int idx[3] = {i, j, k};
  const int ne[3] = {B.ne0, B.ne1, B.ne2};
  for (int nd = 0; nd < 3; ++nd) {
    for (int side = 0; side < 2; ++side) {
      const bool on = side == 0 ? (idx[nd] == 0) : (idx[nd] == ne[nd]);
      if (!on || !(robin & face_bit(nd, side))) continue;
      const int t1 = (nd + 1) % 3, t2 = (nd + 2) % 3;
      for (int f = 0; f < 4; ++f) {
        const int f1 = f & 1, f2 = f >> 1;
        int E1 = idx[t1] - 1 + f1, E2 = idx[t2] - 1 + f2;
        if (E1 < 0 || E1 >= ne[t1] || E2 < 0 || E2 >= ne[t2]) continue;
        const int ca = (1 - f1) + 2 * (1 - f2);
        double Xf[4][3], Tf[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          int fn[3];
          fn[nd] = idx[nd];
          fn[t1] = E1 + (c & 1);
          fn[t2] = E2 + (c >> 1);
          const int64_t ci = brick_cindex(B, fn[0], fn[1], fn[2]);
          Xf[c][0] = B.X0[ci];
          Xf[c][1] = B.X1[ci];
          Xf[c][2] = B.X2[ci];
          Tf[c] = x[brick_xindex(B, 0, fn[0], fn[1], fn[2])];
        }
        for (int q = 0; q < B.ng * B.ng; ++q) {
          const double ws = face_geom(Xf, q, side == 0, nullptr);
          double na = 0.0, Tq = 0.0;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            na = (c == ca) ? c_fN[q][c] : na;
            Tq += c_fN[q][c] * Tf[c];
          }
          r += ws * na * h * (Tenv - Tq);
        }
      }
    }
  }
  return r;
}

// the same terms as a kernel of their own over the boundary control points (after the sweep residual kernel)
__global__ __launch_bounds__(MFEM_BLOCK) void k_thermal_residual_robin(BrickView B, double h, double Tenv, uint32_t robin,
                                                                         const double* __restrict__ x, double* __restrict__ res) {
  int i, j, k;
  if (!boundary_point(B, i, j, k)) return;
  res[(int64_t)(i - B.plo) * B.plane_len + (int64_t)j * B.m2 + k] += thermal_robin_residual(B, i, j, k, h, Tenv, robin, x);
}

// =================================================================================================
// Thermal residual (matrix-free, at x_star):
//   R[a] = sum_q w (-k grad N_a . grad T + N_a s) + sum_q w^s N_a h (Tenv - T)
// Same tiling as k_thermal_matrix: the element load vectors fe[8] of the 5 x 5 x 9 elements touching a 4 x 4 x 8
// tile of control points are integrated once each into LDS, then every control point sums its <= 8 entries and adds
// its own Robin face terms.
// =================================================================================================
__global__ __launch_bounds__(TT_THREADS) void k_thermal_residual(BrickView B, double kcond, double h, double Tenv,
                                                                   uint32_t robin, const double* __restrict__ x,
                                                                   const double* __restrict__ src,
                                                                   double* __restrict__ res) {
  __shared__ double Fe[TT_NEL * 9];
  const int tid = threadIdx.x;
  const TileGeom T = tile_origin(B, blockIdx.x);
  for (int e = tid; e < TT_NEL; e += TT_THREADS) {
    const int ek = e % TT_EK, ej = (e / TT_EK) % TT_EJ, ei = e / (TT_EK * TT_EJ);
    const int I = T.ti - 1 + ei, J = T.tj - 1 + ej, K = T.tk - 1 + ek;
    if (I < 0 || I >= B.ne0 || J < 0 || J >= B.ne1 || K < 0 || K >= B.ne2 || I < B.plo - 1 || I > B.phi - 1) continue;
    double X[8][3], Tn[8], Sn[8], fe[8];
    hex8_load_coords(B, I, J, K, X);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const int64_t xi = brick_xindex(B, 0, I + (b & 1), J + ((b >> 1) & 1), K + (b >> 2));
      Tn[b] = x[xi];
      Sn[b] = src ? src[xi] : 0.0;
      fe[b] = 0.0;
    }
    const int nq = B.ng * B.ng * B.ng;
    for (int q = 0; q < nq; ++q) {
      double g[8][3];
      const double wd = hex8_geom(X, q, g);
      double gT0 = 0.0, gT1 = 0.0, gT2 = 0.0, sq = 0.0;
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        gT0 += g[b][0] * Tn[b];
        gT1 += g[b][1] * Tn[b];
        gT2 += g[b][2] * Tn[b];
        sq += c_N[q][b] * Sn[b];
      }
      const double c = -kcond * wd;
#pragma unroll
      for (int a = 0; a < 8; ++a) fe[a] += c * (g[a][0] * gT0 + g[a][1] * gT1 + g[a][2] * gT2) + wd * c_N[q][a] * sq;
    }
#pragma unroll
    for (int a = 0; a < 8; ++a) Fe[e * 9 + a] = fe[a];
  }
  __syncthreads();
  const int lk = tid % TT_NK, lj = (tid / TT_NK) % TT_NJ, li = tid / (TT_NK * TT_NJ);
  const int i = T.ti + li, j = T.tj + lj, k = T.tk + lk;
  if (i >= B.phi || j >= B.m1 || k >= B.m2) return;
  const int64_t node = (int64_t)(i - B.plo) * B.plane_len + (int64_t)j * B.m2 + k;
  double r = 0.0;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ex = e & 1, ey = (e >> 1) & 1, ez = e >> 2;
    const int I = i - 1 + ex, J = j - 1 + ey, K = k - 1 + ez;
    if (I < 0 || I >= B.ne0 || J < 0 || J >= B.ne1 || K < 0 || K >= B.ne2) continue;
    const int a = (1 - ex) + 2 * (1 - ey) + 4 * (1 - ez);
    r += Fe[((li + ex) * (TT_EJ * TT_EK) + (lj + ey) * TT_EK + (lk + ez)) * 9 + a];
  }
  if (h != 0.0 && robin != 0u) r += thermal_robin_residual(B, i, j, k, h, Tenv, robin, x);
  res[node] = r;
}


// =================================================================================================
// Sweep form of the two thermal kernels (default for 2- and 3-point Gauss rules).
// A workgroup of 256 threads owns a 15 x 15 tile of control points in (j, k) and sweeps it through a segment of lattice
// planes along i.  Per element plane I (the elements between node planes I and I + 1):
//   phase A  thread <-> element of the 16 x 16 element tile: the element's upper four nodes are loaded (the lower four are
//            last step's upper ones, kept in registers), the element is integrated with the sum-factorised routines of
//            hex8_sumfac.h and its fe[8] / 36 unique Ke entries go to LDS;
//   phase B  thread <-> control point: the point of node plane I adds the four elements above it to what it carried over
//            from the four elements below (element plane I - 1) and writes its residual entry / its <= 27 CSR values once;
//            the same four elements' contributions to the point above it (node plane I + 1) become the next carry.
// Element evaluations per control point: 256 / 225 x (L + 1) / L = 1.17 (L = 32 planes per segment) against 1.76 for the
// 4 x 4 x 8 tiles, each at ~40 % of the table form's arithmetic.  Row-owner as before: no atomics, no colours, every value
// written once; the sums over the shared elements of a mirrored pair of entries run in the same order, so the matrix stays
// bitwise symmetric (the symmetric-sweep SpMV depends on that).
// =================================================================================================
#define SW_E 16
#define SW_N (SW_E - 1)
#define SW_THREADS (SW_E * SW_E)
#define SW_KSTRIDE 37

struct SweepTile {
  int tj0, tk0, i0, i1;
};
__device__ __forceinline__ SweepTile sweep_tile(const BrickView& B, int L) {
  const int ntk = (B.m2 + SW_N - 1) / SW_N, ntj = (B.m1 + SW_N - 1) / SW_N;
  SweepTile t;
  t.tk0 = (int)(blockIdx.x % ntk) * SW_N;
  t.tj0 = (int)((blockIdx.x / ntk) % ntj) * SW_N;
  t.i0 = B.plo + (int)(blockIdx.x / (ntk * ntj)) * L;
  t.i1 = min(t.i0 + L, B.phi);
  return t;
}
// the four nodes (J + by, K + bz) of lattice plane ip -> slot [bx] of the element's nodal arrays
__device__ __forceinline__ void sweep_load_coords(const BrickView& B, int ip, int J, int K, int bx, double (&X)[3][2][4]) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int64_t ci = brick_cindex(B, ip, J + (c & 1), K + (c >> 1));
    X[0][bx][c] = B.X0[ci];
    X[1][bx][c] = B.X1[ci];
    X[2][bx][c] = B.X2[ci];
  }
}

template <int NG>
__global__ __launch_bounds__(SW_THREADS) __attribute__((amdgpu_waves_per_eu(NG == 2 ? 2 : 1))) void k_thermal_residual_sweep(BrickView B, int L, double kcond, const double* __restrict__ x,
                                                                         const double* __restrict__ src,
                                                                         double* __restrict__ res, int affine_fast) {
  __shared__ double Fe[SW_THREADS * 9];
  const int tid = threadIdx.x, ek = tid % SW_E, ej = tid / SW_E;
  const SweepTile T = sweep_tile(B, L);
  if (T.i0 >= T.i1) return;
  const int J = T.tj0 - 1 + ej, K = T.tk0 - 1 + ek;  // phase A: this thread's element column
  const bool el_ok = J >= 0 && J < B.ne1 && K >= 0 && K < B.ne2;
  const int j = T.tj0 + ej, k = T.tk0 + ek;          // phase B: this thread's control point column
  const bool nd_ok = ej < SW_N && ek < SW_N && j < B.m1 && k < B.m2;
  const bool has_src = src != nullptr;
  // nodal data of lattice plane ip at the element's four (j, k) corners
  double Xn[3][4], Tq[4], Sq[4];  // the plane requested last (consumed at the top of the next step)
#pragma unroll
  for (int c = 0; c < 4; ++c) Sq[c] = 0.0;
  const int Jc = min(max(J, 0), B.ne1 - 1), Kc = min(max(K, 0), B.ne2 - 1);  // (threads without an element column load what a neighbour loads)
  // (every thread requests on every step, from clamped positions: behind a condition the requested registers meet their old values in a phi, and the copies that resolves into wait for the loads AT ONCE -- the prefetch then costs a memory round trip per plane instead of hiding one: round 5, found in the ISA)
  auto request = [&](int ip) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int64_t ci = brick_cindex(B, ip, Jc + (c & 1), Kc + (c >> 1));
      Xn[0][c] = B.X0[ci];
      Xn[1][c] = B.X1[ci];
      Xn[2][c] = B.X2[ci];
      const int64_t xi = brick_xindex(B, 0, ip, Jc + (c & 1), Kc + (c >> 1));
      Tq[c] = x[xi];
      if (has_src) Sq[c] = src[xi];
    }
  };
  double X[3][2][4], Tn[2][4], Sn[2][4];
  const int Ifirst = max(T.i0 - 1, 0), Ilast = min(T.i1 - 1, B.ne0 - 1);  // element planes of this segment
  {
    request(Ifirst);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      X[0][1][c] = Xn[0][c]; X[1][1][c] = Xn[1][c]; X[2][1][c] = Xn[2][c];
      Tn[1][c] = Tq[c]; Sn[1][c] = Sq[c];
    }
    request(Ifirst + 1);
  }
  double carry = 0.0;
  for (int I = T.i0 - 1; I < T.i1; ++I) {
    const bool plane = I >= 0 && I < B.ne0;  // workgroup-uniform
    if (plane && el_ok) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {  // last step's upper nodes are this step's lower ones; the requested plane arrives
        X[0][0][c] = X[0][1][c]; X[1][0][c] = X[1][1][c]; X[2][0][c] = X[2][1][c];
        Tn[0][c] = Tn[1][c]; Sn[0][c] = Sn[1][c];
        X[0][1][c] = Xn[0][c]; X[1][1][c] = Xn[1][c]; X[2][1][c] = Xn[2][c];
        Tn[1][c] = Tq[c]; Sn[1][c] = Sq[c];
      }
      double fe[2][4];
      if (affine_fast && sf_is_affine(X)) sf_thermal_fe<NG, true>(X, Tn, Sn, has_src, kcond, fe);  // (one adjugate for a parallelepiped: see k_thermal_matrix_sweep)
      else sf_thermal_fe<NG, false>(X, Tn, Sn, has_src, kcond, fe);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        Fe[tid * 9 + 2 * c] = fe[0][c];
        Fe[tid * 9 + 2 * c + 1] = fe[1][c];
      }
    }
    request(min(max(I + 2, Ifirst), Ilast + 1));  // in flight during the rest of this plane's step: gather, write-out, barriers (mfem_lds_barrier does not wait for it)
    mfem_lds_barrier();
    if (nd_ok) {
      double lo = 0.0, up = 0.0;
      if (plane) {
#pragma unroll
        for (int ez = 0; ez < 2; ++ez)
#pragma unroll
          for (int ey = 0; ey < 2; ++ey) {
            const int Jn = j - 1 + ey, Kn = k - 1 + ez;
            if (Jn < 0 || Jn >= B.ne1 || Kn < 0 || Kn >= B.ne2) continue;
            const double* f = Fe + ((ej + ey) * SW_E + ek + ez) * 9 + 2 * ((1 - ey) + 2 * (1 - ez));
            lo += f[0];
            up += f[1];
          }
      }
      if (I >= T.i0) res[(int64_t)(I - B.plo) * B.plane_len + (int64_t)j * B.m2 + k] = carry + lo;
      carry = up;
    }
    mfem_lds_barrier();
  }
}

// The matrix sweep on its own, smaller tile (round 6): 8 x 16 elements, 128 threads, 38 KB of LDS -- FOUR workgroups per CU instead of two (the waves per
// CU stay 8: 250 VGPRs), so that the load / integrate / gather / write-out phases of more independent workgroups overlap (ablation: the phases of two did not).
#define MS_EJ 8
#define MS_EK 16
#define MS_NJ (MS_EJ - 1)
#define MS_NK (MS_EK - 1)
#define MS_THREADS (MS_EJ * MS_EK)
__device__ __forceinline__ SweepTile sweep_tile_m(const BrickView& B, int L) {
  const int ntk = (B.m2 + MS_NK - 1) / MS_NK, ntj = (B.m1 + MS_NJ - 1) / MS_NJ;
  SweepTile t;
  t.tk0 = (int)(blockIdx.x % ntk) * MS_NK;
  t.tj0 = (int)((blockIdx.x / ntk) % ntj) * MS_NJ;
  t.i0 = B.plo + (int)(blockIdx.x / (ntk * ntj)) * L;
  t.i1 = min(t.i0 + L, B.phi);
  return t;
}
template <int NG>
__global__ __launch_bounds__(MS_THREADS) __attribute__((amdgpu_waves_per_eu(NG == 2 ? 2 : 1))) void k_thermal_matrix_sweep(BrickView B, int L, double kcond, double* __restrict__ vals, int stage_rows) {
  __shared__ double Ke[MS_THREADS * SW_KSTRIDE];
  const int tid = threadIdx.x, ek = tid % MS_EK, ej = tid / MS_EK;
  const SweepTile T = sweep_tile_m(B, L);
  if (T.i0 >= T.i1) return;
  const int J = T.tj0 - 1 + ej, K = T.tk0 - 1 + ek;
  const bool el_ok = J >= 0 && J < B.ne1 && K >= 0 && K < B.ne2;
  const int j = T.tj0 + ej, k = T.tk0 + ek;
  const bool nd_ok = ej < MS_NJ && ek < MS_NK && j < B.m1 && k < B.m2;
  int lj0 = 0, lk0 = 0, cj = 1, ck = 1;
  if (nd_ok) {
    lj0 = B.lo1[j];
    lk0 = B.lo2[k];
    cj = B.c1[j];
    ck = B.c2[k];
  }
  const bool jk_inner = nd_ok && j > 0 && j < B.ne1 && k > 0 && k < B.ne2;  // all nine in-plane neighbours exist
  const bool staged = (stage_rows & 1) != 0;  // (kernel argument, uniform: the per-thread write-out is kept behind bit 1 of mfem_debug_set_hex8_thermal)
  double Xn[3][4];
  const int Jc = min(max(J, 0), B.ne1 - 1), Kc = min(max(K, 0), B.ne2 - 1);  // (threads without an element column load what a neighbour loads)
  // (every thread requests on every step, from clamped positions: behind a condition the requested registers meet their old values in a phi, and the copies that resolves into wait for the loads AT ONCE -- the prefetch then costs a memory round trip per plane instead of hiding one: round 5, found in the ISA)
  auto request = [&](int ip) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int64_t ci = brick_cindex(B, ip, Jc + (c & 1), Kc + (c >> 1));
      Xn[0][c] = B.X0[ci];
      Xn[1][c] = B.X1[ci];
      Xn[2][c] = B.X2[ci];
    }
  };
  double X[3][2][4];
  const int Ifirst = max(T.i0 - 1, 0), Ilast = min(T.i1 - 1, B.ne0 - 1);
  {
    request(Ifirst);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      X[0][1][c] = Xn[0][c]; X[1][1][c] = Xn[1][c]; X[2][1][c] = Xn[2][c];
    }
    request(Ifirst + 1);
  }
  double carry[18];
#pragma unroll
  for (int t = 0; t < 18; ++t) carry[t] = 0.0;
  for (int I = T.i0 - 1; I < T.i1; ++I) {
    const bool plane = I >= 0 && I < B.ne0;
    if (plane && el_ok) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        X[0][0][c] = X[0][1][c]; X[1][0][c] = X[1][1][c]; X[2][0][c] = X[2][1][c];
        X[0][1][c] = Xn[0][c]; X[1][1][c] = Xn[1][c]; X[2][1][c] = Xn[2][c];
      }
      double k36[36];
      // affine elements (bit 1 of the kernel's flag word): one adjugate instead of NG^3 -- the FP64 work is what this kernel's time follows (a table of
      // reference integrals, ke = sum_t g0[t] K6[t], was measured too: 216 constant loads per element made the kernel 1.7x SLOWER)
      if (stage_rows & 4) {
#pragma unroll
        for (int t = 0; t < 36; ++t) k36[t] = 0.0;
      } else if ((stage_rows & 2) && sf_is_affine(X)) sf_thermal_ke<NG, true>(X, kcond, k36);
      else sf_thermal_ke<NG, false>(X, kcond, k36);
#pragma unroll
      for (int t = 0; t < 36; ++t) Ke[tid * SW_KSTRIDE + t] = k36[t];
    }
    request(min(max(I + 2, Ifirst), Ilast + 1));  // in flight during the rest of this plane's step: gather, write-out, barriers (mfem_lds_barrier does not wait for it)
    mfem_lds_barrier();
    // now[bx * 9 + dj * 3 + dk]: entries of this plane's point towards node plane I + bx; up[...]: entries of the point above it
    // (node plane I + 1) towards node plane I + bx
    double now[18], up[18];
#pragma unroll
    for (int t = 0; t < 18; ++t) now[t] = up[t] = 0.0;
    if (nd_ok && plane && !(stage_rows & 8)) {
#pragma unroll
      for (int ez = 0; ez < 2; ++ez)
#pragma unroll
        for (int ey = 0; ey < 2; ++ey) {
          const int Jn = j - 1 + ey, Kn = k - 1 + ez;
          if (Jn < 0 || Jn >= B.ne1 || Kn < 0 || Kn >= B.ne2) continue;
          const double* ke = Ke + ((ej + ey) * MS_EK + ek + ez) * SW_KSTRIDE;
          const int a0 = 2 * (1 - ey) + 4 * (1 - ez);
#pragma unroll
          for (int b = 0; b < 8; ++b) {
            const int slot = (b & 1) * 9 + (ey + ((b >> 1) & 1)) * 3 + ez + (b >> 2);
            now[slot] += ke[sym36(a0, b)];
            up[slot] += ke[sym36(a0 + 1, b)];
          }
        }
    }
    // Write-out.  A point away from the lattice boundary has its full 27-entry row, one contiguous run in (di, dj, dk) order -- and the rows of the
    // points of one tile line are contiguous in CSR too.  Writing row[t] per thread puts 64 lanes 216 bytes apart: 27 x 64 eight-byte requests per
    // wave and plane, and at 4.6e8 requests per assembly (256^3) the L2's request rate, not its bytes, set the kernel's time (round 4).  The rows now go
    // through LDS (the element matrices of this plane are dead once every point has gathered them) and leave as contiguous streams: a wave per tile line.
    // (Every barrier below is reached by ALL threads of the workgroup: `staged` is a kernel argument, nothing here sits inside a per-thread branch.)
    const bool full_row = nd_ok && jk_inner && I > 0 && I < B.ne0 && I >= T.i0;
    if (staged) {
      mfem_lds_barrier();  // every point has gathered from Ke
      if (full_row) {
        double* st = Ke + tid * 27;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          st[t] = carry[t];
          st[9 + t] = carry[9 + t] + now[t];
          st[18 + t] = now[9 + t];
        }
      }
    }
    if (nd_ok) {
      if (I >= T.i0 && !(staged && full_row)) {
        double* row = vals + brick_prefix(B, I, j, k);
        if (jk_inner && I > 0 && I < B.ne0) {  // the full 27-entry row is one contiguous run in (di, dj, dk) order
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            row[t] = carry[t];
            row[9 + t] = carry[9 + t] + now[t];
            row[18 + t] = now[9 + t];
          }
        } else {
          const int li0 = B.lo0[I];
#pragma unroll
          for (int di = 0; di < 3; ++di)
#pragma unroll
            for (int dj = 0; dj < 3; ++dj)
#pragma unroll
              for (int dk = 0; dk < 3; ++dk) {
                const int ni = I - 1 + di, nj = j - 1 + dj, nk = k - 1 + dk;
                if (ni < 0 || ni >= B.m0 || nj < 0 || nj >= B.m1 || nk < 0 || nk >= B.m2) continue;
                const int t = dj * 3 + dk;
                const double v = di == 0 ? carry[t] : di == 1 ? carry[9 + t] + now[t] : now[9 + t];
                row[((ni - li0) * cj + (nj - lj0)) * ck + (nk - lk0)] = v;
              }
        }
      }
#pragma unroll
      for (int t = 0; t < 18; ++t) carry[t] = up[t];
    }
    if (staged) {
      mfem_lds_barrier();  // the staged rows are complete
      if (I >= T.i0 && I > 0 && I < B.ne0) {
        const int k0 = max(T.tk0, 1), k1 = min(min(T.tk0 + MS_NK, B.ne2), B.m2);  // points of a tile line with a full row: k in [k0, k1)
        const int cnt = (k1 - k0) * 27;
        for (int line = tid >> 6; line < MS_NJ; line += MS_THREADS / 64) {
          const int jl = T.tj0 + line;
          if (jl < 1 || jl >= B.ne1 || jl >= B.m1 || cnt <= 0 || (stage_rows & 16)) continue;
          double* dst = vals + sw1_prefix(B, I, jl, k0);
          const double* src = Ke + (line * MS_EK + (k0 - T.tk0)) * 27;
          // 16 bytes per lane from the destination's first 16-byte boundary on (round 6: a wave's store instruction covers 1 KB of the run instead of 512 bytes)
          const int head = (int)(((uintptr_t)dst >> 3) & 1), np = (cnt - head) >> 1, ln = tid & 63;
          typedef double sw_d2 __attribute__((ext_vector_type(2)));
          for (int m = ln; m < np; m += 64) {
            const int idx = head + 2 * m;
            __builtin_nontemporal_store(sw_d2{src[idx], src[idx + 1]}, reinterpret_cast<sw_d2*>(dst + idx));
          }
          if (ln == 0 && head) __builtin_nontemporal_store(src[0], dst);
          if (ln == 1 && ((cnt - head) & 1)) __builtin_nontemporal_store(src[cnt - 1], dst + cnt - 1);
        }
      }
    }
    mfem_lds_barrier();
  }
}

static dim3 boundary_grid(const mfem_brick_s* m) {  // boundary_point(): one thread per boundary control point of the owned planes
  const int64_t cnt = boundary_count(m->plo, m->phi, m->ne[0], m->plane_len, m->m[1], m->m[2]);
  return dim3((unsigned)((cnt + MFEM_BLOCK - 1) / MFEM_BLOCK > 0 ? (cnt + MFEM_BLOCK - 1) / MFEM_BLOCK : 1));
}
static std::atomic<int> g_thermal_variant{0};  // 1: the 4 x 4 x 8 tile kernels for every Gauss order (kept: 1- and 4-point rules use them)
static std::atomic<int> g_thermal_stage_rows{3};  // the matrix sweep kernel's flag word: bit 0 rows staged through LDS (off: bit 1 of mfem_debug_set_hex8_thermal), bit 1 the affine-element shortcut (off: bit 2)
extern "C" int mfem_debug_set_hex8_thermal(int variant) try {
  ++mfem_debug_epoch;
  g_thermal_variant = (variant & 1) ? 1 : 0;
  g_thermal_stage_rows = ((variant & 2) ? 0 : 1) | ((variant & 4) ? 0 : 2) | ((variant >> 3) & 7) << 2;  // bits 3-5: TIMING-ONLY ablations of the matrix sweep (no integration / no gather from LDS / no write-out)
  return MFEM_OK;
} MFEM_API_CATCH("mfem_debug_set_hex8_thermal")
static int sweep_planes_m(const mfem_brick_s* m, int64_t* grid) {  // (the matrix sweep's tile)
  const int64_t ntj = (m->m[1] + MS_NJ - 1) / MS_NJ, ntk = (m->m[2] + MS_NK - 1) / MS_NK;
  const int planes = m->phi - m->plo;
  int L = 32;
  while (L > 4 && ntj * ntk * ((planes + L - 1) / L) < 4096) L /= 2;
  *grid = ntj * ntk * ((planes + L - 1) / L);
  return L;
}
// planes per sweep segment: 32, shorter when the (j, k) tiles alone cannot fill the chip
static int sweep_planes(const mfem_brick_s* m, int64_t* grid) {
  const int64_t ntj = (m->m[1] + SW_N - 1) / SW_N, ntk = (m->m[2] + SW_N - 1) / SW_N;
  const int planes = m->phi - m->plo;
  int L = 32;
  while (L > 4 && ntj * ntk * ((planes + L - 1) / L) < 2048) L /= 2;
  *grid = ntj * ntk * ((planes + L - 1) / L);
  return L;
}


// =================================================================================================
// Linear elasticity (examples/linear_elasticity/cantilever/3D_Script.jl:52-63), 3 fields, field-major:
//   K[(i,a),(k,b)] = -sum_q w [ lam d_iN_a d_kN_b + mu d_kN_a d_iN_b + mu delta_ik gradN_a.gradN_b ]
//                    - tau sum_q w^s N_a N_b delta_ik   on penalty faces
// i.e. the 21 _Kval_Basic launches of the reference in one pass.  Row-owner: a thread
// owns the 3 rows of its control point and accumulates them in place (exclusive => no atomics); `vals`
// must be zero on entry (the launcher memsets it).
// =================================================================================================
__global__ __launch_bounds__(MFEM_BLOCK) void k_elasticity_matrix(BrickView B, double lam, double mu, double tau,
                                                                    uint32_t penalty, int64_t T,
                                                                    double* __restrict__ vals) {
  const int64_t node = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (node >= B.n_owned) return;
  int i, j, k;
  node_ijk(B, node, i, j, k);
  const int li = B.lo0[i], lj = B.lo1[j], lk = B.lo2[k];
  const int cj = B.c1[j], ck = B.c2[k];
  const int64_t cn = (int64_t)B.c0[i] * cj * ck;
  const int64_t pre = brick_prefix(B, i, j, k);
  // row (f,node) starts at f*3*T + 3*pre; block g at + g*cn
  double* row[3];
#pragma unroll
  for (int f = 0; f < 3; ++f) row[f] = vals + (int64_t)f * 3 * T + 3 * pre;
  for (int e = 0; e < 8; ++e) {
    const int ex = e & 1, ey = (e >> 1) & 1, ez = e >> 2;
    const int I = i - 1 + ex, J = j - 1 + ey, K = k - 1 + ez;
    if (I < 0 || I >= B.ne0 || J < 0 || J >= B.ne1 || K < 0 || K >= B.ne2) continue;
    const int a = (1 - ex) + 2 * (1 - ey) + 4 * (1 - ez);
    double X[8][3];
    hex8_load_coords(B, I, J, K, X);
    // G[b][s][t] = sum_q w d_sN_a d_tN_b
    double G[8][3][3];
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
      for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int t = 0; t < 3; ++t) G[b][s][t] = 0.0;
    const int nq = B.ng * B.ng * B.ng;
    for (int q = 0; q < nq; ++q) {
      double g[8][3];
      const double wd = hex8_geom(X, q, g);
      double ga[3] = {0.0, 0.0, 0.0};
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const bool me = (b == a);
        ga[0] = me ? g[b][0] : ga[0];
        ga[1] = me ? g[b][1] : ga[1];
        ga[2] = me ? g[b][2] : ga[2];
      }
#pragma unroll
      for (int b = 0; b < 8; ++b)
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
          for (int t = 0; t < 3; ++t) G[b][s][t] += wd * ga[s] * g[b][t];
    }
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const int ni = I + (b & 1), nj = J + ((b >> 1) & 1), nk = K + (b >> 2);
      const int slot = ((ni - li) * cj + (nj - lj)) * ck + (nk - lk);
      const double tr = G[b][0][0] + G[b][1][1] + G[b][2][2];
#pragma unroll
      for (int fi = 0; fi < 3; ++fi)
#pragma unroll
        for (int fk = 0; fk < 3; ++fk) {
          double v = lam * G[b][fi][fk] + mu * G[b][fk][fi];
          if (fi == fk) v += mu * tr;
          row[fi][fk * cn + slot] -= v;
        }
    }
  }
  if (tau != 0.0 && penalty != 0u) {
    visit_boundary_faces(B, i, j, k, penalty, [&](int nd, int side, int ca, const int (&fn)[4][3], const double (&Xf)[4][3]) {
      double mab[4] = {0.0, 0.0, 0.0, 0.0};
      for (int q = 0; q < B.ng * B.ng; ++q) {
        const double ws = face_geom(Xf, q, side == 0, nullptr);
        double na = 0.0;
#pragma unroll
        for (int c = 0; c < 4; ++c) na = (c == ca) ? c_fN[q][c] : na;
#pragma unroll
        for (int c = 0; c < 4; ++c) mab[c] += ws * na * c_fN[q][c];
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int slot = ((fn[c][0] - li) * cj + (fn[c][1] - lj)) * ck + (fn[c][2] - lk);
#pragma unroll
        for (int fi = 0; fi < 3; ++fi) row[fi][fi * cn + slot] -= tau * mab[c];
      }
    });
  }
}

// Same operator, one thread per (control point, adjacent element): a workgroup owns 32 consecutive control points, its
// 256 threads integrate the 8 adjacent elements of each of them concurrently (the row-owner kernel above walks them one after
// the other and read-modify-writes global memory 576 times per thread).  The three rows of a control point are accumulated
// in LDS in eight conflict-free steps (see below; fixed summation order: a second assembly gives the same bits, the row-owner kernel's
// order is the reverse, equal to round-off) and leave as contiguous runs, every CSR value written exactly once -- no memset, no atomics,
// no colours.  128^3: 1.80 ms (profiles/r03_elasticity_matrix_steps.txt).
#define EL2_NODES 32
#define EL2_ROW 243  // 3 fields x 81 slots; an ODD number of doubles: the lanes of a step (one control point each) then spread over all banks (244: 1.94 -> 2.04 ms)
#define EL2_CH (EL2_NODES * 81)  // doubles of one row field's chunk of the LDS copy (32 rows of up to 81 values; a node's row starts at 3 x its prefix: an odd stride too)
// G[b][s][t] = sum_q w det d_sN_a d_tN_b for the row node a = AH + 1 - ex of the thread's element
template <int AH>
__device__ __forceinline__ void el2_integrate(const double (&C)[7][3], int nq, int ex, double (&G)[8][3][3]) {
  double A[8][3][3];  // the loop's own accumulators, handed over once after the loop (accumulating into G itself makes the register allocator
                      // rotate the 72 sums through ~140 copies per pass)
#pragma unroll
  for (int b = 0; b < 8; ++b)
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int t = 0; t < 3; ++t) A[b][s][t] = 0.0;
  for (int q = 0; q < nq; ++q) {
    double g[8][3];
    const double wd = hex8_geom_trilinear(C, q, g);
    double ga[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) ga[s] = wd * (ex ? g[AH][s] : g[AH + 1][s]);
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
      for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int t = 0; t < 3; ++t) A[b][s][t] += ga[s] * g[b][t];
  }
#pragma unroll
  for (int b = 0; b < 8; ++b)
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int t = 0; t < 3; ++t) G[b][s][t] = A[b][s][t];
}
// The same sums on an AFFINE element (a parallelepiped: the bilinear and trilinear coefficients of its map vanish -- every element of make_Brick until a caller
// moves coordinates): the Jacobian is one matrix, so G[b][s][t] = det sum_mn Jinv[m][s] Jinv[n][t] M[a][b][m][n] with the reference integrals M of the
// quadrature in constant memory: 54 multiply-adds per neighbour b instead of 8 Gauss points x (148 of geometry + 75): 0.55 k against 1.8 k FP64 instructions per thread.
template <int AH>
__device__ __forceinline__ void el2_affine(const double (&C)[7][3], int ex, double (&G)[8][3][3]) {
  double J[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    J[i][0] = C[0][i];
    J[i][1] = C[1][i];
    J[i][2] = C[3][i];
  }
  const double det = J[0][0] * J[1][1] * J[2][2] - J[0][0] * J[1][2] * J[2][1] - J[0][1] * J[1][0] * J[2][2] +
                     J[0][1] * J[1][2] * J[2][0] + J[0][2] * J[1][0] * J[2][1] - J[0][2] * J[1][1] * J[2][0];
  const double id = 1.0 / det;
  double I[3][3];
  I[0][0] = (J[1][1] * J[2][2] - J[1][2] * J[2][1]) * id;
  I[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * id;
  I[0][2] = (J[0][1] * J[1][2] - J[1][1] * J[0][2]) * id;
  I[1][0] = (J[1][2] * J[2][0] - J[2][2] * J[1][0]) * id;
  I[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id;
  I[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * id;
  I[2][0] = (J[1][0] * J[2][1] - J[1][1] * J[2][0]) * id;
  I[2][1] = (J[0][1] * J[2][0] - J[2][1] * J[0][0]) * id;
  I[2][2] = (J[0][0] * J[1][1] - J[1][0] * J[0][1]) * id;
  double Id[3][3];  // det * Jinv: the left factor
#pragma unroll
  for (int m = 0; m < 3; ++m)
#pragma unroll
    for (int s = 0; s < 3; ++s) Id[m][s] = det * I[m][s];
  const double (*Ma)[3][3] = c_M[ex ? AH : AH + 1];
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    double T[3][3];  // T[s][n] = sum_m Id[m][s] M[a][b][m][n]
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int n = 0; n < 3; ++n) T[s][n] = Id[0][s] * Ma[b][0][n] + Id[1][s] * Ma[b][1][n] + Id[2][s] * Ma[b][2][n];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int t = 0; t < 3; ++t) G[b][s][t] = T[s][0] * I[0][t] + T[s][1] * I[1][t] + T[s][2] * I[2][t];
  }
}
// affine to 16 ulp of the coordinates' magnitude (what the Jacobian's own cancellation error is made of): the four mixed coefficients against the element's scale
__device__ __forceinline__ bool hex8_is_affine(const double (&X)[8][3], const double (&C)[7][3]) {
  bool ok = true;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double tol = 3.6e-15 * (fabs(X[0][i]) + fabs(C[0][i]) + fabs(C[1][i]) + fabs(C[3][i]));
    ok = ok && fabs(C[2][i]) <= tol && fabs(C[4][i]) <= tol && fabs(C[5][i]) <= tol && fabs(C[6][i]) <= tol;
  }
  return ok;
}
__global__ __launch_bounds__(MFEM_BLOCK) __attribute__((amdgpu_waves_per_eu(2))) void k_elasticity_matrix_lds(BrickView B, double lam, double mu, double tau,
                                                                        uint32_t penalty, int64_t T, double* __restrict__ vals, int abl) {
  __shared__ double rows[3 * EL2_CH];
  __shared__ int64_t s_pre[EL2_NODES];
  __shared__ int32_t s_cn[EL2_NODES];
  // lane <-> control point, half-wave <-> adjacent element (two elements that differ in dimension 0 per wave): neighbouring lanes load
  // neighbouring elements' coordinates, and the row node of a thread inside its element is known per wave up to one bit
  const int tid = threadIdx.x, nl = tid & (EL2_NODES - 1), e = tid / EL2_NODES;
  const int64_t node = (int64_t)blockIdx.x * EL2_NODES + nl;
  const bool live = node < B.n_owned;
  for (int t = tid; t < 3 * EL2_CH; t += MFEM_BLOCK) rows[t] = 0.0;
  int i = 0, j = 0, k = 0, li = 0, lj = 0, lk = 0, cj = 1, ck = 1, cn = 0;
  if (live) {
    node_ijk(B, node, i, j, k);
    li = i > 0 ? i - 1 : 0; lj = j > 0 ? j - 1 : 0; lk = k > 0 ? k - 1 : 0;  // (closed forms of the order-1 row boxes instead of table loads: sw1_prefix above)
    cj = sw1_cnt(j, B.m1); ck = sw1_cnt(k, B.m2);
    cn = sw1_cnt(i, B.m0) * cj * ck;
    if (e == 0) {
      s_pre[nl] = sw1_prefix(B, i, j, k);
      s_cn[nl] = cn;
    }
  } else if (e == 0) {
    s_pre[nl] = 0;
    s_cn[nl] = 0;
  }
  const int ex = e & 1, ey = (e >> 1) & 1, ez = e >> 2;
  const int I = i - 1 + ex, J = j - 1 + ey, K = k - 1 + ez;
  const bool valid = live && I >= 0 && I < B.ne0 && J >= 0 && J < B.ne1 && K >= 0 && K < B.ne2;
  double G[8][3][3];
#pragma unroll
  for (int b = 0; b < 8; ++b)
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int t = 0; t < 3; ++t) G[b][s][t] = 0.0;
  if (valid) {
    double C[7][3];
    bool affine;
    {
      double X[8][3];
      hex8_load_coords_pairs(B, I, J, K, X);
      hex8_trilinear(X, C);
      affine = !(abl & 8) && hex8_is_affine(X, C);  // (bit 5 of mfem_debug_set_elasticity: every element takes the general path)
    }
    // the row node a = (1 - ex) + 2 (1 - ey) + 4 (1 - ez) of this thread in its element: (ey, ez) is the wave's, ex the half-wave's -- the
    // loop is instantiated per wave with the node pair as a constant, so its gradient is one select instead of a search through all eight
    const int nq = (abl & 4) ? 0 : B.ng * B.ng * B.ng;
    if (affine && nq > 0) {
      switch (e >> 1) {
        case 0: el2_affine<6>(C, ex, G); break;
        case 1: el2_affine<4>(C, ex, G); break;
        case 2: el2_affine<2>(C, ex, G); break;
        default: el2_affine<0>(C, ex, G); break;
      }
    } else {
    switch (e >> 1) {
      case 0: el2_integrate<6>(C, nq, ex, G); break;
      case 1: el2_integrate<4>(C, nq, ex, G); break;
      case 2: el2_integrate<2>(C, nq, ex, G); break;
      default: el2_integrate<0>(C, nq, ex, G); break;
    }
    }
  }
  __syncthreads();
  // (round 6) the LDS copy is laid out IN MEMORY ORDER -- per row field one chunk holding the nodes' rows back to back, 3 cn values each (a node's row
  // sits at 3 (pre(node) - pre(first node)), as in the value array) -- so that the write-out below is a plain linear copy with unit-stride lanes.  The old
  // layout (a fixed 243-double block per node, a half-wave per row at the write-out: 256-byte pieces, 8 bytes per lane) spent 0.54 - 0.77 of the
  // kernel's 1.50 ms issuing stores.
  const int relp = 3 * (int)(s_pre[nl] - s_pre[0]);
  double* R = rows + relp;  // + fi * EL2_CH: [block fk * cn + slot]
  // Accumulation without conflicts: in step b EVERY thread adds its element's block towards the element's node b.  The eight threads of a
  // control point sit in eight different elements (offsets e), so in one step they touch the eight different neighbours e + b -- all 256
  // threads work in every step, and a barrier orders the steps (an entry's contributions arrive by decreasing e: a fixed order, the same
  // for the two mirrored entries of a pair of control points).
  if (!(abl & 1)) {
    const int slot0 = ((I - li) * cj + (J - lj)) * ck + (K - lk);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      if (valid) {
        const int slot = slot0 + ((b & 1) * cj + ((b >> 1) & 1)) * ck + (b >> 2);
        double cur[3][3];
#pragma unroll
        for (int fi = 0; fi < 3; ++fi)
#pragma unroll
          for (int fk = 0; fk < 3; ++fk) cur[fi][fk] = R[fi * EL2_CH + fk * cn + slot];
        const double tr = G[b][0][0] + G[b][1][1] + G[b][2][2];
#pragma unroll
        for (int fi = 0; fi < 3; ++fi)
#pragma unroll
          for (int fk = 0; fk < 3; ++fk) {
            double v = lam * G[b][fi][fk] + mu * G[b][fk][fi];
            if (fi == fk) v += mu * tr;
            R[fi * EL2_CH + fk * cn + slot] = cur[fi][fk] - v;
          }
      }
      // steps b and b + 1 (b even) meet only in neighbours e + b == e' + b + 1, i.e. between the two threads of a control point whose
      // elements differ in dimension 0 alone: the two halves of ONE wave, whose LDS instructions execute in program order -- no barrier
      if (b & 1) __syncthreads();
      else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"), __builtin_amdgcn_wave_barrier();
    }
  }
  if (live && e == 0 && tau != 0.0 && penalty != 0u) {
    visit_boundary_faces(B, i, j, k, penalty, [&](int nd, int side, int ca, const int (&fn)[4][3], const double (&Xf)[4][3]) {
      double mab[4] = {0.0, 0.0, 0.0, 0.0};
      for (int q = 0; q < B.ng * B.ng; ++q) {
        const double ws = face_geom(Xf, q, side == 0, nullptr);
        double na = 0.0;
#pragma unroll
        for (int c = 0; c < 4; ++c) na = (c == ca) ? c_fN[q][c] : na;
#pragma unroll
        for (int c = 0; c < 4; ++c) mab[c] += ws * na * c_fN[q][c];
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int slot = ((fn[c][0] - li) * cj + (fn[c][1] - lj)) * ck + (fn[c][2] - lk);
#pragma unroll
        for (int fi = 0; fi < 3; ++fi) R[fi * EL2_CH + fi * cn + slot] -= tau * mab[c];
      }
    });
  }
  __syncthreads();
  // write-out: per row field f the chunk [3 pre(first node), 3 (pre(last) + cn(last))) of the value array = the LDS chunk f, copied with unit-stride lanes
  if (!(abl & 2)) {
    // 3 x the couplings of the block's live nodes (the nodes behind the mesh's last one carry pre = 0, cn = 0)
    const int64_t nlive = B.n_owned - (int64_t)blockIdx.x * EL2_NODES;
    const int lastlive = nlive < EL2_NODES ? (int)nlive - 1 : EL2_NODES - 1;
    const int L3 = 3 * (int)(s_pre[lastlive] + s_cn[lastlive] - s_pre[0]);
#pragma unroll
    for (int f = 0; f < 3; ++f) {
      double* dst = vals + (int64_t)f * 3 * T + 3 * s_pre[0];
      const double* src = rows + f * EL2_CH;
      // 16 bytes per lane from the first 16-byte boundary of the destination on (a wave's store = 1 KB of one run); the odd ends by single lanes
      const int head = L3 > 0 ? (int)(((uintptr_t)dst >> 3) & 1) : 0;
      const int np = (L3 - head) >> 1;
      typedef double el2_d2 __attribute__((ext_vector_type(2)));
      for (int m = tid; m < np; m += MFEM_BLOCK) {
        const int idx = head + 2 * m;
        __builtin_nontemporal_store(el2_d2{src[idx], src[idx + 1]}, reinterpret_cast<el2_d2*>(dst + idx));
      }
      if (tid == 0 && head) __builtin_nontemporal_store(src[0], dst);
      if (tid == 64 && ((L3 - head) & 1)) __builtin_nontemporal_store(src[L3 - 1], dst + L3 - 1);
    }
  }
}

// Surface terms of the elasticity residual at control point (i, j, k): tau (0 - u_i) on penalty faces, sig_ij n_j on traction faces
// (cantilever/3D_Script.jl:60-61), added to r[3].
__device__ __forceinline__ void elasticity_face_residual(const BrickView& B, int i, int j, int k, double tau, uint32_t penalty, uint32_t traction,
                                                         double s11, double s22, double s33, double s23, double s13, double s12,
                                                         const double* __restrict__ x, double (&r)[3]) {
  const uint32_t both = penalty | traction;
  if (both != 0u) {
    const double sg[3][3] = {{s11, s12, s13}, {s12, s22, s23}, {s13, s23, s33}};
    visit_boundary_faces(B, i, j, k, both, [&](int nd, int side, int ca, const int (&fn)[4][3], const double (&Xf)[4][3]) {
      const bool pen = (penalty & face_bit(nd, side)) != 0u && tau != 0.0;
      const bool tra = (traction & face_bit(nd, side)) != 0u;
      double uf[4][3];
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int f = 0; f < 3; ++f) uf[c][f] = x[brick_xindex(B, f, fn[c][0], fn[c][1], fn[c][2])];
      for (int q = 0; q < B.ng * B.ng; ++q) {
        double nrm[3];
        const double ws = face_geom(Xf, q, side == 0, nrm);
        double na = 0.0, uq[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          na = (c == ca) ? c_fN[q][c] : na;
#pragma unroll
          for (int f = 0; f < 3; ++f) uq[f] += c_fN[q][c] * uf[c][f];
        }
#pragma unroll
        for (int f = 0; f < 3; ++f) {
          double v = 0.0;
          if (pen) v += tau * (0.0 - uq[f]);
          if (tra) v += sg[f][0] * nrm[0] + sg[f][1] * nrm[1] + sg[f][2] * nrm[2];
          r[f] += ws * na * v;
        }
      }
    });
  }
}

// Residual at x_star (matrix-free):
//   R[(i,a)] = -sum_q w sigma_ij(u) d_jN_a + sum_q w^s N_a [ tau (0 - u_i) |penalty faces + sig_ij n_j |traction faces ]
__global__ __launch_bounds__(MFEM_BLOCK) void k_elasticity_residual(BrickView B, double lam, double mu, double tau,
                                                                      uint32_t penalty, uint32_t traction,
                                                                      double s11, double s22, double s33, double s23,
                                                                      double s13, double s12,
                                                                      const double* __restrict__ x,
                                                                      double* __restrict__ res) {
  const int64_t node = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (node >= B.n_owned) return;
  int i, j, k;
  node_ijk(B, node, i, j, k);
  double r[3] = {0.0, 0.0, 0.0};
  for (int e = 0; e < 8; ++e) {
    const int ex = e & 1, ey = (e >> 1) & 1, ez = e >> 2;
    const int I = i - 1 + ex, J = j - 1 + ey, K = k - 1 + ez;
    if (I < 0 || I >= B.ne0 || J < 0 || J >= B.ne1 || K < 0 || K >= B.ne2) continue;
    const int a = (1 - ex) + 2 * (1 - ey) + 4 * (1 - ez);
    double X[8][3], u[8][3];
    hex8_load_coords(B, I, J, K, X);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const int ni = I + (b & 1), nj = J + ((b >> 1) & 1), nk = K + (b >> 2);
#pragma unroll
      for (int f = 0; f < 3; ++f) u[b][f] = x[brick_xindex(B, f, ni, nj, nk)];
    }
    const int nq = B.ng * B.ng * B.ng;
    for (int q = 0; q < nq; ++q) {
      double g[8][3];
      const double wd = hex8_geom(X, q, g);
      double du[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};  // du[i][j] = d_j u_i
      double ga[3] = {0.0, 0.0, 0.0};
#pragma unroll
      for (int b = 0; b < 8; ++b) {
#pragma unroll
        for (int fi = 0; fi < 3; ++fi)
#pragma unroll
          for (int fj = 0; fj < 3; ++fj) du[fi][fj] += u[b][fi] * g[b][fj];
        const bool me = (b == a);
        ga[0] = me ? g[b][0] : ga[0];
        ga[1] = me ? g[b][1] : ga[1];
        ga[2] = me ? g[b][2] : ga[2];
      }
      const double tr = du[0][0] + du[1][1] + du[2][2];
#pragma unroll
      for (int fi = 0; fi < 3; ++fi) {
        double acc = 0.0;
#pragma unroll
        for (int fj = 0; fj < 3; ++fj) {
          double sg = mu * (du[fi][fj] + du[fj][fi]);
          if (fi == fj) sg += lam * tr;
          acc += sg * ga[fj];
        }
        r[fi] -= wd * acc;
      }
    }
  }
  elasticity_face_residual(B, i, j, k, tau, penalty, traction, s11, s22, s33, s23, s13, s12, x, r);
#pragma unroll
  for (int f = 0; f < 3; ++f) res[(int64_t)f * B.n_owned + node] = r[f];
}

// The same residual in the plane-sweep form of the thermal kernels (k_thermal_residual_sweep): a workgroup owns a 15 x 15 tile of
// control points in (j, k) and sweeps it through a segment of lattice planes; per element plane, phase A: thread <-> element of the
// 16 x 16 element tile -- ONE sum-factorised integration per element (sf_elasticity_fe; the kernel above integrates an element once
// per adjacent control point: 8 x the geometry), fe[3][8] to LDS; phase B: thread <-> control point adds the four elements above
// it to what it carried over from the four below.  Fixed summation order (elements (ey, ez) ascending, lower plane first), so a slab
// computes bitwise what the global mesh computes.  Surface terms: k_elasticity_residual_faces, one thread per boundary control point.
#define ESW_STRIDE 25
template <int NG>
__global__ __launch_bounds__(SW_THREADS) __attribute__((amdgpu_waves_per_eu(NG == 2 ? 2 : 1))) void k_elasticity_residual_sweep(BrickView B, int L, double lam, double mu,
                                                                                                                 const double* __restrict__ x,
                                                                                                                 double* __restrict__ res) {
  __shared__ double Fe[SW_THREADS * ESW_STRIDE];
  const int tid = threadIdx.x, ek = tid % SW_E, ej = tid / SW_E;
  const SweepTile T = sweep_tile(B, L);
  if (T.i0 >= T.i1) return;
  const int J = T.tj0 - 1 + ej, K = T.tk0 - 1 + ek;  // phase A: this thread's element column
  const bool el_ok = J >= 0 && J < B.ne1 && K >= 0 && K < B.ne2;
  const int j = T.tj0 + ej, k = T.tk0 + ek;          // phase B: this thread's control point column
  const bool nd_ok = ej < SW_N && ek < SW_N && j < B.m1 && k < B.m2;
  double Xn[3][4], Un[3][4];  // the plane requested last (consumed at the top of the next step)
  const int Jc = min(max(J, 0), B.ne1 - 1), Kc = min(max(K, 0), B.ne2 - 1);  // (threads without an element column load what a neighbour loads)
  // (every thread requests on every step, from clamped positions: behind a condition the requested registers meet their old values in a phi, and the copies that resolves into wait for the loads AT ONCE -- the prefetch then costs a memory round trip per plane instead of hiding one: round 5, found in the ISA)
  auto request = [&](int ip) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int64_t ci = brick_cindex(B, ip, Jc + (c & 1), Kc + (c >> 1));
      Xn[0][c] = B.X0[ci];
      Xn[1][c] = B.X1[ci];
      Xn[2][c] = B.X2[ci];
#pragma unroll
      for (int f = 0; f < 3; ++f) Un[f][c] = x[brick_xindex(B, f, ip, Jc + (c & 1), Kc + (c >> 1))];
    }
  };
  double X[3][2][4], U[3][2][4];
  const int Ifirst = max(T.i0 - 1, 0), Ilast = min(T.i1 - 1, B.ne0 - 1);  // element planes of this segment
  {
    request(Ifirst);
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        X[d][1][c] = Xn[d][c];
        U[d][1][c] = Un[d][c];
      }
    request(Ifirst + 1);
  }
  double carry[3] = {0.0, 0.0, 0.0};
  for (int I = T.i0 - 1; I < T.i1; ++I) {
    const bool plane = I >= 0 && I < B.ne0;  // workgroup-uniform
    if (plane && el_ok) {
#pragma unroll
      for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int c = 0; c < 4; ++c) {  // last step's upper nodes are this step's lower ones; the requested plane arrives
          X[d][0][c] = X[d][1][c];
          U[d][0][c] = U[d][1][c];
          X[d][1][c] = Xn[d][c];
          U[d][1][c] = Un[d][c];
        }
      double fe[3][2][4];
      sf_elasticity_fe<NG>(X, U, lam, mu, fe);
#pragma unroll
      for (int f = 0; f < 3; ++f)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          Fe[tid * ESW_STRIDE + f * 8 + 2 * c] = fe[f][0][c];
          Fe[tid * ESW_STRIDE + f * 8 + 2 * c + 1] = fe[f][1][c];
        }
    }
    request(min(max(I + 2, Ifirst), Ilast + 1));  // in flight during the rest of this plane's step: gather, write-out, barriers (mfem_lds_barrier does not wait for it)
    mfem_lds_barrier();
    if (nd_ok) {
      double lo[3] = {0.0, 0.0, 0.0}, up[3] = {0.0, 0.0, 0.0};
      if (plane) {
#pragma unroll
        for (int ez = 0; ez < 2; ++ez)
#pragma unroll
          for (int ey = 0; ey < 2; ++ey) {
            const int Jn = j - 1 + ey, Kn = k - 1 + ez;
            if (Jn < 0 || Jn >= B.ne1 || Kn < 0 || Kn >= B.ne2) continue;
            const double* fp = Fe + ((ej + ey) * SW_E + ek + ez) * ESW_STRIDE + 2 * ((1 - ey) + 2 * (1 - ez));
#pragma unroll
            for (int f = 0; f < 3; ++f) {
              lo[f] += fp[f * 8];
              up[f] += fp[f * 8 + 1];
            }
          }
      }
#pragma unroll
      for (int f = 0; f < 3; ++f) {
        if (I >= T.i0) res[(int64_t)f * B.n_owned + (int64_t)(I - B.plo) * B.plane_len + (int64_t)j * B.m2 + k] = carry[f] + lo[f];
        carry[f] = up[f];
      }
    }
    mfem_lds_barrier();
  }
}

__global__ __launch_bounds__(MFEM_BLOCK) void k_elasticity_residual_faces(BrickView B, double tau, uint32_t penalty, uint32_t traction,
                                                                            double s11, double s22, double s33, double s23, double s13,
                                                                            double s12, const double* __restrict__ x, double* __restrict__ res) {
  int i, j, k;
  if (!boundary_point(B, i, j, k)) return;
  double r[3] = {0.0, 0.0, 0.0};
  elasticity_face_residual(B, i, j, k, tau, penalty, traction, s11, s22, s33, s23, s13, s12, x, r);
  const int64_t node = (int64_t)(i - B.plo) * B.plane_len + (int64_t)j * B.m2 + k;
#pragma unroll
  for (int f = 0; f < 3; ++f) res[(int64_t)f * B.n_owned + node] += r[f];
}

int mfem_hex8_upload_tables(int ng);
int mfem_hex27_assemble_thermal(mfem_context_s* ctx, mfem_brick_s* m, mfem_csr_s* A, const mfem_thermal_params* p, double* vals);
int mfem_hex27_residual_thermal(mfem_context_s* ctx, mfem_brick_s* m, const mfem_thermal_params* p, const double* x_star,
                                const double* s, double* residue);
int mfem_brick_nitsche_faces(mfem_context_s* ctx, mfem_brick_s* m, bool matrix, const mfem_thermal_params* p, const double* xstar, double* out);

extern "C" int mfem_brick_assemble_thermal(mfem_context ctx, mfem_brick m, mfem_csr A, const mfem_thermal_params* p,
                                           double* vals) try {
  MFEM_REQUIRE(ctx && m && A && p && vals, "null argument");
  MFEM_REQUIRE(A->n == m->n_owned, "pattern was not built for 1 field on this brick");
  if (m->p == 2) {  // FP64 MFMA Ke = B^T D B path
    const int rc27 = mfem_hex27_assemble_thermal(ctx, m, A, p, vals);
    return rc27 ? rc27 : mfem_brick_nitsche_faces(ctx, m, true, p, nullptr, vals);
  }
  int rc = mfem_hex8_upload_tables(m->ng);
  if (rc) return rc;
  BrickView B = mfem_brick_view(m, 1);
  if (g_thermal_variant == 0 && (m->ng == 2 || m->ng == 3)) {
    int64_t grid;
    const int L = sweep_planes_m(m, &grid);
    if (m->ng == 2)
      hipLaunchKernelGGL(k_thermal_matrix_sweep<2>, dim3((unsigned)grid), dim3(MS_THREADS), 0, ctx->stream, B, L, p->k, vals, g_thermal_stage_rows.load());
    else
      hipLaunchKernelGGL(k_thermal_matrix_sweep<3>, dim3((unsigned)grid), dim3(MS_THREADS), 0, ctx->stream, B, L, p->k, vals, g_thermal_stage_rows.load());
  } else {
    const int64_t nti = ((m->phi - m->plo) + TT_NI - 1) / TT_NI, ntj = (m->m[1] + TT_NJ - 1) / TT_NJ, ntk = (m->m[2] + TT_NK - 1) / TT_NK;
    hipLaunchKernelGGL(k_thermal_matrix, dim3((unsigned)(nti * ntj * ntk)), dim3(TT_THREADS), 0, ctx->stream, B, p->k, vals);
  }
  MFEM_CHECK_LAUNCH();
  if (p->h != 0.0 && p->robin_faces != 0u) {
    hipLaunchKernelGGL(k_thermal_matrix_robin, boundary_grid(m), dim3(MFEM_BLOCK), 0, ctx->stream, B, p->h, p->robin_faces, vals);
    MFEM_CHECK_LAUNCH();
  }
  return mfem_brick_nitsche_faces(ctx, m, true, p, nullptr, vals);
} MFEM_API_CATCH("mfem_brick_assemble_thermal")

extern "C" int mfem_brick_residual_thermal(mfem_context ctx, mfem_brick m, const mfem_thermal_params* p,
                                           const double* x_star, const double* s, double* residue) try {
  MFEM_REQUIRE(ctx && m && p && x_star && residue, "null argument");
  if (m->p == 2) {
    const int rc27 = mfem_hex27_residual_thermal(ctx, m, p, x_star, s, residue);
    return rc27 ? rc27 : mfem_brick_nitsche_faces(ctx, m, false, p, x_star, residue);
  }
  int rc = mfem_hex8_upload_tables(m->ng);
  if (rc) return rc;
  BrickView B = mfem_brick_view(m, 1);
  if (g_thermal_variant == 0 && (m->ng == 2 || m->ng == 3)) {
    int64_t grid;
    const int L = sweep_planes(m, &grid);
    if (m->ng == 2)
      hipLaunchKernelGGL(k_thermal_residual_sweep<2>, dim3((unsigned)grid), dim3(SW_THREADS), 0, ctx->stream, B, L, p->k, x_star, s,
                         residue, (g_thermal_stage_rows.load() >> 1) & 1);
    else
      hipLaunchKernelGGL(k_thermal_residual_sweep<3>, dim3((unsigned)grid), dim3(SW_THREADS), 0, ctx->stream, B, L, p->k, x_star, s,
                         residue, (g_thermal_stage_rows.load() >> 1) & 1);
    if (p->h != 0.0 && p->robin_faces != 0u) {
      MFEM_CHECK_LAUNCH();
      hipLaunchKernelGGL(k_thermal_residual_robin, boundary_grid(m), dim3(MFEM_BLOCK), 0, ctx->stream, B, p->h, p->Tenv, p->robin_faces,
                         x_star, residue);
    }
  } else {
    const int64_t nti = ((m->phi - m->plo) + TT_NI - 1) / TT_NI, ntj = (m->m[1] + TT_NJ - 1) / TT_NJ, ntk = (m->m[2] + TT_NK - 1) / TT_NK;
    hipLaunchKernelGGL(k_thermal_residual, dim3((unsigned)(nti * ntj * ntk)), dim3(TT_THREADS), 0, ctx->stream, B, p->k, p->h,
                       p->Tenv, p->robin_faces, x_star, s, residue);
  }
  MFEM_CHECK_LAUNCH();
  return mfem_brick_nitsche_faces(ctx, m, false, p, x_star, residue);
} MFEM_API_CATCH("mfem_brick_residual_thermal")

static std::atomic<int> g_elasticity_variant{0};  // bit 0: the matrix row-owner kernel with in-place global accumulation; bit 1: the residual kernel that integrates per adjacent control point (both kept for comparison)
extern "C" int mfem_debug_set_elasticity(int variant) try {
  ++mfem_debug_epoch;
  g_elasticity_variant = variant & 0x3f;  // bits 2-4: timing-only ablations of the matrix kernel (no phases / no write-out / no integration); bit 5: no affine-element shortcut
  return MFEM_OK;
} MFEM_API_CATCH("mfem_debug_set_elasticity")

extern "C" int mfem_brick_assemble_elasticity(mfem_context ctx, mfem_brick m, mfem_csr A, const mfem_elasticity_params* p,
                                              double* vals) try {
  MFEM_REQUIRE(ctx && m && A && p && vals, "null argument");
  MFEM_REQUIRE(m->p == 1, "fused elasticity assembly is implemented for hex-8 (itp_order 1)");
  MFEM_REQUIRE(A->n == 3 * m->n_owned, "pattern was not built for 3 fields on this brick");
  int rc = mfem_hex8_upload_tables(m->ng);
  if (rc) return rc;
  BrickView B = mfem_brick_view(m, 3);
  const int64_t T = A->nnz / 9;
  if (!(g_elasticity_variant & 1)) {  // one thread per (control point, element), rows accumulated in LDS, written once
    const int grid = (int)((m->n_owned + EL2_NODES - 1) / EL2_NODES);
    hipLaunchKernelGGL(k_elasticity_matrix_lds, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, B, p->lambda, p->mu, p->tau,
                       p->penalty_faces, T, vals, g_elasticity_variant >> 2);
    MFEM_CHECK_LAUNCH();
    return MFEM_OK;
  }
  MFEM_CHECK_HIP(hipMemsetAsync(vals, 0, sizeof(double) * (size_t)A->nnz, ctx->stream));
  const int grid = (int)((m->n_owned + MFEM_BLOCK - 1) / MFEM_BLOCK);
  hipLaunchKernelGGL(k_elasticity_matrix, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, B, p->lambda, p->mu, p->tau,
                     p->penalty_faces, T, vals);
  MFEM_CHECK_LAUNCH();
  return MFEM_OK;
} MFEM_API_CATCH("mfem_brick_assemble_elasticity")

extern "C" int mfem_brick_residual_elasticity(mfem_context ctx, mfem_brick m, const mfem_elasticity_params* p,
                                              const double* x_star, double* residue) try {
  MFEM_REQUIRE(ctx && m && p && x_star && residue, "null argument");
  MFEM_REQUIRE(m->p == 1, "fused elasticity residual is implemented for hex-8 (itp_order 1)");
  int rc = mfem_hex8_upload_tables(m->ng);
  if (rc) return rc;
  BrickView B = mfem_brick_view(m, 3);
  // plane sweep, one integration per element: the 2-point rule (itg_order 2-3, every BASELINE config); the 3-point rule's 27 Gauss points
  // of three fields do not fit the register file of this form (it spills 2 KB per lane) and keep the kernel below
  if (!(g_elasticity_variant & 2) && m->ng == 2) {
    int64_t grid;
    const int L = sweep_planes(m, &grid);
    hipLaunchKernelGGL(k_elasticity_residual_sweep<2>, dim3((unsigned)grid), dim3(SW_THREADS), 0, ctx->stream, B, L, p->lambda, p->mu, x_star, residue);
    MFEM_CHECK_LAUNCH();
    if (((p->penalty_faces && p->tau != 0.0) || p->traction_faces) != 0) {
      hipLaunchKernelGGL(k_elasticity_residual_faces, boundary_grid(m), dim3(MFEM_BLOCK), 0, ctx->stream, B, p->tau, p->penalty_faces,
                         p->traction_faces, p->sig[0], p->sig[1], p->sig[2], p->sig[3], p->sig[4], p->sig[5], x_star, residue);
      MFEM_CHECK_LAUNCH();
    }
    return MFEM_OK;
  }
  const int grid = (int)((m->n_owned + MFEM_BLOCK - 1) / MFEM_BLOCK);
  hipLaunchKernelGGL(k_elasticity_residual, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, B, p->lambda, p->mu, p->tau,
                     p->penalty_faces, p->traction_faces, p->sig[0], p->sig[1], p->sig[2], p->sig[3], p->sig[4], p->sig[5],
                     x_star, residue);
  MFEM_CHECK_LAUNCH();
  return MFEM_OK;
} MFEM_API_CATCH("mfem_brick_residual_elasticity")
