// Weakly imposed (Nitsche-type) Dirichlet faces of the thermal form on a structured brick, hex-8 and hex-27:
//   fix_boundary = h_penalty*Bilinear(T, Tw - T) + k*Bilinear(T, n{i}*T{;i})      examples/thermal_conduction/2D_Script.jl:58
// i.e. the linear gradients   K[a][b] += sum_q w^s N_a (-h_penalty N_b + k n.grad N_b)       (05_CodeGenerator.jl:52-91 on a boundary
// and the residual            R[a]    += sum_q w^s N_a (h_penalty (Tw - T) + k n.grad T)       workpiece, :93-154)
// with the basis of ALL nodes of the host element evaluated on the face (05_CodeGenerator.jl:175-189: the normal derivative of the
// host element's off-face nodes does not vanish on the face) -- the term that makes the reference's own Dirichlet problems NONSYMMETRIC:
// row a (a face node) receives entries in the columns of the element's interior nodes, the mirrored rows receive nothing.
//
// Replaces, for these faces: update_BasicBoundary_3D + tangents / normals (4_Update_Integrator.jl:35-75,173-226) and the 1 + 3 _Kval_Basic /
// 1 + 3 _Res_Basic launches of the form on the stored facet tables (06_FEM_Kernel.jl:28-45,65-79).  Nothing per facet is stored: one
// thread per (boundary face element, face node a) rebuilds J, J^-1, the outward normal and the surface weight at the face's Gauss points
// from the host element's nodal coordinates and owns row a of the face element's contribution.  Face elements of one launch share no node
// (parity colouring in the two tangential directions, 4 launches per face), so the read-modify-write of row a is race-free without atomics.
// O(n^(2/3)) work: no tuning beyond that.
#include "brick.h"

struct NitscheArgs {
  BrickView B;
  double k, hp, Tw;
  int nd, side, colour, ng;
  double gp[4], gw[4];  // 1-D Gauss points / weights on [0, 1] (103_Integrations.jl:1-12)
};

template <int P>
__device__ __forceinline__ void lagrange_1d(double x, double (&L)[P + 1], double (&dL)[P + 1]) {
  if (P == 1) {  // 102_Interpolations.jl:3-23, nodes i / p
    L[0] = 1.0 - x;
    L[1] = x;
    dL[0] = -1.0;
    dL[1] = 1.0;
  } else {
    L[0] = 2.0 * (x - 0.5) * (x - 1.0);
    L[1] = -4.0 * x * (x - 1.0);
    L[P] = 2.0 * x * (x - 0.5);
    dL[0] = 4.0 * x - 3.0;
    dL[1] = -8.0 * x + 4.0;
    dL[P] = 4.0 * x - 1.0;
  }
}

template <int P, bool MATRIX>
__global__ __launch_bounds__(MFEM_BLOCK) void k_brick_nitsche(NitscheArgs A, const double* __restrict__ xstar, double* __restrict__ out) {
  constexpr int N1 = P + 1, NF = N1 * N1, NN = N1 * N1 * N1;
  const BrickView& B = A.B;
  const int ne[3] = {B.ne0, B.ne1, B.ne2};
  const int nd = A.nd, t1 = (nd + 1) % 3, t2 = (nd + 2) % 3;
  const int c1 = A.colour & 1, c2 = A.colour >> 1;
  const int n1 = (ne[t1] - c1 + 1) >> 1, n2 = (ne[t2] - c2 + 1) >> 1;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int a = (int)(tid % NF);
  const int64_t f = tid / NF;
  if (n1 <= 0 || n2 <= 0 || f >= (int64_t)n1 * n2) return;
  int E[3];
  E[nd] = A.side ? ne[nd] - 1 : 0;
  E[t1] = 2 * (int)(f % n1) + c1;
  E[t2] = 2 * (int)(f / n1) + c2;
  // the row this thread owns: face node a = a1 + N1 a2 in the two tangential directions
  int ga[3], la[3];
  la[nd] = A.side ? P : 0;
  la[t1] = a % N1;
  la[t2] = a / N1;
  for (int d = 0; d < 3; ++d) ga[d] = P * E[d] + la[d];
  if (ga[0] < B.plo || ga[0] >= B.phi) return;  // only owned rows (every node of the host element then lies within the stored planes)
  double X[NN][3], Tn[NN];
  for (int b = 0; b < NN; ++b) {
    const int b0 = b % N1, b1 = (b / N1) % N1, b2 = b / (N1 * N1);
    const int gi = P * E[0] + b0, gj = P * E[1] + b1, gk = P * E[2] + b2;
    const int64_t ci = brick_cindex(B, gi, gj, gk);
    X[b][0] = B.X0[ci];
    X[b][1] = B.X1[ci];
    X[b][2] = B.X2[ci];
    Tn[b] = MATRIX ? 0.0 : xstar[brick_xindex(B, 0, gi, gj, gk)];
  }
  double macc[NN];
  for (int b = 0; b < NN; ++b) macc[b] = 0.0;
  double racc = 0.0;
  for (int q = 0; q < A.ng * A.ng; ++q) {
    const int q1 = q % A.ng, q2 = q / A.ng;  // first in-face coordinate fastest (103_Integrations.jl:16-17,37)
    double xi[3];
    xi[nd] = A.side ? 1.0 : 0.0;
    xi[t1] = A.gp[q1];
    xi[t2] = A.gp[q2];
    double L[3][N1], dL[3][N1];
    for (int d = 0; d < 3; ++d) lagrange_1d<P>(xi[d], L[d], dL[d]);
    double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};  // J[i][m] = dx_i / dxi_m (4_Update_Integrator.jl:9)
    for (int b = 0; b < NN; ++b) {
      const int b0 = b % N1, b1 = (b / N1) % N1, b2 = b / (N1 * N1);
      const double d0 = dL[0][b0] * L[1][b1] * L[2][b2], d1 = L[0][b0] * dL[1][b1] * L[2][b2], d2 = L[0][b0] * L[1][b1] * dL[2][b2];
      for (int i = 0; i < 3; ++i) {
        J[i][0] += d0 * X[b][i];
        J[i][1] += d1 * X[b][i];
        J[i][2] += d2 * X[b][i];
      }
    }
    const double det = J[0][0] * J[1][1] * J[2][2] - J[0][0] * J[1][2] * J[2][1] - J[0][1] * J[1][0] * J[2][2] + J[0][1] * J[1][2] * J[2][0] +
                       J[0][2] * J[1][0] * J[2][1] - J[0][2] * J[1][1] * J[2][0];
    const double id = 1.0 / det;
    double I[3][3];  // I[m][s] = dxi_m / dx_s (inv_Jac_3D, :90-121)
    I[0][0] = (J[1][1] * J[2][2] - J[1][2] * J[2][1]) * id;
    I[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * id;
    I[0][2] = (J[0][1] * J[1][2] - J[1][1] * J[0][2]) * id;
    I[1][0] = (J[1][2] * J[2][0] - J[2][2] * J[1][0]) * id;
    I[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id;
    I[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * id;
    I[2][0] = (J[1][0] * J[2][1] - J[1][1] * J[2][0]) * id;
    I[2][1] = (J[0][1] * J[2][0] - J[2][1] * J[0][0]) * id;
    I[2][2] = (J[0][0] * J[1][1] - J[1][0] * J[0][1]) * id;
    // tangents J t_ref, the first one negated on the low face so that t1 x t2 points outward (103_Integrations.jl:40-47; :173-226)
    double ta[3], tb[3];
    for (int i = 0; i < 3; ++i) {
      ta[i] = A.side ? J[i][t1] : -J[i][t1];
      tb[i] = J[i][t2];
    }
    const double r0 = ta[1] * tb[2] - ta[2] * tb[1], r1 = -ta[0] * tb[2] + ta[2] * tb[0], r2 = ta[0] * tb[1] - ta[1] * tb[0];
    const double ld = sqrt(r0 * r0 + r1 * r1 + r2 * r2);
    const double nrm[3] = {r0 / ld, r1 / ld, r2 / ld};
    const double ws = A.gw[q1] * A.gw[q2] * ld;
    double v[3];  // n . grad N_b = sum_m dN_b/dxi_m v[m],  v[m] = sum_s I[m][s] n_s
    for (int m = 0; m < 3; ++m) v[m] = I[m][0] * nrm[0] + I[m][1] * nrm[1] + I[m][2] * nrm[2];
    const double Na = L[0][la[0]] * L[1][la[1]] * L[2][la[2]];
    double Tq = 0.0, dTn = 0.0;
    for (int b = 0; b < NN; ++b) {
      const int b0 = b % N1, b1 = (b / N1) % N1, b2 = b / (N1 * N1);
      const double Nb = L[0][b0] * L[1][b1] * L[2][b2];
      const double gb = dL[0][b0] * L[1][b1] * L[2][b2] * v[0] + L[0][b0] * dL[1][b1] * L[2][b2] * v[1] + L[0][b0] * L[1][b1] * dL[2][b2] * v[2];
      if (MATRIX) {
        macc[b] += ws * Na * (-A.hp * Nb + A.k * gb);
      } else {
        Tq += Nb * Tn[b];
        dTn += gb * Tn[b];
      }
    }
    if (!MATRIX) racc += ws * Na * (A.hp * (A.Tw - Tq) + A.k * dTn);
  }
  if (MATRIX) {
    const int64_t base = brick_prefix(B, ga[0], ga[1], ga[2]);
    const int lo0 = B.lo0[ga[0]], lo1 = B.lo1[ga[1]], lo2 = B.lo2[ga[2]], cc1 = B.c1[ga[1]], cc2 = B.c2[ga[2]];
    for (int b = 0; b < NN; ++b) {
      const int b0 = b % N1, b1 = (b / N1) % N1, b2 = b / (N1 * N1);
      out[base + ((int64_t)(P * E[0] + b0 - lo0) * cc1 + (P * E[1] + b1 - lo1)) * cc2 + (P * E[2] + b2 - lo2)] += macc[b];
    }
  } else {
    out[(int64_t)(ga[0] - B.plo) * B.plane_len + (int64_t)ga[1] * B.m2 + ga[2]] += racc;
  }
}

static const double NGP[4][4] = {{0.0, 0, 0, 0},
                                 {-0.57735026918962576451, 0.57735026918962576451, 0, 0},
                                 {-0.77459666924148337704, 0.0, 0.77459666924148337704, 0},
                                 {-0.86113631159405257522, -0.33998104358485626480, 0.33998104358485626480, 0.86113631159405257522}};
static const double NGW[4][4] = {{2.0, 0, 0, 0},
                                 {1.0, 1.0, 0, 0},
                                 {5.0 / 9.0, 8.0 / 9.0, 5.0 / 9.0, 0},
                                 {0.34785484513745385737, 0.65214515486254614263, 0.65214515486254614263, 0.34785484513745385737}};

// matrix: out = vals (CSR order, accumulated into); residual: out = residue (accumulated into), xstar = the local solution vector
int mfem_brick_nitsche_faces(mfem_context_s* ctx, mfem_brick_s* m, bool matrix, const mfem_thermal_params* p, const double* xstar, double* out) {
  if (p->fixed_faces == 0u || (p->h_penalty == 0.0 && p->k == 0.0)) return MFEM_OK;
  MFEM_REQUIRE(m->p == 1 || m->p == 2, "Nitsche faces: itp_order 1 or 2");
  MFEM_REQUIRE(m->ng >= 1 && m->ng <= 4, "Nitsche faces: 1..4 Gauss points per direction");
  BrickView B = mfem_brick_view(m, 1);
  NitscheArgs A;
  A.B = B;
  A.k = p->k;
  A.hp = p->h_penalty;
  A.Tw = p->Tw;
  A.ng = m->ng;
  for (int i = 0; i < 4; ++i) {
    A.gp[i] = i < m->ng ? NGP[m->ng - 1][i] / 2.0 + 0.5 : 0.0;
    A.gw[i] = i < m->ng ? NGW[m->ng - 1][i] / 2.0 : 0.0;
  }
  const int nf = (m->p + 1) * (m->p + 1);
  for (int nd = 0; nd < 3; ++nd) {
    const int t1 = (nd + 1) % 3, t2 = (nd + 2) % 3;
    for (int side = 0; side < 2; ++side) {
      // reference local face ids (002_Initialization.jl:8): 1 z=0, 2 y=0, 3 x=L, 4 y=L, 5 x=0, 6 z=L
      const int id = (nd == 0) ? (side ? 3 : 5) : (nd == 1) ? (side ? 4 : 2) : (side ? 6 : 1);
      if (!(p->fixed_faces & (1u << (id - 1)))) continue;
      if (nd == 0) {  // slab: the face lies in one control-point plane
        const int gp = side ? m->m[0] - 1 : 0;
        if (gp < m->plo || gp >= m->phi) continue;
      }
      for (int colour = 0; colour < 4; ++colour) {
        const int n1 = (m->ne[t1] - (colour & 1) + 1) >> 1, n2 = (m->ne[t2] - (colour >> 1) + 1) >> 1;
        if (n1 <= 0 || n2 <= 0) continue;
        A.nd = nd;
        A.side = side;
        A.colour = colour;
        const int64_t nthreads = (int64_t)n1 * n2 * nf;
        const unsigned grid = (unsigned)((nthreads + MFEM_BLOCK - 1) / MFEM_BLOCK);
        if (m->p == 1) {
          if (matrix)
            hipLaunchKernelGGL((k_brick_nitsche<1, true>), dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, A, xstar, out);
          else
            hipLaunchKernelGGL((k_brick_nitsche<1, false>), dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, A, xstar, out);
        } else {
          if (matrix)
            hipLaunchKernelGGL((k_brick_nitsche<2, true>), dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, A, xstar, out);
          else
            hipLaunchKernelGGL((k_brick_nitsche<2, false>), dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, A, xstar, out);
        }
        MFEM_CHECK_LAUNCH();
      }
    }
  }
  return MFEM_OK;
}
