// Structured hex mesh (make_Brick ordering) shared declarations.
#pragma once
#include "common.h"

#define BRICK_MAX_NG 4
#define BRICK_MAX_Q (BRICK_MAX_NG * BRICK_MAX_NG * BRICK_MAX_NG)

// Per-dimension lattice tables: for lattice index t (0..m-1):
//   lo[t]  first coupled lattice index, c[t] number of coupled indices, P[t] = sum_{t'<t} c[t']
struct BrickDim {
  int32_t m;        // lattice points
  int32_t ne;       // elements
  const int32_t* lo;
  const int32_t* c;
  const int64_t* P;
  int64_t S;        // sum of c
};

struct mfem_brick_s {
  mfem_context_s* ctx;
  int32_t ne[3];      // elements per dim
  int32_t p;          // itp_order
  int32_t itg_order, ng;  // Gauss points per dim
  double len[3];
  int32_t m[3];       // lattice points per dim
  int64_t plane_len;  // m[1]*m[2]
  // slab (in lattice planes along dim 0)
  int32_t plo, phi;   // owned planes [plo, phi)
  int32_t clo, chi;   // planes with coordinates stored [clo, chi)
  int64_t n_owned;    // (phi-plo)*plane_len
  double* coords[3];  // SoA, (chi-clo)*plane_len each
  // device tables (3 dims packed)
  int32_t* d_lo[3];
  int32_t* d_c[3];
  int64_t* d_P[3];
  int64_t S[3];
  int64_t Pplo;       // P_x[plo]
};

struct BrickView {  // POD passed to kernels
  int32_t ne0, ne1, ne2;
  int32_t m0, m1, m2;
  int32_t p, ng;
  int32_t plo, phi, clo, chi;
  int32_t gw;         // ghost planes per side of a slab = interpolation order (a row couples to nodes up to p planes away)
  int64_t plane_len, n_owned;
  const double* X0;
  const double* X1;
  const double* X2;
  const int32_t* lo0; const int32_t* lo1; const int32_t* lo2;
  const int32_t* c0;  const int32_t* c1;  const int32_t* c2;
  const int64_t* P0;  const int64_t* P1;  const int64_t* P2;
  int64_t S1, S2, Pplo;
  int32_t nfields;
};

BrickView mfem_brick_view(const mfem_brick_s* m, int nfields);

// node (i,j,k) -> index into the local solution vector of a field-major slab vector
//   owned: f*n_owned + (i-plo)*PL + j*m2 + k ; ghosts behind all owned entries, per field a low block (planes plo-gw .. plo-1)
//   and a high block (planes phi .. phi+gw-1) of gw planes each
__device__ __forceinline__ int64_t brick_xindex(const BrickView& B, int f, int i, int j, int k) {
  const int64_t inplane = (int64_t)j * B.m2 + k;
  if (i >= B.plo && i < B.phi) return (int64_t)f * B.n_owned + (int64_t)(i - B.plo) * B.plane_len + inplane;
  const int side = (i < B.plo) ? 0 : 1;
  const int off = side ? i - B.phi : i - (B.plo - B.gw);
  return (int64_t)B.nfields * B.n_owned + ((int64_t)(f * 2 + side) * B.gw + off) * B.plane_len + inplane;
}
__device__ __forceinline__ int64_t brick_cindex(const BrickView& B, int i, int j, int k) {
  return (int64_t)(i - B.clo) * B.plane_len + (int64_t)j * B.m2 + k;
}
// number of matrix entries of ONE field block in the rows of all owned nodes before node (i,j,k)
__device__ __forceinline__ int64_t brick_prefix(const BrickView& B, int i, int j, int k) {
  return (B.P0[i] - B.Pplo) * B.S1 * B.S2 + (int64_t)B.c0[i] * (B.P1[j] * B.S2 + (int64_t)B.c1[j] * B.P2[k]);
}
