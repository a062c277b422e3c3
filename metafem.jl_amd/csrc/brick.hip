// Structured brick mesh: make_Brick + mesh_Classical(:Lagrange) + assemble_SparseID! replacement.
//   make_Brick                     reference mesh/ref_geometry/201_Helper_TM.jl:36-51
//   control-point lattice          mesh/unstructured_mesh/3_InitializeMesh.jl:70-163 (Lagrange cube)
//   assemble_SparseID!/KIJ/coosort solver/03_GlobalAssembly.jl:77-168, misc/04_GPU_Utils.jl:87-118
// The reference finds the unique (cp_i, cp_j) pairs with a GPU hash table sized from itp^2*nel keys,
// stores a per-element slot table and sorts the COO (F7: > 140 GB at 256^3).  On a lattice the pattern
// is separable per dimension, so rowptr and colidx are written directly in row-sorted CSR order from
// three 1-D tables (first coupled index, count, prefix count) -- no keys, no table, no sort.
#include "brick.h"

#include <vector>

#include "blas1.h"

BrickView mfem_brick_view(const mfem_brick_s* m, int nfields) {
  BrickView B;
  B.ne0 = m->ne[0]; B.ne1 = m->ne[1]; B.ne2 = m->ne[2];
  B.m0 = m->m[0]; B.m1 = m->m[1]; B.m2 = m->m[2];
  B.p = m->p; B.ng = m->ng;
  B.plo = m->plo; B.phi = m->phi; B.clo = m->clo; B.chi = m->chi;
  B.gw = m->p;
  B.plane_len = m->plane_len; B.n_owned = m->n_owned;
  B.X0 = m->coords[0]; B.X1 = m->coords[1]; B.X2 = m->coords[2];
  B.lo0 = m->d_lo[0]; B.lo1 = m->d_lo[1]; B.lo2 = m->d_lo[2];
  B.c0 = m->d_c[0]; B.c1 = m->d_c[1]; B.c2 = m->d_c[2];
  B.P0 = m->d_P[0]; B.P1 = m->d_P[1]; B.P2 = m->d_P[2];
  B.S1 = m->S[1]; B.S2 = m->S[2]; B.Pplo = m->Pplo;
  B.nfields = nfields;
  return B;
}

__global__ __launch_bounds__(MFEM_BLOCK) void k_brick_coords(BrickView B, double h0, double h1, double h2,
                                                               double* __restrict__ X0, double* __restrict__ X1,
                                                               double* __restrict__ X2) {
  const int64_t total = (int64_t)(B.chi - B.clo) * B.plane_len;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += stride) {
    const int i = (int)(t / B.plane_len) + B.clo;
    const int64_t rem = t % B.plane_len;
    const int j = (int)(rem / B.m2), k = (int)(rem % B.m2);
    // make_Brick: coors = dx * index (201_Helper_TM.jl:37-41), dx = x/n; the order-p lattice subdivides dx by p
    X0[t] = h0 * i;
    X1[t] = h1 * j;
    X2[t] = h2 * k;
  }
}

static int upload_dim_tables(mfem_brick_s* b, int d) {
  const int m = b->m[d], p = b->p;
  mfem_host_alloc_probe();
  std::vector<int32_t> lo(m), c(m);
  std::vector<int64_t> P(m + 1);
  int64_t acc = 0;
  for (int t = 0; t < m; ++t) {
    int l, h;
    if (t % p != 0) {  // interior node of one element
      l = (t / p) * p;
      h = l + p;
    } else {
      l = t - p < 0 ? 0 : t - p;
      h = t + p > m - 1 ? m - 1 : t + p;
    }
    lo[t] = l;
    c[t] = h - l + 1;
    P[t] = acc;
    acc += c[t];
  }
  P[m] = acc;
  b->S[d] = acc;
  MFEM_CHECK_HIP(hipMalloc(&b->d_lo[d], sizeof(int32_t) * m));
  MFEM_CHECK_HIP(hipMalloc(&b->d_c[d], sizeof(int32_t) * m));
  MFEM_CHECK_HIP(hipMalloc(&b->d_P[d], sizeof(int64_t) * (m + 1)));
  MFEM_CHECK_HIP(hipMemcpy(b->d_lo[d], lo.data(), sizeof(int32_t) * m, hipMemcpyHostToDevice));
  MFEM_CHECK_HIP(hipMemcpy(b->d_c[d], c.data(), sizeof(int32_t) * m, hipMemcpyHostToDevice));
  MFEM_CHECK_HIP(hipMemcpy(b->d_P[d], P.data(), sizeof(int64_t) * (m + 1), hipMemcpyHostToDevice));
  if (d == 0) b->Pplo = 0;
  return MFEM_OK;
}

static int brick_alloc_coords(mfem_brick_s* b) {
  mfem_context_s* ctx = b->ctx;
  for (int d = 0; d < 3; ++d) {
    if (b->coords[d]) MFEM_CHECK_HIP(hipFree(b->coords[d]));
    b->coords[d] = nullptr;
  }
  const int64_t nc = (int64_t)(b->chi - b->clo) * b->plane_len;
  for (int d = 0; d < 3; ++d) MFEM_CHECK_HIP(hipMalloc(&b->coords[d], sizeof(double) * (nc > 0 ? nc : 1)));
  BrickView B = mfem_brick_view(b, 1);
  const double h0 = b->len[0] / (b->p * b->ne[0]), h1 = b->len[1] / (b->p * b->ne[1]), h2 = b->len[2] / (b->p * b->ne[2]);
  hipLaunchKernelGGL(k_brick_coords, dim3(mfem_grid_for(nc, MFEM_BLOCK, ctx->num_cus * 8)), dim3(MFEM_BLOCK), 0,
                     ctx->stream, B, h0, h1, h2, b->coords[0], b->coords[1], b->coords[2]);
  MFEM_CHECK_LAUNCH();
  return MFEM_OK;
}

extern "C" int mfem_brick_create(mfem_context ctx, int32_t nx, int32_t ny, int32_t nz, double lx, double ly, double lz,
                                 int32_t itp_order, int32_t itg_order, mfem_brick* out) try {
  MFEM_REQUIRE(ctx && out, "null argument");
  MFEM_REQUIRE(nx > 0 && ny > 0 && nz > 0, "element counts must be positive");
  MFEM_REQUIRE(itp_order == 1 || itp_order == 2, "Lagrange cube order must be 1 or 2 (reference 3_InitializeMesh.jl:132-135)");
  MFEM_REQUIRE(itg_order >= 0 && itg_order <= 7, "itg_order out of range (Gauss tables exist for 1..4 points)");
  const int64_t m0 = (int64_t)itp_order * nx + 1, m1 = (int64_t)itp_order * ny + 1, m2 = (int64_t)itp_order * nz + 1;
  MFEM_REQUIRE(m0 * m1 * m2 < ((int64_t)1 << 31), "control-point ids must fit int32 (FEM_Int)");
  mfem_host_alloc_probe();
  mfem_brick_s* b = new mfem_brick_s();
  memset(b, 0, sizeof(*b));
  b->ctx = ctx;
  b->ne[0] = nx; b->ne[1] = ny; b->ne[2] = nz;
  b->p = itp_order;
  b->itg_order = itg_order;
  b->ng = (itg_order + 2) / 2;  // ceil((itg_order+1)/2), spatial_discretization/103_Integrations.jl:15
  if (b->ng < 1) b->ng = 1;
  b->len[0] = lx; b->len[1] = ly; b->len[2] = lz;
  b->m[0] = (int32_t)m0; b->m[1] = (int32_t)m1; b->m[2] = (int32_t)m2;
  b->plane_len = m1 * m2;
  b->plo = 0; b->phi = b->m[0]; b->clo = 0; b->chi = b->m[0];
  b->n_owned = m0 * b->plane_len;
  int rc = MFEM_OK;
  try {
    for (int d = 0; d < 3 && rc == MFEM_OK; ++d) rc = upload_dim_tables(b, d);
  } catch (...) {  // (host vectors of the tables: nothing half-built is left behind; the entry point's handler reports it)
    mfem_brick_destroy(b);
    throw;
  }
  if (rc == MFEM_OK) rc = brick_alloc_coords(b);
  if (rc != MFEM_OK) {
    mfem_brick_destroy(b);
    return rc;
  }
  *out = b;
  return MFEM_OK;
} MFEM_API_CATCH("mfem_brick_create")

extern "C" int mfem_brick_destroy(mfem_brick b) try {
  if (!b) return MFEM_OK;
  for (int d = 0; d < 3; ++d) {
    if (b->coords[d]) hipFree(b->coords[d]);
    if (b->d_lo[d]) hipFree(b->d_lo[d]);
    if (b->d_c[d]) hipFree(b->d_c[d]);
    if (b->d_P[d]) hipFree(b->d_P[d]);
  }
  delete b;
  return MFEM_OK;
} MFEM_API_CATCH("mfem_brick_destroy")

extern "C" int64_t mfem_brick_num_controlpoints(mfem_brick b) { return b ? (int64_t)b->m[0] * b->plane_len : -1; }
extern "C" int64_t mfem_brick_num_elements(mfem_brick b) { return b ? (int64_t)b->ne[0] * b->ne[1] * b->ne[2] : -1; }
extern "C" double* mfem_brick_coords(mfem_brick b, int32_t d) { return (b && d >= 0 && d < 3) ? b->coords[d] : nullptr; }

extern "C" int mfem_brick_set_slab(mfem_brick b, int32_t plane_lo, int32_t plane_hi) try {
  MFEM_REQUIRE(b, "null brick");
  MFEM_REQUIRE(plane_lo >= 0 && plane_hi <= b->m[0] && plane_lo < plane_hi, "bad plane range");
  const int gw = b->p;  // ghost planes per side
  // order 2: a slab starts and ends on element boundaries (even planes), so that the interface element plane is evaluated
  // on both sides and every owned row finds its columns within two planes
  MFEM_REQUIRE(b->p == 1 || (plane_lo % b->p == 0 && (plane_hi % b->p == 0 || plane_hi == b->m[0])),
               "slabs of an order-p lattice start and end on element boundaries (plane index divisible by p)");
  MFEM_REQUIRE((plane_lo == 0 && plane_hi == b->m[0]) || plane_hi - plane_lo >= gw, "a slab owns at least p planes");
  b->plo = plane_lo;
  b->phi = plane_hi;
  b->clo = plane_lo - gw < 0 ? 0 : plane_lo - gw;
  b->chi = plane_hi + gw > b->m[0] ? b->m[0] : plane_hi + gw;
  b->n_owned = (int64_t)(plane_hi - plane_lo) * b->plane_len;
  int64_t pplo = 0;
  MFEM_CHECK_HIP(hipMemcpy(&pplo, b->d_P[0] + plane_lo, sizeof(int64_t), hipMemcpyDeviceToHost));
  b->Pplo = pplo;
  return brick_alloc_coords(b);
} MFEM_API_CATCH("mfem_brick_set_slab")

// ---- pattern ---------------------------------------------------------------------------------
// rows: f*n_owned + node ; row (f,node) holds F blocks of c(node) entries, block g lists the coupled
// lattice nodes in (i,j,k)-lexicographic = ascending column order.
__global__ __launch_bounds__(MFEM_BLOCK) void k_brick_pattern(BrickView B, int64_t T /* one-block entries over owned nodes */,
                                                                int64_t* __restrict__ rowptr, int32_t* __restrict__ col) {
  const int F = B.nfields;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t node = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; node < B.n_owned; node += stride) {
    const int i = (int)(node / B.plane_len) + B.plo;
    const int64_t rem = node % B.plane_len;
    const int j = (int)(rem / B.m2), k = (int)(rem % B.m2);
    const int64_t pre = brick_prefix(B, i, j, k);
    const int ci = B.c0[i], cj = B.c1[j], ck = B.c2[k];
    const int li = B.lo0[i], lj = B.lo1[j], lk = B.lo2[k];
    const int64_t cn = (int64_t)ci * cj * ck;
    for (int f = 0; f < F; ++f) {
      const int64_t row = (int64_t)f * B.n_owned + node;
      const int64_t start = (int64_t)f * F * T + (int64_t)F * pre;
      rowptr[row] = start;
      for (int g = 0; g < F; ++g) {
        int64_t o = start + g * cn;
        for (int a = 0; a < ci; ++a)
          for (int b = 0; b < cj; ++b)
            for (int c = 0; c < ck; ++c) col[o++] = (int32_t)brick_xindex(B, g, li + a, lj + b, lk + c);
      }
    }
    if (node == B.n_owned - 1) rowptr[(int64_t)F * B.n_owned] = (int64_t)F * F * T;
  }
}

int mfem_csr_plan(mfem_context_s* ctx, mfem_csr_s* A);

extern "C" int mfem_brick_pattern(mfem_context ctx, mfem_brick b, int32_t n_fields, mfem_csr* out) try {
  MFEM_REQUIRE(ctx && b && out, "null argument");
  MFEM_REQUIRE(n_fields >= 1 && n_fields <= 8, "n_fields out of range");
  int64_t Pphi = 0;
  MFEM_CHECK_HIP(hipMemcpy(&Pphi, b->d_P[0] + b->phi, sizeof(int64_t), hipMemcpyDeviceToHost));
  const int64_t T = (Pphi - b->Pplo) * b->S[1] * b->S[2];
  const int64_t n = (int64_t)n_fields * b->n_owned;
  const int64_t nnz = (int64_t)n_fields * n_fields * T;
  const int64_t xlen = n + (int64_t)2 * n_fields * b->p * b->plane_len;
  MFEM_REQUIRE(xlen < ((int64_t)1 << 31), "local column ids must fit int32");
  mfem_host_alloc_probe();
  mfem_csr_s* A = new mfem_csr_s();
  memset(A, 0, sizeof(*A));
  A->ctx = ctx;
  A->n = n;
  A->nnz = nnz;
  A->rowptr_bits = 64;
  A->index_base = 0;
  A->ncols = (b->plo > 0 || b->phi < b->m[0]) ? xlen : n;
  // hint for the solver layouts: the rows of one field are the nodes of a lattice with m[1] x m[2] points per plane (row = field * n_owned +
  // (plane * m1 + j) * m2 + k) -- any use of it must stay correct for an arbitrary pattern (it only orders work)
  A->lat_m1 = b->m[1];
  A->lat_m2 = b->m[2];
  A->lat_fields = n_fields;
  A->lat_m0 = b->m[0];
  A->lat_plo = b->plo;
  A->lat_gw = b->p;
  // (from here on every way out releases the half-built handle: a failed 14 GB hipMalloc at 512^3 must not leak the row pointers)
  auto build = [&]() -> int {
    MFEM_CHECK_HIP(hipMalloc(&A->owned_rowptr, sizeof(int64_t) * (n + 1)));
    MFEM_CHECK_HIP(hipMalloc(&A->owned_colidx, sizeof(int32_t) * (nnz > 0 ? nnz : 1)));
    A->rowptr = A->owned_rowptr;
    A->colidx = (const int32_t*)A->owned_colidx;
    BrickView B = mfem_brick_view(b, n_fields);
    hipLaunchKernelGGL(k_brick_pattern, dim3(mfem_grid_for(b->n_owned, MFEM_BLOCK, ctx->num_cus * 16)), dim3(MFEM_BLOCK), 0,
                       ctx->stream, B, T, (int64_t*)A->owned_rowptr, (int32_t*)A->owned_colidx);
    MFEM_CHECK_LAUNCH();
    return mfem_csr_plan(ctx, A);
  };
  int rc = MFEM_OK;
  try {
    rc = build();
  } catch (...) {
    mfem_csr_destroy(A);
    throw;
  }
  if (rc != MFEM_OK) {
    mfem_csr_destroy(A);
    return rc;
  }
  *out = A;
  return MFEM_OK;
} MFEM_API_CATCH("mfem_brick_pattern")
