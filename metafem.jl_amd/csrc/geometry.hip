// Generic geometry update for arbitrary (unstructured) classical meshes -- the tables the S3 operators consume:
//   update_BasicElements_{2,3}D      reference mesh/unstructured_mesh/4_Update_Integrator.jl:2-33
//   inv_Jac_2D / inv_Jac_3D          :77-121
//   update_Basic_itgval_1_{2,3}D     :125-154  (first-order push-forward)
//   update_BasicBoundary_{2,3}D      :35-75, tangents :163-196, normals + surface det :198-227
// The reference does this with dim^2 skinny CUBLAS GEMMs plus two one-thread-per-element kernels that write
// element-strided (uncoalesced) tables.  Here one thread owns one (quadrature point, element) pair, the q index is
// the fastest thread index, so the integral_vals[q, a, s, e] stores of a wave are unit-stride runs of `itg`
// doubles, and J / J^-1 never leave registers.  (The structured fast paths do not use these tables at all.)
#include "common.h"

template <int DIM>
__device__ __forceinline__ double inv_jac(const double (&J)[3][3], double (&I)[3][3]) {
  if (DIM == 2) {
    const double det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
    I[0][0] = J[1][1] / det;
    I[0][1] = -J[0][1] / det;
    I[1][0] = -J[1][0] / det;
    I[1][1] = J[0][0] / det;
    return det;
  }
  const double det = J[0][0] * J[1][1] * J[2][2] - J[0][0] * J[1][2] * J[2][1] - J[0][1] * J[1][0] * J[2][2] +
                     J[0][1] * J[1][2] * J[2][0] + J[0][2] * J[1][0] * J[2][1] - J[0][2] * J[1][1] * J[2][0];
  I[0][0] = (J[1][1] * J[2][2] - J[1][2] * J[2][1]) / det;
  I[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) / det;
  I[0][2] = (J[0][1] * J[1][2] - J[1][1] * J[0][2]) / det;
  I[1][0] = (J[1][2] * J[2][0] - J[2][2] * J[1][0]) / det;
  I[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) / det;
  I[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) / det;
  I[2][0] = (J[1][0] * J[2][1] - J[1][1] * J[2][0]) / det;
  I[2][1] = (J[0][1] * J[2][0] - J[2][1] * J[0][0]) / det;
  I[2][2] = (J[0][0] * J[1][1] - J[1][0] * J[0][1]) / det;
  return det;
}

// work item t = q + itg * host.  ref: [itg, itp, 1+DIM] of the face/element the host uses (ref_stride doubles per
// face id, eindex selects it; eindex == nullptr => elements).
template <int DIM>
__global__ __launch_bounds__(MFEM_BLOCK) void k_update_geometry(
    int itg, int itp, int64_t nhost, int64_t ncp, const double* __restrict__ ref, int64_t ref_stride,
    const double* __restrict__ wq, int64_t w_stride, const double* __restrict__ tan, int64_t tan_stride,
    const double* __restrict__ coords, const int32_t* __restrict__ cp, const int32_t* __restrict__ host_el,
    const int32_t* __restrict__ eindex, int base, double* __restrict__ vals, double* __restrict__ weights,
    double* __restrict__ normals) {
  const int64_t total = (int64_t)itg * nhost;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += stride) {
    const int q = (int)(t % itg);
    const int64_t h = t / itg;
    const int64_t e = host_el ? (int64_t)host_el[h] - base : h;
    const int f = eindex ? eindex[h] - base : 0;
    const double* R = ref + (int64_t)f * ref_stride;
    const int32_t* cpe = cp + (int64_t)itp * e;
    double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int a = 0; a < itp; ++a) {
      const int64_t c = (int64_t)cpe[a] - base;
#pragma unroll
      for (int i = 0; i < DIM; ++i) {
        const double xi = coords[c + (int64_t)i * ncp];
#pragma unroll
        for (int X = 0; X < DIM; ++X) J[i][X] += R[q + itg * (a + itp * (1 + X))] * xi;  // :9
      }
    }
    double I[3][3];
    const double det = inv_jac<DIM>(J, I);
    double* out = vals + (int64_t)itg * itp * (1 + DIM) * h + q;
    for (int a = 0; a < itp; ++a) {
      out[itg * a] = R[q + itg * a];  // integral_vals[..., 1,1,1, el] .= ref_itp_vals[..., 1,1,1]  (:25)
#pragma unroll
      for (int s = 0; s < DIM; ++s) {
        double v = 0.0;
#pragma unroll
        for (int m = 0; m < DIM; ++m) v += R[q + itg * (a + itp * (1 + m))] * I[m][s];  // :133-142
        out[itg * (a + itp * (1 + s))] = v;
      }
    }
    if (!eindex) {
      weights[t] = wq[q] * det;  // :30
    } else {
      // tangents = J * reference tangents (:163-196); normal + surface det (:198-227)
      const double* T = tan + (int64_t)f * tan_stride;  // [itg, DIM, DIM-1]
      double tg[3][2] = {{0, 0}, {0, 0}, {0, 0}};
#pragma unroll
      for (int i = 0; i < DIM; ++i)
#pragma unroll
        for (int k = 0; k < DIM - 1; ++k)
#pragma unroll
          for (int X = 0; X < DIM; ++X) tg[i][k] += J[i][X] * T[q + itg * (X + DIM * k)];
      double nrm[3], ld;
      if (DIM == 2) {
        ld = sqrt(tg[0][0] * tg[0][0] + tg[1][0] * tg[1][0]);
        nrm[0] = tg[1][0] / ld;
        nrm[1] = -tg[0][0] / ld;
      } else {
        const double r0 = tg[1][0] * tg[2][1] - tg[2][0] * tg[1][1];
        const double r1 = -tg[0][0] * tg[2][1] + tg[2][0] * tg[0][1];
        const double r2 = tg[0][0] * tg[1][1] - tg[1][0] * tg[0][1];
        ld = sqrt(r0 * r0 + r1 * r1 + r2 * r2);
        nrm[0] = r0 / ld;
        nrm[1] = r1 / ld;
        nrm[2] = r2 / ld;
      }
      weights[t] = wq[(int64_t)f * w_stride + q] * ld;  // :71
#pragma unroll
      for (int i = 0; i < DIM; ++i) normals[q + (int64_t)itg * (i + (int64_t)DIM * h)] = nrm[i];
    }
  }
}

extern "C" int mfem_update_basic_elements(mfem_context ctx, int32_t dim, int32_t itg, int32_t itp, int64_t nel, int64_t ncp,
                                          const double* ref_itp_vals, const double* itg_weight, const double* coords,
                                          const int32_t* controlpoint_IDs, int32_t index_base, double* integral_vals,
                                          double* integral_weights) try {
  MFEM_REQUIRE(ctx, "null ctx");
  MFEM_REQUIRE(dim == 2 || dim == 3, "dim must be 2 or 3");
  MFEM_REQUIRE(itg > 0 && itp > 0 && nel >= 0 && ncp > 0, "bad sizes");
  MFEM_REQUIRE(index_base == 0 || index_base == 1, "index_base must be 0 or 1");
  if (nel == 0) return MFEM_OK;
  MFEM_REQUIRE(ref_itp_vals && itg_weight && coords && controlpoint_IDs && integral_vals && integral_weights, "null array");
  const int grid = mfem_grid_for((int64_t)itg * nel, MFEM_BLOCK, ctx->num_cus * 16);
  if (dim == 2)
    hipLaunchKernelGGL(k_update_geometry<2>, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, itg, itp, nel, ncp, ref_itp_vals,
                       (int64_t)0, itg_weight, (int64_t)0, (const double*)nullptr, (int64_t)0, coords, controlpoint_IDs,
                       (const int32_t*)nullptr, (const int32_t*)nullptr, index_base, integral_vals, integral_weights,
                       (double*)nullptr);
  else
    hipLaunchKernelGGL(k_update_geometry<3>, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, itg, itp, nel, ncp, ref_itp_vals,
                       (int64_t)0, itg_weight, (int64_t)0, (const double*)nullptr, (int64_t)0, coords, controlpoint_IDs,
                       (const int32_t*)nullptr, (const int32_t*)nullptr, index_base, integral_vals, integral_weights,
                       (double*)nullptr);
  MFEM_CHECK_LAUNCH();
  return MFEM_OK;
} MFEM_API_CATCH("mfem_update_basic_elements")

extern "C" int mfem_update_basic_boundary(mfem_context ctx, int32_t dim, int32_t itg_b, int32_t itp, int32_t n_face_ids,
                                          int64_t n_facets, int64_t ncp, const double* bdy_ref_itp_vals,
                                          const double* bdy_itg_weights, const double* bdy_tangent_directions,
                                          const double* coords, const int32_t* controlpoint_IDs, const int32_t* element_ID,
                                          const int32_t* element_eindex, int32_t index_base, double* integral_vals,
                                          double* integral_weights, double* normal_directions) try {
  MFEM_REQUIRE(ctx, "null ctx");
  MFEM_REQUIRE(dim == 2 || dim == 3, "dim must be 2 or 3");
  MFEM_REQUIRE(itg_b > 0 && itp > 0 && n_face_ids > 0 && n_facets >= 0 && ncp > 0, "bad sizes");
  MFEM_REQUIRE(index_base == 0 || index_base == 1, "index_base must be 0 or 1");
  if (n_facets == 0) return MFEM_OK;
  MFEM_REQUIRE(bdy_ref_itp_vals && bdy_itg_weights && bdy_tangent_directions && coords && controlpoint_IDs && element_ID &&
                   element_eindex && integral_vals && integral_weights && normal_directions, "null array");
  const int grid = mfem_grid_for((int64_t)itg_b * n_facets, MFEM_BLOCK, ctx->num_cus * 16);
  const int64_t rs = (int64_t)itg_b * itp * (1 + dim), ts = (int64_t)itg_b * dim * (dim - 1);
  if (dim == 2)
    hipLaunchKernelGGL(k_update_geometry<2>, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, itg_b, itp, n_facets, ncp,
                       bdy_ref_itp_vals, rs, bdy_itg_weights, (int64_t)itg_b, bdy_tangent_directions, ts, coords,
                       controlpoint_IDs, element_ID, element_eindex, index_base, integral_vals, integral_weights,
                       normal_directions);
  else
    hipLaunchKernelGGL(k_update_geometry<3>, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, itg_b, itp, n_facets, ncp,
                       bdy_ref_itp_vals, rs, bdy_itg_weights, (int64_t)itg_b, bdy_tangent_directions, ts, coords,
                       controlpoint_IDs, element_ID, element_eindex, index_base, integral_vals, integral_weights,
                       normal_directions);
  MFEM_CHECK_LAUNCH();
  return MFEM_OK;
} MFEM_API_CATCH("mfem_update_basic_boundary")
