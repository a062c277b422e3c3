// Fused assembly of constant-coefficient bilinear forms on UNSTRUCTURED classical meshes (any cube / simplex element the
// host hands reference tables for: quad-4/8, hex-8/20/27, tri-3/6, tet-4/10, ...): geometry on the fly + every term of an
// integration domain + scatter in one launch per colour (or one launch with FP64 atomics).  Replaces, for terms whose
// coefficient is a constant,
//   update_BasicElements / update_BasicBoundary + inv_Jac + update_Basic_itgval_1   mesh/unstructured_mesh/4_Update_Integrator.jl:2-154,163-227
//   the `vals = @. coeff * K_params * w[:, ids]` broadcasts and one _Kval_Basic launch per term    solver/05_CodeGenerator.jl:52-91, 06_FEM_Kernel.jl:28-45
// The reference stores the physical basis table of every element (hex-20 with 27 Gauss points: 17 KB per element) and
// re-reads it once per term (21 launches for 3-D elasticity).  Here a WAVE owns an element (or boundary facet):
//   1. the nodes' coordinates go to the wave's LDS block;
//   2. J, det, J^-1 (facets: tangents, surface det) per Gauss point from the reference table; the physical table
//      T[q][a][s] (s = value, d/dx_1.. d/dx_dim) is built in LDS and never leaves it;
//   3. every node pair (a, b) -- pairs spread over the lanes -- accumulates the small matrix
//      M_ab[s][s'] = sum_q w_q det_q T[q][a][s] T[q][b][s'], and each term is then  coef * M_ab[dual_s][base_s]:
//      the q loop runs once per pair, not once per term;
//   4. the sums of the terms of one sparse block are added to K through the slot table of mfem_pattern_build --
//      plain read-modify-write inside a colour (no two elements of a colour share a control point), FP64 atomics otherwise.
#include "common.h"

#define MA_MAX_TERMS 48
struct ConstTerms {
  int n;
  int32_t ds[MA_MAX_TERMS], bs[MA_MAX_TERMS], block[MA_MAX_TERMS];
  double coef[MA_MAX_TERMS];
};

// The terms as one dense coefficient row per sparse block (run of the term list): block value = sum_c c[k][c] * M[c], M = the NS x NS products of a node
// pair.  Built on the host per launch; read with uniform indices (scalar loads from the kernel arguments, no branch per term).
#define MA_MAX_RUNS 16
struct TermMatrix {
  int nruns;
  int32_t block[MA_MAX_RUNS];
  double c[MA_MAX_RUNS][16];
};

struct MeshItems {
  int itg, itp;
  int64_t ncp;
  const double* ref;     // [n_face_ids][itg, itp, 1 + dim]
  int64_t ref_stride;
  const double* wq;      // [n_face_ids][itg]
  int64_t w_stride;
  const double* tan;     // facets: [n_face_ids][itg, dim, dim - 1]; elements: nullptr
  int64_t tan_stride;
  const double* coords;  // SoA
  const int32_t* cp;     // [itp, nel]
  const int32_t* host_el;   // facets: element of item h; elements: nullptr
  const int32_t* eindex;    // facets: local face id of item h
  const int32_t* order;     // item processed by work unit t (colour order); nullptr = identity
  int base;
};

int g_mesh_stage_min_itp = 16;  // mfem_debug_set("mesh_stage_min_itp"): elements from this many nodes take the staged persistent form of the row-owner kernel
int g_mesh_term_matrix = 1;  // mfem_debug_set("mesh_term_matrix"): 0 = the non-staged element kernel walks the term list (A/B)
int g_mesh_abl = 0;  // mfem_debug_set("mesh_abl"): ablation of k_mesh_assemble phases (tools/u20_assembly_ab.py): 1 no pair products, 2 no stores, 4 no geometry, 8 no table, 16 no coordinate gather

template <int DIM>
__device__ __forceinline__ double ma_inv(const double (&J)[3][3], double (&I)[3][3]) {
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

// S0 = first physical-table slot the terms use (0 value, 1 first derivative), NS = number of consecutive slots from S0:
// (0, 1 + DIM) everything, (1, DIM) gradients only, (0, 1) values only.
// OUT: 0 = add into K through the slot table (plain read-modify-write: colour batches), 1 = the same with FP64 atomics,
//      2 = write the element matrix to an element-major scratch  S[((el * itp + a) * nb + k) * itp + b]  (k = index of the
//          term run / sparse block): unit-stride stores, read back row by row by k_mesh_gather.
// STAGE (round 6, elements of the row-owner form): PERSISTENT workgroups -- the elements' shared reference table (hex-20, 27 Gauss points: 17 KB) is copied to
// LDS once per workgroup and every wave then walks its share of the elements.  With a wave per element and the table read from the L2 an element took
// 38 us of a wave's time at 96^3 (three dependent phases of L2 round trips: 20 x 3 table reads per lane for the Jacobian alone); staging it per ELEMENT had
// been measured slower in round 2 (1.25 against 0.91 ms at 32^3) -- the copy has to be amortised over many elements.
//   The pair products of the STAGE form: a lane takes 4 dual x 2 base nodes (hex-20: 50 lanes, ONE round; 19 LDS reads per Gauss point for 8 pairs where the
//   4 x 1 form of the first version read 16 for 4), and the terms are applied as a dense coefficient matrix per sparse block, built once per workgroup in
//   LDS (the term list in the kernel arguments cost three dependent scalar loads and a branch per term and pair group: 4.8 of 15.3 ms on hex-20
//   elasticity, 21 terms).  DIAGT: every term pairs a word with itself (thermal: grad . grad) -- 3 products per pair and Gauss point instead of 9.
template <int DIM, int S0, int NS, int OUT, bool STAGE = false, bool DIAGT = false>
__global__ __launch_bounds__(MFEM_BLOCK) void k_mesh_assemble(MeshItems V, ConstTerms T, const int32_t* __restrict__ slots,
                                                                int64_t block_stride, double* __restrict__ K, int64_t t0,
                                                                int64_t t1, int nb, int abl, TermMatrix TM) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int itg = V.itg, itp = V.itp;
  const size_t per_wave = (size_t)itg * itp * NS + (size_t)itg * (1 + DIM * DIM) + (size_t)itp * DIM;
  const size_t ref_doubles = STAGE ? (size_t)itg * itp * (1 + DIM) : 0;
  constexpr int NM = DIAGT ? NS : NS * NS;  // products kept per pair
  const size_t coef_doubles = STAGE ? (size_t)nb * NM : 0;
  double* Rs = lds;                                          // [1 + DIM][itp][itg]: the reference table (STAGE)
  double* Cs = lds + ref_doubles;                            // [nb][NM]: block k = sum_c Cs[k][c] M[c] (STAGE)
  double* Tt = lds + ref_doubles + coef_doubles + (size_t)w * per_wave;     // [itg][itp][NS]
  double* wd = Tt + (size_t)itg * itp * NS;     // [itg]
  double* Ji = wd + itg;                        // [itg][DIM*DIM]  J^-1
  double* X = Ji + (size_t)itg * DIM * DIM;     // [itp][DIM]
  if (STAGE) {
    for (int i = threadIdx.x; i < (int)ref_doubles; i += blockDim.x) Rs[i] = V.ref[i];
    for (int c = threadIdx.x; c < (int)coef_doubles; c += blockDim.x) {  // terms in list order: a fixed sum
      const int k = c / NM, m = c - k * NM;
      double sum = 0.0;
      int run = -1;
      for (int i = 0; i < T.n; ++i) {
        if (i == 0 || T.block[i] != T.block[i - 1]) ++run;
        const int sel = DIAGT ? T.ds[i] - S0 : (T.ds[i] - S0) * NS + (T.bs[i] - S0);
        if (run == k && sel == m) sum += T.coef[i];
      }
      Cs[c] = sum;
    }
    __syncthreads();  // (the only workgroup barrier: every wave reaches it)
  }
  for (int64_t t = t0 + (int64_t)blockIdx.x * nw + w; t < t1; t += (int64_t)gridDim.x * nw) {  // (one trip unless STAGE: the grid covers the items)
  const int64_t h = V.order ? (int64_t)V.order[t] - V.base : t;
  const int64_t el = V.host_el ? (int64_t)V.host_el[h] - V.base : h;
  const int f = V.eindex ? V.eindex[h] - V.base : 0;
  const double* R = STAGE ? Rs : V.ref + (int64_t)f * V.ref_stride;
  const int32_t* cpe = V.cp + (int64_t)itp * el;
  for (int i = lane; i < ((abl & 16) ? 0 : itp * DIM); i += 64) {
    const int a = i / DIM, d = i - a * DIM;
    X[i] = V.coords[((int64_t)cpe[a] - V.base) + (int64_t)d * V.ncp];
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_wave_barrier();
  // ---- geometry per Gauss point (lane <-> q)
  const bool split_j = STAGE && 2 * itg <= 64;  // (hex-20 / hex-27 with 27 Gauss points: two lane groups share a point's Jacobian sum, half the nodes each)
  for (int q0 = 0; q0 < ((abl & 4) ? 0 : itg); q0 += 64) {
    const int hq = split_j ? lane / itg : 0, q = split_j ? lane - hq * itg : q0 + lane;
    const int half = split_j ? (itp + 1) >> 1 : itp;
    const int a_lo = hq < 2 ? hq * half : 0, a_hi = hq < 2 ? (a_lo + half < itp ? a_lo + half : itp) : 0;
    const bool qon = q < itg;
    double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int a = a_lo; a < a_hi; ++a) {
#pragma unroll
      for (int m = 0; m < DIM; ++m) {
        const double r = R[(qon ? q : 0) + itg * (a + itp * (1 + m))];
#pragma unroll
        for (int i = 0; i < DIM; ++i) J[i][m] += r * X[a * DIM + i];
      }
    }
    if (split_j) {
#pragma unroll
      for (int i = 0; i < DIM; ++i)
#pragma unroll
        for (int m = 0; m < DIM; ++m) J[i][m] += __shfl_down(J[i][m], itg);  // (group 0 takes group 1's half)
    }
    if (!qon || hq != 0) continue;
    double I[3][3];
    const double det = ma_inv<DIM>(J, I);
#pragma unroll
    for (int m = 0; m < DIM; ++m)
#pragma unroll
      for (int s = 0; s < DIM; ++s) Ji[q * DIM * DIM + m * DIM + s] = I[m][s];
    if (!V.eindex) {
      wd[q] = V.wq[q] * det;
    } else {  // surface weight: |J t1 x J t2| (3-D) or |J t1| (2-D)   4_Update_Integrator.jl:163-227
      const double* Tn = V.tan + (int64_t)f * V.tan_stride;
      double tg[3][2] = {{0, 0}, {0, 0}, {0, 0}};
#pragma unroll
      for (int i = 0; i < DIM; ++i)
#pragma unroll
        for (int k = 0; k < DIM - 1; ++k)
#pragma unroll
          for (int m = 0; m < DIM; ++m) tg[i][k] += J[i][m] * Tn[q + itg * (m + DIM * k)];
      double ld;
      if (DIM == 2) {
        ld = sqrt(tg[0][0] * tg[0][0] + tg[1][0] * tg[1][0]);
      } else {
        const double r0 = tg[1][0] * tg[2][1] - tg[2][0] * tg[1][1];
        const double r1 = -tg[0][0] * tg[2][1] + tg[2][0] * tg[0][1];
        const double r2 = tg[0][0] * tg[1][1] - tg[1][0] * tg[0][1];
        ld = sqrt(r0 * r0 + r1 * r1 + r2 * r2);
      }
      wd[q] = V.wq[(int64_t)f * V.w_stride + q] * ld;
    }
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_wave_barrier();
  // ---- physical table T[q][a][s - S0]  (lane <-> (q, a))
  for (int i = lane; i < ((abl & 8) ? 0 : itg * itp); i += 64) {
    const int q = i % itg, a = i / itg;
    double* o = Tt + ((size_t)q * itp + a) * NS;
    if (S0 == 0) o[0] = R[q + itg * a];
    if (NS > 1 || S0 == 1) {
      double r[3];
#pragma unroll
      for (int m = 0; m < DIM; ++m) r[m] = R[q + itg * (a + itp * (1 + m))];
#pragma unroll
      for (int s = 0; s < DIM; ++s) {
        double v = 0.0;
#pragma unroll
        for (int m = 0; m < DIM; ++m) v += r[m] * Ji[q * DIM * DIM + m * DIM + s];
        o[(1 - S0) + s] = v;
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_wave_barrier();
  // ---- node pairs over the lanes
  const int npair = itp * itp;
  if (STAGE) {
    constexpr int TA = 4, TB = 2;
    const int ng = (itp + TA - 1) / TA, nbp = (itp + TB - 1) / TB;
    const bool pair16 = (itp & 1) == 0 && (((uintptr_t)K) & 15) == 0;  // (even row length: every (row, pair) of the scratch starts on a 16-byte boundary)
    for (int w_ = lane; w_ < ng * nbp; w_ += 64) {
      const int ag = w_ / nbp, bp = w_ - ag * nbp, a0 = ag * TA, b0 = bp * TB;
      double M[TA][TB][NM];
#pragma unroll
      for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j)
#pragma unroll
          for (int c = 0; c < NM; ++c) M[i][j][c] = 0.0;
      {
        const double* ta[TA];
        const double* tb[TB];
#pragma unroll
        for (int i = 0; i < TA; ++i) ta[i] = Tt + (a0 + i < itp ? a0 + i : itp - 1) * NS;  // (a group cut by the element: its last node again, not stored)
#pragma unroll
        for (int j = 0; j < TB; ++j) tb[j] = Tt + (b0 + j < itp ? b0 + j : itp - 1) * NS;
        const int qs = itp * NS;
        const int nq = (abl & 1) ? 0 : itg;
#pragma unroll 3
        for (int q = 0; q < nq; ++q) {
          const double wq_ = wd[q];
          double vb[TB][NS];
#pragma unroll
          for (int j = 0; j < TB; ++j)
#pragma unroll
            for (int s = 0; s < NS; ++s) vb[j][s] = tb[j][q * qs + s] * wq_;
#pragma unroll
          for (int i = 0; i < TA; ++i) {
            double va[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) va[s] = ta[i][q * qs + s];
#pragma unroll
            for (int j = 0; j < TB; ++j) {
              if (DIAGT) {
#pragma unroll
                for (int s = 0; s < NS; ++s) M[i][j][s] += va[s] * vb[j][s];
              } else {
#pragma unroll
                for (int s = 0; s < NS; ++s)
#pragma unroll
                  for (int u = 0; u < NS; ++u) M[i][j][s * NS + u] += va[s] * vb[j][u];
              }
            }
          }
        }
      }
      for (int k = 0; k < nb; ++k) {
        double ck[NM];
#pragma unroll
        for (int c = 0; c < NM; ++c) ck[c] = Cs[k * NM + c];
#pragma unroll
        for (int i = 0; i < TA; ++i) {
          double* dst = K + (((int64_t)el * itp + (a0 + i)) * nb + k) * itp + b0;
          double sum[TB];
#pragma unroll
          for (int j = 0; j < TB; ++j) {
            sum[j] = 0.0;
#pragma unroll
            for (int c = 0; c < NM; ++c) sum[j] += ck[c] * M[i][j][c];
          }
          if (a0 + i < itp && !(abl & 2)) {
            // the lane's two base nodes are neighbours in the scratch row: one 16-byte store where the row length is even (hex-20: every pair)
            if (TB == 2 && pair16 && b0 + 1 < itp) {
              typedef double ma_d2 __attribute__((ext_vector_type(2)));
              *reinterpret_cast<ma_d2*>(dst) = ma_d2{sum[0], sum[1]};
            } else {
#pragma unroll
              for (int j = 0; j < TB; ++j)
                if (b0 + j < itp) dst[j] = sum[j];
            }
          }
        }
      }
    }
    continue;
  }
  if (OUT == 2 && itp >= 16) {
    // Row-owner (scratch) form on elements of 16+ nodes (hex-20, hex-27): a lane takes FOUR dual nodes a for one base node b.  The pair loop is bound by
    // LDS bandwidth -- 7 reads per pair and Gauss point with a pair per lane (w, 3 + 3 table entries; hex-20: 677 KB per element) --; with the b side and
    // the weight shared by four pairs it is 4 reads per pair.  b runs fastest over the lanes: unit-stride stores into the element-major scratch.
    constexpr int TA = 4;
    const int ng = (itp + TA - 1) / TA;
    for (int w_ = lane; w_ < ng * itp; w_ += 64) {
      const int ag = w_ / itp, b = w_ - ag * itp, a0 = ag * TA;
      double M[TA][NS * NS];
#pragma unroll
      for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int s = 0; s < NS * NS; ++s) M[i][s] = 0.0;
      {
        const double* tb = Tt + b * NS;
        const double* ta[TA];
#pragma unroll
        for (int i = 0; i < TA; ++i) ta[i] = Tt + (a0 + i < itp ? a0 + i : itp - 1) * NS;  // (a group cut by the element: its last node again, not stored)
        const int qs = itp * NS;
        const int nq = (abl & 1) ? 0 : itg;
#pragma unroll 3
        for (int q = 0; q < nq; ++q) {
          const double wq_ = wd[q];
          double vb[NS];
#pragma unroll
          for (int s = 0; s < NS; ++s) vb[s] = tb[q * qs + s] * wq_;
#pragma unroll
          for (int i = 0; i < TA; ++i) {
            double va[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) va[s] = ta[i][q * qs + s];
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
              for (int u = 0; u < NS; ++u) M[i][s * NS + u] += va[s] * vb[u];
          }
        }
      }
      int i = 0, krun = 0;
      while (i < T.n) {  // runs of terms with the same sparse block: one value per run and pair
        const int block = T.block[i];
        double sum[TA];
#pragma unroll
        for (int t = 0; t < TA; ++t) sum[t] = 0.0;
        for (; i < T.n && T.block[i] == block; ++i) {
          const int sel = (T.ds[i] - S0) * NS + (T.bs[i] - S0);  // wave-uniform: a scalar branch, not a select chain
          const double cf = T.coef[i];
          switch (sel) {
#define MA_CASE(c) case c: if (c < NS * NS) { _Pragma("unroll") for (int t = 0; t < TA; ++t) sum[t] += cf * M[t][c < NS * NS ? c : 0]; } break;
            MA_CASE(0) MA_CASE(1) MA_CASE(2) MA_CASE(3) MA_CASE(4) MA_CASE(5) MA_CASE(6) MA_CASE(7)
            MA_CASE(8) MA_CASE(9) MA_CASE(10) MA_CASE(11) MA_CASE(12) MA_CASE(13) MA_CASE(14) MA_CASE(15)
#undef MA_CASE
            default: break;
          }
        }
#pragma unroll
        for (int t = 0; t < TA; ++t)
          if (a0 + t < itp && !(abl & 2)) K[(((int64_t)el * itp + (a0 + t)) * nb + krun) * itp + b] = sum[t];
        ++krun;
      }
    }
    continue;
  }
  for (int p = lane; p < npair; p += 64) {
    // slot table order: a fastest; scratch order: b fastest (the lanes' stores are unit-stride)
    const int a = OUT == 2 ? p / itp : p % itp, b = OUT == 2 ? p % itp : p / itp;
    double M[NS * NS];
#pragma unroll
    for (int s = 0; s < NS * NS; ++s) M[s] = 0.0;
    {
      const double* ta = Tt + a * NS;  // 32-bit LDS offsets; one pointer bump per Gauss point
      const double* tb = Tt + b * NS;
      const int qs = itp * NS;
#pragma unroll 3
      for (int q = 0; q < itg; ++q, ta += qs, tb += qs) {
        const double wq_ = wd[q];
        double va[NS], vb[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          va[s] = ta[s] * wq_;
          vb[s] = tb[s];
        }
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
          for (int u = 0; u < NS; ++u) M[s * NS + u] += va[s] * vb[u];
      }
    }
    if (TM.nruns > 0) {
      // (round 6) one dense coefficient row per sparse block instead of the term list: the list cost three dependent scalar loads and a branch per term and
      // pair group -- on tet-10 elasticity (21 terms) nearly all of the kernel's 7.3 ms
      for (int k = 0; k < TM.nruns; ++k) {
        double sum = 0.0;
#pragma unroll
        for (int c = 0; c < NS * NS; ++c) sum += TM.c[k][c] * M[c];
        if (OUT == 2) {
          K[(((int64_t)el * itp + a) * nb + k) * itp + b] = sum;
        } else {
          double* dst = K + ((int64_t)slots[TM.block[k] * block_stride + (int64_t)npair * el + p] - V.base);
          if (OUT == 1) atomicAdd(dst, sum); else *dst += sum;
        }
      }
      continue;
    }
    int i = 0, krun = 0;
    while (i < T.n) {  // runs of terms with the same sparse block: one accumulate per run
      const int block = T.block[i];
      double sum = 0.0;
      for (; i < T.n && T.block[i] == block; ++i) {
        const int sel = (T.ds[i] - S0) * NS + (T.bs[i] - S0);  // wave-uniform: a scalar branch, not a select chain
        double m = 0.0;
        switch (sel) {
#define MA_CASE(c) case c: if (c < NS * NS) m = M[c < NS * NS ? c : 0]; break;
          MA_CASE(0) MA_CASE(1) MA_CASE(2) MA_CASE(3) MA_CASE(4) MA_CASE(5) MA_CASE(6) MA_CASE(7)
          MA_CASE(8) MA_CASE(9) MA_CASE(10) MA_CASE(11) MA_CASE(12) MA_CASE(13) MA_CASE(14) MA_CASE(15)
#undef MA_CASE
          default: break;
        }
        sum += T.coef[i] * m;
      }
      if (OUT == 2) {
        K[(((int64_t)el * itp + a) * nb + krun) * itp + b] = sum;
      } else {
        double* dst = K + ((int64_t)slots[block * block_stride + (int64_t)npair * el + p] - V.base);
        if (OUT == 1) atomicAdd(dst, sum); else *dst += sum;
      }
      ++krun;
    }
  }
  }  // items of this wave
}

static int ma_launch(mfem_context_s* ctx, int dim, const MeshItems& V, const ConstTerms& T, const int32_t* slots,
                     int64_t block_stride, double* K, int64_t n_items, int n_colours, const int64_t* colour_offsets,
                     int scratch_blocks = 0) {
  int smin = 1 << 30, smax = -1;
  for (int i = 0; i < T.n; ++i) {
    smin = T.ds[i] < smin ? T.ds[i] : smin;
    smin = T.bs[i] < smin ? T.bs[i] : smin;
    smax = T.ds[i] > smax ? T.ds[i] : smax;
    smax = T.bs[i] > smax ? T.bs[i] : smax;
  }
  const int mode = smax == 0 ? 2 : smin >= 1 ? 1 : 0;  // values only | gradients only | everything
  const int NS = mode == 2 ? 1 : mode == 1 ? dim : 1 + dim;
  const size_t per_wave = sizeof(double) * ((size_t)V.itg * V.itp * NS + (size_t)V.itg * (1 + dim * dim) + (size_t)V.itp * dim);
  // the row-owner form on elements with a table worth staging (16+ nodes): persistent workgroups, the reference table in LDS (k_mesh_assemble: STAGE)
  const bool stage = scratch_blocks > 0 && !V.eindex && !V.order && V.itp >= g_mesh_stage_min_itp && n_colours == 0;
  TermMatrix TM;
  memset(&TM, 0, sizeof(TM));
  {
    int run = -1;
    bool fits = true;
    for (int i = 0; i < T.n; ++i) {
      if (i == 0 || T.block[i] != T.block[i - 1]) {
        ++run;
        if (run >= MA_MAX_RUNS) { fits = false; break; }
        TM.block[run] = T.block[i];
      }
      const int S0_ = mode == 1 ? 1 : 0, sel = (T.ds[i] - S0_) * NS + (T.bs[i] - S0_);
      if (sel < 0 || sel >= 16) { fits = false; break; }
      TM.c[run][sel] += T.coef[i];
    }
    TM.nruns = fits && g_mesh_term_matrix ? run + 1 : 0;  // (0: the kernel walks the term list)
  }
  bool diag = true;  // every term pairs a word with itself
  for (int i = 0; i < T.n; ++i) diag = diag && T.ds[i] == T.bs[i];
  const size_t shared_ref = stage ? sizeof(double) * ((size_t)V.itg * V.itp * (1 + dim) + (size_t)scratch_blocks * (diag ? NS : NS * NS)) : 0;
  int waves = 4;
  const size_t lds_cap = stage ? 96 * 1024 : 64 * 1024;  // (a workgroup may take up to 160 KB on gfx950; two staged workgroups per CU at hex-20)
  while (waves > 1 && shared_ref + per_wave * waves > lds_cap) waves >>= 1;
  MFEM_REQUIRE(shared_ref + per_wave * waves <= lds_cap, "element table too large for the fused mesh assembly (about 2 * itg * itp * (1 + dim) doubles must fit 64 KB)");
  const size_t ldsb = shared_ref + per_wave * waves;
  const bool atomic = n_colours == 0;
  const int out_mode = scratch_blocks > 0 ? 2 : atomic ? 1 : 0;
  const int nbatch = atomic ? 1 : n_colours;
  for (int c = 0; c < nbatch; ++c) {
    const int64_t a = atomic ? 0 : colour_offsets[c], b = atomic ? n_items : colour_offsets[c + 1];
    if (b <= a) continue;
    int grid = (int)((b - a + waves - 1) / waves);
    if (stage) {  // persistent: what is resident (LDS-bound: 160 KB per CU)
      const int per_cu = (int)(160 * 1024 / (ldsb > 0 ? ldsb : 1));
      const int cap = ctx->num_cus * (per_cu < 1 ? 1 : per_cu > 4 ? 4 : per_cu);
      if (grid > cap) grid = cap;
    }
#define MA_LAUNCH(D, S0, NSS, AT)                                                                                             \
  do {                                                                                                                        \
    if (stage && AT == 2 && diag) {                                                                                           \
      if (ldsb > 64 * 1024)                                                                                                   \
        MFEM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mesh_assemble<D, S0, NSS, 2, true, true>),       \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));                         \
      hipLaunchKernelGGL((k_mesh_assemble<D, S0, NSS, 2, true, true>), dim3(grid), dim3(64 * waves), ldsb, ctx->stream, V, T, \
                         slots, block_stride, K, a, b, scratch_blocks, g_mesh_abl, TM);                                          \
    } else if (stage && AT == 2) {                                                                                            \
      if (ldsb > 64 * 1024)                                                                                                   \
        MFEM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mesh_assemble<D, S0, NSS, 2, true>),             \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));                         \
      hipLaunchKernelGGL((k_mesh_assemble<D, S0, NSS, 2, true>), dim3(grid), dim3(64 * waves), ldsb, ctx->stream, V, T, slots, \
                         block_stride, K, a, b, scratch_blocks, g_mesh_abl, TM);                                                             \
    } else                                                                                                                    \
      hipLaunchKernelGGL((k_mesh_assemble<D, S0, NSS, AT>), dim3(grid), dim3(64 * waves), ldsb, ctx->stream, V, T, slots,     \
                         block_stride, K, a, b, scratch_blocks, g_mesh_abl, TM);                                                             \
  } while (0)
#define MA_MODE(D, AT)                                  \
  do {                                                  \
    if (mode == 2) MA_LAUNCH(D, 0, 1, AT);              \
    else if (mode == 1) MA_LAUNCH(D, 1, D, AT);         \
    else MA_LAUNCH(D, 0, 1 + D, AT);                    \
  } while (0)
#define MA_OUT(D)                                  \
  do {                                             \
    if (out_mode == 2) MA_MODE(D, 2);              \
    else if (out_mode == 1) MA_MODE(D, 1);         \
    else MA_MODE(D, 0);                            \
  } while (0)
    if (dim == 2) MA_OUT(2); else MA_OUT(3);
#undef MA_OUT
#undef MA_MODE
#undef MA_LAUNCH
    MFEM_CHECK_LAUNCH();
  }
  return MFEM_OK;
}

static int ma_terms(int32_t n_terms, const mfem_const_term* terms, int dim, ConstTerms* out) {
  MFEM_REQUIRE(n_terms > 0 && n_terms <= MA_MAX_TERMS && terms, "n_terms must be 1..48");
  out->n = n_terms;
  for (int i = 0; i < n_terms; ++i) {
    MFEM_REQUIRE(terms[i].dual_sd >= 0 && terms[i].dual_sd <= dim && terms[i].base_sd >= 0 && terms[i].base_sd <= dim,
                 "term words: 0 = value, 1 + j = d/dx_j");
    MFEM_REQUIRE(terms[i].block >= 0 && (i == 0 || terms[i].block >= terms[i - 1].block), "terms must be sorted by block");
    out->ds[i] = terms[i].dual_sd;
    out->bs[i] = terms[i].base_sd;
    out->block[i] = terms[i].block;
    out->coef[i] = terms[i].coef;
  }
  return MFEM_OK;
}

extern "C" int mfem_mesh_assemble_elements(mfem_context ctx, int32_t dim, int32_t itg, int32_t itp, int64_t nel, int64_t ncp,
                                           const double* ref_itp_vals, const double* itg_weight, const double* coords,
                                           const int32_t* controlpoint_IDs, int32_t index_base, int32_t n_terms,
                                           const mfem_const_term* terms, const int32_t* sparse_IDs_by_el,
                                           int64_t slot_block_stride, double* K_val, const int32_t* elIDs, int64_t n_items,
                                           int32_t n_colours, const int64_t* colour_offsets) try {
  MFEM_REQUIRE(ctx, "null ctx");
  MFEM_REQUIRE(dim == 2 || dim == 3, "dim must be 2 or 3");
  MFEM_REQUIRE(itg > 0 && itp > 0 && nel >= 0 && ncp > 0 && n_items >= 0 && n_items <= nel, "bad sizes");
  MFEM_REQUIRE(index_base == 0 || index_base == 1, "index_base must be 0 or 1");
  MFEM_REQUIRE(n_colours >= 0 && (n_colours == 0 || colour_offsets), "colour_offsets missing");
  if (n_items == 0) return MFEM_OK;
  MFEM_REQUIRE(ref_itp_vals && itg_weight && coords && controlpoint_IDs && sparse_IDs_by_el && K_val, "null array");
  MFEM_REQUIRE(n_colours == 0 || (colour_offsets[0] == 0 && colour_offsets[n_colours] == n_items), "colour_offsets must span the items");
  ConstTerms T;
  int rc = ma_terms(n_terms, terms, dim, &T);
  if (rc) return rc;
  MeshItems V{itg, itp, ncp, ref_itp_vals, 0, itg_weight, 0, nullptr, 0, coords, controlpoint_IDs, nullptr, nullptr, elIDs, index_base};
  return ma_launch(ctx, dim, V, T, sparse_IDs_by_el, slot_block_stride, K_val, n_items, n_colours, colour_offsets);
} MFEM_API_CATCH("mfem_mesh_assemble_elements")

// ---- row-owner form of the scatter -----------------------------------------------------------------------------------
// Scattering an element matrix entry by entry is what the assembly of a large mesh spends its time on (hex-20 elasticity:
// 3600 read-modify-writes of 8 bytes per element, each its own 64-byte sector).  Row-owner form: the element matrices go to
// the element-major scratch (unit-stride stores), then a wave owns one CSR row (dual field fd, node i): it stages the row's
// column list in LDS, walks the node's adjacency list (element, local id a) in ascending element order and, for every block
// (fd, fb), adds the itp contiguous scratch entries  S[el][a][k][0 .. itp)  at the positions of the columns
// fb * ncp + node(el, b) (binary search in the staged list) -- distinct positions within a step, steps in sequence: no
// atomics, a fixed summation order (bitwise reproducible), every K entry read and written once, contiguously.
#define MG_MAXROW 2048
static std::atomic<long long> g_mesh_rows_count{0};
int g_mesh_gather_rows = 0;  // mfem_debug_set("mesh_gather_rows"): 1 = the gather by row of round 5 (A/B, tests)  // assemblies that took the row-owner form (tests, bench.py)
extern "C" int64_t mfem_debug_mesh_rows_count(void) { return g_mesh_rows_count; }
// element-matrix scratch the row-owner form may take from the context workspace (288 GB of HBM: hex-20 elasticity at 128^3 needs 60 GB)
static const size_t MG_SCRATCH_BUDGET = (size_t)96 << 30;
struct GatherBlocks {
  int nf;          // fields
  int nb;          // blocks in the scratch (runs of the term list)
  int cnt[4];      // blocks with dual field fd
  int k[4][4];     // their scratch index
  int fb[4][4];    // their base field
};

// ranks[(j * itp) + b] = position of node(el, b) among the control points coupled to node i (ascending ids = the order of
// the columns inside every field segment of a row of node i), for adjacency entry j = (i <- el, a).  Once per pattern.
template <typename RP>
__global__ __launch_bounds__(MFEM_BLOCK) void k_mesh_row_ranks(int itp, int64_t ncp, int nf, const RP* __restrict__ rowptr,
                                                                 const int32_t* __restrict__ colidx, int cbase,
                                                                 const int64_t* __restrict__ adj_ptr, const int32_t* __restrict__ adj,
                                                                 const int32_t* __restrict__ cp, int base, uint16_t* __restrict__ ranks,
                                                                 int32_t* __restrict__ repeated) {
  const int64_t total = adj_ptr[ncp] * itp;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t j = t / itp;
    const int b = (int)(t - j * itp);
    const int32_t ea = adj[j];
    const int64_t el = ea / itp;
    const int a = ea - (int)el * itp;
    const int64_t node = (int64_t)cp[el * itp + a] - base;
    const int64_t lo = (int64_t)rowptr[node] - cbase;  // row (field 0, node): its first len / nf columns are field 0's
    const int L = (int)(((int64_t)rowptr[node + 1] - cbase - lo) / nf);
    const int32_t col = (int32_t)((int64_t)cp[el * itp + b] - base);
    int l = 0, h = L - 1;
    while (l < h) {
      const int mid = (l + h) >> 1;
      if (colidx[lo + mid] - cbase < col) l = mid + 1; else h = mid;
    }
    ranks[t] = (uint16_t)l;
    // an element that lists one control point twice (collapsed / degenerate elements) would make two lanes of the gather add at
    // the same position of a row: found here once per pattern (entries with a == 0 visit every element exactly once)
    if (a == 0) {
      bool rep = false;
      for (int b2 = b + 1; b2 < itp; ++b2) rep |= (int32_t)((int64_t)cp[el * itp + b2] - base) == col;
      if (rep) *repeated = 1;
    }
  }
}

extern "C" int mfem_mesh_row_ranks(mfem_context ctx, int32_t itp, int64_t nel, int64_t ncp, int32_t n_fields, mfem_csr A,
                                   const int64_t* adj_ptr, const int32_t* adj, const int32_t* controlpoint_IDs,
                                   int32_t index_base, uint16_t* ranks) try {
  MFEM_REQUIRE(ctx && A && adj_ptr && adj && controlpoint_IDs && ranks, "null argument");
  MFEM_REQUIRE(itp > 0 && nel >= 0 && ncp > 0 && n_fields >= 1 && n_fields <= 4, "bad sizes");
  MFEM_REQUIRE(A->n == (int64_t)n_fields * ncp, "pattern rows != n_fields * ncp");
  MFEM_REQUIRE(A->max_row_nnz / n_fields < 65536, "more than 65535 coupled control points per row");
  if (nel == 0) return MFEM_OK;
  int32_t* d_rep = ctx->d_flags + 11;
  MFEM_CHECK_HIP(hipMemsetAsync(d_rep, 0, sizeof(int32_t), ctx->stream));
  const int grid = mfem_grid_for(nel * itp * (int64_t)itp, MFEM_BLOCK, ctx->num_cus * 16);
  if (A->rowptr_bits == 64)
    hipLaunchKernelGGL(k_mesh_row_ranks<int64_t>, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, itp, ncp, n_fields,
                       (const int64_t*)A->rowptr, A->colidx, A->index_base, adj_ptr, adj, controlpoint_IDs, index_base, ranks, d_rep);
  else
    hipLaunchKernelGGL(k_mesh_row_ranks<int32_t>, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, itp, ncp, n_fields,
                       (const int32_t*)A->rowptr, A->colidx, A->index_base, adj_ptr, adj, controlpoint_IDs, index_base, ranks, d_rep);
  MFEM_CHECK_LAUNCH();
  MFEM_CHECK_HIP(hipMemcpyAsync(ctx->h_flags + 11, d_rep, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  MFEM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  if (ctx->h_flags[11]) {
    mfem_set_error("an element lists the same control point twice: the row-owner gather adds at distinct positions per element -- use "
                   "mfem_mesh_assemble_elements (scatter through the slot table) for this mesh");
    return MFEM_ERR_UNSUPPORTED;
  }
  return MFEM_OK;
} MFEM_API_CATCH("mfem_mesh_row_ranks")

template <typename RP>
__global__ __launch_bounds__(MFEM_BLOCK) void k_mesh_gather(int itp, int64_t ncp, GatherBlocks B, const RP* __restrict__ rowptr,
                                                              int cbase, const int64_t* __restrict__ adj_ptr,
                                                              const int32_t* __restrict__ adj, const uint16_t* __restrict__ ranks,
                                                              const double* __restrict__ S, double* __restrict__ K, int maxrow, int set) {
  extern __shared__ double gl[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  double* vals = gl + (size_t)w * maxrow;
  const int64_t nrows = (int64_t)B.nf * ncp;
  for (int64_t r = (int64_t)blockIdx.x * nw + w; r < nrows; r += (int64_t)gridDim.x * nw) {
    const int fd = (int)(r / ncp);
    const int64_t node = r - (int64_t)fd * ncp;
    const int64_t lo = (int64_t)rowptr[r] - cbase;
    const int len = (int)((int64_t)rowptr[r + 1] - cbase - lo);
    const int L = len / B.nf;  // columns per field segment
    for (int t = lane; t < len; t += 64) vals[t] = 0.0;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    const int nblk = B.cnt[fd];
    const int work = nblk * itp;  // (block, b) pairs of one adjacency entry
    // the scratch runs and ranks of up to 8 adjacency entries are requested before the first is used: a row waits for
    // memory once per 8 elements, not once per element (a corner node of a hex mesh has 8)
    const int64_t j0 = adj_ptr[node], j1 = adj_ptr[node + 1];
    for (int64_t jb = j0; jb < j1; jb += 8) {
      const int nj = (int)(j1 - jb < 8 ? j1 - jb : 8);
      for (int u0 = 0; u0 < work; u0 += 64) {
        const int u = u0 + lane;
        const bool on = u < work;
        const int kk = on ? u / itp : 0, b = on ? u - kk * itp : 0;
        const int seg = B.fb[fd][kk] * L;
        double v[8];
        int pos[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          v[t] = 0.0;
          pos[t] = 0;
          if (t < nj && on) {
            const int32_t ea = adj[jb + t];
            const int64_t el = ea / itp;
            const int a = ea - (int)el * itp;
            v[t] = S[(((int64_t)el * itp + a) * B.nb + B.k[fd][kk]) * itp + b];
            pos[t] = seg + ranks[(jb + t) * itp + b];
          }
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          if (t < nj) {  // wave-uniform; within a step the lanes hit distinct positions, the steps run in sequence
            if (on) vals[pos[t]] += v[t];
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
          }
        }
      }
    }
    if (set) for (int t = lane; t < len; t += 64) K[lo + t] = vals[t];
    else for (int t = lane; t < len; t += 64) K[lo + t] += vals[t];
    __builtin_amdgcn_wave_barrier();
  }
}

// Round 6: the gather by NODE for blocks x nodes <= 64 (hex-20 with up to three fields, hex-8 with up to four).  The rows of a node's NF fields share the
// adjacency list and the ranks: one pointer chase (adj_ptr -> adj -> scratch) per node instead of one per row, all NF x 8 scratch loads of a lane in flight
// together; and a wave takes G = 64 / (blocks x nodes) nodes side by side (three for one field on hex-20, where a row's 20 lanes left 44 idle).  The
// additions into a row keep the order of k_mesh_gather (adjacency entries in sequence): the same bits.
#ifndef MGN_TB
#define MGN_TB 4
#endif
template <typename RP, int NF>
__global__ __launch_bounds__(MFEM_BLOCK) void k_mesh_gather_nodes(int itp, int64_t ncp, GatherBlocks B, const RP* __restrict__ rowptr,
                                                                    int cbase, const int64_t* __restrict__ adj_ptr,
                                                                    const int32_t* __restrict__ adj, const uint16_t* __restrict__ ranks,
                                                                    const double* __restrict__ S, double* __restrict__ K, int maxrow, int work, int G, int set) {
  extern __shared__ double gl[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  double* wave_vals = gl + (size_t)w * G * NF * maxrow;  // [G][NF][maxrow]
  for (int t = lane; t < G * NF * maxrow; t += 64) wave_vals[t] = 0.0;
  const int g = lane / work, u = lane - g * work;
  const int kk = u / itp, b = u - kk * itp;
  double* vals = wave_vals + (size_t)(g < G ? g : 0) * NF * maxrow;
  bool lane_on[NF];
  int kidx[NF], fbase[NF];
#pragma unroll
  for (int fd = 0; fd < NF; ++fd) {
    lane_on[fd] = g < G && kk < B.cnt[fd];
    kidx[fd] = lane_on[fd] ? B.k[fd][kk] : 0;
    fbase[fd] = lane_on[fd] ? B.fb[fd][kk] : 0;
  }
  // adjacency entries per batch: NF x TB scratch loads of a lane in flight (registers -> resident waves: this kernel lives on the number of pointer chases in flight)
  constexpr int TB = MGN_TB;
  __builtin_amdgcn_wave_barrier();
  for (int64_t node0 = ((int64_t)blockIdx.x * nw + w) * G; node0 < ncp; node0 += (int64_t)gridDim.x * nw * G) {
    const int64_t node = node0 + g;
    const bool live = g < G && node < ncp;
    int64_t lo[NF];
    int len[NF];
#pragma unroll
    for (int fd = 0; fd < NF; ++fd) {
      lo[fd] = live ? (int64_t)rowptr[(int64_t)fd * ncp + node] - cbase : 0;
      len[fd] = live ? (int)((int64_t)rowptr[(int64_t)fd * ncp + node + 1] - cbase - lo[fd]) : 0;
    }
    const int64_t j0 = live ? adj_ptr[node] : 0, j1 = live ? adj_ptr[node + 1] : 0;
    for (int64_t jb = j0; __any(jb < j1); jb += TB) {
      const int nj = (int)(j1 - jb < 0 ? 0 : j1 - jb < TB ? j1 - jb : TB);
      double v[NF][TB];
      int rk[TB];
#pragma unroll
      for (int t = 0; t < TB; ++t) {
        rk[t] = 0;
#pragma unroll
        for (int fd = 0; fd < NF; ++fd) v[fd][t] = 0.0;
        if (t < nj) {
          const int32_t ea = adj[jb + t];
          const int64_t el = ea / itp;
          const int a = ea - (int)el * itp;
          const int64_t srow = ((int64_t)el * itp + a) * B.nb;
#pragma unroll
          for (int fd = 0; fd < NF; ++fd)
            if (lane_on[fd]) v[fd][t] = S[(srow + kidx[fd]) * itp + b];
          rk[t] = ranks[(jb + t) * itp + b];
        }
      }
#pragma unroll
      for (int t = 0; t < TB; ++t) {
        if (t < nj) {  // within a step the lanes of a node hit distinct positions, the steps run in sequence
#pragma unroll
          for (int fd = 0; fd < NF; ++fd)
            if (lane_on[fd]) vals[fd * maxrow + fbase[fd] * (len[fd] / NF) + rk[t]] += v[fd][t];
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    __builtin_amdgcn_wave_barrier();
    // the rows leave as whole waves
    for (int gg = 0; gg < G; ++gg) {
      if (node0 + gg >= ncp) break;
#pragma unroll
      for (int fd = 0; fd < NF; ++fd) {
        const int64_t rlo = __shfl(lo[fd], gg * work);
        const int rlen = __shfl(len[fd], gg * work);
        double* rv = wave_vals + ((size_t)gg * NF + fd) * maxrow;
        if (set) {  // (the rows are the first thing K receives: nothing to read)
          for (int t = lane; t < rlen; t += 64) {
            K[rlo + t] = rv[t];
            rv[t] = 0.0;
          }
        } else {
          for (int t = lane; t < rlen; t += 64) {
            K[rlo + t] += rv[t];
            rv[t] = 0.0;
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

static int mesh_rows(mfem_context_s* ctx, int32_t dim, int32_t itg, int32_t itp, int64_t nel, int64_t ncp, const double* ref_itp_vals,
                     const double* itg_weight, const double* coords, const int32_t* controlpoint_IDs, int32_t index_base, int32_t n_terms,
                     const mfem_const_term* terms, int32_t n_fields, mfem_csr_s* A, const int64_t* adj_ptr, const int32_t* adj,
                     const uint16_t* ranks, double* K_val, int overwrite) {
  MFEM_REQUIRE(ctx && A, "null handle");
  MFEM_REQUIRE(dim == 2 || dim == 3, "dim must be 2 or 3");
  MFEM_REQUIRE(itg > 0 && itp > 0 && nel >= 0 && ncp > 0, "bad sizes");
  MFEM_REQUIRE(n_fields >= 1 && n_fields <= 4, "1..4 fields");
  MFEM_REQUIRE(index_base == 0 || index_base == 1, "index_base must be 0 or 1");
  MFEM_REQUIRE(A->n == (int64_t)n_fields * ncp, "pattern rows != n_fields * ncp");
  if (nel == 0) return MFEM_OK;
  MFEM_REQUIRE(ref_itp_vals && itg_weight && coords && controlpoint_IDs && adj_ptr && adj && ranks && K_val, "null array");
  if (A->max_row_nnz > MG_MAXROW) {
    mfem_set_error("rows of up to %d entries: the row-owner assembly stages a row in LDS (<= %d); use mfem_mesh_assemble_elements", A->max_row_nnz, MG_MAXROW);
    return MFEM_ERR_UNSUPPORTED;
  }
  ConstTerms T;
  int rc = ma_terms(n_terms, terms, dim, &T);
  if (rc) return rc;
  GatherBlocks B;
  memset(&B, 0, sizeof(B));
  B.nf = n_fields;
  for (int i = 0; i < T.n; ++i) {
    if (i > 0 && T.block[i] == T.block[i - 1]) continue;
    const int fd = T.block[i] / n_fields, fb = T.block[i] % n_fields;
    MFEM_REQUIRE(fd < n_fields, "block out of range");
    B.k[fd][B.cnt[fd]] = B.nb;
    B.fb[fd][B.cnt[fd]] = fb;
    ++B.cnt[fd];
    ++B.nb;
  }
  const size_t bytes = sizeof(double) * (size_t)nel * itp * B.nb * itp;
  if (bytes > MG_SCRATCH_BUDGET) {
    mfem_set_error("element-matrix scratch of %zu bytes exceeds the 96 GiB budget; use mfem_mesh_assemble_elements", bytes);
    return MFEM_ERR_UNSUPPORTED;
  }
  rc = mfem_ws_reserve(ctx, bytes);
  if (rc) return rc;
  double* S = (double*)ctx->ws;
  MeshItems V{itg, itp, ncp, ref_itp_vals, 0, itg_weight, 0, nullptr, 0, coords, controlpoint_IDs, nullptr, nullptr, nullptr, index_base};
  rc = ma_launch(ctx, dim, V, T, nullptr, 0, S, nel, 0, nullptr, B.nb);
  if (rc) return rc;
  const int64_t nrows = (int64_t)n_fields * ncp;
  const int maxrow = (A->max_row_nnz + 15) & ~15;
  const int waves = 4;
  int work = 0;
  for (int fd = 0; fd < n_fields; ++fd) work = B.cnt[fd] * itp > work ? B.cnt[fd] * itp : work;
  int G = work > 0 && work <= 64 ? 64 / work : 0;
  auto lds_nodes = [&](int g) { return sizeof(double) * (size_t)maxrow * waves * g * n_fields; };
  while (G > 1 && lds_nodes(G) > 32 * 1024) --G;  // (long rows: fewer nodes side by side)
  if (G > 0 && lds_nodes(G) <= 64 * 1024 && !g_mesh_gather_rows) {  // by node (k_mesh_gather_nodes)
    const size_t ldsn = lds_nodes(G);
    const int gridn = mfem_grid_for((ncp + G - 1) / G, waves, ctx->num_cus * 16);
#define MG_NODES(RP, NF)                                                                                                                    \
  hipLaunchKernelGGL((k_mesh_gather_nodes<RP, NF>), dim3(gridn), dim3(64 * waves), ldsn, ctx->stream, itp, ncp, B, (const RP*)A->rowptr, \
                     A->index_base, adj_ptr, adj, ranks, S, K_val, maxrow, work, G, overwrite)
#define MG_FIELDS(RP)                                \
  do {                                               \
    if (n_fields == 1) MG_NODES(RP, 1);              \
    else if (n_fields == 2) MG_NODES(RP, 2);         \
    else if (n_fields == 3) MG_NODES(RP, 3);         \
    else MG_NODES(RP, 4);                            \
  } while (0)
    if (A->rowptr_bits == 64) MG_FIELDS(int64_t); else MG_FIELDS(int32_t);
#undef MG_FIELDS
#undef MG_NODES
    MFEM_CHECK_LAUNCH();
    ++g_mesh_rows_count;
    return MFEM_OK;
  }
  const size_t ldsb = sizeof(double) * (size_t)maxrow * waves;
  const int grid = mfem_grid_for(nrows, waves, ctx->num_cus * 32);
  if (A->rowptr_bits == 64)
    hipLaunchKernelGGL(k_mesh_gather<int64_t>, dim3(grid), dim3(64 * waves), ldsb, ctx->stream, itp, ncp, B, (const int64_t*)A->rowptr,
                       A->index_base, adj_ptr, adj, ranks, S, K_val, maxrow, overwrite);
  else
    hipLaunchKernelGGL(k_mesh_gather<int32_t>, dim3(grid), dim3(64 * waves), ldsb, ctx->stream, itp, ncp, B, (const int32_t*)A->rowptr,
                       A->index_base, adj_ptr, adj, ranks, S, K_val, maxrow, overwrite);
  MFEM_CHECK_LAUNCH();
  ++g_mesh_rows_count;
  return MFEM_OK;
}

extern "C" int mfem_mesh_assemble_elements_rows(mfem_context ctx, int32_t dim, int32_t itg, int32_t itp, int64_t nel, int64_t ncp,
                                                const double* ref_itp_vals, const double* itg_weight, const double* coords,
                                                const int32_t* controlpoint_IDs, int32_t index_base, int32_t n_terms,
                                                const mfem_const_term* terms, int32_t n_fields, mfem_csr A,
                                                const int64_t* adj_ptr, const int32_t* adj, const uint16_t* ranks,
                                                double* K_val) try {
  return mesh_rows(ctx, dim, itg, itp, nel, ncp, ref_itp_vals, itg_weight, coords, controlpoint_IDs, index_base, n_terms, terms, n_fields, A, adj_ptr,
                   adj, ranks, K_val, 0);
} MFEM_API_CATCH("mfem_mesh_assemble_elements_rows")

extern "C" int mfem_mesh_assemble_elements_rows_set(mfem_context ctx, int32_t dim, int32_t itg, int32_t itp, int64_t nel, int64_t ncp,
                                                    const double* ref_itp_vals, const double* itg_weight, const double* coords,
                                                    const int32_t* controlpoint_IDs, int32_t index_base, int32_t n_terms,
                                                    const mfem_const_term* terms, int32_t n_fields, mfem_csr A,
                                                    const int64_t* adj_ptr, const int32_t* adj, const uint16_t* ranks,
                                                    double* K_val) try {
  MFEM_REQUIRE(nel > 0, "the overwriting form needs elements (an empty mesh leaves K_val untouched)");
  return mesh_rows(ctx, dim, itg, itp, nel, ncp, ref_itp_vals, itg_weight, coords, controlpoint_IDs, index_base, n_terms, terms, n_fields, A, adj_ptr,
                   adj, ranks, K_val, 1);
} MFEM_API_CATCH("mfem_mesh_assemble_elements_rows_set")

extern "C" int mfem_mesh_assemble_facets(mfem_context ctx, int32_t dim, int32_t itg_b, int32_t itp, int32_t n_face_ids,
                                         int64_t n_facets, int64_t ncp, const double* bdy_ref_itp_vals,
                                         const double* bdy_itg_weights, const double* bdy_tangent_directions,
                                         const double* coords, const int32_t* controlpoint_IDs, const int32_t* element_ID,
                                         const int32_t* element_eindex, int32_t index_base, int32_t n_terms,
                                         const mfem_const_term* terms, const int32_t* sparse_IDs_by_el,
                                         int64_t slot_block_stride, double* K_val, const int32_t* facetIDs, int64_t n_items,
                                         int32_t n_colours, const int64_t* colour_offsets) try {
  MFEM_REQUIRE(ctx, "null ctx");
  MFEM_REQUIRE(dim == 2 || dim == 3, "dim must be 2 or 3");
  MFEM_REQUIRE(itg_b > 0 && itp > 0 && n_face_ids > 0 && n_facets >= 0 && ncp > 0 && n_items >= 0 && n_items <= n_facets, "bad sizes");
  MFEM_REQUIRE(index_base == 0 || index_base == 1, "index_base must be 0 or 1");
  MFEM_REQUIRE(n_colours >= 0 && (n_colours == 0 || colour_offsets), "colour_offsets missing");
  if (n_items == 0) return MFEM_OK;
  MFEM_REQUIRE(bdy_ref_itp_vals && bdy_itg_weights && bdy_tangent_directions && coords && controlpoint_IDs && element_ID &&
                   element_eindex && sparse_IDs_by_el && K_val, "null array");
  MFEM_REQUIRE(n_colours == 0 || (colour_offsets[0] == 0 && colour_offsets[n_colours] == n_items), "colour_offsets must span the items");
  ConstTerms T;
  int rc = ma_terms(n_terms, terms, dim, &T);
  if (rc) return rc;
  const int64_t rs = (int64_t)itg_b * itp * (1 + dim), ts = (int64_t)itg_b * dim * (dim - 1);
  MeshItems V{itg_b, itp, ncp, bdy_ref_itp_vals, rs, bdy_itg_weights, (int64_t)itg_b, bdy_tangent_directions, ts, coords,
              controlpoint_IDs, element_ID, element_eindex, facetIDs, index_base};
  return ma_launch(ctx, dim, V, T, sparse_IDs_by_el, slot_block_stride, K_val, n_items, n_colours, colour_offsets);
} MFEM_API_CATCH("mfem_mesh_assemble_facets")
