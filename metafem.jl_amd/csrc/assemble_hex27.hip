// hex-27 (Lagrange order 2) thermal assembly: the one place in this backend where the work is a dense GEMM,
//   Ke = B^T D B,   B[(q,s), a] = dN_a/dx_s (q)  (81 x 27),   D = diag(-k w_q det J_q),
// so it runs on the FP64 matrix cores (v_mfma_f64_16x16x4_f64), one wave per element:
//   * the element's 27 nodal coordinates and its row descriptors are staged in per-wave LDS;
//   * the wave builds J at all quadrature points (one (q,i) row per lane), inverts it (adjugate, inv_Jac_3D),
//     and forms each MFMA fragment of B = dN/dxi * J^-1 on the fly from the reference table in LDS (3 FMAs per
//     fragment element) -- B itself is never materialised, which keeps the per-wave LDS footprint at ~9 KB
//     and lets 16 waves share a CU;
//   * Ke (27 x 27 padded to 32 x 32) = 2 x 2 accumulator tiles; the lower-left tile is the transpose of the
//     upper-right one and is not computed: 3 tiles x 21 k-steps = 63 MFMAs per element;
//   * scatter into the global CSR is race-free WITHOUT atomics through the parity colouring of the structured
//     element grid (8 colours: same-colour elements share no control point), one launch per colour -- the
//     colour-partitioned ordering BASELINE.json's north_star asks for.
// Replaces for this element type: update_BasicElements_3D (4_Update_Integrator.jl:2-33,90-154) + the three
// _Kval_Basic launches of the thermal form (06_FEM_Kernel.jl:28-45) + their 46.7 KB/element basis tables (F7).
// The matrix-free residual and the Robin faces use the same wave-per-element / thread-per-face structure with
// plain FP64 FMAs (matrix-vector work, not GEMM).
#include "brick.h"

typedef double d4_t __attribute__((ext_vector_type(4)));

#define H27_MAXQ 64

struct Hex27Tables {        // device-global, filled once per ng
  double dN[H27_MAXQ][3][27];  // [q][m][a]: lanes that differ in a (or in (q, m)) read different LDS banks
  double N[H27_MAXQ][27];
  double w[H27_MAXQ];
  double tab1[2][4][4];       // 1-D Lagrange-2 values (k = 0) and derivatives (k = 1) at the ng Gauss points: [k][q][a], rows padded to 4
  double T1[4][3][4];         // affine elements: 1-D integrals over the ng Gauss points, [X][a][b] (rows padded to 4): X = 0: sum w l'_a l'_b, 1: sum w l_a l_b,
                              // 2: sum w l'_a l_b, 3: sum w l_a l'_b -- the reference integrals of Ke are products of three of them (tensor-product basis and quadrature)
  // face tables: 2-D Lagrange-2 on [0,1]^2 at ng x ng Gauss points, c = c1 + 3*c2
  double fN[16][9];
  double fdN[16][9][2];
  double fw[16];
};
static Hex27Tables* g_tab = nullptr;
static std::atomic<int> g_tab_ng{0};

static const double GP27[4][4] = {{0.0, 0, 0, 0},
                                  {-0.57735026918962576451, 0.57735026918962576451, 0, 0},
                                  {-0.77459666924148337704, 0.0, 0.77459666924148337704, 0},
                                  {-0.86113631159405257522, -0.33998104358485626480, 0.33998104358485626480, 0.86113631159405257522}};
static const double GW27[4][4] = {{2.0, 0, 0, 0},
                                  {1.0, 1.0, 0, 0},
                                  {5.0 / 9.0, 8.0 / 9.0, 5.0 / 9.0, 0},
                                  {0.34785484513745385737, 0.65214515486254614263, 0.65214515486254614263, 0.34785484513745385737}};

static void lag2(double x, double* L, double* dL) {  // nodes 0, 1/2, 1 (102_Interpolations.jl:3-23)
  L[0] = 2.0 * (x - 0.5) * (x - 1.0);
  L[1] = -4.0 * x * (x - 1.0);
  L[2] = 2.0 * x * (x - 0.5);
  dL[0] = 4.0 * x - 3.0;
  dL[1] = -8.0 * x + 4.0;
  dL[2] = 4.0 * x - 1.0;
}

static int hex27_upload_tables(int ng) {
  static std::mutex mu;  // uploads from two host threads must not interleave (the tables themselves are process-wide: see the threading note in include/metafem_mi355x.h)
  std::lock_guard<std::mutex> lk(mu);
  if (g_tab && g_tab_ng == ng) return MFEM_OK;
  mfem_host_alloc_probe();
  Hex27Tables* h = new Hex27Tables();
  memset(h, 0, sizeof(*h));
  double gp[4], gw[4];
  for (int i = 0; i < ng; ++i) {
    gp[i] = GP27[ng - 1][i] / 2.0 + 0.5;
    gw[i] = GW27[ng - 1][i] / 2.0;
  }
  for (int q1 = 0; q1 < ng; ++q1) lag2(gp[q1], h->tab1[0][q1], h->tab1[1][q1]);
  for (int qz = 0; qz < ng; ++qz)
    for (int qy = 0; qy < ng; ++qy)
      for (int qx = 0; qx < ng; ++qx) {
        const int q = qx + ng * (qy + ng * qz);
        double L[3][3], dL[3][3];
        lag2(gp[qx], L[0], dL[0]);
        lag2(gp[qy], L[1], dL[1]);
        lag2(gp[qz], L[2], dL[2]);
        h->w[q] = gw[qx] * gw[qy] * gw[qz];
        for (int a = 0; a < 27; ++a) {
          const int ax = a % 3, ay = (a / 3) % 3, az = a / 9;
          h->N[q][a] = L[0][ax] * L[1][ay] * L[2][az];
          h->dN[q][0][a] = dL[0][ax] * L[1][ay] * L[2][az];
          h->dN[q][1][a] = L[0][ax] * dL[1][ay] * L[2][az];
          h->dN[q][2][a] = L[0][ax] * L[1][ay] * dL[2][az];
        }
      }
  for (int q2 = 0; q2 < ng; ++q2)
    for (int q1 = 0; q1 < ng; ++q1) {
      const int q = q1 + ng * q2;
      double L1[3], d1[3], L2[3], d2[3];
      lag2(gp[q1], L1, d1);
      lag2(gp[q2], L2, d2);
      h->fw[q] = gw[q1] * gw[q2];
      for (int c = 0; c < 9; ++c) {
        const int c1 = c % 3, c2 = c / 3;
        h->fN[q][c] = L1[c1] * L2[c2];
        h->fdN[q][c][0] = d1[c1] * L2[c2];
        h->fdN[q][c][1] = L1[c1] * d2[c2];
      }
    }
  for (int a = 0; a < 3; ++a)  // 1-D integrals of the affine-element path, with THIS quadrature (what the general path sums on an element with a constant Jacobian)
    for (int b2 = 0; b2 < 3; ++b2) {
      double dd = 0.0, mm = 0.0, cc = 0.0, ct = 0.0;
      for (int q = 0; q < ng; ++q) {
        const double* L = h->tab1[0][q];
        const double* dL = h->tab1[1][q];
        dd += gw[q] * dL[a] * dL[b2];
        mm += gw[q] * L[a] * L[b2];
        cc += gw[q] * dL[a] * L[b2];
        ct += gw[q] * L[a] * dL[b2];
      }
      h->T1[0][a][b2] = dd; h->T1[1][a][b2] = mm; h->T1[2][a][b2] = cc; h->T1[3][a][b2] = ct;
    }
  if (!g_tab) MFEM_CHECK_HIP(hipMalloc(&g_tab, sizeof(Hex27Tables)));
  MFEM_CHECK_HIP(hipMemcpy(g_tab, h, sizeof(Hex27Tables), hipMemcpyHostToDevice));
  delete h;
  g_tab_ng = ng;
  return MFEM_OK;
}

// per-wave LDS carve-up (doubles), sized from ng / nq = ng^3 (even-padded).  NI = components pushed through the
// sum-factorised interpolation: 3 (x1, x2, x3) for the matrix, 5 (+ nodal T and nodal source s) for the residual.
//   X[27][NI] | T1 [2][ng][9][NI] (first stage)                         -- both dead once stage 2 has run, so
//   J -> Jinv [nq][9] | grad_xi T [nq][3] | s at the Gauss points [nq]  -- (written by stage 3) overlay them; the
//                                                                           residual's transposed stages reuse this space again
//   T2 [3][ng][ng][3][NI] (second stage); the residual's flux [nq][3] + source [nq] overlays it later
//   w det [nq] | int64 rowbase[27] + int32 info[27][8] (colour-scatter matrix variant only)
__host__ __device__ inline int h27_pad(int v) { return (v + 1) & ~1; }
__host__ __device__ inline int h27_max(int a, int b) { return a > b ? a : b; }
#define H27_N1 (18 * ng * NI)
#define H27_N2 (9 * ng * ng * NI)
#define H27_N3 ((NI == 3 ? 9 : 13) * nq)
#define H27_NA (9 * ng * ng)     // residual, transposed stage A: [3][ng][ng][3]
#define H27_NB (18 * ng)         // residual, transposed stage B: [2][ng][9]
#define W_X 0
#define W_T1 h27_pad(27 * NI)
#define W_J 0
#define W_GX (9 * nq)
#define W_SV (12 * nq)
#define W_VA 0
#define W_WB h27_pad(H27_NA)
#define W_T2 h27_max(W_T1 + h27_pad(H27_N1), h27_pad(H27_N3))
#define W_G W_T2
#define W_D (W_T2 + h27_max(h27_pad(H27_N2), 4 * h27_pad(nq)))
#define W_INFO (W_D + h27_pad(nq))
#define W_SIZE(with_info) (W_INFO + ((with_info) ? 27 + 27 * 4 + 1 : 0))
// workgroup-shared decode table of the sum-factorised stages (int32 words)
#define H27_NDEC (H27_N1 + H27_N2 + H27_N3 + (NI == 3 ? 0 : H27_NA + H27_NB))
#define H27_WAVES 8                  // waves per workgroup (they share the reference tables in LDS)
#define H27_NQP(nq) (((nq) + 3) & ~3)  // Gauss points padded to whole k-groups of the MFMA loop
#define H27_THREADS (64 * H27_WAVES)

__device__ __forceinline__ double readlane_f64(double v, int l) {  // l wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// Row order of an element's Ke in the scratch: dimension-2 index fastest (a = a0 + 3 a1 + 9 a2 -> a2 + 3 a1 + 9 a0).  The
// gather walks the control points with dimension 2 fastest, so the three rows an element gives to one wave are one
// contiguous 648-byte piece and the nine rows of a dimension-0 layer (1944 bytes) are used within a few workgroups of each
// other -- whole cache lines get used while they are resident.
__device__ __forceinline__ int scratch_row(int a) { return a / 9 + 3 * ((a / 3) % 3) + 9 * (a % 3); }

struct Hex27Args {
  BrickView B;
  const Hex27Tables* tab;
  double kcond;
  int affine_fast;   // matrix: elements whose 27 nodes are an affine image of the reference nodes (to round-off) take the constant-Jacobian shortcut
  int colour;        // 0..7: (I&1) | (J&1)<<1 | (K&1)<<2
  int nq, ng;
  int e_lo, e_cnt, ring;  // element planes [e_lo, e_lo + e_cnt) of dimension 0 this launch covers; scratch variant: plane I kept in ring slot I % ring
  // mixed meshes (round 5): pass 1 over a LIST of elements only -- elist[k] = index (I - e_lo, J, K) of the k-th non-affine element inside the planes
  // above, its Ke goes to scratch slot k; nullptr: every element of the planes
  const int32_t* elist;
  int64_t ecount;
};

// Walk of one wave over its elements: the elements of a launch form an n0 x n1 x n2 grid (one colour's sub-lattice, or the
// planes [e_lo, e_lo + e_cnt) of the whole mesh for the scratch path); the wave starts at running index e and advances by
// the number of waves.  The mixed-radix step is worked out once, so an advance is a few adds instead of three integer
// divisions per element.
struct ElemWalk {
  int ci, cj, ck;      // position in the launch's grid
  int si, sj, sk;      // mixed-radix digits of the stride
  int n0, n1, n2;
  int mul, o0, o1, o2; // element (I, J, K) = mul * (ci, cj, ck) + (o0, o1, o2)
  __device__ __forceinline__ void init(const BrickView& B, int colour, int e_lo, int e_cnt, int64_t e, int stride) {
    if (colour < 0) {
      n0 = e_cnt; n1 = B.ne1; n2 = B.ne2;
      mul = 1; o0 = e_lo; o1 = 0; o2 = 0;
    } else {
      o0 = e_lo + (((colour & 1) - e_lo) & 1);  // first plane of the range with the colour's parity
      o1 = (colour >> 1) & 1; o2 = colour >> 2;
      n0 = (e_lo + e_cnt - o0 + 1) >> 1; n1 = (B.ne1 - o1 + 1) >> 1; n2 = (B.ne2 - o2 + 1) >> 1;
      mul = 2;
    }
    if (n0 <= 0 || n1 <= 0 || n2 <= 0) {
      ci = 0; n0 = 0; cj = ck = si = sj = sk = 0;
      return;
    }
    const int64_t n12 = (int64_t)n1 * n2;
    ci = (int)(e / n12 < n0 ? e / n12 : n0);
    cj = (int)((e % n12) / n2);
    ck = (int)(e % n2);
    const int64_t sI = stride / n12;
    si = (int)(sI < n0 ? sI : n0);
    sj = (int)((stride % n12) / n2);
    sk = (int)(stride % n2);
  }
  __device__ __forceinline__ bool have() const { return ci < n0; }
  __device__ __forceinline__ void get(int& I, int& J, int& K) const {
    I = mul * ci + o0; J = mul * cj + o1; K = mul * ck + o2;
  }
  __device__ __forceinline__ void advance() {
    ck += sk;
    if (ck >= n2) { ck -= n2; cj += 1; }
    cj += sj;
    if (cj >= n1) { cj -= n1; ci += 1; }
    ci += si;
  }
};

// __launch_bounds__(512, 4) (second argument = waves per SIMD in HIP): two workgroups must fit a CU, i.e. <= 128 VGPRs per lane; without it the
// scatter epilogue pushed the kernel to 145 VGPRs and only ONE workgroup (2 waves per SIMD) was resident.
// SCRATCH: the two-pass variant (Ke -> element-major scratch) is its own instantiation, free of the scatter's registers.
template <bool MATRIX, bool SCRATCH = false>
__global__ __launch_bounds__(H27_THREADS, 4) void k_hex27(Hex27Args A, const double* __restrict__ xstar,
                                                        const double* __restrict__ src, double* __restrict__ out) {
  extern __shared__ double lds[];
  constexpr int NI = MATRIX ? 3 : 5;
  // workgroup-shared reference tables
  double* s_dN = lds;                           // [nqp + 1][3][27] (matrix only: MFMA operands; nqp = nq rounded up to the 4 Gauss points of a k-group,
                                                //  rows nq .. nqp zero: the k-rows past the last Gauss point and the columns past node 26 read zeros)
  double* s_w = s_dN + (MATRIX ? (H27_NQP(A.nq) + 1) * 81 : 0); // [nq]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nq = A.nq, ng = A.ng;
  double* s_tab1 = s_w + ((A.nq + 1) & ~1);     // [2][ng][4]
  int32_t* s_dec = reinterpret_cast<int32_t*>(s_tab1 + 8 * ng);
  double* wave_base = s_tab1 + 8 * ng + (h27_pad(H27_NDEC) >> 1);
  const int n1 = H27_N1, n2 = H27_N2, n3 = H27_N3;
  if (MATRIX)
    for (int i = tid; i < (H27_NQP(nq) + 1) * 81; i += H27_THREADS) s_dN[i] = i < nq * 81 ? (&A.tab->dN[0][0][0])[i] : 0.0;
  for (int i = tid; i < nq; i += H27_THREADS) s_w[i] = A.tab->w[i];
  for (int i = tid; i < 8 * ng; i += H27_THREADS) s_tab1[i] = A.tab->tab1[i / (4 * ng)][(i >> 2) % ng][i & 3];
  // decode words of the three sum-factorisation stages: low half = offset of the first of the 3 operands (stride NI),
  // high half = offset of the 1-D table row
  for (int t = tid; t < n1; t += H27_THREADS) {   // T1[k][q0][a12][i] = sum_a0 tab1[k][q0][a0] X[a0 + 3 a12][i]
    const int i = t % NI, a12 = (t / NI) % 9, kq = t / (9 * NI);
    s_dec[t] = (3 * a12 * NI + i) | ((4 * kq) << 16);
  }
  for (int t = tid; t < n2; t += H27_THREADS) {   // T2[kk][q0][q1][a2][i] = sum_a1 tab1[kt][q1][a1] T1[k][q0][a1 + 3 a2][i]
    const int i = t % NI, a2 = (t / NI) % 3, q1 = (t / (3 * NI)) % ng, q0 = (t / (3 * NI * ng)) % ng, kk = t / (3 * NI * ng * ng);
    const int k = kk == 0 ? 1 : 0, kt = kk == 1 ? 1 : 0;  // kk = 0: d/dxi0, 1: d/dxi1, 2: values in both (for d/dxi2 and values)
    s_dec[n1 + t] = (((k * ng + q0) * 9 + 3 * a2) * NI + i) | ((4 * (kt * ng + q1)) << 16);
  }
  for (int t = tid; t < n3; t += H27_THREADS) {
    // t < 9 nq:   J[q][i][m]     = sum_a2 tab1[m == 2][q2][a2] T2[m][q0][q1][a2][i]        (i < 3)
    // then 3 nq:  grad_xi T[q][m] = the same with component 3; then nq: s[q] = sum_a2 L[q2][a2] T2[2][q0][q1][a2][4]
    int m, i, q;
    if (t < 9 * nq) {
      m = t % 3; i = (t / 3) % 3; q = t / 9;
    } else if (t < 12 * nq) {
      m = (t - 9 * nq) % 3; i = 3; q = (t - 9 * nq) / 3;
    } else {
      m = 3; i = 4; q = t - 12 * nq;
    }
    const int q0 = q % ng, q1 = (q / ng) % ng, q2 = q / (ng * ng);
    const int kk = m == 3 ? 2 : m, kt = m == 2 ? 1 : 0;
    s_dec[n1 + n2 + t] = ((((kk * ng + q0) * ng + q1) * 3) * NI + i) | ((4 * (kt * ng + q2)) << 16);
  }
  if (!MATRIX) {
    // transposed stages of the residual: fe[a] = sum_q ( dN[q][a][m] h[q][m] + N[q][a] sq[q] )
    for (int t = tid; t < H27_NA; t += H27_THREADS) {  // VA[v][q0][q1][a2] = sum_q2 tab[q2][a2] h_v[q]  (v = 2: D h_2 + L sq)
      const int a2 = t % 3, q1 = (t / 3) % ng, q0 = (t / (3 * ng)) % ng, v = t / (3 * ng * ng);
      const int qb = q0 + ng * q1;
      s_dec[n1 + n2 + n3 + t] = (3 * qb + v) | (a2 << 12) | ((v == 2 ? 1 : 0) << 14) | (qb << 16);
    }
    for (int t = tid; t < H27_NB; t += H27_THREADS) {  // WB[w][q0][a12] = sum_q1 tab[q1][a1] VA[..]  (w = 1: D VA_1 + L VA_2)
      const int a12 = t % 9, a1 = a12 % 3, a2 = a12 / 3, q0 = (t / 9) % ng, w = t / (9 * ng);
      s_dec[n1 + n2 + n3 + H27_NA + t] = (((w * ng + q0) * ng) * 3 + a2) | (a1 << 12) | (w << 14);
    }
  }
  double* W = wave_base + (size_t)wv * W_SIZE(MATRIX && !SCRATCH);
  if (MATRIX)  // G rows of the padding Gauss points nq .. nqp - 1: multiplied by the table's zero rows, so they only have to be finite -- cleared once
    for (int t = lane; t < 9 * (H27_NQP(nq) - nq); t += 64) W[W_J + 9 * nq + t] = 0.0;  // (later writes to this space are stage-2 sums: finite)
  __syncthreads();
  const BrickView& B = A.B;
  int64_t* rowbase = reinterpret_cast<int64_t*>(W + W_INFO);
  int32_t* info = reinterpret_cast<int32_t*>(W + W_INFO + 27);
  const int nwaves = gridDim.x * H27_WAVES;

  // Node data of an element (lane a < 27: coordinates of node a and, for the scatter, its CSR row descriptor).  The loads
  // of element e + nwaves are issued while element e is still being integrated and land in registers; they are written
  // to the wave's LDS block at the top of the next iteration, so their latency (three dependent table lookups + the
  // coordinate loads) never sits on the wave's critical path.
  struct NodePre {
    double x0, x1, x2;
    int64_t rb;
    int32_t s1, c2;
  };
  // (every lane loads -- lanes 27 .. 63 repeat node 26 -- and the wave fetches on every step, the last one repeating its element: behind a condition the
  // loaded registers meet zeros / their old values in a phi, and the copies that resolves into wait for the loads at once -- the prefetch would hide nothing)
  auto fetch_nodes = [&](int I, int J, int K) -> NodePre {
    NodePre n{0.0, 0.0, 0.0, 0, 0, 0};
    {
      const int ln = lane < 27 ? lane : 26;
      const int gi = 2 * I + ln % 3, gj = 2 * J + (ln / 3) % 3, gk = 2 * K + ln / 9;
      const int64_t c = brick_cindex(B, gi, gj, gk);
      n.x0 = B.X0[c];
      n.x1 = B.X1[c];
      n.x2 = B.X2[c];
      if (MATRIX && !SCRATCH) {
        // slot(a, b) = rowbase[a] + gi_b * s1_a + gj_b * s2_a + gk_b   with the row's box origin folded into rowbase
        const int c1 = B.c1[gj];
        n.c2 = B.c2[gk];
        const int64_t s1 = (int64_t)c1 * n.c2;
        n.s1 = (int32_t)s1;
        n.rb = brick_prefix(B, gi, gj, gk) - ((int64_t)B.lo0[gi] * s1 + (int64_t)B.lo1[gj] * n.c2 + B.lo2[gk]);
      }
    }
    return n;
  };
  int I = 0, J = 0, K = 0;
  ElemWalk walk;
  walk.init(B, A.colour, A.e_lo, A.e_cnt, (int64_t)blockIdx.x * H27_WAVES + wv, nwaves);
  // list mode (mixed meshes): the wave walks the entries lk, lk + nwaves, ... of the element list instead of the launch's grid
  int64_t lk = (int64_t)blockIdx.x * H27_WAVES + wv, lk_cur = 0;
  auto list_get = [&](int64_t k, int& I_, int& J_, int& K_) {
    const uint32_t id = (uint32_t)A.elist[k], n2 = (uint32_t)B.ne2, n12 = (uint32_t)B.ne1 * n2;
    const uint32_t i = id / n12, rem = id - i * n12, j = rem / n2;
    I_ = A.e_lo + (int)i; J_ = (int)j; K_ = (int)(rem - j * n2);
  };
  bool have = A.elist ? lk < A.ecount : walk.have();  // wave-uniform
  if (have) {
    if (A.elist) list_get(lk, I, J, K); else walk.get(I, J, K);
  }
  NodePre cur = fetch_nodes(have ? I : A.e_lo, have ? J : 0, have ? K : 0);  // (a wave without an element loads the first one of the launch's planes: inside the slab's coordinates)
  while (have) {
    // ---- 1. nodes: coordinates + row descriptors (matrix) / nodal values (residual)
    if (lane < 27) {
      W[W_X + NI * lane + 0] = cur.x0;
      W[W_X + NI * lane + 1] = cur.x1;
      W[W_X + NI * lane + 2] = cur.x2;
      const int gi = 2 * I + lane % 3, gj = 2 * J + (lane / 3) % 3, gk = 2 * K + lane / 9;
      if (MATRIX && SCRATCH) {
        // two-pass path: no row descriptors needed
      } else if (MATRIX) {
        rowbase[lane] = cur.rb;
        int32_t* in = info + 8 * lane;
        in[0] = cur.s1; in[1] = cur.c2;
        in[5] = gi; in[6] = gj; in[7] = gk;
      } else {
        const int64_t xi = brick_xindex(B, 0, gi, gj, gk);
        W[W_X + NI * lane + 3] = xstar[xi];
        W[W_X + NI * lane + (NI - 1)] = src ? src[xi] : 0.0;
      }
    }
    // next element of this wave: issue its node loads now
    const int Ic = I, Jc = J, Kc = K;
    lk_cur = lk;
    if (A.elist) {
      lk += nwaves;
      have = lk < A.ecount;
      if (have) list_get(lk, I, J, K);
    } else {
      walk.advance();
      have = walk.have();
      if (have) walk.get(I, J, K);
    }
    cur = fetch_nodes(I, J, K);  // (I, J, K keep the last element when the walk is over)
    __builtin_amdgcn_wave_barrier();
    // ---- 2'. AFFINE elements (round 4).  The counters say pass 1 is bound by the ONE pipe FP64 VALU and FP64 MFMA share (MFMA busy 53 % + FP64 /
    //      integer VALU 33 % of the SIMD cycles: profiles/r04_hex27_wave_counters.txt), and ~45 % of an element's 560 VALU instructions are the
    //      sum-factorised Jacobian, its adjugate and det at 27 Gauss points.  When the element's 27 nodes are an affine image of the reference
    //      nodes -- every element of make_Brick until a caller moves coordinates (mfem_brick_coords) -- J is one matrix: G_q = w_q G0 with
    //      G0 = -k adj(J) adj(J)^T / det, 6 numbers computed once.  The test is made per element on the coordinates themselves (lane a against
    //      x(0) + a0 e0 / 2 + a1 e1 / 2 + a2 e2 / 2 with the edge vectors e_m = x(corner m) - x(0)), to 16 ulp of the coordinates' magnitude:
    //      what J's own cancellation error is made of.  Anything else takes the general path below; both give G to round-off of each other.
    bool affine = false;
    if (A.affine_fast) {
      const double x0 = W[W_X + 0], y0 = W[W_X + 1], z0 = W[W_X + 2];
      const double ex0 = W[W_X + NI * 2 + 0] - x0, ey0 = W[W_X + NI * 2 + 1] - y0, ez0 = W[W_X + NI * 2 + 2] - z0;     // node (2,0,0): d x / d xi0
      const double ex1 = W[W_X + NI * 6 + 0] - x0, ey1 = W[W_X + NI * 6 + 1] - y0, ez1 = W[W_X + NI * 6 + 2] - z0;     // node (0,2,0): d x / d xi1
      const double ex2 = W[W_X + NI * 18 + 0] - x0, ey2 = W[W_X + NI * 18 + 1] - y0, ez2 = W[W_X + NI * 18 + 2] - z0;  // node (0,0,2): d x / d xi2
      bool mine = true;
      if (lane < 27) {
        const double a0 = 0.5 * (lane % 3), a1 = 0.5 * ((lane / 3) % 3), a2 = 0.5 * (lane / 9);
        const double px = x0 + a0 * ex0 + a1 * ex1 + a2 * ex2, py = y0 + a0 * ey0 + a1 * ey1 + a2 * ey2, pz = z0 + a0 * ez0 + a1 * ez1 + a2 * ez2;
        const double mx = W[W_X + NI * lane + 0], my = W[W_X + NI * lane + 1], mz = W[W_X + NI * lane + 2];
        const double tol = 3.6e-15;  // 16 ulp of the largest coordinate magnitude the element spans, per component
        mine = fabs(mx - px) <= tol * (fabs(x0) + fabs(ex0) + fabs(ex1) + fabs(ex2)) &&
               fabs(my - py) <= tol * (fabs(y0) + fabs(ey0) + fabs(ey1) + fabs(ey2)) &&
               fabs(mz - pz) <= tol * (fabs(z0) + fabs(ez0) + fabs(ez1) + fabs(ez2));
      }
      affine = __all(mine);
      if (affine && MATRIX) {
        // J[i][m] = e_m[i]; adjugate rows c_m (as in 2b), G0 = -k / det * C C^T
        const double j00 = ex0, j01 = ex1, j02 = ex2, j10 = ey0, j11 = ey1, j12 = ey2, j20 = ez0, j21 = ez1, j22 = ez2;
        const double det = j00 * j11 * j22 - j00 * j12 * j21 - j01 * j10 * j22 + j01 * j12 * j20 + j02 * j10 * j21 - j02 * j11 * j20;
        const double c00 = j11 * j22 - j12 * j21, c01 = j02 * j21 - j01 * j22, c02 = j01 * j12 - j11 * j02;
        const double c10 = j12 * j20 - j22 * j10, c11 = j00 * j22 - j02 * j20, c12 = j02 * j10 - j00 * j12;
        const double c20 = j10 * j21 - j11 * j20, c21 = j01 * j20 - j21 * j00, c22 = j00 * j11 - j10 * j01;
        const double sc0 = -A.kcond / det;
        const double g0 = sc0 * (c00 * c00 + c01 * c01 + c02 * c02), g1 = sc0 * (c00 * c10 + c01 * c11 + c02 * c12),
                     g2 = sc0 * (c00 * c20 + c01 * c21 + c02 * c22), g3 = sc0 * (c10 * c10 + c11 * c11 + c12 * c12),
                     g4 = sc0 * (c10 * c20 + c11 * c21 + c12 * c22), g5 = sc0 * (c20 * c20 + c21 * c21 + c22 * c22);
        __builtin_amdgcn_wave_barrier();  // (the reads of W_X above are done: W_J overlays it)
        for (int q = lane; q < nq; q += 64) {
          double* Jm = W + W_J + q * 9;
          const double wq = s_w[q];
          Jm[0] = wq * g0; Jm[1] = wq * g1; Jm[2] = wq * g2; Jm[3] = wq * g3; Jm[4] = wq * g4; Jm[5] = wq * g5;
        }
      }
      if (affine && !MATRIX) {
        // Residual on an affine element: only T and s (components 3, 4 of the 5 interleaved ones) go through the three sum-factorised stages -- the
        // same decode words, walked over the compact index u -> t = (u / 2) NI + 3 + (u & 1): 2 + 3 + 2 trips of 64 lanes instead of 5 + 7 + 6 --, and the
        // inverse Jacobian is one matrix, written to every Gauss point's row for the flux stage below (the general path computes 27 of them).
        const double j00 = ex0, j01 = ex1, j02 = ex2, j10 = ey0, j11 = ey1, j12 = ey2, j20 = ez0, j21 = ez1, j22 = ez2;
        const double det = j00 * j11 * j22 - j00 * j12 * j21 - j01 * j10 * j22 + j01 * j12 * j20 + j02 * j10 * j21 - j02 * j11 * j20;
        const double id = 1.0 / det;
        const double i0 = (j11 * j22 - j12 * j21) * id, i1 = (j02 * j21 - j01 * j22) * id, i2 = (j01 * j12 - j11 * j02) * id;
        const double i3 = (j12 * j20 - j22 * j10) * id, i4 = (j00 * j22 - j02 * j20) * id, i5 = (j02 * j10 - j00 * j12) * id;
        const double i6 = (j10 * j21 - j11 * j20) * id, i7 = (j01 * j20 - j21 * j00) * id, i8 = (j00 * j11 - j10 * j01) * id;
        for (int u = lane; u < 2 * (n1 / NI); u += 64) {
          const int t = (u >> 1) * NI + 3 + (u & 1);
          const int d = s_dec[t];
          const double* x = W + W_X + (d & 0xffff);
          const double* tb = s_tab1 + (d >> 16);
          W[W_T1 + t] = tb[0] * x[0] + tb[1] * x[NI] + tb[2] * x[2 * NI];
        }
        __builtin_amdgcn_wave_barrier();
        for (int u = lane; u < 2 * (n2 / NI); u += 64) {
          const int t = (u >> 1) * NI + 3 + (u & 1);
          const int d = s_dec[n1 + t];
          const double* x = W + W_T1 + (d & 0xffff);
          const double* tb = s_tab1 + (d >> 16);
          W[W_T2 + t] = tb[0] * x[0] + tb[1] * x[NI] + tb[2] * x[2 * NI];
        }
        __builtin_amdgcn_wave_barrier();
        for (int t = 9 * nq + lane; t < n3; t += 64) {  // grad_xi T and s at the Gauss points (the J entries in front of them are not needed)
          const int d = s_dec[n1 + n2 + t];
          const double* x = W + W_T2 + (d & 0xffff);
          const double* tb = s_tab1 + (d >> 16);
          W[W_J + t] = tb[0] * x[0] + tb[1] * x[NI] + tb[2] * x[2 * NI];
        }
        for (int q = lane; q < nq; q += 64) {  // (rows 0 .. 9 nq of this space: X and T1 are dead, the outputs above sit behind them)
          double* Jm = W + W_J + q * 9;
          Jm[0] = i0; Jm[1] = i1; Jm[2] = i2; Jm[3] = i3; Jm[4] = i4; Jm[5] = i5; Jm[6] = i6; Jm[7] = i7; Jm[8] = i8;
          W[W_D + q] = s_w[q] * det;
        }
      }
    }
    if (!affine) {
    // ---- 2a. J[q][i][m] = sum_a dN[q][a][m] X[a][i], sum-factorised over the tensor-product basis
    //      (dN[q][a][0] = D(q0,a0) L(q1,a1) L(q2,a2), ...): three stages of 3-term sums through the wave's LDS block,
    //      ~2000 multiply-adds per element instead of 6561 -- FP64 VALU work runs on the same pipe as the FP64 MFMAs
    //      on this chip (the phase ablation is additive), so every VALU instruction saved here is matrix-core time.
    {
      for (int t = lane; t < n1; t += 64) {
        const int d = s_dec[t];
        const double* x = W + W_X + (d & 0xffff);
        const double* tb = s_tab1 + (d >> 16);
        W[W_T1 + t] = tb[0] * x[0] + tb[1] * x[NI] + tb[2] * x[2 * NI];
      }
      __builtin_amdgcn_wave_barrier();
      for (int t = lane; t < n2; t += 64) {
        const int d = s_dec[n1 + t];
        const double* x = W + W_T1 + (d & 0xffff);
        const double* tb = s_tab1 + (d >> 16);
        W[W_T2 + t] = tb[0] * x[0] + tb[1] * x[NI] + tb[2] * x[2 * NI];
      }
      __builtin_amdgcn_wave_barrier();
      for (int t = lane; t < n3; t += 64) {
        const int d = s_dec[n1 + n2 + t];
        const double* x = W + W_T2 + (d & 0xffff);
        const double* tb = s_tab1 + (d >> 16);
        W[W_J + t] = tb[0] * x[0] + tb[1] * x[NI] + tb[2] * x[2 * NI];
      }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- 2b. det, inverse (adjugate, inv_Jac_3D), w det ; Jinv overwrites J as [m][s]
    for (int q = lane; q < nq; q += 64) {
      double* Jm = W + W_J + q * 9;
      const double j00 = Jm[0], j01 = Jm[1], j02 = Jm[2], j10 = Jm[3], j11 = Jm[4], j12 = Jm[5], j20 = Jm[6], j21 = Jm[7],
                   j22 = Jm[8];
      const double det = j00 * j11 * j22 - j00 * j12 * j21 - j01 * j10 * j22 + j01 * j12 * j20 + j02 * j10 * j21 - j02 * j11 * j20;
      if (MATRIX) {
        // matrix: only G = -k w det Jinv Jinv^T (symmetric 3 x 3) is needed: Ke = sum_q dN_q G_q dN_q^T
        const double c00 = j11 * j22 - j12 * j21, c01 = j02 * j21 - j01 * j22, c02 = j01 * j12 - j11 * j02;
        const double c10 = j12 * j20 - j22 * j10, c11 = j00 * j22 - j02 * j20, c12 = j02 * j10 - j00 * j12;
        const double c20 = j10 * j21 - j11 * j20, c21 = j01 * j20 - j21 * j00, c22 = j00 * j11 - j10 * j01;
        const double sc = -A.kcond * s_w[q] / det;
        Jm[0] = sc * (c00 * c00 + c01 * c01 + c02 * c02);
        Jm[1] = sc * (c00 * c10 + c01 * c11 + c02 * c12);
        Jm[2] = sc * (c00 * c20 + c01 * c21 + c02 * c22);
        Jm[3] = sc * (c10 * c10 + c11 * c11 + c12 * c12);
        Jm[4] = sc * (c10 * c20 + c11 * c21 + c12 * c22);
        Jm[5] = sc * (c20 * c20 + c21 * c21 + c22 * c22);
        continue;
      }
      const double id = 1.0 / det;
      Jm[0] = (j11 * j22 - j12 * j21) * id;
      Jm[1] = (j02 * j21 - j01 * j22) * id;
      Jm[2] = (j01 * j12 - j11 * j02) * id;
      Jm[3] = (j12 * j20 - j22 * j10) * id;
      Jm[4] = (j00 * j22 - j02 * j20) * id;
      Jm[5] = (j02 * j10 - j00 * j12) * id;
      Jm[6] = (j10 * j21 - j11 * j20) * id;
      Jm[7] = (j01 * j20 - j21 * j00) * id;
      Jm[8] = (j00 * j11 - j10 * j01) * id;
      W[W_D + q] = s_w[q] * det;
    }
    }  // !affine
    __builtin_amdgcn_wave_barrier();
    if (MATRIX) {
      // ---- 3. Ke = B^T D B on the matrix cores.  MFMA fragments are formed on the fly:
      //      lane (k = lane>>4, c = lane&15) of k-step ks holds row r = 4 ks + k of B at columns c and c + 16,
      //      B[(q,s)][a] = sum_m dN[q][a][m] Jinv[q][m][s]; the A operand is the same row scaled by -k w det.
      d4_t C00 = {0, 0, 0, 0}, C01 = {0, 0, 0, 0}, C11 = {0, 0, 0, 0};
      const int c = lane & 15, kl = lane >> 4;
      const bool hi_ok = (c + 16) < 27;
      // Ke = sum_q dN_q G_q dN_q^T: k-row (q, n) has the B operand dN[q][b][n] straight from the table and the A operand
      // sum_m dN[q][a][m] G_q[m][n] (3 FMAs).  The order of the 3 nq rows along k is free: k-steps 3 g + n (n = 0..2) carry
      // the rows (q = 4 g + kl, n), so a lane keeps one Gauss point for three steps and reads its six table entries and the
      // six entries of G once per group (12 LDS words per group; the first version read 30).
      // Branch-free k-groups (round 4): the k-rows past the last Gauss point read zero rows of the table (and the finite filler behind G), the
      // lanes whose second column block lies past node 26 read the table's zero row with stride 0 -- no per-group zeroing of 24 registers, no
      // exec-mask regions between the LDS reads and the MFMAs.
      const int ngroups = H27_NQP(nq) >> 2;
      const double* dn0 = s_dN + kl * 81 + c;
      const double* dn1 = hi_ok ? dn0 + 16 : s_dN + H27_NQP(nq) * 81;
      const int st1 = hi_ok ? 4 * 81 : 0;
      const double* Gq = W + W_J + kl * 9;
      for (int g = 0; g < ngroups; ++g) {
        const double d0[3] = {dn0[0], dn0[27], dn0[54]};
        const double d1[3] = {dn1[0], dn1[27], dn1[54]};
        const double G[3][3] = {{Gq[0], Gq[1], Gq[2]}, {Gq[1], Gq[3], Gq[4]}, {Gq[2], Gq[4], Gq[5]}};
#pragma unroll
        for (int n = 0; n < 3; ++n) {
          const double a0 = d0[0] * G[0][n] + d0[1] * G[1][n] + d0[2] * G[2][n];
          const double a1 = d1[0] * G[0][n] + d1[1] * G[1][n] + d1[2] * G[2][n];
          C00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, d0[n], C00, 0, 0, 0);
          C01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, d1[n], C01, 0, 0, 0);
          C11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, d1[n], C11, 0, 0, 0);
        }
        dn0 += 4 * 81;
        dn1 += st1;
        Gq += 36;
      }
      if (SCRATCH) {
        // ---- 4'. two-pass assembly: Ke goes to the element-major scratch [e][a][b] (written once, no RMW); the
        //      row-owner gather kernel below turns it into CSR rows.
        double* ke = A.elist ? out + lk_cur * 729 : out + (((int64_t)(Ic % A.ring) * B.ne1 + Jc) * B.ne2 + Kc) * 729;
        const int rc_hi = scratch_row(16 + c > 26 ? 26 : 16 + c);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int ra = kl + 4 * reg;
          if (ra < 27) {
            const int r0 = scratch_row(ra);
            ke[r0 * 27 + c] = C00[reg];
            if (hi_ok) {
              ke[r0 * 27 + 16 + c] = C01[reg];
              ke[rc_hi * 27 + ra] = C01[reg];
            }
          }
          if (16 + ra < 27 && hi_ok) ke[scratch_row(16 + ra) * 27 + 16 + c] = C11[reg];
        }
        __builtin_amdgcn_wave_barrier();
        continue;
      }
      // ---- 4. colour-safe scatter: 16 entries per lane in two batches of 8 (register budget: 128 VGPRs = 4 waves per
      //      SIMD); all slots of one element are distinct, so a batch's loads are issued together, then its stores
      //      (plain read-modify-write, no atomics).
      const int rq = kl;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        int64_t slot[8];
        double val[8];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          const int reg = 2 * half + rr;
          const int ra = rq + 4 * reg;  // f64 MFMA C/D map: row = (lane>>4) + 4*reg, col = lane & 15
          const int aa[4] = {ra, ra, 16 + c, 16 + ra};
          const int bb[4] = {c, 16 + c, ra, 16 + c};  // tiles (0,0), (0,1), (1,0) = transpose of (0,1), (1,1)
          const double vv[4] = {C00[reg], C01[reg], C01[reg], C11[reg]};
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int _a = aa[t], _b = bb[t];
            int64_t sl = -1;
            if (_a < 27 && _b < 27) {
              const int32_t* ia = info + 8 * _a;
              const int32_t* ib = info + 8 * _b;
              sl = rowbase[_a] + (int64_t)ib[5] * ia[0] + (int64_t)ib[6] * ia[1] + ib[7];
            }
            slot[4 * rr + t] = sl;
            val[4 * rr + t] = vv[t];
          }
        }
        if (A.colour == -2) {
          // atomics variant (what the reference's _Kval_Basic does, 06_FEM_Kernel.jl:41): all elements in one launch,
          // FP64 adds resolved in L2, nothing returns to the wave; summation order (and so the last bit) is not fixed
#pragma unroll
          for (int t = 0; t < 8; ++t)
            if (slot[t] >= 0) unsafeAtomicAdd(out + slot[t], val[t]);
          continue;
        }
        double old[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) old[t] = slot[t] >= 0 ? out[slot[t]] : 0.0;
#pragma unroll
        for (int t = 0; t < 8; ++t)
          if (slot[t] >= 0) out[slot[t]] = old[t] + val[t];
      }
    } else {
      // ---- 3'. residual: fe[a] = sum_q w det ( -k gradN_a . gradT + N_a s_q )
      //      gradN_a . gradT = sum_m dN[q][a][m] h[q][m],  h = Jinv (Jinv^T gxi),  gxi[m] = sum_b dN[q][b][m] T_b.
      //      gxi and s at the Gauss points came out of the sum-factorised stages above (components 3 and 4).
      for (int q = lane; q < nq; q += 64) {  // lanes q: gradT = Jinv^T gxi ; h = -k w det * Jinv gradT ; source
        const double* Ji = W + W_J + q * 9;
        const double g0 = W[W_GX + 3 * q], g1 = W[W_GX + 3 * q + 1], g2 = W[W_GX + 3 * q + 2];
        const double t0 = g0 * Ji[0] + g1 * Ji[3] + g2 * Ji[6];
        const double t1 = g0 * Ji[1] + g1 * Ji[4] + g2 * Ji[7];
        const double t2 = g0 * Ji[2] + g1 * Ji[5] + g2 * Ji[8];
        const double wd = W[W_D + q], sc = -A.kcond * wd;
        W[W_G + 3 * q + 0] = sc * (Ji[0] * t0 + Ji[1] * t1 + Ji[2] * t2);
        W[W_G + 3 * q + 1] = sc * (Ji[3] * t0 + Ji[4] * t1 + Ji[5] * t2);
        W[W_G + 3 * q + 2] = sc * (Ji[6] * t0 + Ji[7] * t1 + Ji[8] * t2);
        W[W_G + 3 * h27_pad(nq) + q] = W[W_SV + q] * wd;
      }
      __builtin_amdgcn_wave_barrier();
      // the contraction with dN / N, transposed sum factorisation: over q2, then q1, then q0
      const int oA = n1 + n2 + n3, oB = oA + H27_NA, ngg = ng * ng;
      for (int t = lane; t < H27_NA; t += 64) {
        const int d = s_dec[oA + t];
        const int a2 = (d >> 12) & 3, v2 = (d >> 14) & 1;
        const double* hp = W + W_G + (d & 0xfff);
        const double* sp = W + W_G + 3 * h27_pad(nq) + (d >> 16);
        double acc = 0.0;
        for (int q2 = 0; q2 < ng; ++q2) {
          const double L = s_tab1[q2 * 4 + a2], D = s_tab1[(ng + q2) * 4 + a2];
          acc += (v2 ? D : L) * hp[3 * ngg * q2];
          if (v2) acc += L * sp[ngg * q2];
        }
        W[W_VA + t] = acc;
      }
      __builtin_amdgcn_wave_barrier();
      for (int t = lane; t < H27_NB; t += 64) {
        const int d = s_dec[oB + t];
        const int a1 = (d >> 12) & 3, w = (d >> 14) & 1;
        const double* va = W + W_VA + (d & 0xfff);
        double acc = 0.0;
        for (int q1 = 0; q1 < ng; ++q1) {
          const double L = s_tab1[q1 * 4 + a1], D = s_tab1[(ng + q1) * 4 + a1];
          acc += (w ? D : L) * va[3 * q1];
          if (w) acc += L * va[3 * ngg + 3 * q1];
        }
        W[W_WB + t] = acc;
      }
      __builtin_amdgcn_wave_barrier();
      if (lane < 27) {
        const int a0 = lane % 3, a12 = lane / 3;
        double fe = 0.0;
        for (int q0 = 0; q0 < ng; ++q0)
          fe += s_tab1[(ng + q0) * 4 + a0] * W[W_WB + q0 * 9 + a12] + s_tab1[q0 * 4 + a0] * W[W_WB + (ng + q0) * 9 + a12];
        const int gi = 2 * Ic + lane % 3, gj = 2 * Jc + (lane / 3) % 3, gk = 2 * Kc + lane / 9;
        if (gi >= B.plo && gi < B.phi) out[(int64_t)(gi - B.plo) * B.plane_len + (int64_t)gj * B.m2 + gk] += fe;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// Pass 2 of the two-pass assembly: a wave owns 8 consecutive control points and builds their CSR rows in LDS (four waves
// per workgroup, no workgroup barrier -- the waves never exchange data).
//   A. lane (row, e) works out the row's e-th candidate element (a mid node has one element per dimension, an
//      element-boundary node two), the offset of the 27-entry run Ke_e[la][0..26] in the scratch and the LDS slot of the
//      element's first node; a ballot gives the wave the set of (row, e) pairs that exist (3.4 of 8 on average);
//   B. each half-wave streams the runs of its 4 rows in: lane lb < 27 loads entry lb (one contiguous 216-byte read per
//      run, up to sixteen runs in flight per lane) and adds it to the row buffer at the slot of node lb.  A row's runs are taken by
//      one half-wave in element order e = 0..7, so the summation order is fixed;
//   C. the rows leave as contiguous streams.
// No index arithmetic per CSR slot, every scratch entry read once, every value written once.
#define G27_NODES 32
#define G27_ROW 126  // up to 125 entries per row, padded
#define G27_FLIGHT 16 // runs a lane has in flight (the kernel is latency-bound: 4 -> 5.8 ms, 8 -> 5.1 ms at 128^3)
__global__ __launch_bounds__(MFEM_BLOCK) void k_hex27_gather_lds(BrickView B, const double* __restrict__ ke, double* __restrict__ vals, int64_t row_lo,
                                                                  int64_t row_hi, int ring) {
  __shared__ double rows[G27_NODES * G27_ROW];
  __shared__ int64_t s_pre[G27_NODES];
  __shared__ int64_t s_src[G27_NODES * 8];
  __shared__ int32_t s_b0[G27_NODES * 8];
  __shared__ int32_t s_len[G27_NODES], s_c1[G27_NODES], s_c2[G27_NODES];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int t = lane; t < 8 * G27_ROW; t += 64) rows[wv * 8 * G27_ROW + t] = 0.0;
  uint64_t pairs;
  {
    const int nl = tid >> 3, e = tid & 7;
    const int64_t row = row_lo + (int64_t)blockIdx.x * G27_NODES + nl;
    const bool live = row < row_hi;
    int g[3] = {0, 0, 0}, lo0 = 0, lo1 = 0, lo2 = 0, c1 = 1, c2 = 1;
    if (live) {
      const uint32_t r32 = (uint32_t)row, pl = (uint32_t)B.plane_len, m2 = (uint32_t)B.m2;  // control-point ids fit int32
      const uint32_t q0 = r32 / pl, rem = r32 - q0 * pl, q1 = rem / m2;
      g[0] = (int)q0 + B.plo;
      g[1] = (int)q1;
      g[2] = (int)(rem - q1 * m2);
      lo0 = B.lo0[g[0]]; lo1 = B.lo1[g[1]]; lo2 = B.lo2[g[2]];
      c1 = B.c1[g[1]]; c2 = B.c2[g[2]];
    }
    if (e == 0) {
      s_pre[nl] = live ? brick_prefix(B, g[0], g[1], g[2]) : 0;
      s_len[nl] = live ? B.c0[g[0]] * c1 * c2 : 0;
      s_c1[nl] = c1;
      s_c2[nl] = c2;
    }
    const int ed[3] = {e & 1, (e >> 1) & 1, e >> 2};
    const int ne[3] = {B.ne0, B.ne1, B.ne2};
    int E[3];
    bool valid = live;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      if (g[d] & 1) {
        E[d] = (g[d] - 1) >> 1;
        valid = valid && ed[d] == 0;
      } else {
        E[d] = (g[d] >> 1) - 1 + ed[d];
      }
      valid = valid && E[d] >= 0 && E[d] < ne[d];
    }
    const int la = (g[0] - 2 * E[0]) + 3 * (g[1] - 2 * E[1]) + 9 * (g[2] - 2 * E[2]);
    const int64_t eid = ((int64_t)(valid ? E[0] % ring : 0) * ne[1] + E[1]) * ne[2] + E[2];
    s_src[tid] = valid ? (eid * 27 + scratch_row(valid ? la : 0)) * 27 : 0;
    s_b0[tid] = nl * G27_ROW + ((2 * E[0] - lo0) * c1 + (2 * E[1] - lo1)) * c2 + (2 * E[2] - lo2);  // the element's first node
    pairs = __ballot(valid);
  }
  __builtin_amdgcn_wave_barrier();
  {
    const int lb = lane & 31;
    const int first = wv * 64 + (lane >> 5) * 32;  // this half-wave's 32 (row, e) pairs = 4 rows
    const bool active = lb < 27;
    const int bx = lb % 3, by = (lb / 3) % 3, bz = lb / 9;
    uint32_t todo = (lane >> 5) ? (uint32_t)(pairs >> 32) : (uint32_t)pairs;
    while (todo) {
      double v[G27_FLIGHT];
      int sl[G27_FLIGHT];
#pragma unroll
      for (int j = 0; j < G27_FLIGHT; ++j) {
        sl[j] = -1;
        v[j] = 0.0;
        if (todo) {
          const int pair = first + __builtin_ctz(todo), nl = pair >> 3;
          todo &= todo - 1;
          if (active) {
            v[j] = ke[s_src[pair] + lb];
            sl[j] = s_b0[pair] + (bx * s_c1[nl] + by) * s_c2[nl] + bz;
          }
        }
      }
#pragma unroll
      for (int j = 0; j < G27_FLIGHT; ++j)
        if (sl[j] >= 0) rows[sl[j]] += v[j];
    }
  }
  __builtin_amdgcn_wave_barrier();
  for (int r = 0; r < 8; ++r) {
    const int n2 = wv * 8 + r, len = s_len[n2];
    const int64_t pre = s_pre[n2];
    for (int o = lane; o < len; o += 64) vals[pre + o] = rows[n2 * G27_ROW + o];
  }
}

// ---- Affine meshes (round 4): the matrix without Ke ever being stored.  On an element whose 27 nodes are an affine image of the reference nodes the Jacobian is
// one matrix and Ke = sum_t G0_t S_t -- six numbers per element (G0 = -k adj(J) adj(J)^T / det, as in k_hex27's affine branch) times six 27 x 27 reference
// integrals (products of the 1-D integrals Hex27Tables::T1, summed with the same quadrature).  When EVERY element of the launch is affine (each make_Brick mesh until a caller moves coordinates;
// tested per assembly on the coordinates themselves, k_hex27_affine_g0), the row-owner gather below computes each (row, element) run from G0 and the table instead of
// reading it from the element-major scratch: no pass 1, no 12.2 GB scratch written and read back (128^3: 10.5 -> 3 ms for the matrix).  Any non-affine element sends
// the whole assembly through the two-pass MFMA path above.
// slot (optional): per element -1 (affine) or its place k in the compact scratch of the non-affine elements' Ke; elist[k] = the element's index
// (the places are handed out by an atomic counter: which element gets which place varies from run to run, what is stored there does not)
__global__ __launch_bounds__(MFEM_BLOCK) void k_hex27_affine_g0(BrickView B, double kcond, int elo, int ecnt, double* __restrict__ g,
                                                                int32_t* __restrict__ nonaffine, int32_t* __restrict__ slot,
                                                                int32_t* __restrict__ elist) {
  const int64_t nel = (int64_t)ecnt * B.ne1 * B.ne2;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nel) return;
  const int K = (int)(idx % B.ne2), J = (int)((idx / B.ne2) % B.ne1), I = elo + (int)(idx / ((int64_t)B.ne1 * B.ne2));
  auto node = [&](int a, double& x, double& y, double& z) {
    const int64_t c = brick_cindex(B, 2 * I + a % 3, 2 * J + (a / 3) % 3, 2 * K + a / 9);
    x = B.X0[c]; y = B.X1[c]; z = B.X2[c];
  };
  double x0, y0, z0, ex0, ey0, ez0, ex1, ey1, ez1, ex2, ey2, ez2;
  node(0, x0, y0, z0);
  node(2, ex0, ey0, ez0);
  node(6, ex1, ey1, ez1);
  node(18, ex2, ey2, ez2);
  ex0 -= x0; ey0 -= y0; ez0 -= z0; ex1 -= x0; ey1 -= y0; ez1 -= z0; ex2 -= x0; ey2 -= y0; ez2 -= z0;
  const double tol = 3.6e-15;  // (the test of k_hex27: 16 ulp of the coordinates' magnitude, per component)
  const double tx = tol * (fabs(x0) + fabs(ex0) + fabs(ex1) + fabs(ex2)), ty = tol * (fabs(y0) + fabs(ey0) + fabs(ey1) + fabs(ey2)),
               tz = tol * (fabs(z0) + fabs(ez0) + fabs(ez1) + fabs(ez2));
  bool affine = true;
  for (int a = 0; a < 27; ++a) {
    double mx, my, mz;
    node(a, mx, my, mz);
    const double a0 = 0.5 * (a % 3), a1 = 0.5 * ((a / 3) % 3), a2 = 0.5 * (a / 9);
    const double px = x0 + a0 * ex0 + a1 * ex1 + a2 * ex2, py = y0 + a0 * ey0 + a1 * ey1 + a2 * ey2, pz = z0 + a0 * ez0 + a1 * ez1 + a2 * ez2;
    affine = affine && fabs(mx - px) <= tx && fabs(my - py) <= ty && fabs(mz - pz) <= tz;
  }
  double g0 = 0.0, g1 = 0.0, g2 = 0.0, g3 = 0.0, g4 = 0.0, g5 = 0.0;
  if (affine) {
    const double j00 = ex0, j01 = ex1, j02 = ex2, j10 = ey0, j11 = ey1, j12 = ey2, j20 = ez0, j21 = ez1, j22 = ez2;
    const double det = j00 * j11 * j22 - j00 * j12 * j21 - j01 * j10 * j22 + j01 * j12 * j20 + j02 * j10 * j21 - j02 * j11 * j20;
    const double c00 = j11 * j22 - j12 * j21, c01 = j02 * j21 - j01 * j22, c02 = j01 * j12 - j11 * j02;
    const double c10 = j12 * j20 - j22 * j10, c11 = j00 * j22 - j02 * j20, c12 = j02 * j10 - j00 * j12;
    const double c20 = j10 * j21 - j11 * j20, c21 = j01 * j20 - j21 * j00, c22 = j00 * j11 - j10 * j01;
    const double sc0 = -kcond / det;
    g0 = sc0 * (c00 * c00 + c01 * c01 + c02 * c02); g1 = sc0 * (c00 * c10 + c01 * c11 + c02 * c12);
    g2 = sc0 * (c00 * c20 + c01 * c21 + c02 * c22); g3 = sc0 * (c10 * c10 + c11 * c11 + c12 * c12);
    g4 = sc0 * (c10 * c20 + c11 * c21 + c12 * c22); g5 = sc0 * (c20 * c20 + c21 * c21 + c22 * c22);
    if (slot) slot[idx] = -1;
  } else {
    const int k = atomicAdd(nonaffine, 1);
    if (slot) {
      slot[idx] = k;
      elist[k] = (int32_t)idx;
    }
  }
  double* ge = g + idx * 6;
  ge[0] = g0; ge[1] = g1; ge[2] = g2; ge[3] = g3; ge[4] = g4; ge[5] = g5;
}

// The row-owner gather of k_hex27_gather_lds with the runs computed in place.  A wave owns 8 consecutive control points per trip; thread (row, e) takes the row's
// e-th candidate element (3.4 of 8 exist on average) and computes ITS 27-entry run Ke_e[la][0..26] from registers: G0 (6 numbers) and the twelve
// 3-entry rows of the 1-D integrals that belong to its local node la = (a0, a1, a2) -- the reference integrals factor per direction,
//   Ke[la][lb] = M2 P + D2 Q + Ct2 R + C2 T,  P = g0 D0 M1 + g1 (C0 Ct1 + Ct0 C1) + g3 M0 D1,  Q = g5 M0 M1,  R = g2 C0 M1 + g4 C1 M0,  T = g2 Ct0 M1 + g4 Ct1 M0
// (X_d = the 1-D integral X at (a_d, b_d); D = l'l', M = ll, C = l'l, Ct = ll') -- and adds it into the row's box in LDS (ds_add_f64: the threads of one instruction hold
// different (row, element) pairs and the same local node b, i.e. different entries).  36 LDS reads + 27 additions per run; a first version with one lane per entry
// and the 27 x 27 x 6 table in LDS (12 reads per entry) was bound by LDS bandwidth at 6.5 ms (128^3).  The additions into one entry come in program order: the
// result is reproducible (and differs from the two-pass path's in the last bits: another summation order).
#define D27_WAVES 8
#define D27_NODES (8 * D27_WAVES)
#define D27_THREADS (64 * D27_WAVES)
#define D27_TAB 1024  // lattice planes + lines + points whose row-box tables (lo, c, P per direction) are kept in LDS (16 bytes each); beyond: read from memory
#define D27_LDS_BYTES (sizeof(double) * (48 + D27_NODES * G27_ROW + D27_TAB) + sizeof(int32_t) * (2 * D27_TAB))
// Mixed meshes (round 5): slot_of != nullptr -- element idx is affine where slot_of[idx] < 0 (computed in place, as above) and otherwise has its Ke in the
// compact scratch `ke` at slot_of[idx] (pass 1 ran for those elements only, k_hex27<true, true> in list mode): their runs Ke[la][0..26] are streamed in by the
// half-waves exactly as in k_hex27_gather_lds (one contiguous 216-byte read per run) after the wave's in-place runs have been added.  One distorted element no
// longer sends the whole mesh through the two-pass path: the assembly costs what its affine part costs plus pass 1 + the streamed runs of the rest.
#define D27_FLIGHT 8
__global__ __launch_bounds__(D27_THREADS, 4) void k_hex27_direct(BrickView B, const Hex27Tables* __restrict__ tab, const double* __restrict__ g,
                                                              double* __restrict__ vals, int64_t row_lo, int64_t row_hi, int elo,
                                                              const int32_t* __restrict__ slot_of, const double* __restrict__ ke) {
  extern __shared__ double lds[];
  double* sT = lds;                          // [4][3][4]
  double* rows = sT + 48;                    // [D27_NODES][G27_ROW]
  int64_t* t_P = reinterpret_cast<int64_t*>(rows + D27_NODES * G27_ROW);  // [D27_TAB]: P0 | P1 | P2
  int32_t* t_lo = reinterpret_cast<int32_t*>(t_P + D27_TAB);              // [D27_TAB]: lo0 | lo1 | lo2
  int32_t* t_c = t_lo + D27_TAB;             // [D27_TAB]: c0 | c1 | c2
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid < 48) sT[tid] = (&tab->T1[0][0][0])[tid];
  // the row-box tables: phase A below looks a control point's box up per direction (a chain of dependent loads when they come from memory)
  const bool tabs = B.m0 + B.m1 + B.m2 <= D27_TAB;
  const int32_t *lo0 = B.lo0, *lo1 = B.lo1, *lo2 = B.lo2, *c0 = B.c0, *c1p = B.c1, *c2p = B.c2;
  const int64_t *P0 = B.P0, *P1 = B.P1, *P2 = B.P2;
  if (tabs) {
    for (int i = tid; i < B.m0; i += D27_THREADS) { t_lo[i] = B.lo0[i]; t_c[i] = B.c0[i]; t_P[i] = B.P0[i]; }
    for (int i = tid; i < B.m1; i += D27_THREADS) { t_lo[B.m0 + i] = B.lo1[i]; t_c[B.m0 + i] = B.c1[i]; t_P[B.m0 + i] = B.P1[i]; }
    for (int i = tid; i < B.m2; i += D27_THREADS) { t_lo[B.m0 + B.m1 + i] = B.lo2[i]; t_c[B.m0 + B.m1 + i] = B.c2[i]; t_P[B.m0 + B.m1 + i] = B.P2[i]; }
    lo0 = t_lo; lo1 = t_lo + B.m0; lo2 = t_lo + B.m0 + B.m1;
    c0 = t_c; c1p = t_c + B.m0; c2p = t_c + B.m0 + B.m1;
    P0 = t_P; P1 = t_P + B.m0; P2 = t_P + B.m0 + B.m1;
  }
  __syncthreads();  // (the only workgroup barrier: from here on the waves never exchange data)
  const int64_t nblk = (row_hi - row_lo + D27_NODES - 1) / D27_NODES;
  // Phase A of a block, per thread (row nl = tid / 8, candidate element e = tid % 8): the row's box, the element, its G0 -- in registers
  struct PairPre {
    int64_t pre;
    int32_t len, c1, c2, la, b0;
    bool valid;
    double g[6];
    int32_t slot;  // >= 0: the element's place in the compact scratch (non-affine), < 0: computed in place
  };
  auto phase_a = [&](int64_t blk) -> PairPre {
    PairPre P{0, 0, 1, 1, 0, 0, false, {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, -1};
    const int nl = tid >> 3, e = tid & 7;
    const int64_t row = row_lo + blk * D27_NODES + nl;
    if (row >= row_hi) return P;
    const uint32_t r32 = (uint32_t)row, pl = (uint32_t)B.plane_len, m2 = (uint32_t)B.m2;  // control-point ids fit int32
    const uint32_t q0 = r32 / pl, rem = r32 - q0 * pl, q1 = rem / m2;
    const int gg[3] = {(int)q0 + B.plo, (int)q1, (int)(rem - q1 * m2)};
    const int l0 = lo0[gg[0]], l1 = lo1[gg[1]], l2 = lo2[gg[2]];
    P.c1 = c1p[gg[1]];
    P.c2 = c2p[gg[2]];
    const int cc0 = c0[gg[0]];
    P.pre = (P0[gg[0]] - B.Pplo) * B.S1 * B.S2 + (int64_t)cc0 * (P1[gg[1]] * B.S2 + (int64_t)P.c1 * P2[gg[2]]);  // (brick_prefix)
    P.len = cc0 * P.c1 * P.c2;
    const int ed[3] = {e & 1, (e >> 1) & 1, e >> 2};
    const int ne[3] = {B.ne0, B.ne1, B.ne2};
    int E[3];
    bool valid = true;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      if (gg[d] & 1) {
        E[d] = (gg[d] - 1) >> 1;
        valid = valid && ed[d] == 0;
      } else {
        E[d] = (gg[d] >> 1) - 1 + ed[d];
      }
      valid = valid && E[d] >= 0 && E[d] < ne[d];
    }
    P.valid = valid;
    P.la = valid ? (gg[0] - 2 * E[0]) + 3 * (gg[1] - 2 * E[1]) + 9 * (gg[2] - 2 * E[2]) : 0;
    P.b0 = nl * G27_ROW + ((2 * E[0] - l0) * P.c1 + (2 * E[1] - l1)) * P.c2 + (2 * E[2] - l2);  // the element's first node in the row's box
    if (valid) {
      const int64_t eidx = ((int64_t)(E[0] - elo) * ne[1] + E[1]) * ne[2] + E[2];
      const double* ge = g + eidx * 6;
#pragma unroll
      for (int t = 0; t < 6; ++t) P.g[t] = ge[t];
      if (slot_of) P.slot = slot_of[eidx];
    }
    return P;
  };
  for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    for (int t = lane; t < 8 * G27_ROW; t += 64) rows[wv * 8 * G27_ROW + t] = 0.0;
    PairPre cur = phase_a(blk);  // (its loads wait behind the other waves' arithmetic: two workgroups of eight waves per CU)
    // (round 6) the wave's eight rows -- consecutive control points: back to back in the value array -- sit IN MEMORY ORDER in its LDS block (a row at its
    // prefix minus the first row's, not at a fixed 125-entry stride): the write-out below is one linear copy with 16-byte stores instead of one or two
    // stores of `len` eight-byte lanes per row (27 .. 125 entries: a third of the lanes on average)
    const uint32_t p0lo = __builtin_amdgcn_readlane((uint32_t)(uint64_t)cur.pre, 0), p0hi = __builtin_amdgcn_readlane((uint32_t)((uint64_t)cur.pre >> 32), 0);
    const int64_t pre0 = (int64_t)(((uint64_t)p0hi << 32) | p0lo);
    cur.b0 += (int)(cur.pre - pre0) - (lane >> 3) * G27_ROW;
    __builtin_amdgcn_wave_barrier();
    if (cur.valid && cur.slot < 0) {
      const int a0 = cur.la % 3, a1 = (cur.la / 3) % 3, a2 = cur.la / 9;
      double X0[4][3], X2[4][3];  // [D, M, C, Ct][b]
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          X0[x][b] = sT[(x * 3 + a0) * 4 + b];
          X2[x][b] = sT[(x * 3 + a2) * 4 + b];
        }
      const double g0 = cur.g[0], g1 = cur.g[1], g2 = cur.g[2], g3 = cur.g[3], g4 = cur.g[4], g5 = cur.g[5];
#pragma unroll
      for (int b1 = 0; b1 < 3; ++b1) {
        const double D1 = sT[(0 * 3 + a1) * 4 + b1], M1 = sT[(1 * 3 + a1) * 4 + b1], C1 = sT[(2 * 3 + a1) * 4 + b1], Ct1 = sT[(3 * 3 + a1) * 4 + b1];
#pragma unroll
        for (int b0 = 0; b0 < 3; ++b0) {
          const double D0 = X0[0][b0], M0 = X0[1][b0], C0 = X0[2][b0], Ct0 = X0[3][b0];
          const double Pq = g0 * (D0 * M1) + g1 * (C0 * Ct1 + Ct0 * C1) + g3 * (M0 * D1);
          const double Qq = g5 * (M0 * M1);
          const double Rq = g2 * (C0 * M1) + g4 * (C1 * M0);
          const double Tq = g2 * (Ct0 * M1) + g4 * (Ct1 * M0);
          double* rp = rows + cur.b0 + (b0 * cur.c1 + b1) * cur.c2;
#pragma unroll
          for (int b2 = 0; b2 < 3; ++b2) {
            const double v = X2[1][b2] * Pq + X2[0][b2] * Qq + X2[3][b2] * Rq + X2[2][b2] * Tq;
            __builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double*)(rp + b2), v);
          }
        }
      }
    }
    const uint64_t stored = slot_of ? __ballot(cur.valid && cur.slot >= 0) : 0ull;  // (wave-uniform) pairs whose run sits in the scratch
    if (stored) {
      // a half-wave streams the runs of its 4 rows in, element order e = 0..7 per row: lane lb < 27 loads entry lb of a run and adds it at the slot of local
      // node lb in the row's box; what a lane needs about pair p sits in lane p's registers (the pair's own phase A): fetched with wave shuffles
      __builtin_amdgcn_wave_barrier();
      const int lb = lane & 31, hb = lane & 32;
      const bool active = lb < 27;
      const int bx = lb % 3, by = (lb / 3) % 3, bz = lb / 9;
      const int64_t my_src = ((int64_t)(cur.slot >= 0 ? cur.slot : 0) * 27 + scratch_row(cur.la)) * 27;
      const int my_lo = (int)(uint32_t)(uint64_t)my_src, my_hi = (int)(uint32_t)((uint64_t)my_src >> 32);
      uint32_t todo = hb ? (uint32_t)(stored >> 32) : (uint32_t)stored;
      while (__any(todo != 0u)) {  // (both half-waves take part in every shuffle)
        double v[D27_FLIGHT];
        int sl[D27_FLIGHT];
#pragma unroll
        for (int j = 0; j < D27_FLIGHT; ++j) {
          const bool has = todo != 0u;
          const int p = hb + (has ? __builtin_ctz(todo) : 0);
          if (has) todo &= todo - 1u;
          const uint32_t slo = (uint32_t)__shfl(my_lo, p, MFEM_WAVE), shi = (uint32_t)__shfl(my_hi, p, MFEM_WAVE);
          const int pb0 = __shfl(cur.b0, p, MFEM_WAVE), pc1 = __shfl(cur.c1, p, MFEM_WAVE), pc2 = __shfl(cur.c2, p, MFEM_WAVE);
          sl[j] = -1;
          v[j] = 0.0;
          if (has && active) {
            v[j] = __builtin_nontemporal_load(ke + (int64_t)(((uint64_t)shi << 32) | slo) + lb);
            sl[j] = pb0 + (bx * pc1 + by) * pc2 + bz;
          }
        }
#pragma unroll
        for (int j = 0; j < D27_FLIGHT; ++j)
          if (sl[j] >= 0) __builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double*)(rows + sl[j]), v[j]);
      }
    }
    __builtin_amdgcn_wave_barrier();
    {
      int total = 0;  // entries of the wave's live rows (rows behind row_hi carry len = 0)
#pragma unroll
      for (int r = 0; r < 8; ++r) total += __builtin_amdgcn_readlane(cur.len, 8 * r);
      double* dst = vals + pre0;
      const double* src = rows + wv * 8 * G27_ROW;
      const int head = total > 0 ? (int)(((uintptr_t)dst >> 3) & 1) : 0, np = (total - head) >> 1;
      typedef double d27_d2 __attribute__((ext_vector_type(2)));
      for (int m = lane; m < np; m += 64) {
        const int idx = head + 2 * m;
        __builtin_nontemporal_store(d27_d2{src[idx], src[idx + 1]}, reinterpret_cast<d27_d2*>(dst + idx));
      }
      if (lane == 0 && head) __builtin_nontemporal_store(src[0], dst);
      if (lane == 1 && ((total - head) & 1)) __builtin_nontemporal_store(src[total - 1], dst + total - 1);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- General (non-affine) elements without Ke ever being stored (round 5).  The two-pass path costs 12 ms at 128^3: 129 kflop per element on the matrix cores
// plus a 21.9 GB round trip of element matrices.  Here a row's owner computes the run Ke_e[la][0..26] of each adjacent element e itself, from the element's
// G_q (6 numbers per Gauss point: k_hex27_gq_lane stores them, 1296 bytes per element instead of 5832) by sum factorisation over the tensor-product
// basis: with h[q][n] = sum_m dN[q][la][m] G_q[m][n],
//   Ke[la][b] = sum_q0 ( D(q0,b0) WB0 + L(q0,b0) WB1 ),  WB0 = sum_q1 L(q1,b1) VA0,  WB1 = sum_q1 ( D(q1,b1) VA1 + L(q1,b1) VA2 ),
//   VA0 = sum_q2 h0 L(q2,b2),  VA1 = sum_q2 h1 L(q2,b2),  VA2 = sum_q2 h2 D(q2,b2)
// -- about 1000 FMAs per run, all in the registers of ONE lane (the b-side tables are compile-time constants: ng = 3), 27 k FMAs per element against 65 k MFMA-
// equivalent FMAs + the operand FMAs of the two-pass path, and no scratch round trip.
// Work decomposition: a workgroup takes a tile of 4 x 4 x 4 control points starting on an even lattice point.  Per direction the tile has 6 (node, element) slots
// -- node 0 (even: elements 2T - 1 and 2T), node 1 (element 2T), node 2 (2T and 2T + 1), node 3 (2T + 1) -- so its (row, element) pairs are exactly 6 x 6 x 6 = 216
// jobs on 3 x 3 x 3 elements, whose G_q (35 KB) the workgroup stages in LDS once (2.25 - 3.4 x redundancy against 27 x without staging).  Wave (h0, h1) owns the
// nodes {2 h0, 2 h0 + 1} x {2 h1, 2 h1 + 1} x {0..3}: 3 x 3 x 6 = 54 jobs on 54 of its 64 lanes, and every job of a row sits in the same wave -- the 27 additions of a
// job into the row's box in LDS come in program order (plain read - add - write in batches, see below), lanes of one instruction never meet in an entry (same
// local node b of different elements): the result is reproducible.  The wave's 16 rows leave as four contiguous streams (one per lattice line).  The next tile's G_q
// arrives in a second LDS buffer (global_load_lds) during this tile's arithmetic; the buffer the arithmetic has finished with becomes the tile's row boxes.
namespace r27 {
__host__ __device__ constexpr double gp(int q) { return (q == 0 ? -0.77459666924148337704 : q == 1 ? 0.0 : 0.77459666924148337704) / 2.0 + 0.5; }  // (hex27_upload_tables)
__host__ __device__ constexpr double L(int q, int b) {  // lag2 at Gauss point q
  return b == 0 ? 2.0 * (gp(q) - 0.5) * (gp(q) - 1.0) : b == 1 ? -4.0 * gp(q) * (gp(q) - 1.0) : 2.0 * gp(q) * (gp(q) - 0.5);
}
__host__ __device__ constexpr double D(int q, int b) { return b == 0 ? 4.0 * gp(q) - 3.0 : b == 1 ? -8.0 * gp(q) + 4.0 : 4.0 * gp(q) - 1.0; }
// the row box of a lattice point of an order-2 brick in one direction (what upload_dim_tables of brick.hip tabulates as lo / c / P): first coupled
// point, their number, and the number of box entries of all points in front of it -- closed forms, so that no table load (a memory round trip per tile and
// wave) sits in front of the arithmetic
__device__ __forceinline__ int lo(int g) { return (g & 1) ? g - 1 : (g >= 2 ? g - 2 : 0); }
__device__ __forceinline__ int cnt(int g, int m) { return ((g & 1) ? g + 1 : (g + 2 > m - 1 ? m - 1 : g + 2)) - lo(g) + 1; }
__device__ __forceinline__ int pre(int g) { return g == 0 ? 0 : 3 + 3 * (g >> 1) + 5 * ((g - 1) >> 1); }
}  // namespace r27
// G_q = -k w_q det J_q^-1 J_q^-T of every element (6 numbers per Gauss point, 1296 bytes per element) -> gq[e][q0][q1][q2][6].  ONE LANE per element: the
// sum-factorised Jacobian runs in the lane's registers (contract a0 for this q0: 54 numbers; a1 for this q1: 27; a2 per Gauss point) straight from the
// coordinate arrays -- ~3900 FP64 instructions per element with every lane busy, no LDS traffic, no decode tables (the wave-per-element stages of k_hex27 took
// 2.4 ms for this at 128^3; this kernel: see profiles/r05_hex27_rows.txt).  The 18 numbers of a (q0, q1) pair leave through a small LDS transpose so that the
// stores are runs of 144 bytes per element instead of one 8-byte store per lane 1296 bytes apart.
__global__ __launch_bounds__(256) void k_hex27_gq_lane(BrickView B, const Hex27Tables* __restrict__ tab, double kcond, int elo, int ecnt,
                                                        double* __restrict__ gq) {
  __shared__ double stage[4][64 * 19];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t nel = (int64_t)ecnt * B.ne1 * B.ne2;
  const int64_t e_wave = (int64_t)blockIdx.x * 256 + wv * 64;  // first element of this wave
  if (e_wave >= nel) return;
  const int64_t idx = e_wave + lane < nel ? e_wave + lane : nel - 1;  // (lanes past the end repeat the last element and store nothing)
  const int K = (int)(idx % B.ne2), J = (int)((idx / B.ne2) % B.ne1), I = elo + (int)(idx / ((int64_t)B.ne1 * B.ne2));
  const int64_t c000 = brick_cindex(B, 2 * I, 2 * J, 2 * K);
  double Lt[3][3], Dt[3][3];  // [q][a] (wave-uniform: scalar registers)
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      Lt[q][a] = tab->tab1[0][q][a];
      Dt[q][a] = tab->tab1[1][q][a];
    }
  double* st = stage[wv];
#pragma unroll 1
  for (int q0 = 0; q0 < 3; ++q0) {
    const double l0 = Lt[q0][0], l1 = Lt[q0][1], l2 = Lt[q0][2], d0 = Dt[q0][0], d1 = Dt[q0][1], d2 = Dt[q0][2];
    double TL[9][3], TD[9][3];  // [a1 + 3 a2][i]: values / xi0-derivatives at q0
#pragma unroll
    for (int a12 = 0; a12 < 9; ++a12) {
      const int64_t c = c000 + (int64_t)(a12 % 3) * B.m2 + (a12 / 3);
      const double* Xs[3] = {B.X0, B.X1, B.X2};
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const double x0 = Xs[i][c], x1 = Xs[i][c + B.plane_len], x2 = Xs[i][c + 2 * B.plane_len];
        TL[a12][i] = l0 * x0 + l1 * x1 + l2 * x2;
        TD[a12][i] = d0 * x0 + d1 * x1 + d2 * x2;
      }
    }
#pragma unroll 1
    for (int q1 = 0; q1 < 3; ++q1) {
      const double m0 = Lt[q1][0], m1 = Lt[q1][1], m2 = Lt[q1][2], e0 = Dt[q1][0], e1 = Dt[q1][1], e2 = Dt[q1][2];
      double LL[3][3], LD[3][3], DL[3][3];  // [a2][i]
#pragma unroll
      for (int a2 = 0; a2 < 3; ++a2)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          LL[a2][i] = m0 * TL[3 * a2][i] + m1 * TL[3 * a2 + 1][i] + m2 * TL[3 * a2 + 2][i];
          LD[a2][i] = e0 * TL[3 * a2][i] + e1 * TL[3 * a2 + 1][i] + e2 * TL[3 * a2 + 2][i];
          DL[a2][i] = m0 * TD[3 * a2][i] + m1 * TD[3 * a2 + 1][i] + m2 * TD[3 * a2 + 2][i];
        }
#pragma unroll
      for (int q2 = 0; q2 < 3; ++q2) {
        double Jm[3][3];  // [i][m] = d x_i / d xi_m
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          Jm[i][0] = Lt[q2][0] * DL[0][i] + Lt[q2][1] * DL[1][i] + Lt[q2][2] * DL[2][i];
          Jm[i][1] = Lt[q2][0] * LD[0][i] + Lt[q2][1] * LD[1][i] + Lt[q2][2] * LD[2][i];
          Jm[i][2] = Dt[q2][0] * LL[0][i] + Dt[q2][1] * LL[1][i] + Dt[q2][2] * LL[2][i];
        }
        const double j00 = Jm[0][0], j01 = Jm[0][1], j02 = Jm[0][2], j10 = Jm[1][0], j11 = Jm[1][1], j12 = Jm[1][2], j20 = Jm[2][0], j21 = Jm[2][1],
                     j22 = Jm[2][2];
        const double det = j00 * j11 * j22 - j00 * j12 * j21 - j01 * j10 * j22 + j01 * j12 * j20 + j02 * j10 * j21 - j02 * j11 * j20;
        const double c00 = j11 * j22 - j12 * j21, c01 = j02 * j21 - j01 * j22, c02 = j01 * j12 - j11 * j02;
        const double c10 = j12 * j20 - j22 * j10, c11 = j00 * j22 - j02 * j20, c12 = j02 * j10 - j00 * j12;
        const double c20 = j10 * j21 - j11 * j20, c21 = j01 * j20 - j21 * j00, c22 = j00 * j11 - j10 * j01;
        const double sc = -kcond * tab->w[q0 + 3 * q1 + 9 * q2] / det;  // (as step 2b of k_hex27)
        double* o = st + lane * 19 + 6 * q2;
        o[0] = sc * (c00 * c00 + c01 * c01 + c02 * c02);
        o[1] = sc * (c00 * c10 + c01 * c11 + c02 * c12);
        o[2] = sc * (c00 * c20 + c01 * c21 + c02 * c22);
        o[3] = sc * (c10 * c10 + c11 * c11 + c12 * c12);
        o[4] = sc * (c10 * c20 + c11 * c21 + c12 * c22);
        o[5] = sc * (c20 * c20 + c21 * c21 + c22 * c22);
      }
      __builtin_amdgcn_wave_barrier();
      double* dst = gq + e_wave * 162 + (q0 * 9 + q1 * 3) * 6;
#pragma unroll
      for (int sidx = 0; sidx < 18; ++sidx) {
        const int flat = sidx * 64 + lane, e = flat / 18, j = flat - e * 18;
        if (e_wave + e < nel) dst[(int64_t)e * 162 + j] = st[e * 19 + j];
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// b-side table rows of a Gauss point for the stages whose point index is a loop counter: L(q, 0..2), D(q, 0..2) (read into scalar registers)
__constant__ double c_r27_LD[3][8] = {{r27::L(0, 0), r27::L(0, 1), r27::L(0, 2), r27::D(0, 0), r27::D(0, 1), r27::D(0, 2), 0.0, 0.0},
                                       {r27::L(1, 0), r27::L(1, 1), r27::L(1, 2), r27::D(1, 0), r27::D(1, 1), r27::D(1, 2), 0.0, 0.0},
                                       {r27::L(2, 0), r27::L(2, 1), r27::L(2, 2), r27::D(2, 0), r27::D(2, 1), r27::D(2, 2), 0.0, 0.0}};
#define R27_THREADS 256
#define R27_GD (27 * 162)   // doubles of G_q per tile
#define R27_BUF (R27_GD + 2)  // one LDS buffer: the tile's G_q, later its 64 row boxes ((5 + 3 + 5 + 3)^3 = 4096 doubles)
#define R27_LDS_BYTES (sizeof(double) * (2 * R27_BUF + 32))
__global__ __launch_bounds__(R27_THREADS, 2) void k_hex27_rows_gq(BrickView B, const Hex27Tables* __restrict__ tab, const double* __restrict__ gq,
                                                                   double* __restrict__ vals, int elo, int ecnt, int T0lo, int nT0, int nT1, int nT2, int ablate) {
  extern __shared__ double lds[];
  // two buffers: G_q of this tile (read by the arithmetic; afterwards the same space holds the tile's row boxes) | G_q of the next tile, arriving
  // straight from memory (global_load_lds: no registers in between) while this tile's arithmetic runs
  double* sT = lds + 2 * R27_BUF;  // tab1: [2][4][4]
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), h0 = wv & 1, h1 = wv >> 1;  // (wv in a scalar register: what depends on it alone is scalar arithmetic)
  if (tid < 32) sT[tid] = (&tab->tab1[0][0][0])[tid];
  // ---- this lane's job inside any tile
  const bool job = lane < 54;
  const int s2 = lane % 6, s1 = 3 * h1 + (lane / 6) % 3, s0 = 3 * h0 + (job ? lane / 18 : 0);
  auto slot_t = [](int sl) { return (0xE90 >> (2 * sl)) & 3; };   // node of the slot: 0 0 1 2 2 3
  auto slot_le = [](int sl) { return (0xA54 >> (2 * sl)) & 3; };  // its element, counted from 2T - 1: 0 1 1 1 2 2
  auto slot_a = [](int sl) { return (0x492 >> (2 * sl)) & 3; };   // the node's local index in that element: 2 0 1 2 0 1
  const int t0 = slot_t(s0), t1 = slot_t(s1), t2 = slot_t(s2);
  const int le0 = slot_le(s0), le1 = slot_le(s1), le2 = slot_le(s2);
  const int a0 = slot_a(s0), a1 = slot_a(s1), a2 = slot_a(s2);
  // interior layout of the tile's row boxes per direction: counts 5 3 5 3, offsets 0 5 8 13 (as arithmetic: a table indexed per lane is a load from memory)
  auto CI = [](int tt) { return 5 - 2 * (tt & 1); };
  auto PX = [](int tt) { return 5 * ((tt + 1) >> 1) + 3 * (tt >> 1); };
  const int gjob_off = ((le0 * 3 + le1) * 3 + le2) * 162;
  const int ntiles = nT0 * nT1 * nT2;  // (the host keeps it below 2^31)
  typedef double d2v __attribute__((ext_vector_type(2)));
  auto tile_of = [&](int t, int& T0, int& T1, int& T2) {
    const uint32_t ut = (uint32_t)t, q = ut / (uint32_t)nT2;
    T2 = (int)(ut - q * (uint32_t)nT2);
    const uint32_t q1 = q / (uint32_t)nT1;
    T1 = (int)(q - q1 * (uint32_t)nT1);
    T0 = T0lo + (int)q1;
  };
  // G_q of the tile's 27 elements: the 3 elements of an (l0, l1) column are one contiguous run of 486 doubles = 243 pieces of 16 bytes -- thread tid < 243
  // moves piece tid of each of the 9 columns (column base and validity are wave-uniform; the piece's element along the run is tid / 81).  A wave's 64 pieces
  // land as one 1 KB block (what global_load_lds writes: LDS base + 16 x lane).
  const int l2_of_piece = tid / 81;
  const int64_t sE1 = (int64_t)B.ne2 * 162, sE0 = sE1 * B.ne1;  // doubles between element columns
  auto request = [&](int T0, int T1, int T2, double* buf) {
    const int E2 = 2 * T2 - 1 + l2_of_piece;
    const bool ok2 = tid < 243 && E2 >= 0 && E2 < B.ne2 && !(ablate & 4);
    const double* base = gq + ((int64_t)(2 * T0 - 1 - elo) * sE0 + (int64_t)(2 * T1 - 1) * sE1 + (int64_t)(2 * T2 - 1) * 162) + 2 * tid;
    double* dst = buf + 128 * wv;
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      const int E0 = 2 * T0 - 1 + c / 3, E1 = 2 * T1 - 1 + c % 3;
      if (ok2 && E0 >= elo && E0 < elo + ecnt && E1 >= 0 && E1 < B.ne1)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (c / 3) * sE0 + (c % 3) * sE1),
                                         (__attribute__((address_space(3))) void*)(dst + c * 486), 16, 0, 0);
    }
  };
  // Vector memory operations of a wave complete in the order they were issued (one counter for loads and stores on this chip), so "at most as many
  // outstanding as row stores were issued behind the G_q loads" means the loads have landed -- without waiting for the stores to reach memory, which
  // __syncthreads() does (its s_waitcnt vmcnt(0) cost 1.5 ms of a 5.8 ms kernel: a memory round trip per tile).  The count is wave-uniform: a branch per value.
  auto wait_loads_behind = [](int nstores) {
#define R27_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (nstores) {
      R27_W(0) R27_W(1) R27_W(2) R27_W(3) R27_W(4) R27_W(5) R27_W(6) R27_W(7) R27_W(8) R27_W(9) R27_W(10) R27_W(11) R27_W(12) R27_W(13) R27_W(14) R27_W(15)
      R27_W(16) R27_W(17) R27_W(18) R27_W(19) R27_W(20) R27_W(21) R27_W(22) R27_W(23) R27_W(24) R27_W(25) R27_W(26) R27_W(27) R27_W(28)
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef R27_W
  };
  // Tile order: workgroups with equal blockIdx % 8 share an XCD (round-robin dispatch) and its L2; each XCD walks ONE contiguous eighth of the tiles, its
  // workgroups side by side -- the tiles in flight on an XCD are neighbours along k and the previous lattice line is still in its L2, so most of the 3.4-fold
  // re-reading of G_q (a tile needs 27 elements for the 8 it owns) is served there instead of from memory (grids that are no multiple of 8: plain stride)
  const bool xcd = (gridDim.x & 7) == 0;
  const int tstride = xcd ? (int)(gridDim.x >> 3) : (int)gridDim.x;
  const int tend = xcd ? (int)(((int64_t)ntiles * ((blockIdx.x & 7) + 1)) >> 3) : ntiles;
  int t = xcd ? (int)(((int64_t)ntiles * (blockIdx.x & 7)) >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  int cur = 0, nstores = 0;
  int T0 = 0, T1 = 0, T2 = 0, N0 = 0, N1 = 0, N2 = 0;  // this tile / the next one of this workgroup
  if (t < tend) {
    tile_of(t, N0, N1, N2);
    request(N0, N1, N2, lds);
  }
  for (; t < tend; t += tstride, cur ^= 1) {
    double* sG = lds + (cur ? R27_BUF : 0);
    double* rows = sG;  // (after the arithmetic)
    wait_loads_behind(__builtin_amdgcn_readfirstlane(nstores));  // this wave's share of the tile's G_q has landed
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // ... everybody's; and every wave is done with the other buffer (the previous tile's rows: read before its stores were issued)
    T0 = N0; T1 = N1; T2 = N2;
    if (t + tstride < tend) {
      tile_of(t + tstride, N0, N1, N2);
      request(N0, N1, N2, lds + (cur ? 0 : R27_BUF));
    }
    // ---- the job
    const int g0 = 4 * T0 + t0, g1 = 4 * T1 + t1, g2 = 4 * T2 + t2;
    const int E0 = 2 * T0 - 1 + le0, E1 = 2 * T1 - 1 + le1, E2 = 2 * T2 - 1 + le2;
    const bool valid = job && g0 >= B.plo && g0 < B.phi && g1 < B.m1 && g2 < B.m2 && E0 >= elo && E0 < elo + ecnt && E0 < B.ne0 && E1 >= 0 && E1 < B.ne1 &&
                       E2 >= 0 && E2 < B.ne2 && !(ablate & 1);
    double o[27];
#pragma unroll
    for (int b = 0; b < 27; ++b) o[b] = 0.0;
    if (valid) {
      // 1-D values / derivatives of the row's own local node at the Gauss points: sT[q * 4 + a_d], sT[16 + q * 4 + a_d] -- read where they are used.
      // The loops over q0 and q1 are REAL loops (their b-side table rows come from constant memory into scalar registers): fully unrolled, the compiler moves
      // all 81 G_q reads of the job to the top of one 1200-instruction block and spills the accumulators (484 bytes of scratch per lane); a body of three Gauss
      // points keeps 63 accumulators + 18 G entries + temporaries in registers.
      const double* gjob = sG + gjob_off;
      const double* Ta0 = sT + a0;
      const double* Ta1 = sT + a1;
      const double* Ta2 = sT + a2;
      const double La2[3] = {Ta2[0], Ta2[4], Ta2[8]}, Da2[3] = {Ta2[16], Ta2[20], Ta2[24]};
      // the 9 (q0, q1) pairs in a real loop, two per trip: the 18 G entries and the four table values of the NEXT pair are requested before this pair's
      // arithmetic, into the other of two register sets (A / B: no copies)
      double WB0[9], WB1[9];  // [b1 + 3 b2]
      d2v gA[9], gB[9];
      double tA[4], tB[4];  // La0, Da0, La1, Da1 of the pair
      auto fetch = [&](int it, d2v (&gg)[9], double (&tt)[4]) {
        const int n0 = it / 3, n1 = it - 3 * n0;
#pragma unroll
        for (int u = 0; u < 9; ++u) gg[u] = *reinterpret_cast<const d2v*>(gjob + 18 * it + 2 * u);
        tt[0] = Ta0[n0 * 4]; tt[1] = Ta0[16 + n0 * 4]; tt[2] = Ta1[n1 * 4]; tt[3] = Ta1[16 + n1 * 4];
      };
      auto pair = [&](int it, const d2v (&gc)[9], const double (&tc)[4]) {
        const int q0 = it / 3, q1 = it - 3 * q0;
        if (q1 == 0) {
#pragma unroll
          for (int b = 0; b < 9; ++b) WB0[b] = WB1[b] = 0.0;
        }
        const double p01 = tc[1] * tc[2], p10 = tc[0] * tc[3], p00 = tc[0] * tc[2];
        double VA0[3] = {0.0, 0.0, 0.0}, VA1[3] = {0.0, 0.0, 0.0}, VA2[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int q2 = 0; q2 < 3; ++q2) {
          const d2v g01 = gc[3 * q2], g23 = gc[3 * q2 + 1], g45 = gc[3 * q2 + 2];
          const double u0 = p01 * La2[q2], u1 = p10 * La2[q2], u2 = p00 * Da2[q2];
          const double hh0 = __builtin_fma(u2, g23.x, __builtin_fma(u1, g01.y, u0 * g01.x));
          const double hh1 = __builtin_fma(u2, g45.x, __builtin_fma(u1, g23.y, u0 * g01.y));
          const double hh2 = __builtin_fma(u2, g45.y, __builtin_fma(u1, g45.x, u0 * g23.x));
#pragma unroll
          for (int b2 = 0; b2 < 3; ++b2) {
            VA0[b2] = __builtin_fma(hh0, r27::L(q2, b2), VA0[b2]);
            VA1[b2] = __builtin_fma(hh1, r27::L(q2, b2), VA1[b2]);
            VA2[b2] = __builtin_fma(hh2, r27::D(q2, b2), VA2[b2]);
          }
        }
        const double cL[3] = {c_r27_LD[q1][0], c_r27_LD[q1][1], c_r27_LD[q1][2]}, cD[3] = {c_r27_LD[q1][3], c_r27_LD[q1][4], c_r27_LD[q1][5]};
#pragma unroll
        for (int b2 = 0; b2 < 3; ++b2)
#pragma unroll
          for (int b1 = 0; b1 < 3; ++b1) {
            WB0[b1 + 3 * b2] = __builtin_fma(cL[b1], VA0[b2], WB0[b1 + 3 * b2]);
            WB1[b1 + 3 * b2] = __builtin_fma(cL[b1], VA2[b2], __builtin_fma(cD[b1], VA1[b2], WB1[b1 + 3 * b2]));
          }
        if (q1 == 2) {
          const double eL[3] = {c_r27_LD[q0][0], c_r27_LD[q0][1], c_r27_LD[q0][2]}, eD[3] = {c_r27_LD[q0][3], c_r27_LD[q0][4], c_r27_LD[q0][5]};
#pragma unroll
          for (int b12 = 0; b12 < 9; ++b12)
#pragma unroll
            for (int b0 = 0; b0 < 3; ++b0) o[b0 + 3 * b12] = __builtin_fma(eL[b0], WB1[b12], __builtin_fma(eD[b0], WB0[b12], o[b0 + 3 * b12]));
        }
      };
      fetch(0, gA, tA);
#pragma unroll 1
      for (int k = 0; k < 4; ++k) {
        fetch(2 * k + 1, gB, tB);
        pair(2 * k, gA, tA);
        fetch(2 * k + 2, gA, tA);
        pair(2 * k + 1, gB, tB);
      }
      pair(8, gA, tA);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every wave has read its G_q: the buffer becomes the tile's row boxes (the next tile's loads stay in flight)
    // the wave's rows: four lattice lines (t0, t1) of up to four points each.  The rows of a line follow one another in the CSR values, and so they do in LDS
    // (a row's box starts where the previous one of its line ends: cc0 cc1 x the entries of the line's points in front of it) -- a line is zeroed, and later
    // leaves, as ONE stream of cc0 cc1 (entries along the line) doubles; all of it wave-uniform arithmetic
    const int p2lo = r27::pre(4 * T2);
    const int p2hi = (4 * T2 + 4 < B.m2 ? r27::pre(4 * T2 + 4) : r27::pre(B.m2 - 1) + r27::cnt(B.m2 - 1, B.m2)) - p2lo;  // entries along the line of the tile's points
    int line_len[4], line_base[4];
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int z0 = 2 * h0 + (rg >> 1), z1 = 2 * h1 + (rg & 1), q0 = 4 * T0 + z0, q1 = 4 * T1 + z1;
      line_base[rg] = PX(z0) * 256 + CI(z0) * PX(z1) * 16;
      line_len[rg] = (q0 >= B.plo && q0 < B.phi && q1 < B.m1) ? r27::cnt(q0, B.m0) * r27::cnt(q1, B.m1) * p2hi : 0;
      for (int i = lane; i < line_len[rg]; i += 64) rows[line_base[rg] + i] = 0.0;
    }
    __builtin_amdgcn_wave_barrier();
    if (valid && !(ablate & 2)) {
      // Into the row's box with plain read - add - write (ds_add_f64 costs ~2.4 cycles per LANE on this chip: 150 cycles per instruction, the whole kernel's time
      // when the 27 additions of a job are atomics).  Safe because every job of a row sits in THIS wave and the wave's LDS operations execute in order: the
      // lanes of one instruction hold the same local node b of different (row, element) pairs -- different entries --, and two entries b, b' of different
      // jobs coincide only where b_d = 2 meets b'_d = 0 in some direction.  The 27 entries go in 8 batches by the set of directions with b_d = 2 (8 + 3 x 4 +
      // 3 x 2 + 1): no two entries of one batch can meet, so a batch is all its reads, then the additions, then all its writes.
      const int c1 = r27::cnt(g1, B.m1), c2 = r27::cnt(g2, B.m2), c12 = c1 * c2;
      const int row_off = PX(t0) * 256 + CI(t0) * PX(t1) * 16 + r27::cnt(g0, B.m0) * c1 * (r27::pre(g2) - p2lo);  // the row's box: behind those of its line's points in front of it
      double* rp = rows + row_off + ((2 * E0 - r27::lo(g0)) * c1 + (2 * E1 - r27::lo(g1))) * c2 + (2 * E2 - r27::lo(g2));  // the element's first node in the row's box
#pragma unroll
      for (int M = 0; M < 8; ++M) {
        double curv[8];
        int n = 0;
#pragma unroll
        for (int b = 0; b < 27; ++b) {
          const int b0 = b % 3, b1 = (b / 3) % 3, b2 = b / 9;
          if (((b0 == 2) | ((b1 == 2) << 1) | ((b2 == 2) << 2)) == M) curv[n++] = rp[b0 * c12 + b1 * c2 + b2];
        }
        n = 0;
#pragma unroll
        for (int b = 0; b < 27; ++b) {
          const int b0 = b % 3, b1 = (b / 3) % 3, b2 = b / 9;
          if (((b0 == 2) | ((b1 == 2) << 1) | ((b2 == 2) << 2)) == M) rp[b0 * c12 + b1 * c2 + b2] = curv[n++] + o[b];
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- the wave's four lines leave
    nstores = 0;
    {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        nstores += (line_len[rg] + 63) >> 6;
        const int z0 = 2 * h0 + (rg >> 1), z1 = 2 * h1 + (rg & 1), q0 = 4 * T0 + z0, q1 = 4 * T1 + z1;
        const int cc0 = r27::cnt(q0, B.m0), cc1 = r27::cnt(q1, B.m1);
        double* dst = vals + ((int64_t)r27::pre(q0) - B.Pplo) * B.S1 * B.S2 + (int64_t)cc0 * ((int64_t)r27::pre(q1) * B.S2 + (int64_t)cc1 * p2lo);  // (brick_prefix of the line's first point)
        const double* src = rows + line_base[rg];
        for (int i = lane; i < line_len[rg]; i += 64) __builtin_nontemporal_store(src[i], dst + i);
      }
    }
  }
}

// ---- Robin faces (hex-27): one thread per (boundary face element, face node a) = one row of the 9 x 9 face matrix;
// 9 face nodes, ng x ng Gauss points.  colour = parity of the face element in its two tangential directions; the two
// opposite faces of a direction share no node and go into the same launch (side = -1): 4 launches per direction.
struct Face27Args {
  BrickView B;
  const Hex27Tables* tab;
  double h, Tenv;
  int nd, side, colour, ng;
};

template <bool MATRIX>
__global__ __launch_bounds__(MFEM_BLOCK) void k_hex27_faces(Face27Args A, const double* __restrict__ xstar,
                                                              double* __restrict__ out) {
  const BrickView& B = A.B;
  const int ne[3] = {B.ne0, B.ne1, B.ne2};
  const int t1 = (A.nd + 1) % 3, t2 = (A.nd + 2) % 3;
  const int c1 = A.colour & 1, c2 = A.colour >> 1;
  const int n1 = (ne[t1] - c1 + 1) >> 1, n2 = (ne[t2] - c2 + 1) >> 1;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int a = (int)(tid % 9);
  int64_t f = tid / 9;
  int side = A.side;
  if (side < 0) {
    side = (int)(f & 1);
    f >>= 1;
  }
  if (n1 <= 0 || n2 <= 0 || f >= (int64_t)n1 * n2) return;
  int E[3];
  E[A.nd] = side ? ne[A.nd] - 1 : 0;
  E[t1] = 2 * (int)(f % n1) + c1;
  E[t2] = 2 * (int)(f / n1) + c2;
  if (A.nd == 0) {  // slab: the face lies in one control-point plane, the tangential faces span three
    const int gp = side ? 2 * ne[0] : 0;
    if (gp < B.plo || gp >= B.phi) return;
  } else if (2 * E[0] + 2 < B.plo || 2 * E[0] >= B.phi) {
    return;
  }
  int g[9][3];
  double Xf[9][3], Tf[9];
  for (int c = 0; c < 9; ++c) {
    g[c][A.nd] = side ? 2 * ne[A.nd] : 0;
    g[c][t1] = 2 * E[t1] + c % 3;
    g[c][t2] = 2 * E[t2] + c / 3;
    const int64_t ci = brick_cindex(B, g[c][0], g[c][1], g[c][2]);
    Xf[c][0] = B.X0[ci];
    Xf[c][1] = B.X1[ci];
    Xf[c][2] = B.X2[ci];
    if (!MATRIX) Tf[c] = xstar[brick_xindex(B, 0, g[c][0], g[c][1], g[c][2])];
  }
  {
    if (g[a][0] < B.plo || g[a][0] >= B.phi) return;  // only owned rows
    double macc[9];
    for (int b = 0; b < 9; ++b) macc[b] = 0.0;
    double racc = 0.0;
    for (int q = 0; q < A.ng * A.ng; ++q) {
      double ta[3] = {0, 0, 0}, tb[3] = {0, 0, 0};
      for (int c = 0; c < 9; ++c)
        for (int i = 0; i < 3; ++i) {
          ta[i] += A.tab->fdN[q][c][0] * Xf[c][i];
          tb[i] += A.tab->fdN[q][c][1] * Xf[c][i];
        }
      const double r0 = ta[1] * tb[2] - ta[2] * tb[1], r1 = -ta[0] * tb[2] + ta[2] * tb[0], r2 = ta[0] * tb[1] - ta[1] * tb[0];
      const double ws = A.tab->fw[q] * sqrt(r0 * r0 + r1 * r1 + r2 * r2);
      const double na = A.tab->fN[q][a];
      if (MATRIX) {
        for (int b = 0; b < 9; ++b) macc[b] += -A.h * ws * na * A.tab->fN[q][b];
      } else {
        double Tq = 0.0;
        for (int c = 0; c < 9; ++c) Tq += A.tab->fN[q][c] * Tf[c];
        racc += ws * na * A.h * (A.Tenv - Tq);
      }
    }
    if (MATRIX) {
      const int gi = g[a][0], gj = g[a][1], gk = g[a][2];
      const int64_t base = brick_prefix(B, gi, gj, gk);
      const int lo0 = B.lo0[gi], lo1 = B.lo1[gj], lo2 = B.lo2[gk], cc1 = B.c1[gj], cc2 = B.c2[gk];
      for (int b = 0; b < 9; ++b)
        out[base + ((int64_t)(g[b][0] - lo0) * cc1 + (g[b][1] - lo1)) * cc2 + (g[b][2] - lo2)] += macc[b];
    } else {
      out[(int64_t)(g[a][0] - B.plo) * B.plane_len + (int64_t)g[a][1] * B.m2 + g[a][2]] += racc;
    }
  }
}

// 0 (default): colour-partitioned RMW scatter; 1: Ke -> element-major scratch (MFMA kernel, no colours, no RMW) +
// row-owner gather.  Measured at 128^3 (profiles/r01_hex27_mfma_counters.txt): scatter 19.7-22.9 ms; two-pass 30.9 ms
// (MFMA pass 11.8 ms + gather 17.4 ms, the gather being bound by its per-slot index arithmetic).
static std::atomic<int> g_hex27_two_pass{1};
static std::atomic<long long> g_hex27_direct_count{0};  // assemblies that took the scratch-free path (tests)
extern "C" int64_t mfem_debug_hex27_direct_count(void) { return g_hex27_direct_count; }
static std::atomic<int> g_hex27_direct{1};  // bit 9 of mfem_debug_set_hex27 turns the scratch-free assembly of all-affine meshes off (two-pass MFMA path then)
static std::atomic<int> g_hex27_affine{1};  // bit 8 of mfem_debug_set_hex27 turns the affine-element shortcut of the matrix kernel off (every element then takes the general path)
static std::atomic<int> g_hex27_mixed{1};       // bit 10 of mfem_debug_set_hex27 turns the per-element choice off: a mesh with a non-affine element then takes the two-pass path whole (round 4's behaviour)
static std::atomic<int> g_hex27_mixed_max{80};  // bits 24-30: percentage of non-affine elements up to which the per-element choice is taken (0 = the default 80: profiles/r05_hex27_mixed.txt -- at 75 % the choice takes 11.2 ms against 11.9 for the two-pass path, at 100 % 13.1 against 12.2)
static std::atomic<long long> g_hex27_mixed_count{0};  // assemblies that took it with at least one stored element (tests)
extern "C" int64_t mfem_debug_hex27_mixed_count(void) { return g_hex27_mixed_count; }
static std::atomic<int> g_hex27_rows{1};        // bit 11 of mfem_debug_set_hex27 turns the row-owner kernel of general elements (k_hex27_rows_gq) off
static std::atomic<int> g_hex27_rows_min{0};    // bits 2-7: percentage of non-affine elements FROM which it is taken (0 = the default, R27_MIN_PERCENT; below: the per-element choice)
static std::atomic<int> g_hex27_rows_ablate{0};  // bits 12-14: TIMING-ONLY ablations of k_hex27_rows_gq (wrong values): 1 no arithmetic, 2 no LDS additions, 4 no G_q loads
static std::atomic<long long> g_hex27_rows_count{0};  // assemblies that took it (tests)
extern "C" int64_t mfem_debug_hex27_rows_count(void) { return g_hex27_rows_count; }
#define R27_MIN_PERCENT 30
static std::atomic<int> g_hex27_chunk_planes{0};  // bits 16-23 of mfem_debug_set_hex27: element planes per scratch chunk (0 = from the budget)
static std::atomic<size_t> g_hex27_scratch_budget{(size_t)16 << 30};
extern "C" int mfem_debug_set_hex27(int two_pass) try {
  ++mfem_debug_epoch;
  g_hex27_two_pass = (two_pass & 3) == 0 ? 1 : (two_pass & 3);  // 0 / 1 two-pass (default), 2 FP64 atomics, 3 colour scatter
  g_hex27_chunk_planes = (two_pass >> 16) & 255;
  g_hex27_affine = ((two_pass >> 8) & 1) ? 0 : 1;
  g_hex27_direct = ((two_pass >> 9) & 1) ? 0 : 1;
  g_hex27_mixed = ((two_pass >> 10) & 1) ? 0 : 1;
  g_hex27_rows = ((two_pass >> 11) & 1) ? 0 : 1;
  g_hex27_rows_min = (two_pass >> 2) & 63;
  g_hex27_rows_ablate = (two_pass >> 12) & 7;
  g_hex27_mixed_max = ((two_pass >> 24) & 127) ? ((two_pass >> 24) & 127) : 80;
  return MFEM_OK;
} MFEM_API_CATCH("mfem_debug_set_hex27")

// mode: 0 residual, 1 matrix with colour scatter / atomics (row descriptors per wave), 2 matrix -> scratch
static size_t hex27_lds_bytes(int ng, int mode) {
  const int nq = ng * ng * ng, NI = mode == 0 ? 5 : 3;
  return sizeof(double) * ((size_t)(mode == 0 ? 0 : (H27_NQP(nq) + 1) * 81) + ((nq + 1) & ~1) + 8 * ng + (h27_pad(H27_NDEC) >> 1) +
                           H27_WAVES * (size_t)(W_SIZE(mode == 1)));
}

static int hex27_launch_faces(mfem_context_s* ctx, mfem_brick_s* m, bool matrix, double h, double Tenv, uint32_t robin,
                              const double* xstar, double* out) {
  if (h == 0.0 || robin == 0u) return MFEM_OK;
  BrickView B = mfem_brick_view(m, 1);
  for (int nd = 0; nd < 3; ++nd) {
    const int id_lo = (nd == 0) ? 5 : (nd == 1) ? 2 : 1, id_hi = (nd == 0) ? 3 : (nd == 1) ? 4 : 6;
    const bool lo = robin & (1u << (id_lo - 1)), hi = robin & (1u << (id_hi - 1));
    // both faces of the direction in one launch (side = -1) when both carry the condition
    for (int pass = 0; pass < 2; ++pass) {
      int side;
      if (lo && hi) {
        if (pass) break;
        side = -1;
      } else {
        side = pass;
        if (!(side ? hi : lo)) continue;
      }
      const int t1 = (nd + 1) % 3, t2 = (nd + 2) % 3;
      for (int colour = 0; colour < 4; ++colour) {
        const int n1 = (m->ne[t1] - (colour & 1) + 1) >> 1, n2 = (m->ne[t2] - (colour >> 1) + 1) >> 1;
        if (n1 <= 0 || n2 <= 0) continue;
        Face27Args A{B, g_tab, h, Tenv, nd, side, colour, m->ng};
        const int64_t nthreads = (int64_t)n1 * n2 * 9 * (side < 0 ? 2 : 1);
        const int grid = (int)((nthreads + MFEM_BLOCK - 1) / MFEM_BLOCK);
        if (matrix)
          hipLaunchKernelGGL(k_hex27_faces<true>, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, A, xstar, out);
        else
          hipLaunchKernelGGL(k_hex27_faces<false>, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, A, xstar, out);
        MFEM_CHECK_LAUNCH();
      }
    }
  }
  return MFEM_OK;
}

// Element planes (dimension 0) that touch the owned control-point planes [plo, phi) of a slab.  Slabs start and end on
// element boundaries (mfem_brick_set_slab), so the first owned plane also needs the element plane below it.
static void hex27_element_planes(const mfem_brick_s* m, int* elo, int* ehi) {
  *elo = m->plo / 2 - 1 < 0 ? 0 : m->plo / 2 - 1;
  *ehi = m->phi / 2 > m->ne[0] ? m->ne[0] : m->phi / 2;
}
// elements of one parity colour within the element planes [elo, ehi)
static int64_t hex27_colour_count(const mfem_brick_s* m, int colour, int elo, int ehi) {
  const int o0 = elo + (((colour & 1) - elo) & 1);
  const int64_t n0 = o0 < ehi ? (ehi - o0 + 1) >> 1 : 0, n1 = (m->ne[1] - ((colour >> 1) & 1) + 1) >> 1,
                n2 = (m->ne[2] - (colour >> 2) + 1) >> 1;
  return n0 * n1 * n2;
}

int mfem_hex27_assemble_thermal(mfem_context_s* ctx, mfem_brick_s* m, mfem_csr_s* Acsr, const mfem_thermal_params* p,
                                double* vals) {
  const bool slab = !(m->plo == 0 && m->phi == m->m[0]);
  MFEM_REQUIRE(!slab || g_hex27_two_pass == 1, "hex-27 slabs are assembled by the two-pass variant only");
  int elo, ehi;
  hex27_element_planes(m, &elo, &ehi);
  const int nq = m->ng * m->ng * m->ng;
  int rc = hex27_upload_tables(m->ng);
  if (rc) return rc;
  const size_t lds = hex27_lds_bytes(m->ng, g_hex27_two_pass == 1 ? 2 : 1);
  MFEM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_hex27<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  BrickView B = mfem_brick_view(m, 1);
  if (g_hex27_two_pass == 2) {
    MFEM_CHECK_HIP(hipMemsetAsync(vals, 0, sizeof(double) * (size_t)Acsr->nnz, ctx->stream));
    const int64_t nel = (int64_t)m->ne[0] * m->ne[1] * m->ne[2];
    Hex27Args A{B, g_tab, p->k, g_hex27_affine, -2, nq, m->ng, 0, m->ne[0], 1};
    int64_t grid = (nel + H27_WAVES - 1) / H27_WAVES;
    const int64_t cap = (int64_t)ctx->num_cus * 2;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(k_hex27<true>, dim3((int)grid), dim3(H27_THREADS), lds, ctx->stream, A, nullptr, nullptr, vals);
    MFEM_CHECK_LAUNCH();
    return hex27_launch_faces(ctx, m, true, p->h, p->Tenv, p->robin_faces, nullptr, vals);
  }
  if (g_hex27_two_pass == 1 && g_hex27_direct && g_hex27_affine && m->n_owned < ((int64_t)1 << 31)) {
    // all elements affine?  G0 of every element + a count of the ones that are not (one 4-byte read-back per assembly: the coordinates belong to the caller,
    // mfem_brick_coords, and may have changed since the last call)
    const int64_t nel = (int64_t)(ehi - elo) * m->ne[1] * m->ne[2];
    // workspace: G0 [6 nel] | slot [nel] | elist [nel] (int32) -- the compact scratch of the non-affine elements follows once their number is known
    const size_t g_bytes = sizeof(double) * 6 * (size_t)nel, map_bytes = (sizeof(int32_t) * (size_t)nel + 255) & ~(size_t)255;
    rc = mfem_ws_reserve(ctx, g_bytes + 2 * map_bytes);
    if (rc) return rc;
    int32_t* d_cnt = ctx->d_flags + 14;
    MFEM_CHECK_HIP(hipMemsetAsync(d_cnt, 0, sizeof(int32_t), ctx->stream));
    hipLaunchKernelGGL(k_hex27_affine_g0, dim3((unsigned)((nel + MFEM_BLOCK - 1) / MFEM_BLOCK)), dim3(MFEM_BLOCK), 0, ctx->stream, B, p->k, elo, ehi - elo,
                       (double*)ctx->ws, d_cnt, (int32_t*)((char*)ctx->ws + g_bytes), (int32_t*)((char*)ctx->ws + g_bytes + map_bytes));
    MFEM_CHECK_LAUNCH();
    MFEM_CHECK_HIP(hipMemcpyAsync(ctx->h_flags + 14, d_cnt, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    MFEM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    const int64_t n_stored = ctx->h_flags[14];
    // Per-element choice (round 5; until then ONE distorted element sent the whole mesh through the two-pass path: a 2.6x cliff).  Affine elements are
    // computed in place by the row-owner gather, the others go through pass 1 into a scratch that holds ONLY them and are streamed in by the same gather.
    // Beyond g_hex27_mixed_max (80 % of the elements by default) the plain two-pass path is the faster one (its gather streams every run with no
    // arithmetic beside it); a scratch beyond the budget goes there too (it rings over element planes).
    // Mostly general elements (round 5): G_q of every element (1296 bytes each) -> the row owners compute their runs from it, no Ke anywhere
    // (k_hex27_rows_gq; three Gauss points per direction -- its b-side tables are compile-time constants).
    const size_t gq_bytes = sizeof(double) * 6 * (size_t)nq * (size_t)nel;
    if (g_hex27_rows && m->ng == 3 && n_stored * 100 >= nel * (int64_t)(g_hex27_rows_min ? (int)g_hex27_rows_min : R27_MIN_PERCENT) && n_stored > 0 &&
        gq_bytes <= g_hex27_scratch_budget && g_hex27_chunk_planes == 0) {
      rc = mfem_ws_reserve(ctx, gq_bytes);
      if (rc) return rc;
      hipLaunchKernelGGL(k_hex27_gq_lane, dim3((unsigned)((nel + 255) / 256)), dim3(256), 0, ctx->stream, B, (const Hex27Tables*)g_tab, p->k, elo, ehi - elo,
                         (double*)ctx->ws);
      MFEM_CHECK_LAUNCH();
      MFEM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_hex27_rows_gq), hipFuncAttributeMaxDynamicSharedMemorySize, (int)R27_LDS_BYTES));
      const int T0lo = m->plo / 4, nT0 = (m->phi - 1) / 4 - T0lo + 1, nT1 = (m->m[1] + 3) / 4, nT2 = (m->m[2] + 3) / 4;
      const int64_t ntiles = (int64_t)nT0 * nT1 * nT2;
      const int gridr = (int)(ntiles < (int64_t)ctx->num_cus * 2 ? ntiles : (int64_t)ctx->num_cus * 2);  // two 4-wave workgroups per CU (67 KB of LDS each), persistent
      hipLaunchKernelGGL(k_hex27_rows_gq, dim3(gridr), dim3(R27_THREADS), R27_LDS_BYTES, ctx->stream, B, (const Hex27Tables*)g_tab, (const double*)ctx->ws, vals, elo,
                         ehi - elo, T0lo, nT0, nT1, nT2, (int)g_hex27_rows_ablate);
      MFEM_CHECK_LAUNCH();
      ++g_hex27_rows_count;
      return hex27_launch_faces(ctx, m, true, p->h, p->Tenv, p->robin_faces, nullptr, vals);
    }
    const size_t stored_bytes = sizeof(double) * 729 * (size_t)n_stored;
    if (n_stored == 0 || (g_hex27_mixed && n_stored * 100 <= nel * (int64_t)g_hex27_mixed_max && stored_bytes <= g_hex27_scratch_budget)) {
      const size_t head = (g_bytes + 2 * map_bytes + 255) & ~(size_t)255;
      if (n_stored) {
        // (growing the workspace moves it: the tables just made are small -- made again after the move instead of copied)
        if (ctx->ws_bytes < head + stored_bytes) {
          rc = mfem_ws_reserve(ctx, head + stored_bytes);
          if (rc) return rc;
          MFEM_CHECK_HIP(hipMemsetAsync(d_cnt, 0, sizeof(int32_t), ctx->stream));
          hipLaunchKernelGGL(k_hex27_affine_g0, dim3((unsigned)((nel + MFEM_BLOCK - 1) / MFEM_BLOCK)), dim3(MFEM_BLOCK), 0, ctx->stream, B, p->k, elo,
                             ehi - elo, (double*)ctx->ws, d_cnt, (int32_t*)((char*)ctx->ws + g_bytes), (int32_t*)((char*)ctx->ws + g_bytes + map_bytes));
          MFEM_CHECK_LAUNCH();
        }
        MFEM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_hex27<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        Hex27Args A{B, g_tab, p->k, 0, -1, nq, m->ng, elo, ehi - elo, 1, (const int32_t*)((char*)ctx->ws + g_bytes + map_bytes), n_stored};
        int64_t grid1 = (n_stored + H27_WAVES - 1) / H27_WAVES;
        if (grid1 > (int64_t)ctx->num_cus * 2) grid1 = (int64_t)ctx->num_cus * 2;
        hipLaunchKernelGGL((k_hex27<true, true>), dim3((int)grid1), dim3(H27_THREADS), lds, ctx->stream, A, nullptr, nullptr, (double*)((char*)ctx->ws + head));
        MFEM_CHECK_LAUNCH();
        ++g_hex27_mixed_count;
      }
      MFEM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_hex27_direct), hipFuncAttributeMaxDynamicSharedMemorySize, (int)D27_LDS_BYTES));
      const int64_t nblk = (m->n_owned + D27_NODES - 1) / D27_NODES;
      const int grid = (int)(nblk < (int64_t)ctx->num_cus * 2 ? nblk : (int64_t)ctx->num_cus * 2);  // two 8-wave workgroups per CU (78 KB of LDS each), persistent
      hipLaunchKernelGGL(k_hex27_direct, dim3(grid), dim3(D27_THREADS), D27_LDS_BYTES, ctx->stream, B, (const Hex27Tables*)g_tab, (const double*)ctx->ws, vals,
                         (int64_t)0, m->n_owned, elo, n_stored ? (const int32_t*)((char*)ctx->ws + g_bytes) : (const int32_t*)nullptr,
                         (const double*)((char*)ctx->ws + head));
      MFEM_CHECK_LAUNCH();
      if (!n_stored) ++g_hex27_direct_count;
      return hex27_launch_faces(ctx, m, true, p->h, p->Tenv, p->robin_faces, nullptr, vals);
    }
  }
  if (g_hex27_two_pass == 1) {
    // pass 1: every element's Ke on the matrix cores -> element-major scratch; pass 2: row-owner gather -> CSR.
    // The scratch is a ring of element planes (dimension 0): a chunk computes planes [a, b) and gathers the control-point
    // planes [2a, 2b) (the last chunk also 2b), which need element planes a-1 .. b-1 -- plane a-1 is still in the ring.
    const int64_t plane_el = (int64_t)m->ne[1] * m->ne[2];
    const size_t plane_bytes = sizeof(double) * 729 * (size_t)plane_el;
    const int npl = ehi - elo;
    int P = npl;
    if (g_hex27_chunk_planes > 0) P = g_hex27_chunk_planes;
    else if (plane_bytes * (size_t)npl > g_hex27_scratch_budget) P = (int)(g_hex27_scratch_budget / plane_bytes) - 1;
    if (P < 1) P = 1;
    if (P > npl) P = npl;
    const int ring = P >= npl ? npl : P + 1;
    rc = mfem_ws_reserve(ctx, plane_bytes * (size_t)ring);
    if (rc) return rc;
    MFEM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_hex27<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t plane_rows = B.plane_len;
    for (int a = elo; a < ehi; a += P) {
      const int b = a + P < ehi ? a + P : ehi;
      Hex27Args A{B, g_tab, p->k, g_hex27_affine, -1, nq, m->ng, a, b - a, ring};
      int64_t grid = ((b - a) * plane_el + H27_WAVES - 1) / H27_WAVES;
      const int64_t cap = (int64_t)ctx->num_cus * 2;
      if (grid > cap) grid = cap;
      hipLaunchKernelGGL((k_hex27<true, true>), dim3((int)grid), dim3(H27_THREADS), lds, ctx->stream, A, nullptr, nullptr, (double*)ctx->ws);
      MFEM_CHECK_LAUNCH();
      // control-point planes [2a, 2b) (the last chunk: up to phi), clipped to the owned planes
      const int gp_lo = 2 * a < m->plo ? m->plo : 2 * a, gp_hi = b == ehi ? m->phi : 2 * b;
      const int64_t row_lo = (int64_t)(gp_lo - m->plo) * plane_rows, row_hi = (int64_t)(gp_hi - m->plo) * plane_rows;
      if (row_hi <= row_lo) continue;
      hipLaunchKernelGGL(k_hex27_gather_lds, dim3((unsigned)((row_hi - row_lo + G27_NODES - 1) / G27_NODES)), dim3(MFEM_BLOCK), 0,
                         ctx->stream, B, (const double*)ctx->ws, vals, row_lo, row_hi, ring);
      MFEM_CHECK_LAUNCH();
    }
    return hex27_launch_faces(ctx, m, true, p->h, p->Tenv, p->robin_faces, nullptr, vals);
  }
  MFEM_CHECK_HIP(hipMemsetAsync(vals, 0, sizeof(double) * (size_t)Acsr->nnz, ctx->stream));
  for (int colour = 0; colour < 8; ++colour) {
    const int64_t ne = hex27_colour_count(m, colour, elo, ehi);
    if (ne <= 0) continue;
    Hex27Args A{B, g_tab, p->k, g_hex27_affine, colour, nq, m->ng, elo, ehi - elo, 1};
    int64_t grid = (ne + H27_WAVES - 1) / H27_WAVES;
    const int64_t cap = (int64_t)ctx->num_cus * 2;  // 2 workgroups (16 waves) per CU, persistent over the colour's elements
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(k_hex27<true>, dim3((int)grid), dim3(H27_THREADS), lds, ctx->stream, A, nullptr, nullptr, vals);
    MFEM_CHECK_LAUNCH();
  }
  return hex27_launch_faces(ctx, m, true, p->h, p->Tenv, p->robin_faces, nullptr, vals);
}

int mfem_hex27_residual_thermal(mfem_context_s* ctx, mfem_brick_s* m, const mfem_thermal_params* p, const double* x_star,
                                const double* s, double* residue) {
  int elo, ehi;
  hex27_element_planes(m, &elo, &ehi);
  const int nq = m->ng * m->ng * m->ng;
  int rc = hex27_upload_tables(m->ng);
  if (rc) return rc;
  const size_t lds = hex27_lds_bytes(m->ng, 0);
  MFEM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_hex27<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  MFEM_CHECK_HIP(hipMemsetAsync(residue, 0, sizeof(double) * (size_t)m->n_owned, ctx->stream));
  BrickView B = mfem_brick_view(m, 1);
  for (int colour = 0; colour < 8; ++colour) {
    const int64_t ne = hex27_colour_count(m, colour, elo, ehi);
    if (ne <= 0) continue;
    Hex27Args A{B, g_tab, p->k, g_hex27_affine, colour, nq, m->ng, elo, ehi - elo, 1};
    int64_t grid = (ne + H27_WAVES - 1) / H27_WAVES;
    const int64_t cap = (int64_t)ctx->num_cus * 2;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(k_hex27<false>, dim3((int)grid), dim3(H27_THREADS), lds, ctx->stream, A, x_star, s, residue);
    MFEM_CHECK_LAUNCH();
  }
  return hex27_launch_faces(ctx, m, false, p->h, p->Tenv, p->robin_faces, x_star, residue);
}
