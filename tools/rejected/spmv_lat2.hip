// Order-2 lattice layout for the Krylov loop (solver layout mode 3 when the pattern qualifies; otherwise the row-sorted sliced ELL of
// spmv_sell.hip).  The hex-27 matrix of a structured brick couples a node (i, j, k) of the (2 ne + 1)^3 lattice to the box
// [i - Ei, i + Ei] x [j - Ej, j + Ej] x [k - Ek, k + Ek] with half-width 2 in a direction where the index is even (element boundary)
// and 1 where it is odd -- rows of 125 / 75 / 45 / 27 entries for vertex / edge / face / centre nodes.  The caller's contract stays CSR
// (mul!(b, A, x), 04_GPU_Utils.jl:131); per solve the working values are copied once (like the reference's K_total[K_val_ids] gather,
// 02_Preconditioner.jl:35) into a region-major layout and the SpMV runs like the patch sweep of spmv_ell.hip:
//   * a WAVE owns a region of 8 lattice lines x 32 points and sweeps it through consecutive lattice planes; per plane it handles the
//     four in-plane node types one after the other, lane <-> one of the type's 4 x 16 nodes of the region (two lattice points apart);
//   * values: pv[plane][region][type][slot][64 lanes], slot = position in the type's box in ascending-column order, explicit zero
//     where the box sticks out of the lattice: unit-stride 512-byte loads, no column stream (col = row + offset by construction);
//   * x: the (8 + 4) x (32 + 4) neighbourhood of the region enters LDS once per plane (a ring of five planes) and serves every product
//     of five plane steps -- the sliced-ELL kernel gathered x from memory once per node type and moved 11.3 GB per SpMV for 8.9 GB of
//     matrix and vectors (profiles/r02_sell_experiments.txt);
//   * the row sum runs in slot (= column) order in registers; one-wave workgroups, no barrier beyond the wave's own LDS ordering;
//     runs = (region, segment of planes), XCD c sweeping a contiguous eighth of the regions.
// Rows of the first / last two lattice planes (their boxes leave the vector or reach into the ghost planes of a slab) are computed
// from the caller's CSR arrays by a wave-per-row kernel.  Qualification is an inspection of the CSR pattern (every entry of every swept
// row must sit in its node type's box); nothing about the mesh is assumed.
#include <vector>

#include "blas1.h"

extern int64_t g_layout_min_rows_cols;  // spmv_ell.hip
int g_lat2_enable = 1;                  // bit 2 of mfem_debug_set_sell turns the lattice layout off (spmv_sell.hip)

#define L2_RJ 8    // lattice lines per region
#define L2_RK 32   // lattice points per line segment
#define L2_XL (L2_RJ + 4)
#define L2_XW 36   // 32 + 4 neighbourhood columns
#define L2_XN (L2_XL * L2_XW)   // 432 staged x entries per plane
#define L2_WG_PER_CU 8

struct Lat2Geom {
  int64_t PL, n;
  int m1, m2, p0, p1, NRj, NRk, nseg;
};
__host__ __device__ __forceinline__ int l2_hw(int parity) { return parity ? 1 : 2; }  // half-width of the box in a direction
// doubles of one (plane, region) block: 64 lanes x (2 Ei + 1) x (25 + 15 + 15 + 9) slots
__host__ __device__ __forceinline__ int64_t l2_block(int plane) { return (int64_t)64 * (2 * l2_hw(plane & 1) + 1) * 64; }
// start of plane `plane`'s blocks: planes p0 .. plane - 1 come first, NR regions each
__host__ __device__ __forceinline__ int64_t l2_plane_base(int plane, int p0, int64_t NR) {
  const int cnt = plane - p0, odd_first = p0 & 1;
  const int n_odd = odd_first ? (cnt + 1) / 2 : cnt / 2, n_even = cnt - n_odd;
  return NR * 4096 * ((int64_t)n_even * 5 + (int64_t)n_odd * 3);
}
// first slot of type t = 2 pj + pk inside a block, in units of (2 Ei + 1) * 64 doubles
__host__ __device__ __forceinline__ int l2_type_base(int t) { return t == 0 ? 0 : t == 1 ? 25 : t == 2 ? 40 : 55; }

// ---- pattern check (once per pattern): every entry of every row of the swept planes sits in the row's box
template <typename RP>
__global__ __launch_bounds__(MFEM_BLOCK) void k_lat2_check(Lat2Geom G, const RP* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                             int base, int32_t* __restrict__ bad) {
  const int64_t r0 = (int64_t)G.p0 * G.PL, r1 = (int64_t)G.p1 * G.PL;
  int fail = 0;
  for (int64_t r = r0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < r1; r += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(r / G.PL), j = (int)((r % G.PL) / G.m2), k = (int)(r % G.m2);
    const int Ei = l2_hw(i & 1), Ej = l2_hw(j & 1), Ek = l2_hw(k & 1);
    int64_t p = (int64_t)rowptr[r] - base;
    const int64_t pe = (int64_t)rowptr[r + 1] - base;
    for (int di = -Ei; di <= Ei; ++di)
      for (int dj = -Ej; dj <= Ej; ++dj)
        for (int dk = -Ek; dk <= Ek; ++dk) {
          if (j + dj < 0 || j + dj >= G.m1 || k + dk < 0 || k + dk >= G.m2) continue;
          const int64_t want = r + di * G.PL + (int64_t)dj * G.m2 + dk;
          if (p < pe && (int64_t)col[p] - base == want) ++p;  // an absent entry inside the box is fine (explicit zero)
        }
    if (p != pe) fail = 1;  // an entry outside the box, unsorted or duplicate columns
  }
  if (fail) atomicOr(bad, 1);
}

// ---- per solve: CSR-ordered values -> pv[plane][region][type][slot][lane]; one wave per (plane, region, type)
template <typename RP>
__global__ __launch_bounds__(MFEM_BLOCK) void k_lat2_bind(Lat2Geom G, const RP* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                            const double* __restrict__ vals, int base, double* __restrict__ pv) {
  const int lane = threadIdx.x & 63, jl = lane >> 4, kl = lane & 15;
  const int64_t NR = (int64_t)G.NRj * G.NRk, T = (int64_t)(G.p1 - G.p0) * NR * 4;
  for (int64_t t = (int64_t)blockIdx.x * (MFEM_BLOCK / 64) + (threadIdx.x >> 6); t < T; t += (int64_t)gridDim.x * (MFEM_BLOCK / 64)) {
    const int type = (int)(t & 3), region = (int)((t >> 2) % NR), plane = G.p0 + (int)((t >> 2) / NR);
    const int pj = type >> 1, pk = type & 1;
    const int j = (region / G.NRk) * L2_RJ + 2 * jl + pj, k = (region % G.NRk) * L2_RK + 2 * kl + pk;
    const int Ei = l2_hw(plane & 1), Ej = l2_hw(pj), Ek = l2_hw(pk);
    const bool valid = j < G.m1 && k < G.m2;
    const int64_t r = (int64_t)plane * G.PL + (int64_t)j * G.m2 + k;
    int64_t p = 0, pe = 0;
    if (valid) {
      p = (int64_t)rowptr[r] - base;
      pe = (int64_t)rowptr[r + 1] - base;
    }
    double* out = pv + l2_plane_base(plane, G.p0, NR) + region * l2_block(plane) + (int64_t)l2_type_base(type) * (2 * Ei + 1) * 64 + lane;
    for (int di = -Ei; di <= Ei; ++di)
      for (int dj = -Ej; dj <= Ej; ++dj)
        for (int dk = -Ek; dk <= Ek; ++dk) {
          double v = 0.0;
          if (valid && j + dj >= 0 && j + dj < G.m1 && k + dk >= 0 && k + dk < G.m2) {
            const int64_t want = r + di * G.PL + (int64_t)dj * G.m2 + dk;
            if (p < pe && (int64_t)col[p] - base == want) v = vals[p++];
          }
          *out = v;
          out += 64;
        }
  }
}

// ---- the rows of a few row ranges straight from the CSR arrays: one wave per row, lanes across the row's entries
template <typename RP>
__global__ __launch_bounds__(MFEM_BLOCK) void k_spmv_rows_wave(int64_t a0, int64_t a1, int64_t b0, int64_t b1, const RP* __restrict__ rowptr,
                                                                 const int32_t* __restrict__ col, const double* __restrict__ vals, int base,
                                                                 const double* __restrict__ x, double* __restrict__ y, double alpha, double beta,
                                                                 const double* __restrict__ dotw, double* __restrict__ partials,
                                                                 const int32_t* __restrict__ done_flag) {
  __shared__ double red[16];
  if (done_flag && done_flag[0]) return;
  const int lane = threadIdx.x & 63;
  const int64_t na = a1 - a0, nrows = na + (b1 - b0);
  double dot_acc = 0.0;
  for (int64_t q = (int64_t)blockIdx.x * (MFEM_BLOCK / 64) + (threadIdx.x >> 6); q < nrows; q += (int64_t)gridDim.x * (MFEM_BLOCK / 64)) {
    const int64_t r = q < na ? a0 + q : b0 + (q - na);
    const int64_t lo = (int64_t)rowptr[r] - base, hi = (int64_t)rowptr[r + 1] - base;
    double acc = 0.0;
    for (int64_t p = lo + lane; p < hi; p += 64) acc += vals[p] * x[(int64_t)col[p] - base];
    acc = wave_reduce_sum(acc);
    if (lane == 0) {
      double yv = alpha * acc;
      if (beta != 0.0) yv += beta * y[r];
      y[r] = yv;
      if (dotw) dot_acc += yv * dotw[r];
    }
  }
  if (partials) {
    const double b = block_reduce_sum(dot_acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = b;
  }
}

// ---- the sweep
// The values of a run are consumed as one stream of LAYERS: for every plane the four node types in turn, for every type its 2 Ei + 1
// plane-layers of (2 Ej + 1)(2 Ek + 1) slots (25 / 15 / 15 / 9), all contiguous in the copy.  Two layers beyond the current one are
// always requested -- across type and plane boundaries -- so that 15-25 KB per wave are in flight at any time (one layer ahead inside a
// type only: 3.16 ms per CG iteration at 128^3 against 2.69 ms with the sliced ELL; the kernel was latency-bound).
#define L2_NBMAX 25
__device__ __forceinline__ int l2_nb(int type) { return type == 0 ? 25 : type == 3 ? 9 : 15; }

template <int EJ, int EK>
__device__ __forceinline__ double lat2_layer(const double (&v)[L2_NBMAX], const double (*xp)[L2_XW], int xj, int xk) {
  double s = 0.0;
#pragma unroll
  for (int dj = -EJ; dj <= EJ; ++dj)
#pragma unroll
    for (int dk = -EK; dk <= EK; ++dk) s += v[(dj + EJ) * (2 * EK + 1) + (dk + EK)] * xp[xj + dj][xk + dk];
  return s;
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void k_spmv_lat2(Lat2Geom G, const double* __restrict__ pv, const double* __restrict__ x, double* __restrict__ y,
                                                   double alpha, double beta, const double* __restrict__ dotw, double* __restrict__ partials,
                                                   const int32_t* __restrict__ done_flag) {
  __shared__ double xs[5][L2_XL][L2_XW];  // ring of lattice planes: plane q in slot q % 5
  if (done_flag && done_flag[0]) return;
  const int lane = threadIdx.x, jl = lane >> 4, kl = lane & 15;
  const int64_t NR = (int64_t)G.NRj * G.NRk;
  const int nplanes = G.p1 - G.p0;
  // XCD c (workgroups with blockIdx % 8 == c; the grid is a multiple of 8) sweeps a contiguous eighth of the regions, segment by segment
  const int xcd = blockIdx.x & 7, rc = (int)(NR / 8), rrem = (int)(NR % 8), rcnt = rc + (xcd < rrem ? 1 : 0);
  const int rfirst = xcd * rc + (xcd < rrem ? xcd : rrem);
  double dot_acc = 0.0;
  for (int run = blockIdx.x >> 3; run < rcnt * G.nseg; run += gridDim.x >> 3) {
    const int region = rfirst + run % rcnt, seg = run / rcnt;
    const int pa = G.p0 + (int)((int64_t)nplanes * seg / G.nseg), pb = G.p0 + (int)((int64_t)nplanes * (seg + 1) / G.nseg);
    const int j0 = (region / G.NRk) * L2_RJ, k0 = (region % G.NRk) * L2_RK;
    bool valid[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) valid[t] = j0 + 2 * jl + (t >> 1) < G.m1 && k0 + 2 * kl + (t & 1) < G.m2;
    // x staging: entry e = lane + 64 u of the 12 x 36 neighbourhood (lines j0 - 2 .., columns k0 - 2 ..); positions outside the vector are
    // only ever multiplied by explicit zeros -- clamp
    int xo[7];
#pragma unroll
    for (int u = 0; u < 7; ++u) {
      const int e = lane + 64 * u, xl = e / L2_XW, xc = e - xl * L2_XW;
      xo[u] = (j0 - 2 + xl) * G.m2 + (k0 - 2 + xc);
    }
    auto xidx = [&](int plane, int u) -> int64_t {
      int64_t idx = (int64_t)plane * G.PL + xo[u];
      idx = idx < 0 ? 0 : idx;
      return idx < G.n ? idx : G.n - 1;
    };
    // the layer stream: (plane, layer index) -> pointer of this lane's first slot; type and plane-layer of a layer index
    auto layer_ptr = [&](int plane, int li) -> const double* {
      const int nl = 2 * l2_hw(plane & 1) + 1, type = li / nl, di = li - type * nl;
      return pv + l2_plane_base(plane, G.p0, NR) + region * l2_block(plane) + ((int64_t)l2_type_base(type) * nl + (int64_t)di * l2_nb(type)) * 64 + lane;
    };
    auto request = [&](double (&dst)[L2_NBMAX], int plane, int li) {
      if (plane >= pb) return;  // beyond the run: nothing to fetch
      const int nl = 2 * l2_hw(plane & 1) + 1, type = li / nl;
      const double* p = layer_ptr(plane, li);
      const bool ok = type == 0 ? valid[0] : type == 1 ? valid[1] : type == 2 ? valid[2] : valid[3];
      if (!ok) {
#pragma unroll
        for (int u = 0; u < L2_NBMAX; ++u) dst[u] = 0.0;
        return;
      }
      if (type == 0) {
#pragma unroll
        for (int u = 0; u < 25; ++u) dst[u] = __builtin_nontemporal_load(p + (int64_t)u * 64);
      } else if (type == 3) {
#pragma unroll
        for (int u = 0; u < 9; ++u) dst[u] = __builtin_nontemporal_load(p + (int64_t)u * 64);
      } else {
#pragma unroll
        for (int u = 0; u < 15; ++u) dst[u] = __builtin_nontemporal_load(p + (int64_t)u * 64);
      }
    };
    auto advance = [&](int& plane, int& li) {
      if (++li == 4 * (2 * l2_hw(plane & 1) + 1)) {
        li = 0;
        ++plane;
      }
    };
    __syncthreads();  // the previous run's last products may still be reading the ring
    for (int q = pa - 2; q <= pa + 1; ++q) {
      double* dst = &xs[(q + 5) % 5][0][0];
#pragma unroll
      for (int u = 0; u < 6; ++u) dst[lane + 64 * u] = x[xidx(q, u)];
      if (lane < L2_XN - 384) dst[lane + 384] = x[xidx(q, 6)];
    }
    double xr[7];
#pragma unroll
    for (int u = 0; u < 6; ++u) xr[u] = x[xidx(pa + 2, u)];
    xr[6] = lane < L2_XN - 384 ? x[xidx(pa + 2, 6)] : 0.0;
    double c0[L2_NBMAX], c1[L2_NBMAX], c2[L2_NBMAX];  // current layer and the two after it (three deep: 317 VGPRs, one wave per SIMD)
    int fp = pa, fl = 0;  // the next layer to request
    request(c0, fp, fl); advance(fp, fl);
    request(c1, fp, fl); advance(fp, fl);
    for (int plane = pa; plane < pb; ++plane) {
      {  // the plane two ahead arrives in the ring, the one after it is requested
        double* dst = &xs[(plane + 2) % 5][0][0];
#pragma unroll
        for (int u = 0; u < 6; ++u) dst[lane + 64 * u] = xr[u];
        if (lane < L2_XN - 384) dst[lane + 384] = xr[6];
        if (plane + 1 < pb) {
#pragma unroll
          for (int u = 0; u < 6; ++u) xr[u] = x[xidx(plane + 3, u)];
          xr[6] = lane < L2_XN - 384 ? x[xidx(plane + 3, 6)] : 0.0;
        }
      }
      __syncthreads();  // one wave: orders its LDS writes before the reads of other lanes
      const int Ei = l2_hw(plane & 1), nl = 2 * Ei + 1;
      double acc = 0.0;
      for (int li = 0; li < 4 * nl; ++li) {
        request(c2, fp, fl);
        advance(fp, fl);
        const int type = li / nl, di = li - type * nl - Ei;
        const int pj = type >> 1, pk = type & 1, xj = 2 * jl + pj + 2, xk = 2 * kl + pk + 2;
        const double(*xp)[L2_XW] = xs[(plane + di + 5) % 5];
        if (type == 0) acc += lat2_layer<2, 2>(c0, xp, xj, xk);
        else if (type == 1) acc += lat2_layer<2, 1>(c0, xp, xj, xk);
        else if (type == 2) acc += lat2_layer<1, 2>(c0, xp, xj, xk);
        else acc += lat2_layer<1, 1>(c0, xp, xj, xk);
        if (di == Ei) {  // the type's last layer: the row is complete
          const bool ok = type == 0 ? valid[0] : type == 1 ? valid[1] : type == 2 ? valid[2] : valid[3];
          if (ok) {
            const int64_t r = (int64_t)plane * G.PL + (int64_t)(j0 + 2 * jl + pj) * G.m2 + k0 + 2 * kl + pk;
            double yv = alpha * acc;
            if (beta != 0.0) yv += beta * y[r];
            y[r] = yv;
            if (dotw) dot_acc += yv * (dotw == x ? xs[plane % 5][xj][xk] : dotw[r]);
          }
          acc = 0.0;
        }
#pragma unroll
        for (int u = 0; u < L2_NBMAX; ++u) {
          c0[u] = c1[u];
          c1[u] = c2[u];
        }
      }
      __syncthreads();  // every lane is done with the ring slot the next step overwrites
    }
  }
  if (partials) {
    const double w = wave_reduce_sum(dot_acc);
    if (lane == 0) partials[blockIdx.x] = w;
  }
}

// ---- host side
static Lat2Geom lat2_geom(const mfem_context_s* ctx, const mfem_csr_s* A) {
  Lat2Geom G;
  G.PL = A->lat2_PL;
  G.n = A->n;
  G.m1 = A->lat2_m1;
  G.m2 = A->lat2_m2;
  G.p0 = A->lat2_p0;
  G.p1 = A->lat2_p1;
  G.NRj = A->lat2_NRj;
  G.NRk = A->lat2_NRk;
  // runs per region: the smallest count that fills >= 90 % of the resident one-wave workgroups in whole rounds, runs >= 8 planes long
  const int64_t NR = (int64_t)G.NRj * G.NRk, slots = (int64_t)L2_WG_PER_CU * ctx->num_cus;
  const int nplanes = G.p1 - G.p0;
  int best = 1;
  double best_eff = 0.0;
  for (int ns = 1; ns <= (nplanes / 8 > 1 ? nplanes / 8 : 1) && ns <= 64; ++ns) {
    const int64_t R = NR * ns, rounds = (R + slots - 1) / slots;
    const double eff = (double)R / (double)(rounds * slots);
    if (eff > best_eff + 1e-9) { best_eff = eff; best = ns; }
    if (eff >= 0.9) { best = ns; break; }
  }
  G.nseg = best;
  return G;
}

int mfem_lat2_plan(mfem_context_s* ctx, mfem_csr_s* A) {
  if (A->lat2_state != 0) return MFEM_OK;
  A->lat2_state = -1;
  if (A->max_row_nnz != 125 || A->n < 125 || A->n >= ((int64_t)1 << 31)) return MFEM_OK;
  // a full-length row near the middle of the matrix gives the candidate lattice: 125 offsets (di, dj, dk) in [-2, 2]^3
  const int64_t w0 = A->n / 2 > 2048 ? A->n / 2 - 2048 : 0, wn = (A->n - w0) < 4096 ? (A->n - w0) : 4096;
  std::vector<int64_t> win((size_t)wn + 1);
  if (A->rowptr_bits == 64) {
    MFEM_CHECK_HIP(hipMemcpy(win.data(), (const char*)A->rowptr + w0 * 8, (size_t)(wn + 1) * 8, hipMemcpyDeviceToHost));
  } else {
    std::vector<int32_t> w32((size_t)wn + 1);
    MFEM_CHECK_HIP(hipMemcpy(w32.data(), (const char*)A->rowptr + w0 * 4, (size_t)(wn + 1) * 4, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i <= wn; ++i) win[(size_t)i] = w32[(size_t)i];
  }
  int64_t rm = -1;
  for (int64_t i = 0; i < wn && rm < 0; ++i)
    if (win[(size_t)i + 1] - win[(size_t)i] == 125) rm = i;
  if (rm < 0) return MFEM_OK;
  int32_t cbuf[125];
  MFEM_CHECK_HIP(hipMemcpy(cbuf, A->colidx + (win[(size_t)rm] - A->index_base), sizeof(cbuf), hipMemcpyDeviceToHost));
  int64_t d[125];
  for (int i = 0; i < 125; ++i) d[i] = (int64_t)cbuf[i] - A->index_base - (w0 + rm);
  const int64_t m2 = d[65] + 2, PL = d[75] + 2 * m2 + 2;  // (0, +1, -2) and (+1, -2, -2)
  if (m2 < 5 || PL < 5 * m2 || PL % m2 != 0 || A->n % PL != 0) return MFEM_OK;
  for (int i = 0; i < 125; ++i)
    if (d[i] != (i / 25 - 2) * PL + ((i / 5) % 5 - 2) * m2 + (i % 5 - 2)) return MFEM_OK;
  const int64_t m1 = PL / m2, P = A->n / PL;
  if (m1 < 5 || P < 8 || m1 > 32767 || m2 > 32767) return MFEM_OK;
  A->lat2_m1 = (int)m1;
  A->lat2_m2 = (int)m2;
  A->lat2_PL = PL;
  A->lat2_p0 = 2;
  A->lat2_p1 = (int)P - 2;
  A->lat2_NRj = (int)((m1 + L2_RJ - 1) / L2_RJ);
  A->lat2_NRk = (int)((m2 + L2_RK - 1) / L2_RK);
  const Lat2Geom G = lat2_geom(ctx, A);
  int32_t* d_bad = ctx->d_flags + 9;
  MFEM_CHECK_HIP(hipMemsetAsync(d_bad, 0, sizeof(int32_t), ctx->stream));
  const int grid = mfem_grid_for((int64_t)(G.p1 - G.p0) * G.PL, MFEM_BLOCK, ctx->num_cus * 16);
  if (A->rowptr_bits == 64)
    hipLaunchKernelGGL(k_lat2_check<int64_t>, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, G, (const int64_t*)A->rowptr, A->colidx,
                       A->index_base, d_bad);
  else
    hipLaunchKernelGGL(k_lat2_check<int32_t>, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, G, (const int32_t*)A->rowptr, A->colidx,
                       A->index_base, d_bad);
  MFEM_CHECK_LAUNCH();
  MFEM_CHECK_HIP(hipMemcpyAsync(ctx->h_flags + 9, d_bad, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  MFEM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  if (ctx->h_flags[9]) return MFEM_OK;  // some swept row is not a lattice row (e.g. a slab that does not start on an element boundary)
  const int64_t NR = (int64_t)G.NRj * G.NRk;
  A->lat2_total = l2_plane_base(G.p1, G.p0, NR);
  // regions sticking far out of a small lattice: not worth the padding (the parity tests lift the size limits and take the layout anyway)
  if (g_layout_min_rows_cols != 0 && (double)A->lat2_total > 1.35 * (double)A->nnz) return MFEM_OK;
  A->lat2_state = 1;
  return MFEM_OK;
}

size_t mfem_lat2_vals_bytes(const mfem_csr_s* A) {
  return (A->lat2_state == 1 && g_lat2_enable && A->n >= g_layout_min_rows_cols) ? sizeof(double) * (size_t)A->lat2_total : 0;
}

int mfem_lat2_bind(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, double* buf) {
  A->lat2_vals = nullptr;
  A->lat2_src = nullptr;
  if (!mfem_lat2_vals_bytes(A) || !buf) return MFEM_OK;
  const Lat2Geom G = lat2_geom(ctx, A);
  const int64_t T = (int64_t)(G.p1 - G.p0) * G.NRj * G.NRk * 4;
  const int grid = (int)(T / 4 + 1 < (int64_t)ctx->num_cus * 32 ? T / 4 + 1 : (int64_t)ctx->num_cus * 32);
  if (A->rowptr_bits == 64)
    hipLaunchKernelGGL(k_lat2_bind<int64_t>, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, G, (const int64_t*)A->rowptr, A->colidx, vals,
                       A->index_base, buf);
  else
    hipLaunchKernelGGL(k_lat2_bind<int32_t>, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, G, (const int32_t*)A->rowptr, A->colidx, vals,
                       A->index_base, buf);
  MFEM_CHECK_LAUNCH();
  A->lat2_vals = buf;
  A->lat2_src = vals;
  return MFEM_OK;
}

static int64_t g_lat2_launches = 0;
extern "C" int64_t mfem_debug_lat2_spmv_count(void) { return g_lat2_launches; }

// returns 1 if launched, 0 if another kernel should be used, < 0 on error
int mfem_lat2_launch(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, const double* x, double* y, double alpha, double beta,
                     const double* dotw, double* partials, int* n_partials, const int32_t* done_flag) {
  if (!A->lat2_vals || vals != A->lat2_src) return 0;
  const Lat2Geom G = lat2_geom(ctx, A);
  const int64_t NR = (int64_t)G.NRj * G.NRk;
  int64_t gs = 8 * ((NR + 7) / 8) * G.nseg, cap = ((int64_t)L2_WG_PER_CU * ctx->num_cus) & ~(int64_t)7;
  if (cap > MFEM_MAX_PARTIALS - 1024) cap = MFEM_MAX_PARTIALS - 1024;
  if (gs > cap) gs = cap;
  if (gs < 8) gs = 8;
  ++g_lat2_launches;
  hipLaunchKernelGGL(k_spmv_lat2, dim3((unsigned)gs), dim3(64), 0, ctx->stream, G, (const double*)A->lat2_vals, x, y, alpha, beta, dotw, partials,
                     done_flag);
  MFEM_CHECK_LAUNCH();
  // the first / last two planes from the caller's CSR arrays
  const int64_t a1 = (int64_t)G.p0 * G.PL, b0 = (int64_t)G.p1 * G.PL;
  const int go = (int)mfem_grid_for((a1 + (A->n - b0)) * 64, MFEM_BLOCK, 1024);
  double* pp = partials ? partials + gs : nullptr;
  if (A->rowptr_bits == 64)
    hipLaunchKernelGGL(k_spmv_rows_wave<int64_t>, dim3(go), dim3(MFEM_BLOCK), 0, ctx->stream, (int64_t)0, a1, b0, A->n, (const int64_t*)A->rowptr,
                       A->colidx, vals, A->index_base, x, y, alpha, beta, dotw, pp, done_flag);
  else
    hipLaunchKernelGGL(k_spmv_rows_wave<int32_t>, dim3(go), dim3(MFEM_BLOCK), 0, ctx->stream, (int64_t)0, a1, b0, A->n, (const int32_t*)A->rowptr,
                       A->colidx, vals, A->index_base, x, y, alpha, beta, dotw, pp, done_flag);
  MFEM_CHECK_LAUNCH();
  if (n_partials && partials) *n_partials = (int)gs + go;
  return 1;
}

// matrix values of the swept planes (explicit zeros included) + the CSR entries of the other rows + x as staged + y
int64_t mfem_lat2_bytes(const mfem_csr_s* A) {
  const int64_t swept = (int64_t)(A->lat2_p1 - A->lat2_p0) * A->lat2_PL, out_rows = A->n - swept;
  const int64_t steps = (int64_t)(A->lat2_p1 - A->lat2_p0) * A->lat2_NRj * A->lat2_NRk;
  return A->lat2_total * 8 + steps * L2_XN * 8 + swept * 8 + out_rows * (int64_t)(64 * 12 + 16);
}
