// Timing probe for the per-solve layout copy (k_dia_vals, spmv_ell.hip): what does the memory system give a wave that reads a 64-row tile of a
// 27-entries-per-row CSR value stream (13 824 contiguous bytes), transposes it through LDS and writes 27 slot pieces?  Stand-alone (no library):
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/copy_probe tools/copy_probe.hip && /tmp/copy_probe [rows = 135005697]
// Variants, each timed over the whole value array (best of 3):
//   0 stream read, grid-stride, 8 B per lane            1 tile read: 27 loads per lane in flight, no LDS
//   2 = 1 + LDS transposition (lane = row)               3 = 2 + 27 stores, tile-contiguous output ([tile][slot][64])
//   4 = 2 + 27 stores, two 256-byte pieces per slot at a 1 KB stride (the patch-major copy's shape)
//   5 = 3 with one dependent row-pointer round trip per tile in front of the value loads
//   6 = 3 with the NEXT tile's loads issued before the current tile's stores (register double buffer)
//   7 = 3 with 16-byte loads (13.5 per lane)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int K = 27, RT = 64, TILE = K * RT;  // doubles per tile

__global__ __launch_bounds__(256) void k_stream(int64_t n, const double* __restrict__ v, double* __restrict__ sink) {
  double s = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) s += __builtin_nontemporal_load(v + i);
  if (s == 1.2345e300) sink[0] = s;
}

template <int VAR>
__global__ __launch_bounds__(128) void k_tile(int64_t ntiles, const double* __restrict__ v, const int64_t* __restrict__ rowptr, double* __restrict__ out,
                                               double* __restrict__ sink) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  double* T = lds + (size_t)w * TILE;
  const int64_t stride = (int64_t)gridDim.x * nw;
  double acc = 0.0;
  double nx[K];
  int64_t t = (int64_t)blockIdx.x * nw + w;
  if (VAR == 6 && t < ntiles) {
#pragma unroll
    for (int u = 0; u < K; ++u) nx[u] = __builtin_nontemporal_load(v + t * TILE + lane + 64 * u);
  }
  for (; t < ntiles; t += stride) {
    int64_t s0 = t * TILE;
    if (VAR == 5) s0 = rowptr[t * RT];  // one dependent round trip (= t * TILE)
    double tv[K];
    if (VAR == 7) {
      typedef double d2 __attribute__((ext_vector_type(2)));
      const d2* p = reinterpret_cast<const d2*>(v + s0);
      d2* T2 = reinterpret_cast<d2*>(T);
      d2 q[14];
#pragma unroll
      for (int u = 0; u < 14; ++u) q[u] = (lane + 64 * u) < TILE / 2 ? __builtin_nontemporal_load(p + lane + 64 * u) : (d2){0.0, 0.0};
#pragma unroll
      for (int u = 0; u < 14; ++u) if ((lane + 64 * u) < TILE / 2) T2[lane + 64 * u] = q[u];
    } else if (VAR == 6) {
#pragma unroll
      for (int u = 0; u < K; ++u) tv[u] = nx[u];
      if (t + stride < ntiles) {
#pragma unroll
        for (int u = 0; u < K; ++u) nx[u] = __builtin_nontemporal_load(v + (t + stride) * TILE + lane + 64 * u);
      }
    } else {
#pragma unroll
      for (int u = 0; u < K; ++u) tv[u] = __builtin_nontemporal_load(v + s0 + lane + 64 * u);
    }
    if (VAR == 1) {
#pragma unroll
      for (int u = 0; u < K; ++u) acc += tv[u];
      continue;
    }
    if (VAR != 7) {
#pragma unroll
      for (int u = 0; u < K; ++u) T[lane + 64 * u] = tv[u];
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    double rv[K];
#pragma unroll
    for (int s = 0; s < K; ++s) rv[s] = T[lane * K + s];
    if (VAR == 2) {
#pragma unroll
      for (int s = 0; s < K; ++s) acc += rv[s];
    } else if (VAR == 4) {
      // two "patches" of 32 points: four consecutive tiles fill the four lines of two patch-step blocks (K slots x 128 rows each), slot stride 128
      double* o = out + (((t >> 2) * 2 + (lane >> 5)) * K * 128 + (t & 3) * 32 + (lane & 31));
#pragma unroll
      for (int s = 0; s < K; ++s) __builtin_nontemporal_store(rv[s], o + s * 128);
    } else {
      double* o = out + t * TILE + lane;
#pragma unroll
      for (int s = 0; s < K; ++s) __builtin_nontemporal_store(rv[s], o + s * 64);
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (acc == 1.2345e300) sink[0] = acc;
}

// The real address pattern of the patch-major copy (spmv_symp.h) on an m x m x m lattice: FLAGS bit 0 low part, bit 1 edge entries (nontemporal), bit 2 edge entries
// (plain stores), bit 3 the diagonal to a slot-major array, bit 4 the 27 scaling factors (ssym[r + off]) loaded beside the values
#include "../metafem.jl_amd/csrc/spmv_symp.h"
template <int FLAGS>
__global__ __launch_bounds__(128) void k_exact(int64_t ntiles, int m, const double* __restrict__ v, double* __restrict__ pv, double* __restrict__ dg,
                                                const double* __restrict__ ssym, double* __restrict__ sink) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  double* T = lds + (size_t)w * TILE;
  const int64_t stride = (int64_t)gridDim.x * nw;
  const uint32_t PL = (uint32_t)m * m, m2 = m;
  const int NPk = (m + SP_W - 1) / SP_W, NS = (m + SP_L - 1) / SP_L, spNP = NS * NPk;
  const int64_t spT = (int64_t)spNP * m;
  for (int64_t t = (int64_t)blockIdx.x * nw + w; t < ntiles; t += stride) {
    const int64_t s0 = t * TILE;
    const int64_t r = t * RT + lane;
    double tv[K], sc[K];
#pragma unroll
    for (int u = 0; u < K; ++u) tv[u] = __builtin_nontemporal_load(v + s0 + lane + 64 * u);
    double srow = 1.0;
    if (FLAGS & 16) {
      srow = ssym[r + PL + m2 + 1];
#pragma unroll
      for (int u = 0; u < K; ++u) sc[u] = ssym[r + (int64_t)(u / 9) * PL + (int64_t)((u / 3) % 3) * m2 + u % 3];  // (ssym shifted by PL + m2 + 1: in range)
    }
#pragma unroll
    for (int u = 0; u < K; ++u) T[lane + 64 * u] = tv[u];
    __builtin_amdgcn_wave_barrier();
    const uint32_t ru = (uint32_t)r;
    const uint32_t p = ru / PL, rem = ru - p * PL, jj = rem / m2, kk = rem - jj * m2;
    const int line = (int)(jj % SP_L), pcol = (int)(kk % SP_W);
    const int64_t step = (int64_t)p * spNP + (int)(jj / SP_L) * NPk + (int)(kk / SP_W);
    double* const pm = pv + step * SP_MAIN + line * SP_W + pcol;
    double* const pl = pv + spT * SP_MAIN + step * SP_LOW + line * SP_W + pcol;
    double* const pe = pv + step * SP_MAIN + 14 * SP_ROWS;
    const double* Tr = T + lane * K;
#pragma unroll
    for (int sl = 0; sl < K; ++sl) {
      double x = Tr[sl];
      if (FLAGS & 16) x = x * (srow * sc[sl]);
      if (sl < 13) {
        if (FLAGS & 1) __builtin_nontemporal_store(x, pl + sl * SP_ROWS);
        const int e = sp_edge_of(sl, line, pcol);
        if ((FLAGS & 2) && e >= 0) __builtin_nontemporal_store(x, pe + e);
        if ((FLAGS & 4) && e >= 0) pe[e] = x;
      } else {
        __builtin_nontemporal_store(x, pm + (sl - 13) * SP_ROWS);
        if ((FLAGS & 8) && sl == 13) __builtin_nontemporal_store(x, dg + r);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// The same copy with patch-aligned tiles: a workgroup of two waves per patch step (4 lines x 32 columns), a wave per pair of lines -- every slot store an aligned
// 512-byte piece, the two halves of each 1 KB slot written by the same workgroup.  FLAGS as k_exact; bit 5: the edge block gathered in LDS and written as one
// contiguous 2.5 KB piece by the workgroup (instead of bit 1 / 2's scattered stores)
template <int FLAGS>
__global__ __launch_bounds__(128) void k_exact2(int m, const double* __restrict__ v, double* __restrict__ pv, double* __restrict__ dg,
                                                 const double* __restrict__ ssym, double* __restrict__ sink) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double* T = lds + (size_t)w * TILE;
  double* E = lds + 2 * TILE;
  const uint32_t PL = (uint32_t)m * m, m2 = m;
  const int NPk = (m + SP_W - 1) / SP_W, NS = (m + SP_L - 1) / SP_L, spNP = NS * NPk;
  const int64_t spT = (int64_t)spNP * m, nsteps = spT;
  const int h = lane >> 5, c = lane & 31;
  if (FLAGS & 32) { for (int i = threadIdx.x; i < SP_EPAD; i += 128) E[i] = 0.0; __syncthreads(); }
  for (int64_t step = blockIdx.x; step < nsteps; step += gridDim.x) {
    const int kp = (int)(step % NPk);
    const int64_t q = step / NPk;
    const int strip = (int)(q % NS), p = (int)(q / NS);
    const int jj0 = strip * SP_L + 2 * w, kk0 = kp * SP_W;
    const int ncol = (int)m2 - kk0 < SP_W ? (int)m2 - kk0 : SP_W, nlines = m - jj0 < 2 ? (m - jj0 < 1 ? 0 : 1) : 2;
    const bool valid = c < ncol && h < nlines;
    const int64_t rb = (int64_t)p * PL + (int64_t)jj0 * m2 + kk0, r = rb + (int64_t)h * m2 + c;
    const int line = 2 * w + h;
    if (ncol == SP_W && nlines == 2) {
      const double* vA = v + rb * K + lane;
      const double* vB = v + (rb + m2) * K + lane - SP_W * K;
      const double* v13 = h ? vB : vA;
      double tv[K], sc[K];
#pragma unroll
      for (int u = 0; u < K; ++u) tv[u] = __builtin_nontemporal_load((u < 13 ? vA : u == 13 ? v13 : vB) + 64 * u);
      double srow = 1.0;
      if (FLAGS & 16) {
        srow = ssym[r + PL + m2 + 1];
#pragma unroll
        for (int u = 0; u < K; ++u) sc[u] = ssym[r + (int64_t)(u / 9) * PL + (int64_t)((u / 3) % 3) * m2 + u % 3];
      }
#pragma unroll
      for (int u = 0; u < K; ++u) T[lane + 64 * u] = tv[u];
      __builtin_amdgcn_wave_barrier();
      double* const pm = pv + step * SP_MAIN + line * SP_W + c;
      double* const pl = pv + spT * SP_MAIN + step * SP_LOW + line * SP_W + c;
      double* const pe = pv + step * SP_MAIN + 14 * SP_ROWS;
      const double* Tr = T + lane * K;
#pragma unroll
      for (int sl = 0; sl < K; ++sl) {
        double x = Tr[sl];
        if (FLAGS & 16) x = x * (srow * sc[sl]);
        if (sl < 13) {
          if (FLAGS & 1) __builtin_nontemporal_store(x, pl + sl * SP_ROWS);
          const int e = sp_edge_of(sl, line, c);
          if ((FLAGS & 2) && e >= 0) __builtin_nontemporal_store(x, pe + e);
          if ((FLAGS & 32) && e >= 0) E[e] = x;
        } else {
          __builtin_nontemporal_store(x, pm + (sl - 13) * SP_ROWS);
          if ((FLAGS & 8) && sl == 13) __builtin_nontemporal_store(x, dg + r);
        }
      }
    }
    if (FLAGS & 32) {
      __syncthreads();
      double* const pe = pv + step * SP_MAIN + 14 * SP_ROWS;
      for (int i = threadIdx.x; i < SP_EPAD; i += 128) { __builtin_nontemporal_store(E[i], pe + i); E[i] = 0.0; }
      __syncthreads();
    }
    __builtin_amdgcn_wave_barrier();
  }
}

int main(int argc, char** argv) {
  const int64_t rows = argc > 1 ? atoll(argv[1]) : 135005697ll;
  const int64_t ntiles = rows / RT;
  const int64_t nv = ntiles * TILE;
  double *v, *out, *sink;
  int64_t* rp;
  CK(hipMalloc(&v, nv * 8));
  CK(hipMalloc(&out, nv * 8 + (8 * K * 128 * 8)));
  CK(hipMalloc(&sink, 64));
  CK(hipMalloc(&rp, (rows + 1) * 8));
  CK(hipMemset(v, 0, nv * 8));
  {
    std::vector<int64_t> h(rows + 1);
    for (int64_t i = 0; i <= rows; ++i) h[i] = i * K;
    CK(hipMemcpy(rp, h.data(), (rows + 1) * 8, hipMemcpyHostToDevice));
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const double gb = nv * 8 / 1e9;
  printf("rows %lld, %.1f GB of values\n", (long long)rows, gb);
  auto time = [&](auto launch, const char* name, double bytes_factor) {
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipEventRecord(e0));
      launch();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep && ms < best) best = ms;
    }
    printf("%-58s %8.2f ms  %6.2f TB/s\n", name, best, gb * bytes_factor / best);
    fflush(stdout);
  };
  time([&] { hipLaunchKernelGGL(k_stream, dim3(256 * 8), dim3(256), 0, 0, nv, v, sink); }, "0 stream read (8 waves x 4 per CU)", 1.0);
  for (int rounds : {1, 4, 16}) {   // 5 workgroups of 2 waves are resident per CU (LDS: 13.5 KB per wave)
    const int grid = 256 * 5 * rounds;
    char nm[128];
#define RUN(VARIANT, label, fac)                                                                                                        \
    snprintf(nm, sizeof nm, "%s [grid %d x 2 waves]", label, grid);                                                                     \
    time([&] { hipLaunchKernelGGL((k_tile<VARIANT>), dim3(grid), dim3(128), 2 * TILE * 8, 0, ntiles, v, rp, out, sink); }, nm, fac);
    RUN(1, "1 tile read, 27 loads in flight", 1.0)
    RUN(2, "2 + LDS transposition", 1.0)
    RUN(3, "3 + 27 stores, tile-contiguous", 2.0)
    RUN(4, "4 + 27 stores, 2 x 256 B pieces at 1 KB stride", 2.0)
    RUN(5, "5 = 3 + dependent row-pointer round trip", 2.0)
    RUN(6, "6 = 3 + next tile's loads in flight", 2.0)
    RUN(7, "7 = 3 with 16-byte loads", 2.0)
  }
  {  // the real pattern on the 513^3 lattice (rows / 64 full tiles of it)
    const int m = 513;
    const int64_t mrows = (int64_t)m * m * m, mt = mrows / RT;
    const int NPk = (m + SP_W - 1) / SP_W, NS = (m + SP_L - 1) / SP_L;
    const int64_t steps = (int64_t)NS * NPk * m;
    double *pv, *dg, *ss;
    CK(hipFree(out));
    CK(hipMalloc(&pv, steps * SP_STEP * 8));
    CK(hipMalloc(&dg, mrows * 8));
    CK(hipMalloc(&ss, (mrows + 3ll * m * m) * 8));
    CK(hipMemset(ss, 0, (mrows + 3ll * m * m) * 8));
    if (mt > ntiles) { printf("needs rows >= 513^3\n"); return 1; }
    const int grid = 4096;
    const double gbm = mt * TILE * 8 / 1e9;
    auto timex = [&](auto launch, const char* name, double f) {
      float best = 1e30f;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
      }
      printf("%-70s %8.2f ms  %6.2f TB/s\n", name, best, gbm * f / best); fflush(stdout);
    };
#define RUNX(F, label, fac) timex([&] { hipLaunchKernelGGL((k_exact<F>), dim3(grid), dim3(128), 2 * TILE * 8, 0, mt, m, v, pv, dg, ss, sink); }, label, fac);
    RUNX(0, "x0 main slots only (14 of 27 stored)", 1.0 + 14.0 / 27)
    RUNX(1, "x1 main + low part", 2.0)
    RUNX(3, "x3 main + low + edge entries (nontemporal)", 2.0)
    RUNX(5, "x5 main + low + edge entries (plain stores)", 2.0)
    RUNX(11, "x11 main + low + edges (nt) + diagonal to a slot-major array", 2.0)
    RUNX(27, "x27 = x11 + the 27 scaling factors loaded beside the values", 2.0)
    RUNX(29, "x29 = x27 with plain edge stores", 2.0)
    for (int g2 : {1280, 5120, 20480}) {
      char nm[160];
#define RUNY(F, label, fac) snprintf(nm, sizeof nm, "%s [grid %d]", label, g2); \
      timex([&] { hipLaunchKernelGGL((k_exact2<F>), dim3(g2), dim3(128), 2 * TILE * 8 + SP_EPAD * 8, 0, m, v, pv, dg, ss, sink); }, nm, fac);
      RUNY(0, "y0 aligned: main slots only", 1.0 + 14.0 / 27)
      RUNY(1, "y1 aligned: main + low", 2.0)
      RUNY(3, "y3 aligned: main + low + edges scattered (nt)", 2.0)
      RUNY(33, "y33 aligned: main + low + edge block through LDS", 2.0)
      RUNY(41, "y41 = y33 + diagonal", 2.0)
      RUNY(57, "y57 = y41 + scaling factors", 2.0)
    }
  }
  return 0;
}
