// Micro-probe: what does the x[col] gather of the 27-point stencil SpMV cost, as a function of the lane -> nonzero map?
//   mode 0  CSR order: lane l of a wave reads nonzeros 2l, 2l+1 of a 128-nonzero chunk (what k_spmv_lds does)
//   mode 1  transposed: lane l <-> row r0 + l, one stencil slot per instruction (64 consecutive x entries)
//   mode 2  no gather at all (index stream only), the floor
// Each thread accumulates the gathered values so the loads cannot be dropped; the index stream is read in every mode.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void k_build(int N1, int64_t n, int* __restrict__ col) {  // interior-like stencil with clamping, 27 per row
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n; r += stride) {
    const int k = r % N1, j = (r / N1) % N1, i = r / ((int64_t)N1 * N1);
    int s = 0;
    for (int di = -1; di <= 1; ++di)
      for (int dj = -1; dj <= 1; ++dj)
        for (int dk = -1; dk <= 1; ++dk) {
          const int ii = min(max(i + di, 0), N1 - 1), jj = min(max(j + dj, 0), N1 - 1), kk = min(max(k + dk, 0), N1 - 1);
          col[r * 27 + s++] = (int)(((int64_t)ii * N1 + jj) * N1 + kk);
        }
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_gather(int64_t n, const int* __restrict__ col, const double* __restrict__ x,
                                                 double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  double acc = 0.0;
  // a wave owns 64 rows = 1728 nonzeros per step
  for (int64_t r0 = wave * 64; r0 + 64 <= n; r0 += nwaves * 64) {
    const int* c = col + r0 * 27;
    if (MODE == 0) {
#pragma unroll 9
      for (int it = 0; it < 27; it += 2) {  // 13.5 chunks of 128 nonzeros
        const int p = it * 64 + 2 * lane;
        if (p + 1 < 1728) {
          const int2 cc = *reinterpret_cast<const int2*>(c + p);
          acc += x[cc.x] + x[cc.y];
        }
      }
    } else if (MODE == 3 || MODE == 4 || MODE == 5) {
      // CSR-order lanes, but the address is 3: one line for the whole wave (x[0]) | 4: index folded into a 16 KB window
      // (always L1 resident) | 5: 64 consecutive doubles per instruction (perfectly coalesced)
#pragma unroll 9
      for (int it = 0; it < 27; it += 2) {
        const int p = it * 64 + 2 * lane;
        if (p + 1 < 1728) {
          const int2 cc = *reinterpret_cast<const int2*>(c + p);
          if (MODE == 3) acc += x[cc.x & 1] + x[cc.y & 1];
          if (MODE == 4) acc += x[cc.x & 2047] + x[cc.y & 2047];
          if (MODE == 5) acc += x[(r0 & ~63) + lane + (cc.x & 1)] + x[(r0 & ~63) + 64 + lane + (cc.y & 1)];
        }
      }
    } else if (MODE == 6) {
      // CSR order, but a pair of consecutive columns is fetched with ONE 16-byte load (2/3 of the pairs of the stencil)
      typedef double d2 __attribute__((ext_vector_type(2), aligned(8)));
#pragma unroll 9
      for (int it = 0; it < 27; it += 2) {
        const int p = it * 64 + 2 * lane;
        if (p + 1 < 1728) {
          const int2 cc = *reinterpret_cast<const int2*>(c + p);
          if (cc.y == cc.x + 1) {
            const d2 v = *reinterpret_cast<const d2*>(x + cc.x);
            acc += v.x + v.y;
          } else {
            acc += x[cc.x] + x[cc.y];
          }
        }
      }
    } else if (MODE == 1) {
#pragma unroll 9
      for (int s = 0; s < 27; ++s) acc += x[c[lane * 27 + s]];
    } else {
#pragma unroll 9
      for (int it = 0; it < 27; it += 2) {
        const int p = it * 64 + 2 * lane;
        if (p + 1 < 1728) {
          const int2 cc = *reinterpret_cast<const int2*>(c + p);
          acc += (double)(cc.x ^ cc.y);
        }
      }
    }
  }
  out[blockIdx.x * (int64_t)blockDim.x + threadIdx.x] = acc;
}

int main(int argc, char** argv) {
  const int N1 = argc > 1 ? atoi(argv[1]) : 257;
  const int64_t n = (int64_t)N1 * N1 * N1;
  int* col; double *x, *out;
  CK(hipMalloc(&col, n * 27 * sizeof(int)));
  CK(hipMalloc(&x, n * sizeof(double)));
  const int grid = 256 * 8;
  CK(hipMalloc(&out, (size_t)grid * 256 * sizeof(double)));
  CK(hipMemset(x, 0, n * sizeof(double)));
  hipLaunchKernelGGL(k_build, dim3(4096), dim3(256), 0, 0, N1, n, col);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 7; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      for (int t = 0; t < 10; ++t) {
        if (mode == 0) hipLaunchKernelGGL(k_gather<0>, dim3(grid), dim3(256), 0, 0, n, col, x, out);
        if (mode == 1) hipLaunchKernelGGL(k_gather<1>, dim3(grid), dim3(256), 0, 0, n, col, x, out);
        if (mode == 2) hipLaunchKernelGGL(k_gather<2>, dim3(grid), dim3(256), 0, 0, n, col, x, out);
        if (mode == 3) hipLaunchKernelGGL(k_gather<3>, dim3(grid), dim3(256), 0, 0, n, col, x, out);
        if (mode == 4) hipLaunchKernelGGL(k_gather<4>, dim3(grid), dim3(256), 0, 0, n, col, x, out);
        if (mode == 5) hipLaunchKernelGGL(k_gather<5>, dim3(grid), dim3(256), 0, 0, n, col, x, out);
        if (mode == 6) hipLaunchKernelGGL(k_gather<6>, dim3(grid), dim3(256), 0, 0, n, col, x, out);
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) printf("mode %d: %.3f ms per pass (index stream %.2f GB -> %.0f GB/s)\n", mode, ms / 10, n * 27 * 4 / 1e9, n * 27 * 4 / 1e9 / (ms / 10) * 1e3);
    }
  }
  return 0;
}
