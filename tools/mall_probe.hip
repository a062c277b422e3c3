// Micro-probe: is a re-read whose reuse distance fits the Infinity Cache (but not an XCD's L2) cheaper than a second HBM stream?
// Every wave streams a fresh part of a 4 GiB array (128-bit loads, 4 in flight per lane) and, per fresh load, also loads the
// element `shift` bytes behind it.  shift = 0: same line (L1 / L2 hit); 8 MiB / 32 MiB: beyond one L2, inside the Infinity Cache;
// 1 GiB: a second HBM stream.  Decides whether a symmetric (half-stored) diagonal SpMV could win on this chip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

template <bool SECOND>
__global__ __launch_bounds__(256) void k_stream(const d2* __restrict__ a, int64_t lo, int64_t n, int64_t shift, double* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  d2 acc = {0.0, 0.0};
  int64_t i = lo + blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i + 3 * stride < lo + n; i += 4 * stride) {
    d2 v[4], w[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      v[u] = __builtin_nontemporal_load(a + i + u * stride);
      if (SECOND) w[u] = a[i + u * stride - shift];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc += v[u];
      if (SECOND) acc += w[u];
    }
  }
  if (acc.x + acc.y == 1.2345e300) out[0] = acc.x;
}

// XCD-local variant: workgroup -> XCD is round-robin, so XCD x = blockIdx % 8 sweeps its own contiguous eighth of the range and the
// re-read stays inside that XCD's L2 when `shift` is small enough.
__global__ __launch_bounds__(256) void k_stream_xcd(const d2* __restrict__ a, int64_t lo, int64_t n, int64_t shift, double* __restrict__ out) {
  const int xcd = blockIdx.x & 7;
  const int64_t chunk = n >> 3, base = lo + xcd * chunk;
  const int64_t stride = (int64_t)(gridDim.x >> 3) * blockDim.x;
  d2 acc = {0.0, 0.0};
  int64_t i = base + (blockIdx.x >> 3) * (int64_t)blockDim.x + threadIdx.x;
  for (; i + 3 * stride < base + chunk; i += 4 * stride) {
    d2 v[4], w[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      v[u] = a[i + u * stride];  // plain load: the line should stay in this XCD's L2 for the re-read
      w[u] = a[i + u * stride - shift];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += v[u] + w[u];
  }
  if (acc.x + acc.y == 1.2345e300) out[0] = acc.x;
}

int main() {
  const int64_t total = (int64_t)4 << 30;            // bytes
  const int64_t nel = total / 16;
  d2* a; double* out;
  CK(hipMalloc(&a, total)); CK(hipMalloc(&out, 8));
  CK(hipMemset(a, 0, total));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int64_t lo = ((int64_t)1 << 30) / 16, n = ((int64_t)2 << 30) / 16;  // stream the 2 GiB in [1 GiB, 3 GiB)
  const int grid = 256 * 8;
  auto run = [&](const char* name, bool second, int64_t shift_bytes) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(e0));
      if (second) hipLaunchKernelGGL(k_stream<true>, dim3(grid), dim3(256), 0, 0, a, lo, n, shift_bytes / 16, out);
      else hipLaunchKernelGGL(k_stream<false>, dim3(grid), dim3(256), 0, 0, a, lo, n, (int64_t)0, out);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    printf("%-44s %.3f ms  fresh stream %.0f GB/s  loads issued %.0f GB/s\n", name, best, 2147.483648 / best, (second ? 2 : 1) * 2147.483648 / best);
  };
  auto runx = [&](const char* name, int64_t shift_bytes) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_stream_xcd, dim3(grid), dim3(256), 0, 0, a, lo, n, shift_bytes / 16, out);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    printf("%-44s %.3f ms  fresh stream %.0f GB/s  loads issued %.0f GB/s\n", name, best, 2147.483648 / best, 2 * 2147.483648 / best);
  };
  runx("XCD-local + re-read 256 KiB behind", (int64_t)256 << 10);
  runx("XCD-local + re-read 512 KiB behind", (int64_t)512 << 10);
  runx("XCD-local + re-read 1 MiB behind", (int64_t)1 << 20);
  runx("XCD-local + re-read 2 MiB behind", (int64_t)2 << 20);
  runx("XCD-local + re-read 3 MiB behind", (int64_t)3 << 20);
  runx("XCD-local + re-read 8 MiB behind", (int64_t)8 << 20);
  run("one stream (2 GiB)", false, 0);
  run("+ re-read, shift 0 (same line)", true, 0);
  run("+ re-read, 64 KiB behind", true, (int64_t)64 << 10);
  run("+ re-read, 1 MiB behind", true, (int64_t)1 << 20);
  run("+ re-read, 8 MiB behind", true, (int64_t)8 << 20);
  run("+ re-read, 32 MiB behind", true, (int64_t)32 << 20);
  run("+ re-read, 128 MiB behind", true, (int64_t)128 << 20);
  run("+ re-read, 1 GiB behind (second HBM stream)", true, (int64_t)1 << 30);
  return 0;
}
