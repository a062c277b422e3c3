// Issue rate of the two FP64 MFMA shapes on gfx950 (round 5 probe: would 4x4x4 blocks -- 28 of 49 symmetric blocks of a 28 x 28 Ke instead of 3 of 4 tiles of
// a padded 32 x 32 -- shorten the hex-27 Ke kernel?).  One wave per SIMD, N instructions with four independent accumulators each, cycles from s_memtime.
// build: hipcc -O3 --offload-arch=gfx950 tools/mfma_f64_probe.hip -o tools/bin/mfma_f64_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k16(double* out, int n, long long* cyc) {
  d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, b, c3, 0, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k4(double* out, int n, long long* cyc) {
  double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, a, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, a, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, b, c3, 0, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  double* out; long long* cyc; long long h = 0;
  hipMalloc(&out, 8 * 256 * 1024); hipMalloc(&cyc, 8);
  const int n = 20000;
  for (int waves = 1; waves <= 4; waves *= 2) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms16, ms4;
    k16<<<1024, 64 * waves>>>(out, 100, cyc); hipDeviceSynchronize();
    hipEventRecord(e0); k16<<<1024, 64 * waves>>>(out, n, cyc); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms16, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); long long c16 = h;
    k4<<<1024, 64 * waves>>>(out, 100, cyc); hipDeviceSynchronize();
    hipEventRecord(e0); k4<<<1024, 64 * waves>>>(out, n, cyc); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms4, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); long long c4 = h;
    const double f16 = 1024.0 * waves * 4.0 * n * 2048, f4 = 1024.0 * waves * 4.0 * n * 512;
    printf("waves/WG %d: 16x16x4: %.3f ms = %.1f TFLOP/s (%.1f timer ticks per instruction, first wave)   4x4x4(4 blocks): %.3f ms = %.1f TFLOP/s (%.1f ticks)\n", waves, ms16,
           f16 / ms16 / 1e9, (double)c16 / (4.0 * n), ms4, f4 / ms4 / 1e9, (double)c4 / (4.0 * n));
  }
  return 0;
}
