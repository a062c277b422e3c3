// Which XCD does workgroup b run on?  Reads HW_REG_XCC_ID (hwreg 20, bits 0-3) per workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15;
}
int main() {
  const int n = 4096;
  int* d; hipMalloc(&d, n * 4);
  hipLaunchKernelGGL(k, dim3(n), dim3(256), 0, 0, d);
  int h[4096]; hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
  int ok = 0;
  for (int i = 0; i < n; ++i) ok += (h[i] == (i & 7));
  printf("first 32:"); for (int i = 0; i < 32; ++i) printf(" %d", h[i]); printf("\nblocks with xcc == blockIdx %% 8: %d of %d\n", ok, n);
  int cnt[16] = {0}; for (int i = 0; i < n; ++i) cnt[h[i]]++;
  printf("per xcc:"); for (int i = 0; i < 16; ++i) printf(" %d", cnt[i]); printf("\n");
  return 0;
}
