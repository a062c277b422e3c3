// Micro-probe: cost of a 64-lane gather instruction on gfx950 as a function of the address pattern (L2/L1-resident table, so
// that what is timed is the texture-addresser / L1 tag pipeline, not HBM).  Prints cycles per wave-instruction per CU.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/ta_probe tools/ta_probe.hip ; run: tools/bin/ta_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2), aligned(8)));

template <int MODE>
__global__ __launch_bounds__(256) void k(const double* __restrict__ x, double* __restrict__ out, int iters, int tab) {
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
  double acc = 0.0;
  int base = (wave * 977) % (tab - 16384);
  for (int it = 0; it < iters; ++it) {
    int idx;
    if (MODE == 0) idx = lane;                               // unit stride, 8 B
    else if (MODE == 1) idx = 2 * lane;                      // stride 2
    else if (MODE == 2) idx = 4 * lane;                      // stride 4
    else if (MODE == 3) idx = lane >> 1;                     // adjacent lanes share an address
    else if (MODE == 4) idx = (lane / 3) * 257 + lane % 3;   // runs of 3 consecutive, runs a lattice line apart (CSR order, 1 nnz / lane)
    else if (MODE == 5) idx = 16 * lane;                     // every lane its own 128-B line
    else if (MODE == 6) { const int e = 2 * lane; idx = (e / 3) * 257 + e % 3; }   // CSR order, lane = nonzero pair, first of the pair
    else if (MODE == 7) idx = 2 * lane;                      // 16-byte loads, unit stride (see below)
    else if (MODE == 8) idx = (lane & 3) + 257 * (lane >> 2);  // quads of 4 consecutive, quads a line apart
    else if (MODE == 9) idx = (lane & 7) + 257 * (lane >> 3);  // octets of 8 consecutive
    else if (MODE == 10) idx = (lane & 15) + 257 * (lane >> 4); // 16 consecutive
    else if (MODE == 11) idx = (lane >> 1) + (lane & 1) * 257;  // lane pairs: rows r (slot s) and r (slot s') ... alternate lines
    else idx = lane;
    const int a = base + idx + (it & 63) * 64;
    if (MODE == 7) {
      const d2 v = *reinterpret_cast<const d2*>(x + a);
      acc += v.x + v.y;
    } else {
      acc += x[a];
    }
  }
  if (acc == 123.456) out[0] = acc;
}

template <int MODE>
static void run(const double* x, double* out, int tab, const char* what) {
  const int iters = 2048, grid = 256 * 8;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, x, out, iters, tab);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, x, out, iters, tab);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  const double instr_per_cu = (double)grid * 4 * iters / 256.0;   // wave-instructions per CU
  printf("mode %2d  %-70s %.3f ms  %.1f cycles per wave-instruction per CU (2.4 GHz)\n", MODE, what, ms, ms * 1e-3 * 2.4e9 / instr_per_cu);
}

int main() {
  const int tab = 1 << 19;  // 4 MB of doubles: L2-resident
  double *x, *out;
  hipMalloc(&x, sizeof(double) * tab);
  hipMalloc(&out, 64);
  hipMemset(x, 0, sizeof(double) * tab);
  run<0>(x, out, tab, "8 B, unit stride");
  run<1>(x, out, tab, "8 B, stride 2");
  run<2>(x, out, tab, "8 B, stride 4");
  run<3>(x, out, tab, "8 B, adjacent lane pairs share an address");
  run<4>(x, out, tab, "8 B, runs of 3 consecutive (CSR order, one nonzero per lane)");
  run<5>(x, out, tab, "8 B, every lane its own 128-B line");
  run<6>(x, out, tab, "8 B, CSR order with a nonzero PAIR per lane (first of the pair)");
  run<7>(x, out, tab, "16 B, unit stride");
  run<8>(x, out, tab, "8 B, quads of 4 consecutive, quads a lattice line apart");
  run<9>(x, out, tab, "8 B, octets of 8 consecutive");
  run<10>(x, out, tab, "8 B, 16 consecutive");
  run<11>(x, out, tab, "8 B, lane pairs on alternate lines");
  return 0;
}
