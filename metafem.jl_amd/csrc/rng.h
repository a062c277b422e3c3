// Counter-based U[0,1) generator shared by device kernels; restated bit-for-bit by
// oracle/solvers.py::fem_rand.  Replaces FEM_rand -> CUDA.Random.rand! (reference
// misc/04_GPU_Utils.jl:22), whose stream is unseeded.
#pragma once
#include <stdint.h>

__host__ __device__ inline double mfem_u01(uint64_t seed, uint32_t stream, uint64_t idx) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (idx + 1ull) + 0xD1B54A32D192ED03ull * ((uint64_t)stream + 1ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}
