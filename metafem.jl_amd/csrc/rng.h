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

// Rademacher (+-1) shadow vectors of idrs! (round 6): bit k of the word of row idx is the sign of P_k[idx] (1: +1, 0: -1), k < 64.  The reference draws P
// with an unseeded rand (04_IDRs.jl:35 -> FEM_rand): any full-rank P spans a valid shadow space; signs need no memory -- the s dot products P' g read g only
// (eight streamed U(0,1) vectors were 8 of the ~35 vector streams of an inner step at s = 8).  Restated by oracle/solvers.py::fem_sign.
#define MFEM_SIGN_STREAM 0x5149u
__host__ __device__ inline uint64_t mfem_sign_word(uint64_t seed, uint64_t idx) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (idx + 1ull) + 0xD1B54A32D192ED03ull * ((uint64_t)MFEM_SIGN_STREAM + 1ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
