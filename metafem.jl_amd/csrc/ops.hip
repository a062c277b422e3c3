// Generic element operators with the reference signatures (seam S3): the fallback every generated
// K_linear / K_nonlinear body can call for weak forms that have no fused fast path.
//   _Var_Basic   solver/06_FEM_Kernel.jl:1-13     target[q,t] += sum_a N[q,a,sd,host_t] * x[cp[a,el_t] + shift]
//   _Kval_Basic  :28-45   K[slot[a,b,el_t] + shift] += sum_q N[q,a,dsd,host_t] N[q,b,bsd,host_t] vals[q,t]
//   _Res_Basic   :65-79   residue[cp[a,el_t] + shift] += sum_q N[q,a,dsd,host_t] vals[q,t]
// The reference runs ONE THREAD per work item t, so lane l and lane l+1 read basis tables 4..47 KB apart
// (fully uncoalesced) and every output goes through an FP64 atomic.  Here a wave (kval) or a sub-wave
// group (var/res) owns a work item: the item's itg x itp basis slab is contiguous in memory, so it is
// loaded with unit-stride lanes into LDS once and reused itp (resp. itg) times from there; the slot /
// control-point ids of an item are contiguous too.  Accumulation is race-free without atomics when the
// caller supplies colour batches (no two items of a batch touch the same output); with n_colours = 0 the
// reference's FP64 atomics are used.
#include "common.h"

struct OpView {
  int itg, itp, n_sd, base;
  const double* N;  // [itg, itp, n_sd, n_host]
};

__device__ __forceinline__ const double* slab(const OpView& V, int sd, int64_t host) {
  return V.N + (int64_t)V.itg * V.itp * ((int64_t)sd + (int64_t)V.n_sd * host);
}

// ---- _Var_Basic: GROUP lanes per item, lane -> q, loop over a ------------------------------------------
template <int GROUP>
__global__ __launch_bounds__(MFEM_BLOCK) void k_op_var(OpView V, int sd, int64_t shift, const int32_t* __restrict__ cp,
                                                         const double* __restrict__ x, double* __restrict__ target,
                                                         const int32_t* __restrict__ host_ids,
                                                         const int32_t* __restrict__ el_ids, int64_t t0, int64_t t1) {
  const int g = threadIdx.x % GROUP;
  const int64_t t = t0 + ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / GROUP;
  if (t >= t1) return;
  const int64_t el = (int64_t)el_ids[t] - V.base, host = (int64_t)host_ids[t] - V.base;
  const double* Ns = slab(V, sd, host);
  for (int q = g; q < V.itg; q += GROUP) {
    double acc = 0.0;
    for (int a = 0; a < V.itp; ++a) acc += Ns[q + V.itg * a] * x[(int64_t)cp[a + (int64_t)V.itp * el] + shift - V.base];
    target[q + (int64_t)V.itg * t] += acc;  // each (q,t) is owned by exactly one lane
  }
}

// ---- _Res_Basic: GROUP lanes per item, lane -> a, loop over q ------------------------------------------
template <int GROUP, bool ATOMIC>
__global__ __launch_bounds__(MFEM_BLOCK) void k_op_res(OpView V, int dsd, const double* __restrict__ vals, int64_t shift,
                                                         const int32_t* __restrict__ cp, double* __restrict__ residue,
                                                         const int32_t* __restrict__ host_ids,
                                                         const int32_t* __restrict__ el_ids, int64_t t0, int64_t t1) {
  const int g = threadIdx.x % GROUP;
  const int64_t t = t0 + ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / GROUP;
  if (t >= t1) return;
  const int64_t el = (int64_t)el_ids[t] - V.base, host = (int64_t)host_ids[t] - V.base;
  const double* Ns = slab(V, dsd, host);
  const double* v = vals + (int64_t)V.itg * t;
  for (int a = g; a < V.itp; a += GROUP) {
    double acc = 0.0;
    for (int q = 0; q < V.itg; ++q) acc += Ns[q + V.itg * a] * v[q];
    double* dst = residue + ((int64_t)cp[a + (int64_t)V.itp * el] + shift - V.base);
    if (ATOMIC) atomicAdd(dst, acc); else *dst += acc;
  }
}

// ---- _Kval_Basic: one wave per item; both basis slabs + vals staged in LDS ------------------------------
template <bool ATOMIC>
__global__ __launch_bounds__(MFEM_BLOCK) void k_op_kval(OpView V, int dsd, int bsd, const double* __restrict__ vals,
                                                          const int32_t* __restrict__ slots, int64_t shift,
                                                          double* __restrict__ K, const int32_t* __restrict__ host_ids,
                                                          const int32_t* __restrict__ el_ids, int64_t t0, int64_t t1) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int sz = V.itg * V.itp;
  double* Nd = lds + (size_t)w * (2 * sz + V.itg);
  double* Nb = Nd + sz;
  double* vq = Nb + sz;
  const int64_t t = t0 + (int64_t)blockIdx.x * (blockDim.x >> 6) + w;
  if (t >= t1) return;  // wave-uniform; no workgroup barrier below
  const int64_t el = (int64_t)el_ids[t] - V.base, host = (int64_t)host_ids[t] - V.base;
  const double* gd = slab(V, dsd, host);
  const double* gb = slab(V, bsd, host);
  for (int i = lane; i < sz; i += 64) {
    Nd[i] = gd[i];
    Nb[i] = gb[i];
  }
  for (int q = lane; q < V.itg; q += 64) vq[q] = vals[q + (int64_t)V.itg * t];
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's LDS writes have landed
  const int npair = V.itp * V.itp;
  const int32_t* sl = slots + (int64_t)npair * el;
  for (int p = lane; p < npair; p += 64) {
    const int a = p % V.itp, b = p / V.itp;  // slots are [a, b, el] column-major
    double sum = 0.0;
    for (int q = 0; q < V.itg; ++q) sum += Nd[q + V.itg * a] * Nb[q + V.itg * b] * vq[q];
    double* dst = K + ((int64_t)sl[p] + shift - V.base);
    if (ATOMIC) atomicAdd(dst, sum); else *dst += sum;
  }
}

static int check_layout(const mfem_op_layout* L) {
  MFEM_REQUIRE(L, "null layout");
  MFEM_REQUIRE(L->itg > 0 && L->itp > 0 && L->n_sd > 0 && L->n_host >= 0, "bad operator layout");
  MFEM_REQUIRE(L->index_base == 0 || L->index_base == 1, "index_base must be 0 or 1");
  MFEM_REQUIRE(L->n_colours >= 0 && (L->n_colours == 0 || L->colour_offsets), "colour_offsets missing");
  return MFEM_OK;
}

template <typename Launch>
static int for_each_batch(const mfem_op_layout* L, int64_t n_threads, Launch launch) {
  if (L->n_colours == 0) return launch((int64_t)0, n_threads);
  MFEM_REQUIRE(L->colour_offsets[0] == 0 && L->colour_offsets[L->n_colours] == n_threads, "colour_offsets must span all work items");
  for (int c = 0; c < L->n_colours; ++c) {
    const int64_t a = L->colour_offsets[c], b = L->colour_offsets[c + 1];
    MFEM_REQUIRE(a <= b, "colour_offsets must be non-decreasing");
    if (b > a) {
      int rc = launch(a, b);
      if (rc) return rc;
    }
  }
  return MFEM_OK;
}

static int group_for(int m) {
  int g = 8;
  while (g < m && g < 64) g <<= 1;
  return g;
}

extern "C" int mfem_op_var(mfem_context ctx, const mfem_op_layout* L, const double* itp_vals, int32_t sd, int64_t cpID_shift,
                           const int32_t* el_g_cpIDs, const double* x, double* target, const int32_t* itg_hostIDs,
                           const int32_t* elIDs, int64_t n_threads) {
  MFEM_REQUIRE(ctx, "null ctx");
  int rc = check_layout(L);
  if (rc) return rc;
  if (n_threads == 0) return MFEM_OK;
  MFEM_REQUIRE(n_threads > 0 && itp_vals && el_g_cpIDs && x && target && itg_hostIDs && elIDs, "null array");
  MFEM_REQUIRE(sd >= 0 && sd < L->n_sd, "sd out of range");
  OpView V{L->itg, L->itp, L->n_sd, L->index_base, itp_vals};
  const int G = group_for(L->itg);
  const int per_block = MFEM_BLOCK / G;
  const int grid = (int)((n_threads + per_block - 1) / per_block);
#define LAUNCH_VAR(GG) hipLaunchKernelGGL(k_op_var<GG>, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, V, sd, cpID_shift, \
                                          el_g_cpIDs, x, target, itg_hostIDs, elIDs, (int64_t)0, n_threads)
  if (G == 8) LAUNCH_VAR(8); else if (G == 16) LAUNCH_VAR(16); else if (G == 32) LAUNCH_VAR(32); else LAUNCH_VAR(64);
#undef LAUNCH_VAR
  MFEM_CHECK_LAUNCH();
  return MFEM_OK;
}

extern "C" int mfem_op_res(mfem_context ctx, const mfem_op_layout* L, const double* itp_vals, int32_t dual_sd, const double* vals,
                           int64_t cpID_shift, const int32_t* el_g_cpIDs, double* residue, const int32_t* itg_hostIDs,
                           const int32_t* elIDs, int64_t n_threads) {
  MFEM_REQUIRE(ctx, "null ctx");
  int rc = check_layout(L);
  if (rc) return rc;
  if (n_threads == 0) return MFEM_OK;
  MFEM_REQUIRE(n_threads > 0 && itp_vals && vals && el_g_cpIDs && residue && itg_hostIDs && elIDs, "null array");
  MFEM_REQUIRE(dual_sd >= 0 && dual_sd < L->n_sd, "sd out of range");
  OpView V{L->itg, L->itp, L->n_sd, L->index_base, itp_vals};
  const int G = group_for(L->itp);
  const int per_block = MFEM_BLOCK / G;
  const bool atomic = L->n_colours == 0;
  return for_each_batch(L, n_threads, [&](int64_t a, int64_t b) -> int {
    const int grid = (int)((b - a + per_block - 1) / per_block);
#define LAUNCH_RES(GG, AT) hipLaunchKernelGGL((k_op_res<GG, AT>), dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, V, dual_sd, vals, \
                                              cpID_shift, el_g_cpIDs, residue, itg_hostIDs, elIDs, a, b)
    if (atomic) { if (G == 8) LAUNCH_RES(8, true); else if (G == 16) LAUNCH_RES(16, true); else if (G == 32) LAUNCH_RES(32, true); else LAUNCH_RES(64, true); }
    else        { if (G == 8) LAUNCH_RES(8, false); else if (G == 16) LAUNCH_RES(16, false); else if (G == 32) LAUNCH_RES(32, false); else LAUNCH_RES(64, false); }
#undef LAUNCH_RES
    MFEM_CHECK_LAUNCH();
    return MFEM_OK;
  });
}

extern "C" int mfem_op_kval(mfem_context ctx, const mfem_op_layout* L, const double* itp_vals, int32_t dual_sd, int32_t base_sd,
                            const double* vals, const int32_t* sparse_IDs_by_el, int64_t sparse_ID_shift, double* K_val,
                            const int32_t* itg_hostIDs, const int32_t* elIDs, int64_t n_threads) {
  MFEM_REQUIRE(ctx, "null ctx");
  int rc = check_layout(L);
  if (rc) return rc;
  if (n_threads == 0) return MFEM_OK;
  MFEM_REQUIRE(n_threads > 0 && itp_vals && vals && sparse_IDs_by_el && K_val && itg_hostIDs && elIDs, "null array");
  MFEM_REQUIRE(dual_sd >= 0 && dual_sd < L->n_sd && base_sd >= 0 && base_sd < L->n_sd, "sd out of range");
  OpView V{L->itg, L->itp, L->n_sd, L->index_base, itp_vals};
  const size_t lds = sizeof(double) * 4 * (2 * (size_t)L->itg * L->itp + L->itg);
  MFEM_REQUIRE(lds <= 64 * 1024, "element too large for the LDS-staged operator (itg*itp > 1000)");
  const bool atomic = L->n_colours == 0;
  return for_each_batch(L, n_threads, [&](int64_t a, int64_t b) -> int {
    const int grid = (int)((b - a + 3) / 4);
    if (atomic)
      hipLaunchKernelGGL(k_op_kval<true>, dim3(grid), dim3(MFEM_BLOCK), lds, ctx->stream, V, dual_sd, base_sd, vals,
                         sparse_IDs_by_el, sparse_ID_shift, K_val, itg_hostIDs, elIDs, a, b);
    else
      hipLaunchKernelGGL(k_op_kval<false>, dim3(grid), dim3(MFEM_BLOCK), lds, ctx->stream, V, dual_sd, base_sd, vals,
                         sparse_IDs_by_el, sparse_ID_shift, K_val, itg_hostIDs, elIDs, a, b);
    MFEM_CHECK_LAUNCH();
    return MFEM_OK;
  });
}
