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

int g_op_wave_min_itp = 10;  // mfem_debug_set("op_wave_forms", a, b): b > 0 = elements from b nodes take the wave forms (default 10: tet-10, hex-20, hex-27; measured on tet-10 64^3: residual 7.1 -> 4.9 ms thermal, 16.1 -> 9.2 ms elasticity)
int g_op_wave_forms = 1;  // mfem_debug_set("op_wave_forms"): 0 = the sub-wave forms of the batched var / res operators on every element (A/B, tests); 2 = the wave forms for any item count (tests on small meshes)

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
                           const int32_t* elIDs, int64_t n_threads) try {
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
} MFEM_API_CATCH("mfem_op_var")

extern "C" int mfem_op_res(mfem_context ctx, const mfem_op_layout* L, const double* itp_vals, int32_t dual_sd, const double* vals,
                           int64_t cpID_shift, const int32_t* el_g_cpIDs, double* residue, const int32_t* itg_hostIDs,
                           const int32_t* elIDs, int64_t n_threads) try {
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
} MFEM_API_CATCH("mfem_op_res")

extern "C" int mfem_op_kval(mfem_context ctx, const mfem_op_layout* L, const double* itp_vals, int32_t dual_sd, int32_t base_sd,
                            const double* vals, const int32_t* sparse_IDs_by_el, int64_t sparse_ID_shift, double* K_val,
                            const int32_t* itg_hostIDs, const int32_t* elIDs, int64_t n_threads) try {
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
} MFEM_API_CATCH("mfem_op_kval")


// =====================================================================================================
// Batched operators: all terms of one integration domain in one launch (see include/metafem_mi355x.h).
// =====================================================================================================
struct KvalTerms {
  int n;
  mfem_kval_term t[MFEM_MAX_BATCH_TERMS];
};
struct ResTerms {
  int n;
  mfem_res_term t[MFEM_MAX_BATCH_TERMS];
};
struct VarTerms {
  int n;
  mfem_var_term t[MFEM_MAX_BATCH_TERMS];
};

// WPI waves per item (1, 2 or 4: large elements spread their itp^2 pairs over several waves that share one LDS copy of the
// table).  LDS per item: the item's whole table [itg, itp, n_sd] + vals of all terms [itg, n_terms].
template <bool ATOMIC>
__global__ __launch_bounds__(MFEM_BLOCK) void k_op_kval_batch(OpView V, KvalTerms T, int wpi, const double* __restrict__ vals,
                                                                int64_t term_stride, const int32_t* __restrict__ slots,
                                                                int64_t block_stride, int64_t shift_unit, double* __restrict__ K,
                                                                const int32_t* __restrict__ host_ids,
                                                                const int32_t* __restrict__ el_ids, int64_t t0, int64_t t1) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int ipw = (blockDim.x >> 6) / wpi;  // items per workgroup
  const int item = w / wpi, sub = w - item * wpi;
  const int sz = V.itg * V.itp;
  double* N = lds + (size_t)item * ((size_t)V.n_sd * sz + (size_t)T.n * V.itg);
  double* vq = N + (size_t)V.n_sd * sz;
  const int64_t t = t0 + (int64_t)blockIdx.x * ipw + item;
  const bool valid = t < t1;
  int64_t el = 0;
  if (valid) {
    el = (int64_t)el_ids[t] - V.base;
    const int64_t host = (int64_t)host_ids[t] - V.base;
    const double* g = slab(V, 0, host);
    const int step = 64 * wpi, total = V.n_sd * sz;
    for (int b = sub * 64 + lane; b < total; b += 8 * step) {  // eight loads of a lane in flight (a copy loop of unknown trip count waits for every load)
      double r[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) r[u] = b + u * step < total ? g[b + u * step] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (b + u * step < total) N[b + u * step] = r[u];
    }
    for (int i = sub * 64 + lane; i < T.n * V.itg; i += 64 * wpi) {
      const int term = i / V.itg, q = i - term * V.itg;
      vq[i] = vals[term * term_stride + q + (int64_t)V.itg * t];
    }
  }
  __syncthreads();
  if (!valid) return;
  const int npair = V.itp * V.itp;
  for (int p = sub * 64 + lane; p < npair; p += 64 * wpi) {
    const int a = p % V.itp, b = p / V.itp;
    int i = 0;
    while (i < T.n) {  // runs of terms with the same block: one accumulate per run
      const int block = T.t[i].block;
      double sum = 0.0;
      for (; i < T.n && T.t[i].block == block; ++i) {
        const double* Nd = N + (size_t)T.t[i].dual_sd * sz + V.itg * a;
        const double* Nb = N + (size_t)T.t[i].base_sd * sz + V.itg * b;
        const double* v = vq + i * V.itg;
        for (int q = 0; q < V.itg; ++q) sum += Nd[q] * Nb[q] * v[q];
      }
      double* dst = K + ((int64_t)slots[block * block_stride + (int64_t)npair * el + p] + block * shift_unit - V.base);
      if (ATOMIC) atomicAdd(dst, sum); else *dst += sum;
    }
  }
}

// GROUP lanes per item, lane -> a; terms with the same cpID_shift (dual field) are summed before the accumulate.
template <int GROUP, bool ATOMIC>
__global__ __launch_bounds__(MFEM_BLOCK) void k_op_res_batch(OpView V, ResTerms T, const double* __restrict__ vals, int64_t term_stride,
                                                               const int32_t* __restrict__ cp, double* __restrict__ residue,
                                                               const int32_t* __restrict__ host_ids,
                                                               const int32_t* __restrict__ el_ids, int64_t t0, int64_t t1) {
  const int g = threadIdx.x % GROUP;
  const int64_t t = t0 + ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / GROUP;
  if (t >= t1) return;
  const int64_t el = (int64_t)el_ids[t] - V.base, host = (int64_t)host_ids[t] - V.base;
  for (int a = g; a < V.itp; a += GROUP) {
    const int64_t node = (int64_t)cp[a + (int64_t)V.itp * el] - V.base;
    int i = 0;
    while (i < T.n) {
      const int64_t shift = T.t[i].cpID_shift;
      double acc = 0.0;
      for (; i < T.n && T.t[i].cpID_shift == shift; ++i) {
        const double* Ns = slab(V, T.t[i].dual_sd, host) + V.itg * a;
        const double* v = vals + i * term_stride + (int64_t)V.itg * t;
        for (int q = 0; q < V.itg; ++q) acc += Ns[q] * v[q];
      }
      double* dst = residue + node + shift;
      if (ATOMIC) atomicAdd(dst, acc); else *dst += acc;
    }
  }
}

// GROUP lanes per item, lane -> q; every term writes its own target array.
template <int GROUP>
__global__ __launch_bounds__(MFEM_BLOCK) void k_op_var_batch(OpView V, VarTerms T, const int32_t* __restrict__ cp,
                                                               double* __restrict__ targets, int64_t term_stride,
                                                               const int32_t* __restrict__ host_ids,
                                                               const int32_t* __restrict__ el_ids, int64_t n) {
  const int g = threadIdx.x % GROUP;
  const int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / GROUP;
  if (t >= n) return;
  const int64_t el = (int64_t)el_ids[t] - V.base, host = (int64_t)host_ids[t] - V.base;
  const int32_t* cpe = cp + (int64_t)V.itp * el;
  for (int i = 0; i < T.n; ++i) {
    const double* Ns = slab(V, T.t[i].sd, host);
    const double* x = T.t[i].x + T.t[i].cpID_shift - V.base;
    for (int q = g; q < V.itg; q += GROUP) {
      double acc = 0.0;
      for (int a = 0; a < V.itp; ++a) acc += Ns[q + V.itg * a] * x[cpe[a]];
      targets[i * term_stride + q + (int64_t)V.itg * t] = acc;
    }
  }
}

// ---- wave-per-item forms of the two batched operators for elements of 10..64 nodes (round 6) ------------------------------------------
// The sub-wave forms above give a lane a node (res) or a Gauss point (var) and let it walk its own 216-byte run of the table: 20 lanes of an element pull
// 20 different sectors per load, and every term re-reads its slab -- hex-20 96^3: 12.0 ms per residual of four terms (1.3 TB/s), 9.3 ms for the nine
// inner variables of elasticity.  Here a PERSISTENT wave owns an item: the slabs the terms use are copied to LDS once with unit-stride lanes (each slab
// read once per item however many terms use it), the terms' (slab, shift) sit in lanes and are broadcast with readlane (no scalar loads per term), and
// the short loop dimension is split over the 64 / itp (res: 3 for hex-20) or 64 / itg (var: 2) lane groups, partial sums joined with one shuffle each.
struct SlabUse {
  int lo, n;  // the slabs lo .. lo + n - 1 (first to last one a term uses: contiguous in an item's table) are copied
};
typedef double op_d2 __attribute__((ext_vector_type(2)));
// `count` doubles from global memory to LDS, 8 loads of a lane in flight (a plain copy loop of unknown trip count compiles to load - wait - store per
// trip: 27 round trips per hex-20 item); 16-byte loads when both sides allow
__device__ __forceinline__ void op_stage(double* __restrict__ dst, const double* __restrict__ src, int count, int lane) {
  if ((count & 1) == 0 && ((uintptr_t)src & 15) == 0) {
    const op_d2* s2 = (const op_d2*)src;
    op_d2* d2 = (op_d2*)dst;
    const int c2 = count >> 1;
    for (int b = 0; b < c2; b += 512) {
      op_d2 r[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = b + u * 64 + lane;
        r[u] = i < c2 ? __builtin_nontemporal_load(s2 + i) : op_d2{0.0, 0.0};
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = b + u * 64 + lane;
        if (i < c2) d2[i] = r[u];
      }
    }
  } else {
    for (int b = 0; b < count; b += 512) {
      double r[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = b + u * 64 + lane;
        r[u] = i < count ? src[i] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = b + u * 64 + lane;
        if (i < count) dst[i] = r[u];
      }
    }
  }
}

template <bool ATOMIC>
__global__ __launch_bounds__(MFEM_BLOCK) void k_op_res_batch_wave(OpView V, ResTerms T, SlabUse U, const double* __restrict__ vals, int64_t term_stride,
                                                                    const int32_t* __restrict__ cp, double* __restrict__ residue,
                                                                    const int32_t* __restrict__ host_ids, const int32_t* __restrict__ el_ids,
                                                                    int64_t t0, int64_t t1) {
  extern __shared__ double sm[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int tab = V.itg * V.itp;
  const size_t per_wave = (size_t)U.n * tab + (size_t)T.n * V.itg;
  double* S = sm + (size_t)w * per_wave;  // [slab][a][q]
  double* Vq = S + (size_t)U.n * tab;     // [term][q]
  // the terms: lane i holds term i
  int my_slot = 0;
  long long my_shift = 0;
  if (lane < T.n) {
    my_slot = T.t[lane].dual_sd - U.lo;
    my_shift = T.t[lane].cpID_shift;
  }
  const int H = 64 / V.itp, h = lane / V.itp, a = lane - h * V.itp;
  const int qc = (V.itg + H - 1) / H, q0 = h * qc, q1 = h < H ? (q0 + qc < V.itg ? q0 + qc : V.itg) : q0;
  for (int64_t t = t0 + (int64_t)blockIdx.x * nw + w; t < t1; t += (int64_t)gridDim.x * nw) {
    const int64_t el = (int64_t)el_ids[t] - V.base, host = (int64_t)host_ids[t] - V.base;
    op_stage(S, slab(V, U.lo, host), U.n * tab, lane);
    for (int b = 0; b < T.n * V.itg; b += 256) {  // (four loads of a lane in flight)
      double r[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = b + u * 64 + lane;
        const int term = i / V.itg, q = i - term * V.itg;
        r[u] = i < T.n * V.itg ? vals[term * term_stride + q + (int64_t)V.itg * t] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = b + u * 64 + lane;
        if (i < T.n * V.itg) Vq[i] = r[u];
      }
    }
    const int64_t node = h == 0 ? (int64_t)cp[a + (int64_t)V.itp * el] - V.base : 0;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    int i = 0;
    while (i < T.n) {
      const long long shift = __shfl(my_shift, i);
      double acc = 0.0;
      for (; i < T.n && __shfl(my_shift, i) == shift; ++i) {
        const double* Ns = S + (size_t)__shfl(my_slot, i) * tab + V.itg * a;
        const double* v = Vq + i * V.itg;
        for (int q = q0; q < q1; ++q) acc += Ns[q] * v[q];
      }
      double tot = acc;
      for (int o = 1; o < H; ++o) tot += __shfl(acc, lane + o * V.itp);  // (lanes of group 0 collect; the others' sums are not used)
      if (h == 0) {
        double* dst = residue + node + shift;
        if (ATOMIC) atomicAdd(dst, tot); else *dst += tot;
      }
    }
    __builtin_amdgcn_wave_barrier();  // the next item's copy overwrites S
  }
}

__global__ __launch_bounds__(MFEM_BLOCK) void k_op_var_batch_wave(OpView V, VarTerms T, SlabUse U, const int32_t* __restrict__ cp,
                                                                    double* __restrict__ targets, int64_t term_stride,
                                                                    const int32_t* __restrict__ host_ids, const int32_t* __restrict__ el_ids, int64_t n) {
  extern __shared__ double sm[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int tab = V.itg * V.itp;
  const size_t per_wave = (size_t)U.n * tab + (size_t)T.n * V.itp;
  double* S = sm + (size_t)w * per_wave;  // [slab][a][q]
  double* Xa = S + (size_t)U.n * tab;     // [term][a]: the nodal values of the term's source vector
  int my_slot = 0;
  long long my_x = 0;  // lane i: the source vector of term i, shifted to the term's field (fetched by the gathering lanes with a lane permute)
  if (lane < T.n) {
    my_slot = T.t[lane].sd - U.lo;
    my_x = (long long)(T.t[lane].x + T.t[lane].cpID_shift - V.base);
  }
  const int H = 64 / V.itg, h = lane / V.itg, q = lane - h * V.itg;
  const int ac = (V.itp + H - 1) / H, a0 = h * ac, a1 = h < H ? (a0 + ac < V.itp ? a0 + ac : V.itp) : a0;
  for (int64_t t = (int64_t)blockIdx.x * nw + w; t < n; t += (int64_t)gridDim.x * nw) {
    const int64_t el = (int64_t)el_ids[t] - V.base, host = (int64_t)host_ids[t] - V.base;
    const int32_t* cpe = cp + (int64_t)V.itp * el;
    op_stage(S, slab(V, U.lo, host), U.n * tab, lane);
    for (int b = 0; b < T.n * V.itp; b += 256) {  // (every lane takes part in the permutes; four gathers of a lane in flight)
      double r[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = b + u * 64 + lane;
        const bool on = i < T.n * V.itp;
        const int term = on ? i / V.itp : 0, a = on ? i - term * V.itp : 0;
        const double* xs = (const double*)__shfl(my_x, term);
        r[u] = on ? xs[cpe[a]] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = b + u * 64 + lane;
        if (i < T.n * V.itp) Xa[i] = r[u];
      }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    for (int i = 0; i < T.n; ++i) {
      const double* Ns = S + (size_t)__shfl(my_slot, i) * tab + q;
      const double* xa = Xa + i * V.itp;
      double acc = 0.0;
      for (int a = a0; a < a1; ++a) acc += Ns[V.itg * a] * xa[a];
      double tot = acc;
      for (int o = 1; o < H; ++o) tot += __shfl(acc, lane + o * V.itg);
      if (h == 0) targets[i * term_stride + q + (int64_t)V.itg * t] = tot;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

extern "C" int mfem_op_kval_batch(mfem_context ctx, const mfem_op_layout* L, const double* itp_vals, int32_t n_terms,
                                  const mfem_kval_term* terms, const double* vals, const int32_t* sparse_IDs_by_el,
                                  int64_t slot_block_stride, int64_t sparse_ID_shift_unit, double* K_val,
                                  const int32_t* itg_hostIDs, const int32_t* elIDs, int64_t n_threads) try {
  MFEM_REQUIRE(ctx, "null ctx");
  int rc = check_layout(L);
  if (rc) return rc;
  if (n_threads == 0 || n_terms == 0) return MFEM_OK;
  MFEM_REQUIRE(n_threads > 0 && n_terms > 0 && n_terms <= MFEM_MAX_BATCH_TERMS, "n_terms must be 1..MFEM_MAX_BATCH_TERMS");
  MFEM_REQUIRE(terms && itp_vals && vals && sparse_IDs_by_el && K_val && itg_hostIDs && elIDs, "null array");
  KvalTerms T;
  T.n = n_terms;
  for (int i = 0; i < n_terms; ++i) {
    MFEM_REQUIRE(terms[i].dual_sd >= 0 && terms[i].dual_sd < L->n_sd && terms[i].base_sd >= 0 && terms[i].base_sd < L->n_sd, "sd out of range");
    MFEM_REQUIRE(terms[i].block >= 0 && (i == 0 || terms[i].block >= terms[i - 1].block), "terms must be sorted by block");
    T.t[i] = terms[i];
  }
  OpView V{L->itg, L->itp, L->n_sd, L->index_base, itp_vals};
  const size_t per_item = sizeof(double) * ((size_t)L->n_sd * L->itg * L->itp + (size_t)n_terms * L->itg);
  const int npair = L->itp * L->itp;
  const int wpi = npair > 128 ? 4 : npair > 64 ? 2 : 1;  // waves per item
  const int waves = 4, ipw = waves / wpi;
  MFEM_REQUIRE(per_item * ipw <= 64 * 1024, "element table too large for the LDS-staged batched operator");
  const size_t lds = per_item * ipw;
  const int64_t term_stride = (int64_t)L->itg * n_threads;
  const bool atomic = L->n_colours == 0;
  return for_each_batch(L, n_threads, [&](int64_t a, int64_t b) -> int {
    const int grid = (int)((b - a + ipw - 1) / ipw);
    if (atomic)
      hipLaunchKernelGGL(k_op_kval_batch<true>, dim3(grid), dim3(64 * waves), lds, ctx->stream, V, T, wpi, vals, term_stride,
                         sparse_IDs_by_el, slot_block_stride, sparse_ID_shift_unit, K_val, itg_hostIDs, elIDs, a, b);
    else
      hipLaunchKernelGGL(k_op_kval_batch<false>, dim3(grid), dim3(64 * waves), lds, ctx->stream, V, T, wpi, vals, term_stride,
                         sparse_IDs_by_el, slot_block_stride, sparse_ID_shift_unit, K_val, itg_hostIDs, elIDs, a, b);
    MFEM_CHECK_LAUNCH();
    return MFEM_OK;
  });
} MFEM_API_CATCH("mfem_op_kval_batch")

extern "C" int mfem_op_res_batch(mfem_context ctx, const mfem_op_layout* L, const double* itp_vals, int32_t n_terms,
                                 const mfem_res_term* terms, const double* vals, const int32_t* el_g_cpIDs, double* residue,
                                 const int32_t* itg_hostIDs, const int32_t* elIDs, int64_t n_threads) try {
  MFEM_REQUIRE(ctx, "null ctx");
  int rc = check_layout(L);
  if (rc) return rc;
  if (n_threads == 0 || n_terms == 0) return MFEM_OK;
  MFEM_REQUIRE(n_threads > 0 && n_terms > 0 && n_terms <= MFEM_MAX_BATCH_TERMS, "n_terms must be 1..MFEM_MAX_BATCH_TERMS");
  MFEM_REQUIRE(terms && itp_vals && vals && el_g_cpIDs && residue && itg_hostIDs && elIDs, "null array");
  ResTerms T;
  T.n = n_terms;
  for (int i = 0; i < n_terms; ++i) {
    MFEM_REQUIRE(terms[i].dual_sd >= 0 && terms[i].dual_sd < L->n_sd, "sd out of range");
    MFEM_REQUIRE(i == 0 || terms[i].cpID_shift >= terms[i - 1].cpID_shift, "terms must be sorted by cpID_shift");
    T.t[i] = terms[i];
  }
  OpView V{L->itg, L->itp, L->n_sd, L->index_base, itp_vals};
  const int G = group_for(L->itp);
  const int per_block = MFEM_BLOCK / G;
  const int64_t term_stride = (int64_t)L->itg * n_threads;
  const bool atomic = L->n_colours == 0;
  {  // elements of 16..64 nodes: a persistent wave per item, the slabs in LDS (k_op_res_batch_wave)
    int sd_lo = terms[0].dual_sd, sd_hi = terms[0].dual_sd;
    for (int i = 1; i < n_terms; ++i) {
      sd_lo = terms[i].dual_sd < sd_lo ? terms[i].dual_sd : sd_lo;
      sd_hi = terms[i].dual_sd > sd_hi ? terms[i].dual_sd : sd_hi;
    }
    const SlabUse U{sd_lo, sd_hi - sd_lo + 1};
    const size_t per_wave = sizeof(double) * ((size_t)U.n * L->itg * L->itp + (size_t)n_terms * L->itg);
    if (g_op_wave_forms && L->itp >= g_op_wave_min_itp && L->itp <= 64 && n_terms <= 64 && per_wave * 4 <= 80 * 1024 && n_threads >= (g_op_wave_forms == 2 ? 1 : 256)) {
      const size_t ldsb = per_wave * 4;
      const int per_cu = (int)(160 * 1024 / ldsb);
      return for_each_batch(L, n_threads, [&](int64_t a, int64_t b) -> int {
        int grid = (int)((b - a + 3) / 4);
        const int cap = ctx->num_cus * (per_cu < 1 ? 1 : per_cu > 4 ? 4 : per_cu);
        if (grid > cap) grid = cap;
        if (atomic) {
          if (ldsb > 64 * 1024) MFEM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_op_res_batch_wave<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
          hipLaunchKernelGGL(k_op_res_batch_wave<true>, dim3(grid), dim3(MFEM_BLOCK), ldsb, ctx->stream, V, T, U, vals, term_stride, el_g_cpIDs, residue,
                             itg_hostIDs, elIDs, a, b);
        } else {
          if (ldsb > 64 * 1024) MFEM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_op_res_batch_wave<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
          hipLaunchKernelGGL(k_op_res_batch_wave<false>, dim3(grid), dim3(MFEM_BLOCK), ldsb, ctx->stream, V, T, U, vals, term_stride, el_g_cpIDs, residue,
                             itg_hostIDs, elIDs, a, b);
        }
        MFEM_CHECK_LAUNCH();
        return MFEM_OK;
      });
    }
  }
  return for_each_batch(L, n_threads, [&](int64_t a, int64_t b) -> int {
    const int grid = (int)((b - a + per_block - 1) / per_block);
#define LAUNCH_RESB(GG, AT) hipLaunchKernelGGL((k_op_res_batch<GG, AT>), dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, V, T, vals, \
                                               term_stride, el_g_cpIDs, residue, itg_hostIDs, elIDs, a, b)
    if (atomic) { if (G == 8) LAUNCH_RESB(8, true); else if (G == 16) LAUNCH_RESB(16, true); else if (G == 32) LAUNCH_RESB(32, true); else LAUNCH_RESB(64, true); }
    else        { if (G == 8) LAUNCH_RESB(8, false); else if (G == 16) LAUNCH_RESB(16, false); else if (G == 32) LAUNCH_RESB(32, false); else LAUNCH_RESB(64, false); }
#undef LAUNCH_RESB
    MFEM_CHECK_LAUNCH();
    return MFEM_OK;
  });
} MFEM_API_CATCH("mfem_op_res_batch")

extern "C" int mfem_op_var_batch(mfem_context ctx, const mfem_op_layout* L, const double* itp_vals, int32_t n_terms,
                                 const mfem_var_term* terms, const int32_t* el_g_cpIDs, double* targets,
                                 const int32_t* itg_hostIDs, const int32_t* elIDs, int64_t n_threads) try {
  MFEM_REQUIRE(ctx, "null ctx");
  int rc = check_layout(L);
  if (rc) return rc;
  if (n_threads == 0 || n_terms == 0) return MFEM_OK;
  MFEM_REQUIRE(n_threads > 0 && n_terms > 0 && n_terms <= MFEM_MAX_BATCH_TERMS, "n_terms must be 1..MFEM_MAX_BATCH_TERMS");
  MFEM_REQUIRE(terms && itp_vals && el_g_cpIDs && targets && itg_hostIDs && elIDs, "null array");
  VarTerms T;
  T.n = n_terms;
  for (int i = 0; i < n_terms; ++i) {
    MFEM_REQUIRE(terms[i].sd >= 0 && terms[i].sd < L->n_sd && terms[i].x, "sd out of range or null x");
    T.t[i] = terms[i];
  }
  OpView V{L->itg, L->itp, L->n_sd, L->index_base, itp_vals};
  const int G = group_for(L->itg);
  const int per_block = MFEM_BLOCK / G;
  const int64_t term_stride = (int64_t)L->itg * n_threads;
  {  // elements of 16+ nodes with up to 64 Gauss points: a persistent wave per item, the slabs in LDS (k_op_var_batch_wave)
    int sd_lo = terms[0].sd, sd_hi = terms[0].sd;
    for (int i = 1; i < n_terms; ++i) {
      sd_lo = terms[i].sd < sd_lo ? terms[i].sd : sd_lo;
      sd_hi = terms[i].sd > sd_hi ? terms[i].sd : sd_hi;
    }
    const SlabUse U{sd_lo, sd_hi - sd_lo + 1};
    const size_t per_wave = sizeof(double) * ((size_t)U.n * L->itg * L->itp + (size_t)n_terms * L->itp);
    if (g_op_wave_forms && L->itp >= g_op_wave_min_itp && L->itg <= 64 && n_terms <= 64 && per_wave * 4 <= 80 * 1024 && n_threads >= (g_op_wave_forms == 2 ? 1 : 256)) {
      const size_t ldsb = per_wave * 4;
      const int per_cu = (int)(160 * 1024 / ldsb);
      int gridw = (int)((n_threads + 3) / 4);
      const int cap = ctx->num_cus * (per_cu < 1 ? 1 : per_cu > 4 ? 4 : per_cu);
      if (gridw > cap) gridw = cap;
      if (ldsb > 64 * 1024) MFEM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_op_var_batch_wave), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
      hipLaunchKernelGGL(k_op_var_batch_wave, dim3(gridw), dim3(MFEM_BLOCK), ldsb, ctx->stream, V, T, U, el_g_cpIDs, targets, term_stride, itg_hostIDs, elIDs,
                         n_threads);
      MFEM_CHECK_LAUNCH();
      return MFEM_OK;
    }
  }
  const int grid = (int)((n_threads + per_block - 1) / per_block);
#define LAUNCH_VARB(GG) hipLaunchKernelGGL(k_op_var_batch<GG>, dim3(grid), dim3(MFEM_BLOCK), 0, ctx->stream, V, T, el_g_cpIDs, targets, \
                                           term_stride, itg_hostIDs, elIDs, n_threads)
  if (G == 8) LAUNCH_VARB(8); else if (G == 16) LAUNCH_VARB(16); else if (G == 32) LAUNCH_VARB(32); else LAUNCH_VARB(64);
#undef LAUNCH_VARB
  MFEM_CHECK_LAUNCH();
  return MFEM_OK;
} MFEM_API_CATCH("mfem_op_var_batch")
