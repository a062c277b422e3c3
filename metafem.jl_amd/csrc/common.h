// Shared internals of libmetafem_mi355x.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <mutex>
#include <unordered_map>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "metafem_mi355x is written for gfx950 (MI355X) only: counted s_waitcnt vmcnt(k) pipelines assume loads and stores share one in-order counter, global_load_lds_dwordx4 and v_mfma_f64 are used directly -- build with --offload-arch=gfx950"
#endif
#include "../../include/metafem_mi355x.h"
#include "../../include/metafem_mi355x_debug.h"

#define MFEM_WAVE 64
#define MFEM_BLOCK 256
#define MFEM_MAX_PARTIALS 4096
#define MFEM_NSCALARS 4096       // device-resident Krylov scalars (doubles)  // upper bound on per-launch partial sums of a fused reduction

void mfem_set_error(const char* fmt, ...);

// Every `extern "C" int` entry point is a function-try-block closed by this handler: no C++ exception leaves the library (include/metafem_mi355x.h,
// error convention).  mfem_api_exception (api.hip) rethrows the exception in flight, maps it to a status and sets mfem_last_error().
int mfem_api_exception(const char* entry) noexcept;
#define MFEM_API_CATCH(entry) catch (...) { return mfem_api_exception(entry); }
// Called in front of the library's host allocations (new / std::vector): throws std::bad_alloc when mfem_debug_fail_host_alloc armed it -- the
// test hook that shows the handler above at work (tests/test_gpu_round4_abi.py).
void mfem_host_alloc_probe();

#define MFEM_CHECK_HIP(expr)                                                                  \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      mfem_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));    \
      return MFEM_ERR_HIP;                                                                    \
    }                                                                                         \
  } while (0)

#define MFEM_REQUIRE(cond, msg)                                      \
  do {                                                               \
    if (!(cond)) {                                                   \
      mfem_set_error("%s:%d: %s (%s)", __FILE__, __LINE__, msg, #cond); \
      return MFEM_ERR_INVALID;                                       \
    }                                                                \
  } while (0)

#define MFEM_CHECK_LAUNCH() MFEM_CHECK_HIP(hipGetLastError())

struct mfem_comm_s;

// Row split of a slab SpMV (multi-GPU): the rows that reference ghost columns -- the first / last halo_plane_len owned rows
// of every field next to a neighbour rank ("boundary zones") -- against all other rows.  The interior part runs while the
// halo exchange is in flight, the boundary part after it (spmv.hip: mfem_spmv_halo).  Kernels test whole work units
// (tiles / chunks / a lane's rows): a unit that touches a zone belongs to the boundary part, so every row is computed once.
#define MFEM_MAX_ZONES 8
struct SpmvPart {
  int part;  // 0 = all rows, 1 = interior units only, 2 = boundary units only
  int nz;
  int64_t lo[MFEM_MAX_ZONES], hi[MFEM_MAX_ZONES];
};
__host__ __device__ __forceinline__ bool spmv_part_skip(const SpmvPart& P, int64_t r0, int64_t r1) {
  if (P.part == 0) return false;
  bool bnd = false;
  for (int z = 0; z < P.nz; ++z) bnd = bnd || (r0 < P.hi[z] && r1 > P.lo[z]);
  return P.part == 1 ? bnd : !bnd;
}


#define MFEM_GRAPH_SLOTS 4
struct mfem_context_s {
  int device;
  hipStream_t stream;
  int num_cus;
  // reduction scratch: partial sums (device) + a small block of device scalars + pinned host mirror
  double* d_partials;   // [MFEM_MAX_PARTIALS * 8]
  double* d_scalars;    // [256] device-resident Krylov scalars
  double* h_scalars;    // pinned, [256]
  int32_t* d_flags;     // [24] device flags (16-17: the symmetry fingerprint of k_symp_fill, 64 bits): 0-7 the Krylov loop's two banks (done, iteration count, ...); 9 mirrored-sweep check; 10 ring self-test; 11 mesh
                        // assembly; 12-15 one-shot statistics, each user clears the slots it reads before its launch and synchronises after it (layout binds:
                        // 12-13 max |a| / symmetry measure, krylov.hip: 12-15 extremes of S, assemble_hex27.hip: 14 count of non-affine elements)
  int32_t* h_flags;     // pinned
  // generic workspace (grown on demand, never shrunk)
  void* ws;        // (ws_raw + the placement offset, see mfem_ws_reserve)
  void* ws_raw;    // what hipMalloc returned
  size_t ws_bytes;
  // placement of a large workspace (krylov.hip, solve_inner): the solver SpMV runs at one of two speeds depending on the physical memory an
  // allocation received (profiles/r03_placement_probe.txt); the first big solve on a workspace times it, tries ONE second allocation and keeps
  // the faster.  ws_try: candidates allocated after the first (the best so far kept in ws_alt*); 99: decided.
  int ws_try;
  void* ws_alt;
  void* ws_alt_raw;
  float ws_try_ms;
  float ws_log[4];  // the candidates' times (ms for two SpMVs), in the order tried; [3]: which one was kept (0-based)
  // optional user shadow vectors
  const double* shadow;
  int32_t shadow_count;
  // multi-GPU
  mfem_comm_s* comm;
  int64_t halo_plane_len;
  int32_t halo_fields;
  hipEvent_t ev0, ev1;
  // optional per-launch timing of the SpMV kernel (bench.py roofline): event pairs on ctx->stream
  int prof_on;
  int prof_used;
  hipEvent_t* prof_ev;      // [2 * MFEM_PROF_PAIRS]
  double prof_ms;
  int64_t prof_count;
  // hipGraph replay of launch-bound Krylov cycles (krylov.h: mfem_cycle_run): one cached executable graph, keyed by a
  // hash of everything its kernel arguments depend on; graph_stream stands in for the legacy null stream, which cannot
  // be captured
  hipGraphExec_t graph_exec[MFEM_GRAPH_SLOTS];   // small cache: a coupled problem alternates between a few matrices / solvers
  uint64_t graph_key[MFEM_GRAPH_SLOTS];
  int graph_next;                                // round-robin replacement
  hipStream_t graph_stream;
  hipEvent_t graph_ev;
  int graph_active;   // set by mfem_solve for the duration of a solve when cycles may be captured
  int force_csr;      // set while a product must run the CSR kernel on the caller's arrays whatever layout is bound (the residual a tile solve reports, krylov.hip)
  int probe_active;   // set while a measuring product runs on this context (symmetry probe, placement trial): not an SpMV a solver asked for -- the usage counters skip it
};
#define MFEM_PROF_PAIRS 1024
int mfem_prof_flush(mfem_context_s* ctx);

struct mfem_csr_s {
  mfem_context_s* ctx;
  uint64_t serial;          // unique per created pattern (cycle-graph cache key)
  int64_t n, nnz;
  const void* rowptr;
  int rowptr_bits;
  const int32_t* colidx;
  int index_base;
  // plan for the LDS-staged SpMV
  int32_t max_row_nnz;
  int32_t rows_per_block;  // power of two, 0 => long-row fallback
  // tiles cut by nonzeros for rows of uneven length (spmv.hip: k_spmv_csr_rb): rb_state 1 = planned, -1 = not used
  int rb_state;
  // node-blocked form of a field-major multi-field pattern (round 6): nb_F = fields F > 1 when the F rows of every node list the node's coupled nodes once
  // per column field (mfem_node_block_fields: checked entry by entry once per pattern -- nb_checked), 0 = no or not asked
  int nb_F, nb_checked;
  int64_t rb_ntiles;
  int64_t rb_elided;        // tiles of them whose columns the kernel derives from the tile's first two rows (bit 31 of rb_rows[t])
  int32_t* rb_rows;         // owned, [rb_ntiles + 1]: first row of every tile
  uint8_t* cw_elide;        // owned: one flag per tile of cw_R rows of the fixed-row-count wave-tile kernel (k_spmv_csr_w): columns derivable from the tile's first row
  int cw_R;
  int32_t lat_m1, lat_m2, lat_fields;  // lattice hint of a structured pattern (0 = none): points per lattice plane = lat_m1 * lat_m2 (brick.hip)
  int32_t lat_m0, lat_plo, lat_gw;     // ... planes of the whole lattice, first owned plane of a slab, ghost planes per side (0 = not given)
  int32_t lat_inferred;                // the hint was read off row 0 of a caller-supplied pattern (mfem_lattice_from_first_row), not given by mfem_brick_pattern
  uint16_t* diag_off;       // owned, [n], built on first use: offset of the diagonal entry inside its row (0xFFFF = none stored): |diag| is then an
                            // n-sized gather instead of a scan of all nonzeros (Jacobi_By_Diagonal of every solve)
  // owned storage (mfem_brick_pattern) -- freed in destroy
  void* owned_rowptr;
  void* owned_colidx;
  // slab info (multi-GPU): rows = owned nodes, x has ghost planes; 0 for single GPU
  int64_t x_offset;  // offset of the first owned entry inside the local x (per field)
  int64_t ncols;     // columns the pattern addresses: n, or n + ghost entries for a slab pattern (0 = n)
  // slot-major padded copy for near-uniform rows (spmv_ell.hip): ell_state 0 = not planned, -1 = not eligible, 1 = ready
  int ell_state, ell_K;
  int64_t ell_npad;
  int32_t* ell_cols;        // owned, [K][npad], 0-based
  const double* ell_src;    // the CSR-ordered values the bound copy mirrors (identity of the `vals` argument)
  double* ell_vals;         // not owned (solver workspace), [K][npad]
  // diagonal-slotted variant (all entries on <= 32 diagonals): dia_state 0 = not inspected, -1 = no, 1 = yes
  int dia_state, dia_classes;
  void* dia_dev;            // owned, device copy of the diagonal lists (DiaOffsets, spmv_ell.hip)
  int32_t* dia_flags;       // owned, one int per 128-row block: 1 = regular (diagonal-slotted), 0 = explicit columns
  int32_t dia_regular_blocks;
  int dia_triples;          // the diagonals come in runs of three consecutive offsets
  // symmetric sweep variant of the diagonal-slotted SpMV (27-point lattice stencil): sym_state 0 = not inspected, -1 = no, 1 = structure ok
  int sym_state;
  int64_t sym_c0, sym_c1;   // chunks (512 rows) [c0, c1) whose blocks are all regular
  int sym_cls;              // the diagonal list (class) with the lattice form
  int sym_S;                // chunks per lattice plane (rounded): a workgroup sweeps chunks c, c + S, c + 2 S, ...
  int sym_bound;            // 1 = the bound values passed the bitwise symmetry check of this bind
  int64_t sym_mx, sym_myz;  // matrix entries per chunk the sweep kernel takes from LDS: previous-plane diagonals / in-chunk -y, -z
  int ell_bound_mode;       // 0 none, 1 slot-major with explicit columns, 2 diagonal-slotted
  // wave-private (j, k)-patch form of the symmetric sweep (spmv_ell.hip: k_spmv_symp): symp_state 0 = not inspected, -1 = no, 1 = structure ok
  int symp_state;
  int symp_m1, symp_m2;     // lattice lines per plane, points per line
  int64_t symp_PL;          // rows per lattice plane (m1 * m2)
  int symp_p0, symp_p1;     // regular lattice planes [p0, p1) (plane = row / PL): the rows the sweep computes
  int symp_NS, symp_NPk;    // strips of 4 lines, patches of 32 points per line
  double* symp_vals;        // not owned (solver workspace, behind ell_vals): [plane - p0][patch][27 x 128 + edge block]
  int symp_bound;           // 1 = the bound values passed the bitwise symmetry check of the mirrored pairs
  int64_t symp_pairs;       // value pairs (16 bytes) one SpMV of the sweep reads from memory (accounting)
  // row-sorted sliced ELL for rows of uneven length (spmv_sell.hip): sell_state 0 = not planned, -1 = no, 1 = ready
  int sell_state;
  int64_t sell_total, sell_nblk;
  int64_t sell_nb_int;      // leading blocks without a ghost-reading row (= sell_nblk for a pattern without ghost columns): the interior part of a split SpMV
  int32_t* sell_rowid;      // owned: sorted position -> row
  int64_t* sell_ptr;        // owned: [nblk + 1] start of each 128-row block in the sliced arrays
  int32_t* sell_cols;       // owned, [sell_total], 0-based
  int32_t* sell_flags;      // owned, [nblk]: 1 = all 128 rows share one diagonal list
  int32_t* sell_off;        // owned, [sell_total / 128]: that list, at ptr[b] / 128
  int32_t sell_regular_blocks;
  // field-periodic blocks (round 6): a field-major multi-field matrix repeats the node list of a row once per column field, shifted by the rows of a field;
  // a block whose 128 rows have one length K = F * P and columns col[f * P + t] = col[t] + f * shift reads the first P slots' columns only
  int32_t lat27_det;            // the bound copy of the hex-27 tiles is in the deterministic (lane = row, phase-major) form
  int32_t sell_fields;          // F (0: none found)
  int64_t sell_shift;           // column shift between two fields
  int32_t sell_periodic_blocks;
  int32_t sell_sig_sorted;      // 1: the diagonal-list signature took part in the row sort (lattice patterns); 0: mesh order within a length (unstructured)
  // node-blocked form (round 6, "BSELL"): a field-major F-field matrix whose F rows of a node share the node's coupling list (every FEM pattern of
  // mfem_pattern_build) -- a lane owns a NODE: per coupled node one column index, F gathers of x and F x F values.  bsell_F > 0: the sliced arrays are
  // sell_rowid = sorted position -> node, sell_ptr = [node blocks + 1] in node slots x 64, sell_cols = node-level columns, sell_total = value entries
  int32_t bsell_F;
  int64_t bsell_ncp, bsell_slots;
  const double* sell_src;
  double* sell_vals;        // not owned (solver workspace), [sell_total]
  // symmetric lattice-tile layout of the hex-27 lattice matrix (spmv_lat27.hip): lat27_state 0 = not inspected, -1 = no, 1 = the pattern is the stencil
  int lat27_state;
  const double* lat27_src;
  const double* lat27_dsc;  // not owned: right Jacobi scaling applied to x while it is staged (nullptr: none)
  double* lat27_vals;       // not owned (solver workspace): the stored (diagonal + upper) entries, unit by unit
  double* lat27_dump;       // not owned (behind lat27_vals): one y block per tile
  double lat27_asym;        // max |A[r][c] - A[c][r]| / max |A[r][c]| seen by the last bind
  int lat27_scaled;         // the last bind carried a right Jacobi scaling (accounting)
  int lat_refused;          // a lattice-tile bind has refused values of this pattern once (not symmetric): solves plan the other layouts too from then on
  // the same for the 3-field 27-point lattice matrix (hex-8 elasticity; spmv_lat8.hip)
  int lat8_state;
  const double* lat8_src;
  const double* lat8_dsc;
  double* lat8_vals;
  double* lat8_dump;
  double lat8_asym;
  int lat8_scaled;
  // A = S + N (spmv_rem.hip): the sparse skew remainder a lattice-tile bind carries when the values are nonsymmetric in a few rows only (Nitsche / SUPG
  // faces): owned device storage, grown on demand and kept between solves; rem_active: the CURRENT bind applies it after the tiles' gather pass
  int rem_active;
  int64_t rem_nrows, rem_nent, rem_cap_rows, rem_cap_ent;
  int32_t* rem_rows;   // [rem_nrows] the rows of N
  int64_t* rem_ptr;    // [rem_nrows] first entry of each
  int32_t* rem_len;    // [rem_nrows] entries of each
  int32_t* rem_col;    // [rem_nent] 0-based columns
  double* rem_val;     // [rem_nent] A[r][c] - A[c][r]
  unsigned long long* rem_cnt;  // device counters of the build
  void* rem_sort_tmp;           // owned: radix-sort scratch of the build, grown on demand (a nonsymmetric K rebuilds the remainder at every solve)
  size_t rem_sort_bytes;
  double rem_asym_before;       // what the probe measured on the tiles alone (the asymmetry the remainder repairs)
  int64_t rem_last_rows, rem_last_ent;  // what the last ACCEPTED remainder of the last probe held (0: none) -- survives the unbind at the end of a solve (tests, bench.py)
};
bool mfem_rem_enabled();
bool mfem_rem_diag();
void mfem_rem_clear(mfem_csr_s* A);
void mfem_rem_free(mfem_csr_s* A);
int mfem_rem_build(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, int n_fields, const double* y1, const double* y2, const double* scale,
                   double gate, bool* built);
int mfem_rem_apply(mfem_context_s* ctx, mfem_csr_s* A, const double* x, const double* dsc, double* y, double alpha, const double* dotw,
                   double* partials, int* n_partials, const int32_t* done_flag);
int64_t mfem_rem_design_bytes(const mfem_csr_s* A);
int mfem_lat8_plan(mfem_context_s* ctx, mfem_csr_s* A);
bool mfem_lat8_for_method(const mfem_csr_s* A, bool is_cg);  // one-field matrices: only the solvers that work on A D^-1 (cg! keeps the bitwise patch sweep)
size_t mfem_lat8_bytes(const mfem_csr_s* A);
// binds symmetric values, or -- allow_rem -- values whose asymmetry a sparse remainder (spmv_rem.hip) repairs; scratch: 3 n doubles
int mfem_lat8_bind(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, double* buf, const double* dsc, double* scratch, bool allow_rem = false);
void mfem_lat8_unbind(mfem_csr_s* A);
bool mfem_lat8_bound(const mfem_csr_s* A, const double* vals);
int mfem_spmv_lat8_launch(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, const double* x, double* y, double alpha, double beta,
                          const double* dotw, double* partials, int* n_partials, const int32_t* done_flag, int part);
int64_t mfem_lat8_design_bytes(const mfem_csr_s* A);
int64_t mfem_lat8_entries(const mfem_csr_s* A);
int mfem_lattice_from_first_row(mfem_context_s* ctx, mfem_csr_s* A);  // proposes lat_* for a pattern without a hint (spmv_lat27.hip)
// First i-layer of lattice tiles (8 planes each, `gw` planes of upward reach) that stages a ghost plane of the upper neighbour: the layers below it
// are the interior part of a slab's split SpMV.  m0 = owned planes; without an upper neighbour every layer is interior.
static inline int mfem_lat_first_ghost_layer(int m0, int gw, int nti, bool has_upper) {
  if (!has_upper) return nti;
  int t = m0 - 7 - gw;  // a layer's staged planes end at 8 ti + 7 + gw
  t = t <= 0 ? 0 : (t + 7) / 8;
  return t < nti ? t : nti;
}
int mfem_lat27_plan(mfem_context_s* ctx, mfem_csr_s* A);
size_t mfem_lat27_bytes(const mfem_csr_s* A);
int mfem_lat27_bind(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, double* buf, const double* dsc, double* scratch, bool allow_rem = false);
// rem_fields > 0: rows above the gate may be repaired by a remainder built for that many fields (then *asym is the measure of tiles + remainder and
// A->rem_active is set); 0: symmetric values only
int mfem_sym_probe(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, double* scratch, double amax, void (*unbind)(mfem_csr_s*),
                   void (*rebind)(mfem_csr_s*, void*), void* cookie, double* asym, int rem_fields = 0);
void mfem_lat27_unbind(mfem_csr_s* A);
bool mfem_lat27_bound(const mfem_csr_s* A, const double* vals);
int mfem_spmv_lat27_launch(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, const double* x, double* y, double alpha,
                           double beta, const double* dotw, double* partials, int* n_partials, const int32_t* done_flag, int part);
int64_t mfem_lat27_design_bytes(const mfem_csr_s* A);
int64_t mfem_lat27_entries(const mfem_csr_s* A);
int mfem_node_block_fields(mfem_context_s* ctx, mfem_csr_s* A);  // fills A->nb_F (spmv_sell.hip)
int mfem_sell_plan(mfem_context_s* ctx, mfem_csr_s* A);
size_t mfem_sell_vals_bytes(const mfem_csr_s* A);
int mfem_sell_bind(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, double* buf, const double* dsc);
void mfem_sell_unbind(mfem_csr_s* A);
void mfem_sell_free(mfem_csr_s* A);
int mfem_spmv_sell_launch(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, const double* x, double* y, double alpha,
                          double beta, const double* dotw, double* partials, int* n_partials, const int32_t* done_flag, int part);
bool mfem_sell_bound(const mfem_csr_s* A, const double* vals);  // the sliced layout (rows permuted; ghost-reading rows sorted last) serves these values
int mfem_ell_plan(mfem_context_s* ctx, mfem_csr_s* A);
size_t mfem_ell_vals_bytes(const mfem_csr_s* A);
int mfem_ell_bind(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, double* buf, const double* dsc, const double* ssym = nullptr);
bool mfem_dia_layout_planned(const mfem_csr_s* A);  // mfem_ell_bind would make the diagonal-slotted copy (mode 2)
bool mfem_symp_wanted(const mfem_csr_s* A);  // the symmetric patch sweep (mode 2) would be tried for this pattern
void mfem_ell_unbind(mfem_csr_s* A);
int mfem_ell_diag(mfem_context_s* ctx, mfem_csr_s* A, double* d);
void mfem_ell_free(mfem_csr_s* A);
int mfem_spmv_ell_launch(mfem_context_s* ctx, mfem_csr_s* A, const double* vals, const double* x, double* y, double alpha,
                         double beta, const double* dotw, double* partials, int* n_partials, const int32_t* done_flag,
                         const SpmvPart& part);

int mfem_ws_reserve(mfem_context_s* ctx, size_t bytes);
int mfem_ws_next_candidate(mfem_context_s* ctx);
int mfem_ws_decide(mfem_context_s* ctx, bool keep_current);
uint64_t mfem_next_csr_serial();
bool mfem_context_alive(mfem_context_s* ctx);       // false once mfem_context_destroy has run (api.hip)
void mfem_graphs_invalidate(mfem_context_s* ctx);  // drops every cached cycle graph of the context (api.hip)
extern std::atomic<int> mfem_debug_epoch;  // bumped by every mfem_debug_set_*: part of the cycle-graph cache key (api.hip)

// ---- device helpers ---------------------------------------------------------------------
__device__ __forceinline__ double wave_reduce_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, MFEM_WAVE);
  return v;
}

// Block-wide sum for blockDim.x == MFEM_BLOCK (4 waves). Result valid in thread 0.
// Workgroup barrier for data the waves exchange through LDS ONLY.  __syncthreads() is s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier: it also waits until every
// global load of the wave has returned and every global STORE has reached memory -- in a kernel that requests the next plane's data early, or streams its
// results out with stores, that is a memory round trip per barrier (round 5: found in k_hex27_rows_gq with the counters, then in the plane sweeps).  Here only
// the LDS counter is drained; global loads into registers are waited for where the registers are used (the compiler's own s_waitcnt), stores never.
// NOT for barriers that order global-memory accesses between the waves of a workgroup.
__device__ __forceinline__ void mfem_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ double block_reduce_sum(double v, double* smem /* >= 4 doubles */) {
  v = wave_reduce_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();  // a slower wave may still be reading smem from a preceding reduce_partials_bcast / block_reduce_sum
  if (lane == 0) smem[w] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int i = 0; i < nw; ++i) r += smem[i];
  }
  __syncthreads();
  return r;
}

// Workgroups of `kernel` a CU holds at once (registers, LDS, waves), from the runtime, cached per kernel; `fallback` when the query fails.  A PERSISTENT grid must
// be a multiple of what is resident: one workgroup more per CU than fits runs as a second round while the others idle (round 5: k_lat8_gather<F = 1> at 75
// VGPRs holds 6 workgroups per CU and was launched with 8 -- two rounds for the work of 1.33).
static inline int mfem_resident_per_cu(const void* kernel, int block_threads, size_t dyn_lds, int fallback) {
  static std::mutex mu;
  static std::unordered_map<const void*, int> cache;
  std::lock_guard<std::mutex> lk(mu);
  auto it = cache.find(kernel);
  if (it != cache.end()) return it->second;
  int occ = 0;
  const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, block_threads, dyn_lds);
  const int v = (e == hipSuccess && occ >= 1) ? occ : fallback;
  cache[kernel] = v;
  return v;
}
static inline int mfem_grid_for(int64_t work_items, int per_block, int cap) {
  int64_t g = (work_items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}
