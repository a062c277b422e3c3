/*
 * metafem_mi355x_debug.h -- tuning / diagnostic hooks of libmetafem_mi355x.so.
 *
 * NOT part of the drop-in surface (that is include/metafem_mi355x.h, which a MetaFEM.jl maintainer binds): nothing here
 * replaces a reference interface.  These are the knobs and probes the benchmarks (bench.py), the profiling scripts (tools/)
 * and the parity tests use to select kernel variants and to time the solver's SpMV.
 *
 * Every mfem_debug_set_* knob is PROCESS-WIDE state read at launch time: set it only while no call is in flight on any
 * context of the process (the "one context per host thread" rule of the main header covers the seams, not these knobs).
 * Each call bumps an epoch that is part of the cycle-graph cache key, so cached graphs never outlive a knob change.
 * None of them changes results beyond round-off; kernel variants that compute WRONG results for timing purposes live in
 * tools/, not in this library.
 */
#ifndef METAFEM_MI355X_DEBUG_H
#define METAFEM_MI355X_DEBUG_H

#include "metafem_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

/* launches of the symmetric-sweep SpMV kernels (mfem_csr_solver_layout_entries) so far, process-wide */
int64_t mfem_debug_sym_spmv_count(void);
/* binds of the patch sweep whose symmetry verdict came from the fingerprint the fill pass sums (round 6) instead of the separate check pass (process-wide) */
long long mfem_debug_symp_fingerprint_count(void);
/* THE tuning knobs: one entry point (round 6; until round 5 one function per knob).  Process-wide state read at launch time: set only while no call is
 * in flight on any context; every call bumps the epoch that is part of the cycle-graph cache key.  No knob changes results beyond round-off.  Keys and
 * the meaning of (a, b) are documented where the knobs are described below (lines marked `^ key "..."`); an unknown key returns MFEM_ERR_INVALID. */
int mfem_debug_set(const char* key, int64_t a, int64_t b);

/* CSR kernels behind mul!: tiles per XCD run (0 = dispatcher round-robin) | variant << 16 (0 default, 1 product tile, 3 wave tiles cut by
 * nonzeros -- set before the pattern is created --, 4 workgroup-wide transposing tile, 6 / 7 wave tiles of a fixed row count) | bit 27:
 * without the 2688-entry wave tile (rows of 64..83 entries then share 1792-entry tiles 16 at a time) | bit 26: row-block tiles round-robin
 * over the XCDs instead of a contiguous eighth each | bit 25: no column-offset inspection of the row-block tiles (before the pattern is
 * created), persistent workgroups per CU. */
/* ^ key "spmv": mfem_debug_set("spmv", a, b) with (int xcd_aware, int grid_mult) = (a[, b]) */
/* modes 1/2: bit 0 on/off; bit 1 never use diagonal slots; bits 4-7 / 16-19 kernel variants; bits 8-15 workgroups per CU;
 * bit 20 XCD-contiguous row chunks; bit 22 symmetric sweep kernels off; bit 23 the workgroup-tile sweep (k_spmv_sym27) instead of
 * the wave-private patch sweep (k_spmv_symp);
 * bits 24-25 workgroup size of the diagonal-slotted kernel (0: 256, 1: 512, 2: 1024, 3: 128); bit 26 rows outside the swept planes
 * in a launch of their own; bit 27 the patch-major copy made from the slot-major copy in a second pass. */
/* ^ key "ell": mfem_debug_set("ell", a, b) with (int enable) = (a[, b]) */
/* mode 3: bit 0 on/off; bit 1 always read explicit columns; bit 2 every XCD walks a contiguous eighth of the blocks; bits 4-7 (x 8 = R)
 * rows sorted inside lattice regions of R^3 points when the pattern carries a lattice hint (mfem_brick_pattern) -- both measured slower than
 * the default at hex-27 128^3 (profiles/r03_sell_regions.txt), same results; bits 8-13 sort rows within windows of 2^w rows (0 = whole
 * matrix); bits 16-20 slots in flight per lane (4, 5 = default, 8, 9, 10, 15); bits 24-28 workgroups per CU (default 8). */
/* (round 6) bit 3 of "sell": field-periodic blocks of a field-major multi-field matrix read their whole column stream again (A/B) */
/* ^ key "sell": mfem_debug_set("sell", a, b) with (int enable) = (a[, b]) */
/* mode 4 (symmetric lattice tiles, hex-27): 0 = off (mode 3 serves those solves), 1 = on (default). */
/* (round 6) bit 3 of "lat27": pass 1 by the four-lanes-per-row kernel of rounds 3-5 (not bitwise reproducible) instead of the deterministic lane = row form */
/* ^ key "lat27": mfem_debug_set("lat27", a, b) with (int enable) = (a[, b]) */  /* bit 1: pass 2 (the gather of the tiles' y blocks) by the kernel that walks the covering blocks one memory round trip at a time (same y bit for bit) */
/* bit 2 of mfem_debug_set_lat27: CG iterations on the tiles as SpMV (pass 1 + pass 2) + residual update; by default, on one rank, pass 2 runs INSIDE the residual
 * update (k_lat27_gather_cg: A p is never stored, p . A p comes from pass 1) -- the same iterates to round-off.  mfem_debug_lat27_cg_fused: that switch;
 * mfem_debug_lat27_pass1_bytes: what pass 1 alone moves by design (the SpMV launch bench.py times in that mode). */
int mfem_debug_lat27_cg_fused(void);
int64_t mfem_debug_lat27_pass1_bytes(mfem_csr A);
/* SpMVs mode 4 has served so far, process-wide */
long long mfem_debug_lat27_spmv_count(void);
/* max |layout x - CSR x| / max |A[r][c]| of the probe product of the last mode-4 bind on this pattern (mode 4 is taken up to 4e-13) */
double mfem_debug_lat27_asymmetry(mfem_csr A);
/* mode 5 (symmetric lattice tiles, F-field 27-point matrix): the same three entry points; bit 1 of `enable`: mfem_csr_solver_layout and
 * mfem_spmv_solver_layout report / take mode 5 for ONE field too (they answer for cg!, which keeps mode 2 there) */
/* ^ key "lat8": mfem_debug_set("lat8", a, b) with (int enable) = (a[, b]) */  /* bit 2: pass 2 by the staged gather (all loads of a tile in flight at once, sums in the same order; measured slower on these tiles, off by default) */
long long mfem_debug_lat8_spmv_count(void);
double mfem_debug_lat8_asymmetry(mfem_csr A);
/* Modes 1-3 are used from these row counts on (defaults 262 144 for mode 2 -- and mode 5 --, 1 000 000 for modes 1 and 3; mode 4 from
 * min(explicit_columns, 180 000) rows: profiles/r03_lat_tiles_thresholds.txt): smaller systems
 * are launch-bound and stay on the CSR tile kernel.  The parity tests set both to 0. */
/* ^ key "layout_min_rows": mfem_debug_set("layout_min_rows", a, b) with (int64_t diagonal_slots, int64_t explicit_columns) = (a[, b]) */
/* hipGraph replay of solver cycles inside mfem_solve (default on for n <= 4 000 000 without a communicator): an IDR(s)
 * cycle, a BiCGStab(l) sweep, a CGS2 step or a CG iteration pair is captured once and replayed; results are identical to
 * the plain launch sequence.  bit 0 of `on`: 0 disables; max_n > 0 changes the size limit.  bit 1 (round 6; also MFEM_GRAPH_COMM=1 in the
 * environment): cycles are captured WITH an RCCL communicator attached too (ncclAllReduce on the context stream, the halo exchange's fork / join of
 * the halo stream) -- off by default: RCCL with more than one rank has not executed on this pool; a capture that fails falls back to direct launches. */
/* ^ key "graphs": mfem_debug_set("graphs", a, b) with (int on, int64_t max_n) = (a[, b]) */
/* cycles captured with a communicator attached so far (process-wide) */
int mfem_debug_graph_comm_count(void);
/* idrs!: 1 = the literal bi-orthogonalisation loop of 04_IDRs.jl:62-66 (k dependent dot products and 2 k vector updates per inner step) instead of
 * the merged form (one multi-dot pass, the alphas by forward substitution with M, one vector kernel): the same numbers in exact arithmetic.
 * Round 6 -- bits: 1 = the literal loop (above); 2 = shadow vectors as U(0,1) vectors from mfem_rand, streamed (the default until round 5; now P is the
 * +-1 family of the seed's sign words and P' g reads g only); 4 = update of step k and combination of step k + 1 as two kernels (fused by default). */
/* ^ key "idrs": mfem_debug_set("idrs", a, b) with (int bits) = (a[, b]) */
/* bicgstabl_GS!: 1 = the literal operation sequence of 03_BiCGstabl.jl:41-94 (one pass over the vectors per dot product and per update) instead of the
 * fused form (dot products produced by the SpMVs, the minimal-residual part on the Gram matrix of R[0..l], the updates of a sweep in one kernel). */
/* ^ key "bicgstabl": mfem_debug_set("bicgstabl", a, b) with (int literal_sequence) = (a[, b]) */
/* persistent workgroups per CU of the streaming vector kernels (axpy family, fused CG updates, dots); default 3. */
/* ^ key "vec_grid": mfem_debug_set("vec_grid", a, b) with (int workgroups_per_cu) = (a[, b]) */
/* Multi-rank SpMV: 1 (default) the halo exchange runs on a second stream beside the rows that read no ghost column and the
 * boundary rows follow in a second launch; 0 the exchange completes before a single launch (same results bitwise). */
/* ^ key "halo_overlap": mfem_debug_set("halo_overlap", a, b) with (int on) = (a[, b]) */
/* hex-27 matrix assembly: bits 0-1: 0 / 1 (default) two-pass -- MFMA Ke -> element-major scratch (a ring of element
 * planes) + LDS row-building gather; 2 FP64 atomics in one launch; 3 colour-partitioned read-modify-write scatter straight
 * from the MFMA accumulators (8 launches).  Bits 16-23: element planes per scratch chunk (0 = whole mesh if it fits the
 * 16 GiB scratch budget).  Bit 8: the constant-Jacobian shortcut of affine elements off (every element takes the general path).
 * Bit 9: the scratch-free assembly of meshes whose elements are ALL affine off (such a mesh then takes the two-pass MFMA path;
 * by default its matrix is built by a row-owner gather from per-element G0 and the reference integrals, Ke never stored).
 * Bit 10 (round 5): the PER-ELEMENT choice off -- a mesh with at least one non-affine element then takes the two-pass path whole (round 4's behaviour;
 * by default its affine elements are computed in place and only the others go through pass 1, into a scratch that holds only them).  Bits 24-30:
 * percentage of non-affine elements up to which the per-element choice is taken (0 = the default 80; beyond it the two-pass path is faster).
 * Bit 11 (round 5): the row-owner kernel of GENERAL elements off (k_hex27_rows_gq: rows computed in place from per-element G_q by sum factorisation, no Ke
 * stored anywhere; taken by default from bits 2-7 percent of non-affine elements on -- 0 = the default 30 -- when the mesh has three Gauss points per direction).
 * Bits 12-14: TIMING-ONLY ablations of that kernel (wrong values; tools/hex27_rows_ablate.py). */
/* ^ key "hex27": mfem_debug_set("hex27", a, b) with (int two_pass) = (a[, b]) */
/* number of mfem_mesh_assemble_elements_rows calls that ran the row-owner form (process-wide) */
int64_t mfem_debug_mesh_rows_count(void);
/* 128-row blocks of the sliced layout (mode 3) that are field-periodic -- col[f P + t] = col[t] + f shift: one column slot per node is read for the F
 * column fields -- (-1: null handle; 0: none / not planned) */
int64_t mfem_debug_sell_periodic_blocks(mfem_csr A);
/* number of hex-27 matrix assemblies that took the row-owner kernel of general elements (process-wide) */
int64_t mfem_debug_hex27_rows_count(void);
/* number of hex-27 matrix assemblies that took the per-element choice with at least one stored (non-affine) element (process-wide) */
int64_t mfem_debug_hex27_mixed_count(void);
/* number of hex-27 matrix assemblies that took the scratch-free path so far (process-wide) */
int64_t mfem_debug_hex27_direct_count(void);
/* hex-8 elasticity kernels.  Bit 0: matrix -- 0 (default) thread per (control point, element) with the rows accumulated in LDS and
 * written once; 1 the earlier row-owner kernel accumulating in global memory (equal to round-off: other summation order, table form of
 * the Jacobian).  Bit 1: residual -- 0 (default) the plane-sweep kernel with one sum-factorised integration per element (2-point Gauss
 * rule); 1 the kernel that integrates an element once per adjacent control point (table form; equal to round-off).  Bits 2-4: TIMING-ONLY
 * ablations of the default matrix kernel (wrong values): no accumulation steps / no write-out / no integration (tools/el_time.py). */
/* ^ key "elasticity": mfem_debug_set("elasticity", a, b) with (int variant) = (a[, b]) */
/* hex-8 thermal matrix / residual kernels: 0 (default) the plane-sweep kernels with sum-factorised element integration
 * (2- and 3-point Gauss rules; other rules always use the tile kernels); 1 the 4 x 4 x 8 tile kernels with the table form. */
/* ^ key "hex8_thermal": mfem_debug_set("hex8_thermal", a, b) with (int variant) = (a[, b]) */
/* Per-launch timing of the solver's SpMV kernel with hip events on the context stream (bench.py's roofline).
 * read: total device ms and launch count since the last reset. */
int mfem_prof_spmv_enable(mfem_context ctx, int on);
int mfem_prof_spmv_read(mfem_context ctx, double* total_ms /* [host] */, int64_t* launches /* [host] */, int reset);
/* Communication the solver's stream is EXPOSED to, per rank (communicator attached to ctx; zeros without one): halo_wait = time the context stream
 * spent waiting for the halo exchange after the interior rows were done (RCCL: hip-event pair around the wait for the halo stream -- ~0 when the
 * exchange finished beside the interior rows; host callbacks: the whole staged exchange), allreduce = the all-reduces of the reduction groups
 * (event pair around ncclAllReduce on the context stream: includes waiting for the slowest rank).  Off by default; the events cost a few
 * microseconds per operation, and captured cycle graphs are not used with a communicator anyway. */
int mfem_prof_comm_enable(mfem_context ctx, int on);
int mfem_prof_comm_read(mfem_context ctx, double* halo_wait_ms /* [host] */, int64_t* halo_waits /* [host] */, double* allreduce_ms /* [host] */,
                        int64_t* allreduces /* [host] */, int reset);
/* RCCL transport check on the communicator attached to ctx (mfem_comm_create, not the host-callback one): `rounds` times the call
 * sequence one overlapped SpMV + reduction group issues -- grouped ncclSend / ncclRecv of `count` doubles on the halo stream fenced
 * by events, a kernel on the context stream beside it, the stream wait, ncclAllReduce of 3 scalars on the context stream with the
 * same communicator -- on a RING (to rank + 1, from rank - 1, modulo world), so that a one-rank communicator runs every call as
 * well (self send / receive).  MFEM_OK when every received entry and the reduced scalars are right. */
int mfem_debug_comm_selftest(mfem_context ctx, int64_t count, int32_t rounds);

/* Placement experiment: the solver workspace's base becomes align_up(hipMalloc's pointer, align) + offset (0, 0: off).  tools/placement_probe.py;
 * mfem_debug_ws_address returns the base in use. */
/* ^ key "ws_placement": mfem_debug_set("ws_placement", a, b) with (long long align, long long offset) = (a[, b]) */
unsigned long long mfem_debug_ws_address(mfem_context ctx);
/* Workspace placement trial, OFF by default (round 4: a drop-in mfem_solve must not hide seconds of allocation work).  1: the first solve on a
 * workspace of 8 GB or more times the solver SpMV, tries up to two more allocations of the workspace the same way (at most two alive: PEAK MEMORY
 * = TWICE THE WORKSPACE, 90 GB at 512^3) and keeps the fastest -- the speed follows the physical memory an allocation received (7 % of every later
 * iteration at 512^3, profiles/r04_placement_counters.txt).  Paid once per workspace: 5 - 7 s at 512^3 (a hipMalloc of 45 GB takes 2 s, a hipFree
 * about as long; MFEM_WS_TRIAL_VERBOSE=1 prints the steps' times to stderr).  Rank-local: with a communicator attached every rank runs its own
 * trial (the candidates' timing uses no collective).  bench.py opts in with --ws-trial 1 and says so in its line. */
/* ^ key "ws_trial": mfem_debug_set("ws_trial", a, b) with (int on) = (a[, b]) */
/* 1 (default): the two vector kernels of the classic CG recurrences use streaming (nontemporal) loads, and from 4e7 rows on streaming stores too
 * (csrc/krylov.hip: cg_ld / cg_st); 0: plain accesses.  Same values either way. */
/* ^ key "cg_streaming": mfem_debug_set("cg_streaming", a, b) with (int on) = (a[, b]) */
/* cg_variant 0 (auto) with a communicator of more than one rank: the single-reduction CG (one all-reduce, 9 vector streams per iteration) below this many
 * rows per rank (n_global / world; default 2e7), the classic recurrence (two all-reduces, 8 streams) from there on -- at 512^3 per rank a vector stream
 * costs 0.2 ms, an all-reduce ~0.03 ms.  Set 0 for always-classic, a huge value for always-single. */
/* ^ key "cg_single_max_rows": mfem_debug_set("cg_single_max_rows", a, b) with (int64_t rows) = (a[, b]) */

/* out4[0..2]: the times (ms for two SpMVs) of the workspace candidates tried by that choice, in order; 0 = not tried. */
int mfem_debug_ws_trial_log(mfem_context ctx, double* out4);

/* Fault injection for the error convention (metafem_mi355x.h: nothing is thrown across the boundary): the nth host allocation the library
 * probes from now on (handle structs, planning vectors) throws std::bad_alloc once; the entry point that hits it returns MFEM_ERR_ALLOC with
 * mfem_last_error() set and leaves every handle valid.  0 disarms.  tests/test_gpu_round4_abi.py. */
int mfem_debug_fail_host_alloc(int nth);

/* CSR kernel behind mul! on rows of near-uniform length (k_spmv_csr_w), a round-5 experiment, OFF by default: on = 1 makes the workgroups of an XCD walk an eighth of
 * every lattice plane through all planes (one-field lattice patterns whose two planes of x exceed `min_bytes`; < 0 keeps the threshold, default 3 MiB) instead of
 * sharing one front with the other XCDs, so that an L2 holds the x window it re-reads.  Measured: 8.34 against 8.23 ms at 512^3, 1.113 against 1.092 at 256^3
 * (profiles/r05_csr_strips.txt) -- the kernel's time does not follow the re-read x (Infinity-Cache hits).  Same tiles, same sums: bitwise the same y. */
/* ^ key "csr_strips": mfem_debug_set("csr_strips", a, b) with (int on, int64_t min_bytes) = (a[, b]) */

/* A = S + N (round 5, csrc/spmv_rem.hip): a symmetric lattice-tile bind (modes 4 / 5) whose values fail the symmetry gate in at most n / 8 rows keeps
 * the tiles and carries the mirrored entries' differences N[r][c] = A[r][c] - A[c][r] of those rows as a small CSR applied after the tiles' gather pass
 * (Nitsche / SUPG faces: the reference's nonsymmetric K).  Bit 0 (default 1): on; 0 -- such values send the solve to the layouts that read every entry, as
 * before.  Bit 1 (default 0): the diagnostic product mfem_spmv_solver_layout, which answers for cg!, takes a remainder too (tests).
 * mfem_debug_remainder_info: rows / entries of the remainder the CURRENT or last bind on this pattern carries (0 / 0: none), and the asymmetry the
 * probe measured on the tiles alone; mfem_debug_rem_spmv_count: products that applied one (process-wide). */
/* ^ key "remainder": mfem_debug_set("remainder", a, b) with (int enable) = (a[, b]) */
/* TEST HOOK: the residual a single-rank tile solve recomputes from the caller's CSR values before it ends the passes is multiplied by `scale`
 * (default 1; <= 0 resets): lets a test put the tiles' residual and the caller's on the two sides of the tolerance */
/* ^ key "recheck_scale_ppm": mfem_debug_set("recheck_scale_ppm", a, b) with (a = scale in millionths) = (a[, b]) */
/* Node-blocked sliced layout (round 6, csrc/spmv_sell.hip "BSELL"): mode 3 on a field-major F-field matrix (F = 2, 3, 4) whose F rows of a node share
 * the node's coupling list -- a lane owns a node: one column index and F gathers of x per F x F values.  Taken for patterns without a lattice hint and
 * without ghost columns.  Bit 0: on (default 1; 0 = the row-sorted form, as before; read when the pattern's layout is planned).  Bit 1 (default 0): the
 * per-solve copy by lane quads per row instead of the LDS transpose (A/B).
 * mfem_debug_bsell_fields: F of the pattern's planned layout (0: row-sorted form, -1: null handle); mfem_debug_bsell_spmv_count: products so far. */
/* ^ key "bsell": mfem_debug_set("bsell", a, b) with (int on) = (a[, b]) */
int mfem_debug_bsell_fields(mfem_csr A);
long long mfem_debug_bsell_spmv_count(void);
/* The gather of mfem_mesh_assemble_elements_rows runs by NODE when a node's blocks x element nodes fit a wave (round 6: the rows of a node's fields share one
 * adjacency walk, several nodes per wave); 1 = by row, as in round 5 (A/B and the bitwise comparison in tests/test_gpu_unstructured.py). */
/* ^ key "mesh_gather_rows": mfem_debug_set("mesh_gather_rows", a, b) with (int by_row) = (a[, b]) */
/* mfem_op_var_batch / mfem_op_res_batch on elements of 10+ nodes (b > 0: of b+ nodes) run a persistent wave per item with the table slabs in LDS (round 6); 0 = the sub-wave
 * forms of rounds 2-5 on every element (A/B, tests); 2 = the wave forms for any item count (default: from 256 items). */
/* ^ key "op_wave_forms": mfem_debug_set("op_wave_forms", a, b) with (int on) = (a[, b]) */
/* Elements with at least this many nodes take the staged persistent form of the row-owner element kernel (k_mesh_assemble<.., STAGE>; default 16: hex-20,
 * hex-27; 0 restores the default). */
/* ^ key "mesh_stage_min_itp": mfem_debug_set("mesh_stage_min_itp", a, b) with (int nodes) = (a[, b]) */
/* The non-staged element kernel of the fused mesh assembly applies the terms as one dense coefficient row per sparse block (default 1); 0 = it walks the
 * term list, as in rounds 2-5 (A/B). */
/* ^ key "mesh_term_matrix": mfem_debug_set("mesh_term_matrix", a, b) with (int on) = (a[, b]) */
/* TIMING ONLY (wrong results): phases of k_mesh_assemble left out -- 1 the pair products, 2 the stores of the row-owner form, 4 the geometry, 8 the
 * physical table, 16 the coordinate gather (tools/u20_assembly_ab.py). */
/* ^ key "mesh_abl": mfem_debug_set("mesh_abl", a, b) with (int bits) = (a[, b]) */
int mfem_debug_remainder_info(mfem_csr A, int64_t* rows /* [host] */, int64_t* entries /* [host] */, double* asym_before /* [host] */);
long long mfem_debug_rem_spmv_count(void);

#ifdef __cplusplus
}
#endif
#endif /* METAFEM_MI355X_DEBUG_H */
