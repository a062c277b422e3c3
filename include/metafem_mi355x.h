/*
 * metafem_mi355x.h -- C ABI of libmetafem_mi355x.so, the MI355X (gfx950) assembly-and-solve
 * backend for MetaFEM.jl's hot path.
 *
 * Every entry point replaces one seam of the reference (jxx2/MetaFEM.jl v0.1.4; citations are
 * file:line under the reference's src/).  The reference has no FFI: its seams are Julia
 * Function-typed fields and fixed-name callees of generated code, so the ABI
 * below is what a `ccall` shim binds -- see INTEGRATION.md for the Julia side.
 *
 * Conventions
 *   - all array arguments are DEVICE pointers owned by the caller unless marked [host];
 *   - Float = double (FEM_Float, misc/02_Global_Macros.jl:124); column indices are int32
 *     (FEM_Int, :123); row pointers are int32 or int64 (512^3 hex-8 has nnz > 2^31, F8);
 *   - `index_base` is 1 for arrays coming straight from Julia (CUSPARSE 'O' convention,
 *     misc/04_GPU_Utils.jl:131) and 0 otherwise;
 *   - every function returns 0 on success, a negative mfem_status otherwise and never
 *     throws/aborts across the boundary; mfem_last_error() gives the message (thread local);
 *   - work is enqueued on the context's stream; functions that return a scalar to the host
 *     synchronise that stream, nothing else does;
 *   - a context is bound to one device and used by one host thread at a time.
 */
#ifndef METAFEM_MI355X_H
#define METAFEM_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MFEM_ABI_VERSION 5

typedef enum {
  MFEM_OK = 0,
  MFEM_ERR_INVALID = -1,   /* bad argument (null pointer, negative size, unknown enum) */
  MFEM_ERR_HIP = -2,       /* a HIP runtime call failed */
  MFEM_ERR_UNSUPPORTED = -3,
  MFEM_ERR_NOT_CONVERGED = -4, /* never returned by mfem_solve (reference semantics: it reports, not fails) */
  MFEM_ERR_COMM = -5,
  MFEM_ERR_ALLOC = -6,     /* host memory exhausted inside the library (std::bad_alloc caught at the boundary) */
  MFEM_ERR_INTERNAL = -7   /* any other C++ exception caught at the boundary (mfem_last_error() carries its what()) */
} mfem_status;
/* Error convention of every entry point that returns int: MFEM_OK or a negative mfem_status, with mfem_last_error() (thread-local text) set.
 * NOTHING is thrown or aborted across this boundary: each entry point is a function-try-block whose handler maps std::bad_alloc to
 * MFEM_ERR_ALLOC and anything else to MFEM_ERR_INTERNAL (csrc/common.h: MFEM_API_CATCH), so a host that calls through ccall / ctypes / cgo
 * never sees a C++ exception unwind into its frames.  After a non-zero return the handles passed in stay valid and destroyable. */

typedef struct mfem_context_s* mfem_context;
typedef struct mfem_csr_s* mfem_csr;       /* CSR pattern + SpMV plan (no values) */
typedef struct mfem_brick_s* mfem_brick;   /* structured hex mesh (make_Brick ordering) */
typedef struct mfem_mesh_s* mfem_mesh;     /* general (unstructured) mesh: coords + controlpoint_IDs */
typedef struct mfem_comm_s* mfem_comm;     /* RCCL communicator + slab neighbours */

/* ---- context -------------------------------------------------------------------------- */
int mfem_abi_version(void);
const char* mfem_last_error(void);
/* device: HIP ordinal.  stream: hipStream_t to enqueue on (NULL = the null stream). */
int mfem_context_create(int device, void* stream, mfem_context* out);
int mfem_context_set_stream(mfem_context ctx, void* stream);
int mfem_context_destroy(mfem_context ctx);
/* Lifetimes: destroy patterns, bricks and communicators before the context they were created on.  A pattern that is destroyed
 * after its context (finalisers of a garbage-collected host run in any order) only releases its own memory. */
int mfem_context_sync(mfem_context ctx);

/* ---- S1 primitives: the CUSPARSE/CUBLAS call sites of the Krylov loop -------------------- */
/* CSR pattern (replaces FEM_SpMat_CSR(K_J_ptr, K_J, ...), misc/04_GPU_Utils.jl:120).  The
 * arrays are borrowed, not copied: they must outlive the handle AND keep their contents -- the handle caches what it learnt from them
 * when it was created (longest row, row blocks, which tiles of rows repeat one column-offset list; later the solver layouts), like the
 * reference's K_J_ptr / K_J, which are written once by assemble_SparseID!.  A changed pattern needs a new handle.  rowptr_bits = 32 | 64. */
int mfem_csr_create(mfem_context ctx, int64_t n, int64_t nnz, const void* rowptr, int rowptr_bits,
                    const int32_t* colidx, int index_base, mfem_csr* out);
int mfem_csr_destroy(mfem_csr A);
/* Re-run the inspection after the borrowed rowptr / colidx were rewritten IN PLACE (same n, nnz, addresses): drops everything the
 * handle cached (row blocks, column-offset flags, solver layouts, cached solver graphs) and plans again.  The stream must have
 * finished producing the arrays (the call synchronises the context stream).  assemble_SparseID! writes K_J_ptr / K_J once per mesh
 * (03_GlobalAssembly.jl:77-140), so a reference-style host never needs it; it exists so that a stale cache cannot be the only option. */
int mfem_csr_replan(mfem_context ctx, mfem_csr A);
/* y = alpha*A*x + beta*y : mul!(b, A, x, alpha, beta), misc/04_GPU_Utils.jl:131 (CUSPARSE mv! 'N'). */
int mfem_spmv_csr(mfem_context ctx, mfem_csr A, const double* vals, const double* x, double* y,
                  double alpha, double beta);
/* y = a*x + b*y  (the broadcast axpy family of every solver body, e.g. 03_BiCGstabl.jl:50,57). */
int mfem_axpby(mfem_context ctx, int64_t n, double a, const double* x, double b, double* y);
/* *out [host] = sum x_i*y_i  (LinearAlgebra.dot on CuArray -> CUBLAS dot; 03_BiCGstabl.jl:45). */
int mfem_dot(mfem_context ctx, int64_t n, const double* x, const double* y, double* out);
/* *out [host] = ||x||_2      (norm -> CUBLAS nrm2; solver/04_Time_Domain.jl:51). */
int mfem_nrm2(mfem_context ctx, int64_t n, const double* x, double* out);
/* x_i = U[0,1) from the counter-based generator (seed, stream_id); replaces FEM_rand ->
 * CUDA.Random.rand! (misc/04_GPU_Utils.jl:22), which is unseeded in the reference (F9). */
int mfem_rand(mfem_context ctx, int64_t n, uint64_t seed, uint32_t stream_id, double* x);

/* Jacobi kernels of Pr_Jacobi!/Pl_Jacobi (linear_solver/02_Preconditioner.jl). */
int mfem_jacobi_by_diagonal(mfem_context ctx, mfem_csr A, const double* vals, double* d);   /* :122-130, d preset to 1 */
int mfem_jacobi2_by_column(mfem_context ctx, mfem_csr A, const double* vals, double* d);    /* :132-139 (+ sqrt :112) */
int mfem_jacobi_by_row(mfem_context ctx, mfem_csr A, const double* vals, double* d);        /* :170-177 */
int mfem_mat_div_jacobi(mfem_context ctx, mfem_csr A, double* vals, const double* d);       /* :141-148, in place */

/* ---- solver layouts (inspector-executor; internal to mfem_solve, the caller's contract stays CSR) ------------------
 * Layout the Krylov loop of mfem_solve uses for this pattern (inspector result, computed on first use):
 *   mode 0  the CSR tile kernel (small systems -- see mfem_debug_set("layout_min_rows", ...) -- and matrices with < 128 rows)
 *   mode 1  slot-major padded copy of the working values + explicit columns (rows of near-uniform length)
 *   mode 2  as 1, and 128-row blocks whose entries all sit on a common list of <= 96 diagonals store their values by
 *           diagonal and do not read columns at all (any lattice stencil; up to 4 lists, e.g. one per row field of a
 *           field-major multi-field matrix; detected from the CSR pattern, nothing is assumed)
 *   mode 3  rows of uneven length (hex-27, unstructured meshes): rows stably sorted by decreasing length and diagonal-list
 *           signature, sliced ELL with a slot count per 128-row block (SELL-128); blocks whose rows share one diagonal
 *           list store it once instead of columns; y is written through the row permutation
 *   mode 4  the hex-27 lattice matrix (order-2 Lagrange brick, one field, one rank; the pattern is checked entry by entry) on a solver that
 *           keeps the matrix unscaled (cg!, or no Jacobi scaling): symmetric lattice tiles -- only the diagonal and the entries with
 *           column > row are stored (14..63 of a row's 27..125), 16 rows of each of the 8 node types per wave, x and y of a tile of
 *           8 x 8 x 32 lattice points staged in LDS, the mirrored products added there; a second pass sums the tiles' y blocks in a
 *           fixed order (for cg! on one rank that pass runs inside the residual update of the iteration and p . A p comes from the first pass: A p
 *           is never stored; same iterates to round-off).  Taken per solve only if the values are symmetric: the bind compares one probe product of the layout with the CSR
 *           kernel's (x in [0.75, 1.25): an entry pair that differs by delta shows up as >= 0.75 |delta|) and requires
 *           max |difference| <= 4e-13 max |A[r][c]| (mode 3 serves the solve otherwise).  y agrees with the CSR kernel to round-off (other summation order), not bitwise, and not
 *           bitwise from run to run.  A right Jacobi scaling (A D^-1) is applied to x while it is staged; the stored matrix stays A.
 *   mode 5  the same construction for the F-field 27-point lattice matrix, F = 1..3 (hex-8; field-major rows; F = 3: elasticity): lane = node, per
 *           node the upper entries of its own F x F block and the blocks towards its 13 upper neighbours (F = 3: 123 of 243 values; F = 1: 14
 *           of 27), tiles of 8 x 8 x 16 nodes; mode 2 serves the solve if the values are not symmetric.  One field: only for the solvers that
 *           work on A D^-1 (idrs!, bicgstabl_GS!, cgs2! with Pr_Jacobi!: the scaled copy mode 2 would make is not symmetric, so its mirrored
 *           sweep cannot run); cg! keeps mode 2's bitwise patch sweep, and this query answers for cg!.
 * The copy is made once per solve from the caller's CSR-ordered values, like the reference's K_total[K_val_ids] gather
 * (02_Preconditioner.jl:35).  slots = padded row length, regular_rows = rows in diagonal-slotted blocks (mode 2). */
int mfem_csr_solver_layout(mfem_context ctx, mfem_csr A, int32_t* mode, int32_t* slots, int64_t* padded_rows,
                           int64_t* regular_rows);
/* Symmetric sweep (mode 2, 27-point lattice stencil, one field): when the values of a solve are bitwise symmetric where it
 * matters (checked once per solve), a row's entries on the 13 lower diagonals are taken from the mirror entry of the neighbouring
 * row, kept in LDS, instead of from memory: same products, same summation order, bitwise the same y as the plain kernel.  Two
 * forms: 1 = a workgroup sweeps an in-plane tile of 512 rows through consecutive lattice planes (about 2/3 of the matrix traffic
 * while a lattice line fits the tile twice); 2 = a wave sweeps a (j, k) patch of 4 lattice lines x 32 points on a patch-major copy
 * (14 of a row's 27 entries + the patch-rim entries, whatever the line length; x staged per plane in LDS) -- used from 2.4e7 rows /
 * 257-point lines on.  entries = 8-byte matrix values one SpMV of the planned layout reads from memory, bytes (may be NULL) = all
 * bytes the planned kernel reads and writes per SpMV by design: those entries, 4-byte columns of rows in generic blocks, x as
 * often as the kernel stages / gathers it from memory by design (form 2: the patch neighbourhoods overlap, 1.6 n entries), y.
 * symmetric_sweep = 0 / 1 / 2 (assuming the values pass the check); 3 = modes 4 and 5. */
int mfem_csr_solver_layout_entries(mfem_context ctx, mfem_csr A, int64_t* entries, int32_t* symmetric_sweep);
int mfem_csr_solver_layout_bytes(mfem_context ctx, mfem_csr A, int64_t* bytes);
/* The same accounting for the CSR kernel behind mul! (mfem_spmv_csr: the caller's arrays, no copy): bytes one launch moves BY DESIGN =
 * nnz * 8 (values) + 4 * column_entries_read + n * 16 (x once, y once) + the row pointers + the tile table.  column_entries_read
 * (may be NULL) < nnz where the inspection found tiles whose rows repeat the column offsets of their first row(s): the kernel reads
 * only those rows' columns there.  SURVEY 8(d)'s CSR formula (nnz * 12 + n * 16 + row pointers) is what a kernel without that
 * inspection moves; bench.py reports both. */
int mfem_csr_spmv_bytes(mfem_context ctx, mfem_csr A, int64_t* bytes, int64_t* column_entries_read);
/* y = alpha A x + beta y through that layout, conversion of `vals` included (diagnostic: what the Krylov loop computes). */
int mfem_spmv_solver_layout(mfem_context ctx, mfem_csr A, const double* vals, const double* x, double* y, double alpha,
                            double beta);

/* ---- S1: the linear solver seam  fem_domain.linear_solver(globalfield) -------------------- */
typedef enum {
  MFEM_SOLVER_CG = 0,          /* added (not in the reference, F5); symmetric definite K only */
  MFEM_SOLVER_BICGSTABL_GS = 1,/* bicgstabl_GS!  linear_solver/03_BiCGstabl.jl:18-96 */
  MFEM_SOLVER_IDRS = 2,        /* idrs!          linear_solver/04_IDRs.jl:26-95 */
  MFEM_SOLVER_CGS2 = 3         /* cgs2!          linear_solver/07_CGS.jl:54-105 */
} mfem_solver_kind;

typedef enum {
  MFEM_PRECOND_NONE = 0,            /* Identity, 02_Preconditioner.jl:78-86 */
  MFEM_PRECOND_JACOBI_RIGHT_DIAG = 1,   /* Pr_Jacobi!(normalized_by_column=false), :103-120 */
  MFEM_PRECOND_JACOBI_RIGHT_COLNORM = 2 /* Pr_Jacobi!(normalized_by_column=true) */
} mfem_precond_kind;

typedef enum {
  MFEM_LEFT_NONE = 0,         /* Pl_func = Identity (iterative_Solve! default) */
  MFEM_LEFT_JACOBI_DIAG = 1,  /* Pl_Jacobi(A), 02_Preconditioner.jl:155-168: b ./= |diag| of the (already Pr-scaled) matrix */
  MFEM_LEFT_JACOBI_ROWNORM = 2/* Pl_Jacobi(A; normalized_by_row = true), :160-162,170-177 */
} mfem_left_precond_kind;

typedef struct {
  int32_t method;        /* mfem_solver_kind */
  int32_t precond;       /* mfem_precond_kind.  For CG, JACOBI_* means M = |diag K| (standard PCG). */
  int32_t l_or_s;        /* `s` kwarg: BiCGStab l (default 2) / IDR s (default 4) */
  int32_t maxiter;       /* per pass */
  int32_t max_pass;      /* restart passes, iterative_Solve! default 4 */
  int32_t check_every;   /* host polls the device convergence flag every this many iterations (>=1) */
  double converge_tol;   /* ABSOLUTE on ||r||_2/sqrt(n): globalfield.converge_tol (F10) */
  uint64_t seed;         /* shadow-vector seed (reference is unseeded, F9).  bicgstabl_GS! / cgs2!: mfem_rand(seed, stream) vectors; idrs! (round 6): the +-1 vectors of the seed's sign words (csrc/rng.h; oracle: fem_sign), never stored */
  int32_t fixed_iterations; /* !=0: ignore converge_tol, run exactly maxiter iterations in ONE pass (benchmark mode) */
  int32_t scale_in_place;   /* !=0: Pr_Jacobi! semantics -- `vals` is overwritten by the column-scaled matrix
                               (the reference scales its private gather K_total[K_val_ids], :35,118).
                               ==0: `vals` is left untouched.  On the solver layouts the scaling happens while the layout copy is made
                               (no scaled CSR copy exists); small systems on the CSR kernel and solves with a left preconditioner
                               scale a private copy (nnz*8 B of workspace). */
  int32_t left_precond;     /* mfem_left_precond_kind (Pl_func).  The reference applies Pl to every mat-vec result inside
                               the Krylov body; here the rows of the working matrix and b are scaled once, which is the
                               same operator.  The restart wrapper follows :57-60 (true residual un-scaled, tol_factor).
                               Not valid with MFEM_SOLVER_CG (it would break symmetry). */
  int32_t cg_variant;       /* MFEM_SOLVER_CG only.  1: classic PCG recurrence (two dependent reduction groups per iteration:
                               p.Ap, then r.z and r.r).  2: single-reduction form (Chronopoulos-Gear: s = A p carried by
                               recurrence, so r.u, w.u and r.r are reduced together right after the SpMV -- one all-reduce per
                               iteration on several GPUs; same iterates in exact arithmetic, round-off-level differences).
                               3: the classic recurrence carrying z = M^-1 r instead of r (z -= alpha M^-1 A p; r = M z only inside the
                               kernel that needs r.z and r.r): the p update then reads neither r nor M^-1 -- 9 vector streams per
                               iteration instead of 10; same iterates in exact arithmetic, round-off-level differences.
                               4: plain CG on S^-1 A S^-1, S = sqrt|diag A| -- the Jacobi-preconditioned iteration in the variables
                               S x, 8 vector streams per iteration (no M^-1 stream).  The scaling is folded into the solver layout's
                               copy as a * (t_r t_c), t = 1 / S, with the product t_r t_c formed first (a bitwise symmetric matrix stays so; an asymmetry
                               of one ulp can vanish in that rounding -- the iteration then runs on a symmetric matrix); the kernels
                               stop on the TRUE residual norm (they read S once the bound smax |S^-1 r| comes near the tolerance).
                               Taken on the diagonal-slotted solver layout (mode 2, mirrored or plain kernels), Jacobi by the diagonal,
                               a positive finite diagonal on every rank (ghost columns are divided by their owners' S); otherwise 3
                               runs.  With more than one rank and cg_variant 0 the single-reduction form runs on the scaled system
                               (9 vector streams instead of 10: its u IS r).
                               0 = auto: the single-reduction form when a communicator with more than one rank is attached AND the system has
                               fewer than 2e7 rows per rank (above that the vector stream it adds costs more than the all-reduce it saves), the
                               two-reduction form otherwise -- each on the scaled system where that applies. */
} mfem_solve_options;

typedef struct {
  int32_t passes;
  int32_t iterations;      /* sum over passes of the solver's returned iteration count */
  double final_res;        /* true residual ||b - A x||/sqrt(n) of the last pass */
  double initial_res;
  double solve_ms;         /* device time, hip events on the context stream */
  int32_t converged;
  int32_t spmv_count;      /* matrix-vector products that ran (the first pass starts from r = b: x0 = 0, no product) */
} mfem_solve_stats;

/* What a solve does to the matrix it is handed -- read before relying on bit patterns:
 *  - mfem_solve copies `vals` once per call into a SOLVER LAYOUT chosen from the pattern and its size (mfem_csr_solver_layout reports the mode);
 *    the Krylov loop's SpMVs read that copy, `vals` itself is never modified (scale_in_place apart).
 *  - Modes 0-3 (CSR kernel, slot-major / diagonal-slotted copy incl. its mirrored sweep, sliced layout) are BITWISE REPRODUCIBLE run to run: a
 *    fixed summation order, no floating-point atomics.  The mirrored sweep of mode 2 is taken only after a per-solve BITWISE symmetry check of
 *    the values and returns bit for bit what the plain kernel returns.
 *  - Mode 5 (symmetric lattice tiles of the hex-8 lattice matrices, 1-3 fields; the default for idrs! / bicgstabl_GS! / cgs2! from 2.6e5 rows on, one
 *    rank or slabs) is BITWISE REPRODUCIBLE since round 6: pass 1 runs its steps phase-major with workgroup barriers between the phases, so every LDS
 *    cell receives its mirrored products from one wave per phase in program order (tests/test_gpu_lat8.py::test_tiles_are_bitwise_reproducible: 20
 *    products and repeated idrs!(8) / bicgstabl_GS!(2) solves identical bit for bit).  Its y equals the CSR kernel's to round-off (<= 1e-13 relative).
 *  - Mode 4 (the tiles of the hex-27 one-field lattice matrix, from 1.8e5 rows on) is BITWISE REPRODUCIBLE since round 6 as well: pass 1 runs lane = row
 *    (a wave owns the rows of four node types in a cube of 8^3 lattice points, the two waves of a cube split the types by the parity of their (j, k)
 *    column), steps phase-major with barriers (tests/test_gpu_lat27.py::test_tiles_are_bitwise_reproducible).  The four-lanes-per-row kernel of rounds
 *    3-5 (ds_add_f64 across waves: ~1e-16 relative, not bitwise) is still selectable: mfem_debug_set("lat27", 1 | 8, 0).
 *  - SYMMETRY GATE of modes 4 / 5: they store one triangle, so every bind measures whether THESE values are symmetric -- one probe product
 *    (entries of magnitude in [0.75, 1.25), random signs) through the layout against the CSR kernel on the caller's values; the layout is taken
 *    when max over rows r of |difference|_r <= 4e-13 |a_rr| (rows without a stored non-zero diagonal: 4e-13 max|a|).  Consequence: an
 *    asymmetry BELOW that level (about 2 000 ulp of the row's diagonal) is not detected and the solve then runs on the symmetrised matrix
 *    (upper triangle mirrored); anything above it sends the solve to modes 3 / 2 on the caller's exact values.  The residual a solve on
 *    ONE rank reports (stats.final_res, and with it `converged`) is recomputed at the end with the CSR kernel on the caller's own values (one extra
 *    SpMV per solve; not in fixed_iterations mode) -- it cannot hide a substitution.  mfem_debug_lat27_asymmetry / mfem_debug_lat8_asymmetry return the last measure.
 * Threading: one context per host thread; handles are not shared between threads while a call is in flight.  Quadrature / lattice tables live
 * in __constant__ memory per PROCESS (uploads are serialised): assemblies with different Gauss orders must not run concurrently from
 * different host threads.  The mfem_debug_* knobs are process-wide atomics, to be changed only while no call is in flight. */

/* Speed of a large solve depends on the ALLOCATION its workspace received, by up to ~10 %: at 512^3 (45 GB of workspace) the Krylov loop's SpMV runs at
 * 3.9 - 4.05 ms or at 4.3 - 4.5 ms depending on the physical memory behind the workspace (address-translation reach: UTCL2 misses 1.1 - 1.3e5 against 2.0e5
 * per launch with identical requests and bytes, profiles/r04_placement_counters.txt); which kind a hipMalloc returns is outside the library's control (the
 * virtual-memory API with 2 MiB / 1 GiB chunks showed the same spread).  The library does NOT hunt for the fast kind by default -- that costs seconds of
 * allocation work and twice the workspace (mfem_debug_set("ws_trial", 1, 0), opt-in; bench.py opts in and says so) -- so a caller sees either speed, about evenly.
 * Results are identical.
 * A nonsymmetric K (round 5): the symmetric lattice tiles of modes 4 / 5 keep serving idrs! / bicgstabl_GS! / cgs2! solves whose values fail the symmetry gate in
 * at most n / 8 rows -- the reference's Nitsche / SUPG boundary terms -- by carrying the mirrored entries' differences of those rows as a small CSR applied
 * behind the tiles (A = S + N, csrc/spmv_rem.hip); the acceptance test is the same probe, applied to S + N.  cg! is never offered that.
 * Between passes: when the tiles' copy says a pass has converged, the residual that ENDS the passes is recomputed with the CSR kernel on the
 * caller's own values (one rank; with a communicator the tiles' residual decides, and stats.final_res is the tiles').
 *
 * delta_x = iterative_Solve!(globalfield; Sv_func!, Pr_func!, max_pass, maxiter, s)
 * (linear_solver/02_Preconditioner.jl:32-76).  b = residue, x_out = the returned vector
 * (caller-allocated, length n; x0 = 0 as in :45).  K_val_ids is the identity in this backend. */
int mfem_solve(mfem_context ctx, mfem_csr A, double* vals, const double* b, double* x_out,
               const mfem_solve_options* opts, mfem_solve_stats* stats);
/* Optional: caller-supplied shadow vectors for the next mfem_solve on this context
 * (count = 1 for BiCGStab/CGS2, s for IDR; each length n, contiguous).  NULL restores the RNG. */
int mfem_solve_set_shadow(mfem_context ctx, const double* shadow, int32_t count);

/* ---- structured brick mesh: make_Brick + mesh_Classical(Lagrange) + update_Mesh -------- */
/* Lattice of (p*nx+1)(p*ny+1)(p*nz+1) control points, id = (i*(p*ny+1) + j)*(p*nz+1) + k
 * (make_Brick vertex order, mesh/ref_geometry/201_Helper_TM.jl:36-41, applied to the order-p
 * lattice); elements i-outer/k-inner (:43-51).  Coordinates are generated on device and may be
 * overwritten through mfem_brick_coords (isoparametric geometry is evaluated on the fly from
 * them -- nothing of update_BasicElements_3D's per-element tables is stored, F7). */
int mfem_brick_create(mfem_context ctx, int32_t nx, int32_t ny, int32_t nz, double lx, double ly, double lz,
                      int32_t itp_order, int32_t itg_order, mfem_brick* out);
int mfem_brick_destroy(mfem_brick m);
int64_t mfem_brick_num_controlpoints(mfem_brick m);
int64_t mfem_brick_num_elements(mfem_brick m);
/* SoA device coords x1|x2|x3, each ncp doubles (controlpoints.x1/x2/x3). */
double* mfem_brick_coords(mfem_brick m, int32_t dim_id);
/* Slab restriction for domain decomposition along i: this handle then owns node
 * planes [i_lo, i_hi) of the global lattice and assembles only those rows; columns refer to the
 * LOCAL numbering [owned | ghosts] with itp_order ghost planes per side and field (a row couples to control points up
 * to itp_order planes away; see mfem_context_set_comm).  Slabs of an order-2 lattice start and end on element
 * boundaries (even plane index, or the last plane) and own at least 2 planes.  Default = whole mesh. */
int mfem_brick_set_slab(mfem_brick m, int32_t plane_lo, int32_t plane_hi);

/* assemble_SparseID! + sort + generate_J_ptr (solver/03_GlobalAssembly.jl:77-140;
 * misc/04_GPU_Utils.jl:87-118) for n_fields field-major blocks (all (dual,base) pairs present):
 * analytic row-sorted CSR, no hash table, no COO sort.  Outputs are library-owned device arrays
 * valid until mfem_csr_destroy: rowptr is int64, 0-based. */
int mfem_brick_pattern(mfem_context ctx, mfem_brick m, int32_t n_fields, mfem_csr* out);
/* The same for UNSTRUCTURED connectivity: controlpoint_IDs[itp, nel] (column-major, ids per index_base) ->
 * row-sorted CSR of n_fields field-major blocks (radix sort + unique of the packed (cp_i, cp_j) keys; no hash
 * table).  sparse_IDs_by_el (optional, device, [n_fields^2][itp, itp, nel] int32, ids per index_base) receives
 * the CSR slot of every element entry for every (dual, base) field block u = dual*n_fields + base -- what
 * `sparse_IDs_by_el + sparse_mapping[(dual,base)]*sparse_unitsize` addresses in the reference (:111-118,148-151). */
int mfem_pattern_build(mfem_context ctx, int32_t itp, int64_t nel, int64_t ncp, const int32_t* controlpoint_IDs,
                       int32_t index_base, int32_t n_fields, mfem_csr* out, int32_t* sparse_IDs_by_el);
const int64_t* mfem_csr_rowptr64(mfem_csr A);
const int32_t* mfem_csr_colidx(mfem_csr A);
int64_t mfem_csr_nnz(mfem_csr A);
int64_t mfem_csr_n(mfem_csr A);
/* Number of columns = length of the x (and of a per-column vector such as the Jacobi d) this pattern addresses: n for a
 * square pattern; n + ghost entries for a slab pattern (mfem_brick_set_slab), whose ghost columns are numbered behind the
 * owned ones.  mfem_jacobi2_by_column zeroes and fills that many entries of d; mfem_mat_div_jacobi reads them. */
int64_t mfem_csr_ncols(mfem_csr A);

/* ---- S2 fused assembly closures (constant-coefficient fast paths) ------------------------ */
/* Faces of the brick in the reference's local face numbering (ref_geometry/002_Initialization.jl:8;
 * spatial_discretization/103_Integrations.jl:24-30): bit f-1 set <=> face f,
 * 1: z=0, 2: y=0, 3: x=L, 4: y=L, 5: x=0, 6: z=L. */
typedef struct {
  double k;            /* -k*Bilinear(T{;i},T{;i})                   thermal_conduction/3D_Script.jl:30 */
  double h;            /* h*Bilinear(T, Tenv - T) on robin_faces     :31 */
  double Tenv;
  uint32_t robin_faces;
  /* ABI 5: weakly imposed (Nitsche-type) Dirichlet faces, the reference's own way of fixing a temperature --
   *   h_penalty*Bilinear(T, Tw - T) + k*Bilinear(T, n{i}*T{;i}) on fixed_faces     thermal_conduction/2D_Script.jl:58
   * with k the conductivity above.  The second term couples a face node's row to ALL nodes of the host element and has no mirrored
   * counterpart: K is NONSYMMETRIC wherever fixed_faces != 0 (cg! is then not applicable; idrs! / bicgstabl_GS! as in the script). */
  uint32_t fixed_faces;
  double h_penalty;
  double Tw;
} mfem_thermal_params;

/* K_linear_func for the thermal weak form: vals (CSR order of mfem_brick_pattern(.., 1, ..)) =
 *   sum_el sum_q w (-k) dN_a.dN_b + sum_robin sum_q w^s (-h) N_a N_b + sum_fixed sum_q w^s N_a (-h_penalty N_b + k n.grad N_b).
 * Overwrites vals. */
int mfem_brick_assemble_thermal(mfem_context ctx, mfem_brick m, mfem_csr A, const mfem_thermal_params* p,
                                double* vals);
/* K_nonlinear_func (residual part) for the same form, evaluated matrix-free at x_star:
 *   residue[a] = sum_q w(-k dN_a.dT + N_a s) + sum_robin w^s N_a h (Tenv - T) + sum_fixed w^s N_a (h_penalty (Tw - T) + k n.grad T).
 * s = nodal source (CONTROLPOINT_VAR `s`, may be NULL = 0).  Overwrites residue. */
int mfem_brick_residual_thermal(mfem_context ctx, mfem_brick m, const mfem_thermal_params* p,
                                const double* x_star, const double* s, double* residue);

typedef struct {
  double lambda, mu;       /* sigma = lambda*delta*eps_mm + 2 mu eps   cantilever/3D_Script.jl:53-57 */
  double tau;              /* tau*Bilinear(d{i}, dw{i} - d{i}) on penalty_faces, dw = 0   :60 */
  uint32_t penalty_faces;
  uint32_t traction_faces; /* Bilinear(d{i}, sig{i,j} n{j}) with constant sig   :61 */
  double sig[6];           /* symmetric tensor 11,22,33,23,13,12 */
} mfem_elasticity_params;
int mfem_brick_assemble_elasticity(mfem_context ctx, mfem_brick m, mfem_csr A, const mfem_elasticity_params* p,
                                   double* vals);
int mfem_brick_residual_elasticity(mfem_context ctx, mfem_brick m, const mfem_elasticity_params* p,
                                   const double* x_star, double* residue);

/* ---- generic geometry update (the tables the S3 operators consume; not used by the fused fast paths) -------- */
/* update_BasicElements_{2,3}D + inv_Jac + update_Basic_itgval_1 (mesh/unstructured_mesh/4_Update_Integrator.jl:2-33,
 * 77-154).  ref_itp_vals[itg, itp, 1+dim]: reference value (slot 0) and first derivatives d/dxi_m (slot 1+m);
 * coords: SoA x1|x2|x3, ncp each; controlpoint_IDs[itp, nel].  Outputs (column-major, caller-allocated):
 * integral_vals[itg, itp, 1+dim, nel] (slot 0 value, 1+s = d/dx_s), integral_weights[itg, nel] = w_q det J. */
int mfem_update_basic_elements(mfem_context ctx, int32_t dim, int32_t itg, int32_t itp, int64_t nel, int64_t ncp,
                               const double* ref_itp_vals, const double* itg_weight, const double* coords,
                               const int32_t* controlpoint_IDs, int32_t index_base, double* integral_vals,
                               double* integral_weights);
/* update_BasicBoundary_{2,3}D + tangents + normals (:35-75, 163-227) for facets bound to (element_ID, element_eindex).
 * Per local face id f (n_face_ids of them, contiguous): bdy_ref_itp_vals[f][itg_b, itp, 1+dim], bdy_itg_weights[f][itg_b],
 * bdy_tangent_directions[f][itg_b, dim, dim-1].  Outputs: integral_vals[itg_b, itp, 1+dim, nf],
 * integral_weights[itg_b, nf] = w_q * surface det, normal_directions[itg_b, dim, nf] (outward unit normal). */
int mfem_update_basic_boundary(mfem_context ctx, int32_t dim, int32_t itg_b, int32_t itp, int32_t n_face_ids,
                               int64_t n_facets, int64_t ncp, const double* bdy_ref_itp_vals,
                               const double* bdy_itg_weights, const double* bdy_tangent_directions, const double* coords,
                               const int32_t* controlpoint_IDs, const int32_t* element_ID, const int32_t* element_eindex,
                               int32_t index_base, double* integral_vals, double* integral_weights,
                               double* normal_directions);

/* ---- S3 generic element operators (reference signatures, accumulate semantics) ----------- */
/* itp_vals is integral_vals[itg, itp, n_sd, n_host] column-major with the derivative hyper-cube
 * flattened to n_sd slots; *_sd are flat 0-based slot offsets (the reference's sd_IDs tuple).
 * ids are 0- or 1-based per index_base.  n_threads = length of elIDs (one work item each).
 * colour_offsets (n_colours+1, [host]) partitions the work items into race-free batches; pass
 * n_colours = 0 to use FP64 atomics instead (the reference's behaviour, 06_FEM_Kernel.jl:10,42,77). */
typedef struct {
  int32_t itg, itp, n_sd;
  int64_t n_host;          /* last dim of itp_vals */
  int32_t index_base;
  int32_t n_colours;
  const int64_t* colour_offsets; /* [host] */
} mfem_op_layout;
/* _Var_Basic  solver/06_FEM_Kernel.jl:1-13 : target[q,t] += sum_a N[q,a,sd,host_t] x[cp[a,el_t]+shift] */
int mfem_op_var(mfem_context ctx, const mfem_op_layout* L, const double* itp_vals, int32_t sd, int64_t cpID_shift,
                const int32_t* el_g_cpIDs, const double* x, double* target, const int32_t* itg_hostIDs,
                const int32_t* elIDs, int64_t n_threads);
/* _Kval_Basic :28-45 : K[slot[a,b,el_t]+shift] += sum_q N[q,a,dsd] N[q,b,bsd] vals[q,t] */
int mfem_op_kval(mfem_context ctx, const mfem_op_layout* L, const double* itp_vals, int32_t dual_sd, int32_t base_sd,
                 const double* vals, const int32_t* sparse_IDs_by_el, int64_t sparse_ID_shift, double* K_val,
                 const int32_t* itg_hostIDs, const int32_t* elIDs, int64_t n_threads);
/* _Res_Basic :65-79 : residue[cp[a,el_t]+shift] += sum_q N[q,a,dsd] vals[q,t] */
int mfem_op_res(mfem_context ctx, const mfem_op_layout* L, const double* itp_vals, int32_t dual_sd, const double* vals,
                int64_t cpID_shift, const int32_t* el_g_cpIDs, double* residue, const int32_t* itg_hostIDs,
                const int32_t* elIDs, int64_t n_threads);

/* ---- batched forms of the three operators (new) ------------------------------------------------
 * The generated updaters of the reference launch _Kval_Basic / _Res_Basic / _Var_Basic once per term and re-read the
 * item's whole basis table every time (21 launches for 3-D elasticity, 05_CodeGenerator.jl:52-154).  These entry points
 * take the term list of one integration domain at once: the item's table is staged in LDS once, terms that hit the same
 * sparse block / dual field are summed in registers before the single accumulate into K / residue.  Semantics are the
 * sum of the corresponding single-term calls (to round-off: the summation order over terms differs).
 * Term arrays are [host]; `vals` / `targets` are term-major: term t at + t * itg * n_threads. */
#define MFEM_MAX_BATCH_TERMS 48
typedef struct {
  int32_t dual_sd, base_sd; /* flat derivative ids of the dual / base word */
  int32_t block;            /* sparse block u = sparse_mapping[(dual_pos, base_pos)]; terms must be sorted by block */
  int32_t reserved;
} mfem_kval_term;
/* slot(a, b, el; u) = sparse_IDs_by_el[u * slot_block_stride + a + itp*(b + itp*el)] + u * sparse_ID_shift_unit */
int mfem_op_kval_batch(mfem_context ctx, const mfem_op_layout* L, const double* itp_vals, int32_t n_terms,
                       const mfem_kval_term* terms, const double* vals, const int32_t* sparse_IDs_by_el,
                       int64_t slot_block_stride, int64_t sparse_ID_shift_unit, double* K_val, const int32_t* itg_hostIDs,
                       const int32_t* elIDs, int64_t n_threads);
typedef struct {
  int32_t dual_sd, reserved;
  int64_t cpID_shift;       /* dual_pos * variable_size; terms must be sorted by cpID_shift */
} mfem_res_term;
int mfem_op_res_batch(mfem_context ctx, const mfem_op_layout* L, const double* itp_vals, int32_t n_terms,
                      const mfem_res_term* terms, const double* vals, const int32_t* el_g_cpIDs, double* residue,
                      const int32_t* itg_hostIDs, const int32_t* elIDs, int64_t n_threads);
typedef struct {
  int32_t sd, reserved;
  int64_t cpID_shift;
  const double* x;          /* [device] source vector of this word (x_star, or a control-point array for externals) */
} mfem_var_term;
/* targets[t][q, item] = sum_a N[q,a,sd_t,host] x_t[cp[a,el] + shift_t]   (OVERWRITES: the reference passes fresh zeros) */
int mfem_op_var_batch(mfem_context ctx, const mfem_op_layout* L, const double* itp_vals, int32_t n_terms,
                      const mfem_var_term* terms, const int32_t* el_g_cpIDs, double* targets, const int32_t* itg_hostIDs,
                      const int32_t* elIDs, int64_t n_threads);

/* ---- fused assembly on unstructured meshes (constant-coefficient terms; new) ------------------------------------------
 * The generic path above stores every element's physical basis table (update_BasicElements: hex-20 with 27 Gauss points =
 * 17 KB per element) and re-reads it once per term.  For terms whose coefficient is a constant -- every linear gradient of
 * the thermal, elasticity, inertia, convection and penalty forms of the example scripts -- these entry points do geometry,
 * all terms of the integration domain and the scatter in one launch per colour (one launch with atomics): a wave owns an
 * element (facet), builds J, det, J^-1 and the physical table in LDS and accumulates, per node pair (a, b),
 *     M_ab[s][s'] = sum_q w_q det_q D^s N_a(q) D^s' N_b(q)      (s = 0 value, 1 + j = d/dx_j)
 * once; term t then adds coef_t * M_ab[dual_sd_t][base_sd_t] to K at slot(a, b, el; block_t).  Semantics = the sum of
 * the corresponding _Kval_Basic calls with vals = coef * w (06_FEM_Kernel.jl:28-45; 05_CodeGenerator.jl:52-91), to
 * round-off.  Works for any classical element the host has reference tables for (itg * itp * (1 + dim) doubles <= 64 KB).
 * elIDs / facetIDs (optional, ids per index_base): the item each work unit processes, e.g. elements sorted by colour;
 * n_colours = 0 -> FP64 atomics, else colour_offsets[n_colours + 1] [host] partitions the n_items work units. */
typedef struct {
  int32_t dual_sd, base_sd; /* 0 = value, 1 + j = d/dx_j */
  int32_t block;            /* sparse block u = dual_pos * n_fields + base_pos; terms sorted by block */
  int32_t reserved;
  double coef;              /* the constant coefficient, K_params factor included */
} mfem_const_term;
int mfem_mesh_assemble_elements(mfem_context ctx, int32_t dim, int32_t itg, int32_t itp, int64_t nel, int64_t ncp,
                                const double* ref_itp_vals, const double* itg_weight, const double* coords,
                                const int32_t* controlpoint_IDs, int32_t index_base, int32_t n_terms,
                                const mfem_const_term* terms, const int32_t* sparse_IDs_by_el, int64_t slot_block_stride,
                                double* K_val, const int32_t* elIDs, int64_t n_items, int32_t n_colours,
                                const int64_t* colour_offsets);
/* Row-owner form of the same element assembly (large meshes: scattering a hex-20 elasticity element matrix is 3600
 * read-modify-writes of 8 bytes, each its own memory sector).  Pass 1 writes the element matrices to a library-owned
 * element-major scratch (unit-stride stores), pass 2 gives every CSR row (dual field, node) to a wave that walks the node's
 * adjacency list in ascending element order and adds the contiguous scratch runs at the row's columns: no atomics, no
 * colours, a fixed summation order (bitwise reproducible), K read and written once, contiguously.  K_val is ACCUMULATED
 * into (zero it first; facet terms are added by mfem_mesh_assemble_facets before or after).
 * adj_ptr [ncp + 1], adj [nel * itp]: for every control point the (element * itp + local node id) pairs that reference it,
 * 0-based, ascending.  A = the pattern of mfem_pattern_build for n_fields fields.  MFEM_ERR_UNSUPPORTED if a row is longer
 * than 2048 entries or the scratch (nel * itp^2 * blocks * 8 B) exceeds 16 GiB: use mfem_mesh_assemble_elements then. */
int mfem_mesh_assemble_elements_rows(mfem_context ctx, int32_t dim, int32_t itg, int32_t itp, int64_t nel, int64_t ncp,
                                     const double* ref_itp_vals, const double* itg_weight, const double* coords,
                                     const int32_t* controlpoint_IDs, int32_t index_base, int32_t n_terms,
                                     const mfem_const_term* terms, int32_t n_fields, mfem_csr A, const int64_t* adj_ptr,
                                     const int32_t* adj, const uint16_t* ranks, double* K_val);
/* The same, K_val OVERWRITTEN: every entry of every row of the pattern receives the sum of its element contributions (zero where none) -- K_val need
 * not be initialised.  What K_linear_func does first (05_CodeGenerator.jl:282: K_linear starts from zero): the memset of K and the read of the zeros
 * are 2 x 8 bytes per nonzero that the accumulating form pays (hex-20 elasticity at 96^3: 30 GB of 70).  nel must be > 0. */
int mfem_mesh_assemble_elements_rows_set(mfem_context ctx, int32_t dim, int32_t itg, int32_t itp, int64_t nel, int64_t ncp,
                                         const double* ref_itp_vals, const double* itg_weight, const double* coords,
                                         const int32_t* controlpoint_IDs, int32_t index_base, int32_t n_terms,
                                         const mfem_const_term* terms, int32_t n_fields, mfem_csr A, const int64_t* adj_ptr,
                                         const int32_t* adj, const uint16_t* ranks, double* K_val);
/* ranks [nel * itp * itp] (device, once per pattern): for adjacency entry j = (node i <- element el, local a) and local node
 * b, the position of node(el, b) among the control points coupled to i -- the column offset inside every field segment of a
 * row of node i (the role the reference's sparse_IDs_by_el plays for its scatter, read unit-stride by the row-owner pass).
 * MFEM_ERR_UNSUPPORTED when an element lists one control point twice (collapsed elements): the row-owner pass assumes distinct
 * positions per element; use mfem_mesh_assemble_elements there. */
int mfem_mesh_row_ranks(mfem_context ctx, int32_t itp, int64_t nel, int64_t ncp, int32_t n_fields, mfem_csr A,
                        const int64_t* adj_ptr, const int32_t* adj, const int32_t* controlpoint_IDs, int32_t index_base,
                        uint16_t* ranks);
/* The same on boundary facets (element_ID, element_eindex as in mfem_update_basic_boundary): the basis of ALL nodes of the
 * host element evaluated on the face (05_CodeGenerator.jl:175-189), weight = w_q * surface det. */
int mfem_mesh_assemble_facets(mfem_context ctx, int32_t dim, int32_t itg_b, int32_t itp, int32_t n_face_ids, int64_t n_facets,
                              int64_t ncp, const double* bdy_ref_itp_vals, const double* bdy_itg_weights,
                              const double* bdy_tangent_directions, const double* coords, const int32_t* controlpoint_IDs,
                              const int32_t* element_ID, const int32_t* element_eindex, int32_t index_base, int32_t n_terms,
                              const mfem_const_term* terms, const int32_t* sparse_IDs_by_el, int64_t slot_block_stride,
                              double* K_val, const int32_t* facetIDs, int64_t n_items, int32_t n_colours,
                              const int64_t* colour_offsets);

/* ---- multi-GPU (new; the reference is single-GPU, F6) ------------------------------------ */
/* 128-byte RCCL unique id, created on rank 0 and shipped to the other ranks by the host
 * (torch.distributed / MPI / a file).  */
int mfem_comm_unique_id(void* out128 /* [host] */);
int mfem_comm_create(mfem_context ctx, int32_t rank, int32_t world, const void* unique_id128, mfem_comm* out);
/* The same communicator behind host callbacks instead of RCCL: the library stages device data through pinned host memory
 * and calls back for the actual exchange (gloo, MPI, ... -- whatever the host has).  Every solver code path is identical to
 * the RCCL one (same kernels, same schedule: halo begin / interior rows / halo end / boundary rows, same reduction groups);
 * only the transport differs.  For ranks that share one GPU (RCCL rejects duplicate devices: this is how the multi-rank
 * path is tested on a single-GPU box) and for hosts without GPU-aware MPI.  Callbacks return 0 on success and are called on
 * the thread that called into the library, with the context stream idle. */
typedef struct {
  void* user;
  /* in-place sum over all ranks of `count` doubles in host memory */
  int (*allreduce_sum)(void* user, double* buf, int32_t count);
  /* one exchange with both slab neighbours: send_lo -> rank - 1, recv_lo <- rank - 1, send_hi -> rank + 1, recv_hi <- rank + 1,
   * `count` doubles each; the pointers of a missing neighbour (first / last rank) are NULL.  Must not deadlock when all ranks
   * call it at the same time (post the receives and sends, then wait). */
  int (*neighbour_exchange)(void* user, const double* send_lo, double* recv_lo, const double* send_hi, double* recv_hi,
                            int64_t count);
  uint32_t flags; /* MFEM_COMM_HOST_* */
  uint32_t reserved;
} mfem_comm_host_ops;
#define MFEM_COMM_HOST_POISON_GHOSTS 1u /* test aid: ghost entries are NaN between halo begin and end */
int mfem_comm_create_host(mfem_context ctx, int32_t rank, int32_t world, const mfem_comm_host_ops* ops, mfem_comm* out);
int mfem_comm_destroy(mfem_comm c);
/* Attach to a context (c = NULL detaches): subsequent mfem_solve calls on slab matrices all-reduce their
 * scalars over the communicator and, before each SpMV, exchange one ghost block of `plane_len` doubles per
 * field with ranks +-1 (the first / last plane_len owned entries of the field go out).  plane_len = itp_order control-
 * point planes: m1*m2 for hex-8, 2*m1*m2 for hex-27.  Local vectors are [owned: n_fields x n_owned_nodes, field-major |
 * ghosts: (field 0 lo, field 0 hi, field 1 lo, ...) x plane_len]  (see mfem_brick_set_slab). */
int mfem_context_set_comm(mfem_context ctx, mfem_comm c, int64_t n_owned_nodes, int64_t plane_len, int32_t n_fields);
int mfem_allreduce_sum(mfem_context ctx, double* dev_scalars, int32_t count);
int mfem_halo_exchange(mfem_context ctx, double* x_local);
/* Reverse exchange: the ghost blocks of x_local (contributions this rank accumulated for entries its neighbours own) are
 * sent to the owners and ADDED to their first / last plane_len owned entries per field.  With it a column sum over a slab
 * matrix (mfem_jacobi2_by_column on [owned | ghost] columns) becomes the global column sum. */
int mfem_halo_reduce(mfem_context ctx, double* x_local);

#ifdef __cplusplus
}
#endif
#endif /* METAFEM_MI355X_H */
