# MI355X.jl -- Julia side of the drop-in boundary: binds libmetafem_mi355x.so (include/metafem_mi355x.h) into MetaFEM.jl.
#
# STATUS: WRITTEN, NOT EXECUTED.  The build image of this backend has no Julia (SURVEY.md F2), so this file has never been
# parsed or run by a Julia process; it is the binding a MetaFEM.jl maintainer would `include` from src/MetaFEM.jl after
# src/solver/linear_solver/*.jl, kept in step with the header by hand.  The very same entry points, struct layouts and argument
# orders are exercised from Python (metafem.jl_amd/_lib.py, ctypes) by the GPU test-suite and from plain C by
# tests/c_abi_smoke.c.  Citations `file:line` are relative to the reference's src/ (jxx2/MetaFEM.jl v0.1.4).
#
# What it replaces (SURVEY.md section 8b):
#   S1  fem_domain.linear_solver(globalfield)          solver/01_Types.jl:166, called solver/04_Time_Domain.jl:76
#       -> iterative_Solve!(gf; Sv_func!, Pl_func, ...) linear_solver/02_Preconditioner.jl:32-76          => mfem_solve
#       mul! / dot / norm / FEM_rand                    misc/04_GPU_Utils.jl:131,22; LinearAlgebra         => mfem_spmv_csr, mfem_dot, ...
#   S2  K_linear_func / K_nonlinear_func closures       solver/01_Types.jl:164-165 (assigned 05_CodeGenerator.jl:287-288)
#                                                                                                          => mfem_brick_assemble_* / _residual_*
#   S3  _Var_Basic / _Kval_Basic / _Res_Basic           solver/06_FEM_Kernel.jl:1,28,65                    => mfem_op_var / _kval / _res
#   S0  array backend entry                             misc/04_GPU_Utils.jl:1-38                           => ROCArray methods below
module MI355X

using AMDGPU
import LinearAlgebra

# ---- library handle and error convention ---------------------------------------------------------------------------------
const lib = get(ENV, "METAFEM_MI355X_LIB", joinpath(@__DIR__, "..", "metafem.jl_amd", "libmetafem_mi355x.so"))
const ABI_VERSION = 5    # == MFEM_ABI_VERSION

struct MFEMError <: Exception
    rc::Cint
    msg::String
end

function check(rc::Cint)
    rc == 0 && return nothing
    throw(MFEMError(rc, unsafe_string(ccall((:mfem_last_error, lib), Cstring, ()))))
end

"Raw device pointer of a ROCArray as the `void*` the C ABI takes."
dptr(a) = Ptr{Cvoid}(UInt(pointer(a)))
dptr(::Nothing) = C_NULL

mutable struct Context
    h::Ptr{Cvoid}
end

"One context per device and host thread; work is enqueued on Julia's current HIP stream."
function Context(device::Integer = 0)
    got = ccall((:mfem_abi_version, lib), Cint, ())
    got == ABI_VERSION || error("libmetafem_mi355x ABI $got, this binding was written for $ABI_VERSION")
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:mfem_context_create, lib), Cint, (Cint, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), device, AMDGPU.stream().stream, h))
    c = Context(h[])
    finalizer(x -> ccall((:mfem_context_destroy, lib), Cint, (Ptr{Cvoid},), x.h), c)
    return c
end

const CTX = Ref{Context}()
ctx() = (isassigned(CTX) || (CTX[] = Context(0)); CTX[].h)

# ---- struct mirrors of include/metafem_mi355x.h (isbits, C layout) ---------------------------------------------------------
Base.@kwdef struct SolveOptions           # == mfem_solve_options
    method::Int32 = 2                     # 0 cg (new), 1 bicgstabl_GS!, 2 idrs!, 3 cgs2!
    precond::Int32 = 1                    # 0 Identity, 1 Pr_Jacobi!, 2 Pr_Jacobi!(normalized_by_column = true)
    l_or_s::Int32 = 0                     # the `s` kwarg
    maxiter::Int32 = 2000
    max_pass::Int32 = 4
    check_every::Int32 = 32
    converge_tol::Float64 = 1e-6          # ABSOLUTE on norm(r)/sqrt(n): globalfield.converge_tol
    seed::UInt64 = 0x5EED
    fixed_iterations::Int32 = 0
    scale_in_place::Int32 = 0
    left_precond::Int32 = 0               # Pl_func: 0 Identity, 1 Pl_Jacobi, 2 Pl_Jacobi(normalized_by_row = true)
    cg_variant::Int32 = 0
end

struct SolveStats                         # == mfem_solve_stats
    passes::Int32
    iterations::Int32
    final_res::Float64
    initial_res::Float64
    solve_ms::Float64
    converged::Int32
    spmv_count::Int32
end

Base.@kwdef struct ThermalParams          # == mfem_thermal_params
    k::Float64
    h::Float64 = 0.0
    Tenv::Float64 = 0.0
    robin_faces::UInt32 = 0x00
    fixed_faces::UInt32 = 0x00            # ABI 5: h_penalty*Bilinear(T, Tw - T) + k*Bilinear(T, n{i}*T{;i}) (thermal_conduction/2D_Script.jl:58)
    h_penalty::Float64 = 0.0
    Tw::Float64 = 0.0
end

struct ElasticityParams                   # == mfem_elasticity_params
    lambda::Float64
    mu::Float64
    tau::Float64
    penalty_faces::UInt32
    traction_faces::UInt32
    sig::NTuple{6, Float64}               # 11, 22, 33, 23, 13, 12
end

struct OpLayout                           # == mfem_op_layout
    itg::Int32
    itp::Int32
    n_sd::Int32
    n_host::Int64
    index_base::Int32
    n_colours::Int32
    colour_offsets::Ptr{Int64}            # [host]; C_NULL with n_colours = 0 -> FP64 atomics (the reference's behaviour)
end

struct KvalTerm                           # == mfem_kval_term
    dual_sd::Int32
    base_sd::Int32
    block::Int32
    reserved::Int32
end

struct ResTerm                            # == mfem_res_term
    dual_sd::Int32
    reserved::Int32
    cpID_shift::Int64
end

struct VarTerm                            # == mfem_var_term
    sd::Int32
    reserved::Int32
    cpID_shift::Int64
    x::Ptr{Cvoid}
end

struct ConstTerm                          # == mfem_const_term
    dual_sd::Int32
    base_sd::Int32
    block::Int32
    reserved::Int32
    coef::Float64
end

# ---- S1 primitives: FEM_SpMat_CSR, mul!, dot, norm, FEM_rand ----------------------------------------------------------------
"""
CSR pattern handle (no values): what `FEM_SpMat_CSR(K_J_ptr, K_J, K_vals, dims)` (misc/04_GPU_Utils.jl:120) wraps.
`rowptr` / `colidx` are BORROWED and frozen while the handle lives (assemble_SparseID! writes them once per mesh).
"""
mutable struct CSRPattern
    h::Ptr{Cvoid}
    rowptr
    colidx
    n::Int64
end

function CSRPattern(rowptr::ROCArray{Int32}, colidx::ROCArray{Int32}, n::Integer; index_base::Integer = 1)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:mfem_csr_create, lib), Cint, (Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Ref{Ptr{Cvoid}}),
                ctx(), n, length(colidx), dptr(rowptr), 32, dptr(colidx), index_base, h))
    A = CSRPattern(h[], rowptr, colidx, n)
    finalizer(x -> ccall((:mfem_csr_destroy, lib), Cint, (Ptr{Cvoid},), x.h), A)
    return A
end

"The reference's matrix object: pattern + the values of this Newton step."
struct SpMat_CSR
    A::CSRPattern
    vals::ROCArray{Float64}
end
FEM_SpMat_CSR(J_ptr::ROCArray{Int32}, Js::ROCArray{Int32}, Ks::ROCArray{Float64}, dim::Tuple) =
    SpMat_CSR(CSRPattern(J_ptr, Js, dim[1]), Ks)

"mul!(b, A, x, alpha, beta): b = alpha*A*x + beta*b  (misc/04_GPU_Utils.jl:131, CUSPARSE mv! 'N')."
function mul!(b::ROCVector{Float64}, M::SpMat_CSR, x::ROCVector{Float64}, alpha::Number = 1.0, beta::Number = 0.0)
    check(ccall((:mfem_spmv_csr, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Float64),
                ctx(), M.A.h, dptr(M.vals), dptr(x), dptr(b), Float64(alpha), Float64(beta)))
    return b
end

function dot(x::ROCVector{Float64}, y::ROCVector{Float64})
    out = Ref{Float64}(0.0)
    check(ccall((:mfem_dot, lib), Cint, (Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Float64}), ctx(), length(x), dptr(x), dptr(y), out))
    return out[]
end

function norm(x::ROCVector{Float64})
    out = Ref{Float64}(0.0)
    check(ccall((:mfem_nrm2, lib), Cint, (Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ref{Float64}), ctx(), length(x), dptr(x), out))
    return out[]
end

"normalized_norm (solver/04_Time_Domain.jl:51)."
normalized_norm(x::ROCVector{Float64}) = norm(x) / sqrt(length(x))

"FEM_rand (misc/04_GPU_Utils.jl:22) with an explicit seed (the reference's CUDA.Random.rand! is unseeded, SURVEY F9)."
function FEM_rand!(x::ROCVector{Float64}; seed::UInt64 = 0x5EED, stream_id::Integer = 0)
    check(ccall((:mfem_rand, lib), Cint, (Ptr{Cvoid}, Int64, UInt64, UInt32, Ptr{Cvoid}), ctx(), length(x), seed, stream_id, dptr(x)))
    return x
end

# ---- S1: the linear-solver seam --------------------------------------------------------------------------------------------
const SOLVER_ID = Dict(:cg! => 0, :bicgstabl_GS! => 1, :idrs! => 2, :cgs2! => 3)
const PR_ID = Dict(:Identity => 0, :Pr_Jacobi! => 1)      # (install! passes nameof(Pr_func!))
const PL_ID = Dict(:Identity => 0, :Pl_Jacobi => 1)      # cylinder_flow/3D_MetaFEM_Script.jl:90 passes Pl_func = Pl_Jacobi

"""
    iterative_Solve!(globalfield; Sv_func! = :idrs!, Pr_func! = :Pr_Jacobi!, Pl_func = :Identity, max_pass = 4, maxiter = 2000, s = 0)

Same contract as linear_solver/02_Preconditioner.jl:32-76: reads `basicfield_size, K_J_ptr, K_J, K_val_ids, K_total, residue,
converge_tol`, returns a NEW device vector (x0 = 0, right-Jacobi un-scaled at exit, up to `max_pass` restarts with the true residual
recomputed in between, absolute tolerance on norm(r)/sqrt(n)).  Install with
`fem_domain.linear_solver = gf -> MI355X.iterative_Solve!(gf; Sv_func! = :idrs!, maxiter = 2000, max_pass = 10, s = 8)`.
"""
function iterative_Solve!(gf; Sv_func!::Symbol = :idrs!, Pr_func!::Symbol = :Pr_Jacobi!, Pl_func::Symbol = :Identity,
                          max_pass = 4, maxiter = 2000, s = 0, normalized_by_column::Bool = false, seed::UInt64 = 0x5EED)
    n = gf.basicfield_size
    A = get!(PATTERNS, objectid(gf.K_J_ptr)) do       # one handle per assemble_Global_Variables! (the arrays are written once)
        CSRPattern(gf.K_J_ptr, gf.K_J, n)
    end
    # the reference gathers K_total[K_val_ids] (:35); with a pattern from mfem_brick_pattern / mfem_pattern_build the values are
    # already in CSR order (K_val_ids == 1:nnz) and no copy is made
    K_vals = gf.K_val_ids isa UnitRange ? gf.K_total : gf.K_total[gf.K_val_ids]
    x = AMDGPU.zeros(Float64, n)
    pr = Pr_func! == :Pr_Jacobi! ? (normalized_by_column ? 2 : 1) : 0
    opts = Ref(SolveOptions(method = SOLVER_ID[Sv_func!], precond = pr, l_or_s = s, maxiter = maxiter, max_pass = max_pass,
                            converge_tol = gf.converge_tol, seed = seed, left_precond = PL_ID[Pl_func]))
    stats = Ref{SolveStats}()
    check(ccall((:mfem_solve, lib), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{SolveOptions}, Ref{SolveStats}),
                ctx(), A.h, dptr(K_vals), dptr(gf.residue), dptr(x), opts, stats))
    println("pass $(stats[].passes) with res = $(stats[].final_res) iter = $(stats[].iterations).")
    return x
end
const PATTERNS = Dict{UInt, CSRPattern}()

# ---- S3: the three element operators, reference argument order (solver/06_FEM_Kernel.jl:1,28,65) -----------------------------
"sd_IDs tuple (per-dimension derivative order + 1) -> flat 0-based slot of the flattened derivative hyper-cube."
sd_flat(sd_IDs::NTuple{N, <:Integer}, max_sd_order::Integer) where {N} =
    Int32(sum((sd_IDs[i] - 1) * (max_sd_order + 1)^(i - 1) for i in 1:N))

function op_layout(itp_vals::ROCArray{Float64})
    nd = ndims(itp_vals)                 # [itg, itp, (sd+1)^dim..., n_host]
    n_sd = prod(size(itp_vals)[3:nd-1])
    OpLayout(size(itp_vals, 1), size(itp_vals, 2), n_sd, size(itp_vals, nd), 1, 0, C_NULL)
end
max_sd(itp_vals) = size(itp_vals, 3) - 1

function _Var_Basic(itp_vals, sd_IDs, cpID_shift, el_g_cpIDs, x_star, target, itg_hostIDs, elIDs)
    check(ccall((:mfem_op_var, lib), Cint,
                (Ptr{Cvoid}, Ref{OpLayout}, Ptr{Cvoid}, Int32, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64),
                ctx(), Ref(op_layout(itp_vals)), dptr(itp_vals), sd_flat(sd_IDs, max_sd(itp_vals)), cpID_shift, dptr(el_g_cpIDs),
                dptr(x_star), dptr(target), dptr(itg_hostIDs), dptr(elIDs), length(elIDs)))
end

function _Kval_Basic(itp_vals, dual_sd_IDs, base_sd_IDs, vals, sparse_IDs_by_el, sparse_ID_shift, K_val, itg_hostIDs, elIDs)
    m = max_sd(itp_vals)
    check(ccall((:mfem_op_kval, lib), Cint,
                (Ptr{Cvoid}, Ref{OpLayout}, Ptr{Cvoid}, Int32, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64),
                ctx(), Ref(op_layout(itp_vals)), dptr(itp_vals), sd_flat(dual_sd_IDs, m), sd_flat(base_sd_IDs, m), dptr(vals),
                dptr(sparse_IDs_by_el), sparse_ID_shift, dptr(K_val), dptr(itg_hostIDs), dptr(elIDs), length(elIDs)))
end

function _Res_Basic(itp_vals, dual_sd_IDs, vals, cpID_shift, el_g_cpIDs, residue, itg_hostIDs, elIDs)
    check(ccall((:mfem_op_res, lib), Cint,
                (Ptr{Cvoid}, Ref{OpLayout}, Ptr{Cvoid}, Int32, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64),
                ctx(), Ref(op_layout(itp_vals)), dptr(itp_vals), sd_flat(dual_sd_IDs, max_sd(itp_vals)), dptr(vals), cpID_shift,
                dptr(el_g_cpIDs), dptr(residue), dptr(itg_hostIDs), dptr(elIDs), length(elIDs)))
end

"""
All `_Kval_Basic` terms of one integration domain in one launch (05_CodeGenerator.jl:52-91 emits one launch per term):
`vals_all[:, :, t] = @. expr_t * K_params[td_t + 1] * w[:, ids]`, `terms` sorted by `block = sparse_mapping[(dual_pos, base_pos)]`.
`slot_block_stride = 0`, `sparse_ID_shift_unit = sparse_unitsize` reproduce `sparse_IDs_by_el[...] + u * sparse_unitsize`
(03_GlobalAssembly.jl:148-151).
"""
function kval_batch!(K_val, itp_vals, terms::Vector{KvalTerm}, vals_all, sparse_IDs_by_el, sparse_unitsize, itg_hostIDs, elIDs)
    check(ccall((:mfem_op_kval_batch, lib), Cint,
                (Ptr{Cvoid}, Ref{OpLayout}, Ptr{Cvoid}, Int32, Ptr{KvalTerm}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid},
                 Ptr{Cvoid}, Ptr{Cvoid}, Int64),
                ctx(), Ref(op_layout(itp_vals)), dptr(itp_vals), length(terms), terms, dptr(vals_all), dptr(sparse_IDs_by_el),
                0, sparse_unitsize, dptr(K_val), dptr(itg_hostIDs), dptr(elIDs), length(elIDs)))
end

# ---- structured bricks: make_Brick + mesh_Classical(:Lagrange) + update_Mesh + assemble_SparseID! ----------------------------
mutable struct Brick
    h::Ptr{Cvoid}
end

"make_Brick(size, num, :CUBE) (mesh/ref_geometry/201_Helper_TM.jl:36-51) + mesh_Classical(itp_type = :Lagrange) on the device."
function Brick(size::NTuple{3, Real}, num::NTuple{3, Integer}; itp_order::Integer = 1, itg_order::Integer = 3)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:mfem_brick_create, lib), Cint,
                (Ptr{Cvoid}, Int32, Int32, Int32, Float64, Float64, Float64, Int32, Int32, Ref{Ptr{Cvoid}}),
                ctx(), num[1], num[2], num[3], size[1], size[2], size[3], itp_order, itg_order, h))
    b = Brick(h[])
    finalizer(x -> ccall((:mfem_brick_destroy, lib), Cint, (Ptr{Cvoid},), x.h), b)
    return b
end

"Row-sorted CSR of `n_fields` field-major blocks, library-owned (int64 row pointers, 0-based): K_val_ids = 1:nnz."
function pattern(b::Brick, n_fields::Integer)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:mfem_brick_pattern, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ref{Ptr{Cvoid}}), ctx(), b.h, n_fields, h))
    n = ccall((:mfem_csr_n, lib), Int64, (Ptr{Cvoid},), h[])
    A = CSRPattern(h[], nothing, nothing, n)
    finalizer(x -> ccall((:mfem_csr_destroy, lib), Cint, (Ptr{Cvoid},), x.h), A)
    return A
end
nnz(A::CSRPattern) = ccall((:mfem_csr_nnz, lib), Int64, (Ptr{Cvoid},), A.h)

# ---- S2: fused assembly closures for constant-coefficient forms on bricks ------------------------------------------------------
"""
Installs the fused kernels as `K_linear_func` / `K_nonlinear_func` (solver/01_Types.jl:164-165) for the thermal form of
examples/thermal_conduction/3D_Script.jl:30-31.  Contract of the generated closures (05_CodeGenerator.jl:265-291): the linear
function overwrites `globalfield.K_linear`; the nonlinear one sets `residue` and `K_total`.
"""
function install_thermal_fastpath!(fem_domain, brick::Brick, A::CSRPattern, p::ThermalParams)
    gf = fem_domain.globalfield
    fem_domain.K_linear_func = (td; fem_domain) ->
        check(ccall((:mfem_brick_assemble_thermal, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{ThermalParams}, Ptr{Cvoid}),
                    ctx(), brick.h, A.h, Ref(p), dptr(gf.K_linear)))
    fem_domain.K_nonlinear_func = (td; fem_domain) -> begin
        gf.K_total = gf.K_linear                                      # no nonlinear gradients in this form
        s = fem_domain.workpieces[1].mesh.controlpoints.s
        check(ccall((:mfem_brick_residual_thermal, lib), Cint,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Ref{ThermalParams}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                    ctx(), brick.h, Ref(p), dptr(gf.x_star), dptr(s), dptr(gf.residue)))
    end
    return fem_domain
end

"The same for the elasticity form of examples/linear_elasticity/cantilever/3D_Script.jl:52-63 (3 fields, penalty + traction faces)."
function install_elasticity_fastpath!(fem_domain, brick::Brick, A::CSRPattern, p::ElasticityParams)
    gf = fem_domain.globalfield
    fem_domain.K_linear_func = (td; fem_domain) ->
        check(ccall((:mfem_brick_assemble_elasticity, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{ElasticityParams}, Ptr{Cvoid}),
                    ctx(), brick.h, A.h, Ref(p), dptr(gf.K_linear)))
    fem_domain.K_nonlinear_func = (td; fem_domain) -> begin
        gf.K_total = gf.K_linear
        check(ccall((:mfem_brick_residual_elasticity, lib), Cint,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Ref{ElasticityParams}, Ptr{Cvoid}, Ptr{Cvoid}),
                    ctx(), brick.h, Ref(p), dptr(gf.x_star), dptr(gf.residue)))
    end
    return fem_domain
end

"""
Unstructured meshes (read_Mesh / mesh_Classical): every constant-coefficient `K_linear` term of an integration domain in one call
(row-owner form).  `terms`: word 0 = value, 1 + j = d/dx_j; block = sparse_mapping[(dual_pos, base_pos)]; coef includes K_params.
`adj_ptr`, `adj`, `ranks`: built once per assemble_Global_Variables! (see `row_ranks!`).  K_linear is ACCUMULATED into; `overwrite = true`
(the element domain is the first thing K_linear receives after `K_linear .= 0`, 05_CodeGenerator.jl:282) writes every row instead: no memset needed.
"""
function assemble_const_terms!(K_linear, dim, itg, itp, nel, ncp, ref_itp_vals, itg_weight, coords, controlpoint_IDs,
                               terms::Vector{ConstTerm}, n_fields, A::CSRPattern, adj_ptr, adj, ranks; overwrite::Bool = false)
    sort!(terms, by = t -> t.block)
    if overwrite && nel > 0
        check(ccall((:mfem_mesh_assemble_elements_rows_set, lib), Cint,
                    (Ptr{Cvoid}, Int32, Int32, Int32, Int64, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32, Int32,
                     Ptr{ConstTerm}, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                    ctx(), dim, itg, itp, nel, ncp, dptr(ref_itp_vals), dptr(itg_weight), dptr(coords), dptr(controlpoint_IDs), 1,
                    length(terms), terms, n_fields, A.h, dptr(adj_ptr), dptr(adj), dptr(ranks), dptr(K_linear)))
        return
    end
    check(ccall((:mfem_mesh_assemble_elements_rows, lib), Cint,
                (Ptr{Cvoid}, Int32, Int32, Int32, Int64, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32, Int32,
                 Ptr{ConstTerm}, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                ctx(), dim, itg, itp, nel, ncp, dptr(ref_itp_vals), dptr(itg_weight), dptr(coords), dptr(controlpoint_IDs), 1,
                length(terms), terms, n_fields, A.h, dptr(adj_ptr), dptr(adj), dptr(ranks), dptr(K_linear)))
end

function row_ranks!(ranks::ROCArray{UInt16}, itp, nel, ncp, n_fields, A::CSRPattern, adj_ptr, adj, controlpoint_IDs)
    check(ccall((:mfem_mesh_row_ranks, lib), Cint,
                (Ptr{Cvoid}, Int32, Int64, Int64, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ptr{Cvoid}),
                ctx(), itp, nel, ncp, n_fields, A.h, dptr(adj_ptr), dptr(adj), dptr(controlpoint_IDs), 1, dptr(ranks)))
    return ranks
end

# ---- several GPUs: one Julia process per GPU (MPI.jl), slab decomposition along the first dimension ---------------------------
"Rank 0 creates the 128-byte RCCL id; the host broadcasts it (MPI.Bcast!)."
function comm_unique_id()
    id = Vector{UInt8}(undef, 128)
    check(ccall((:mfem_comm_unique_id, lib), Cint, (Ptr{UInt8},), id))
    return id
end

function attach_comm!(rank::Integer, nranks::Integer, id::Vector{UInt8}, n_owned_nodes::Integer, plane_len::Integer, n_fields::Integer)
    c = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:mfem_comm_create, lib), Cint, (Ptr{Cvoid}, Int32, Int32, Ptr{UInt8}, Ref{Ptr{Cvoid}}), ctx(), rank, nranks, id, c))
    check(ccall((:mfem_context_set_comm, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int32), ctx(), c[], n_owned_nodes, plane_len, n_fields))
    return c[]
end

set_slab!(b::Brick, plane_lo::Integer, plane_hi::Integer) =
    check(ccall((:mfem_brick_set_slab, lib), Cint, (Ptr{Cvoid}, Int32, Int32), b.h, plane_lo, plane_hi))

# ---- S0: array backend entry (misc/04_GPU_Utils.jl:1-38) ----------------------------------------------------------------------
# MetaFEM.jl side (to be added next to GPU_DeviceArray):
#     ROC_DeviceArray{T, N} = ROCArray{T, N}
#     FEM_ArrayTypes = (:Array, :GPU_DeviceArray, :GPU_UnifiedArray, :ROC_DeviceArray)
#     FEM_zeros(::Type{ROC_DeviceArray}, ::Type{T}, dims::Number...) where T = AMDGPU.zeros(T, dims...)
#     FEM_ones(::Type{ROC_DeviceArray}, ::Type{T}, dims::Number...) where T = AMDGPU.ones(T, dims...)
#     FEM_rand(::Type{ROC_DeviceArray}, ::Type{T}, dims::Number...) where T = MI355X.FEM_rand!(ROCArray{T}(undef, dims...))
#     FEM_convert(::Type{ROC_DeviceArray}, src::AbstractArray{T, N}) where {T, N} = ROCArray{T, N}(src)
# The solver `@eval` loops (02_Preconditioner.jl:30-31, 03_BiCGstabl.jl:16) then specialise for it; iterative_Solve! above replaces
# their bodies for this array type.

# ---- install!: the seams above as METHODS of MetaFEM's own functions, so that a script written for the reference runs as it is -------------------------
"""
    MI355X.install!(MetaFEM)

Adds, inside the MetaFEM module, the ROCArray methods of the four seams (SURVEY.md section 8b) and makes ROCArray the default array type -- after it
`fem_domain.linear_solver = x -> iterative_Solve!(x; Sv_func! = idrs!, maxiter = 2000, max_pass = 10, s = 8)` (every example script's line) lands in
`mfem_solve`, and the generated updaters' `_Var_Basic` / `_Kval_Basic` / `_Res_Basic` calls (05_CodeGenerator.jl:9-10,33-34,77-78,112-113,138-139) in
`mfem_op_*`.  julia/examples/thermal_3D.jl is the whole recipe: two lines, then `include` of the reference's own script.

  S0  misc/04_GPU_Utils.jl:1-38        DEFAULT_ARRAYINFO._type, FEM_ArrayTypes, FEM_zeros / FEM_ones / FEM_rand / FEM_buffer / FEM_convert
  S1  misc/04_GPU_Utils.jl:120,131     FEM_SpMat_CSR, mul!;  linear_solver/02_Preconditioner.jl:30-76  iterative_Solve!(::GlobalField{ROCArray})
  S3  solver/06_FEM_Kernel.jl:1,28,65  the three operators on ROCArray arguments
The solver selectors arrive as the reference's FUNCTIONS (`idrs!`, `Pr_Jacobi!`, `Pl_Jacobi`); they are mapped to the enum by name.

Still Julia-side work for a maintainer (not bound here because their table layouts are MetaFEM-internal): update_BasicElements_{2,3}D /
update_BasicBoundary_{2,3}D (mesh/unstructured_mesh/4_Update_Integrator.jl:2-75) -> mfem_update_basic_elements / _boundary, and assemble_SparseID!
(solver/03_GlobalAssembly.jl:77-140) -> mfem_pattern_build; INTEGRATION.md shows both calls.
"""
function install!(MetaFEM::Module)
    me = @__MODULE__
    @eval MetaFEM begin
        const ROC_DeviceArray{T, N} = $(AMDGPU.ROCArray){T, N}
        DEFAULT_ARRAYINFO._type = ROC_DeviceArray
        FEM_zeros(::Type{ROC_DeviceArray}, ::Type{T}, dims::Number...) where {T} = $(AMDGPU).zeros(T, dims...)
        FEM_ones(::Type{ROC_DeviceArray}, ::Type{T}, dims::Number...) where {T} = $(AMDGPU).ones(T, dims...)
        FEM_rand(::Type{ROC_DeviceArray}, ::Type{T}, dims::Number...) where {T} = $me.FEM_rand!($(AMDGPU.ROCArray){T}(undef, dims...))
        FEM_buffer(::Type{ROC_DeviceArray}, ::Type{T}, dims::Number...) where {T} = $(AMDGPU).zeros(T, dims...)
        FEM_convert(::Type{ROC_DeviceArray}, src::$(AMDGPU.ROCArray)) = src
        FEM_convert(::Type{ROC_DeviceArray}, src::AbstractArray{T, N}) where {T, N} = $(AMDGPU.ROCArray){T, N}(src)
        FEM_SpMat_CSR(J_ptr::$(AMDGPU.ROCArray), Js::$(AMDGPU.ROCArray), Ks::$(AMDGPU.ROCArray), dim::Tuple) = $me.FEM_SpMat_CSR(J_ptr, Js, Ks, dim)
        mul!(b::$(AMDGPU.ROCArray){Float64, 1}, A::$me.SpMat_CSR, x::$(AMDGPU.ROCArray){Float64, 1}, alpha::Number = 1., beta::Number = 0.) =
            $me.mul!(b, A, x, alpha, beta)
        function iterative_Solve!(globalfield::GlobalField{ROC_DeviceArray}; Sv_func!::Function = idrs!, Pr_func!::Function = Pr_Jacobi!,
                                  Pl_func::Function = Identity, max_pass = 4, maxiter = 2000, s = 0, kwargs...)
            $me.iterative_Solve!(globalfield; Sv_func! = nameof(Sv_func!), Pr_func! = nameof(Pr_func!), Pl_func = nameof(Pl_func), max_pass = max_pass,
                                 maxiter = maxiter, s = s)
        end
        _Var_Basic(itp_vals::$(AMDGPU.ROCArray), sd_IDs, cpID_shift, el_g_cpIDs::$(AMDGPU.ROCArray), x_star::$(AMDGPU.ROCArray),
                   target::$(AMDGPU.ROCArray), itg_hostIDs::$(AMDGPU.ROCArray), elIDs::$(AMDGPU.ROCArray)) =
            $me._Var_Basic(itp_vals, sd_IDs, cpID_shift, el_g_cpIDs, x_star, target, itg_hostIDs, elIDs)
        _Kval_Basic(itp_vals::$(AMDGPU.ROCArray), dual_sd_IDs, base_sd_IDs, vals::$(AMDGPU.ROCArray), sparse_IDs_by_el::$(AMDGPU.ROCArray),
                    sparse_ID_shift, K_val::$(AMDGPU.ROCArray), itg_hostIDs::$(AMDGPU.ROCArray), elIDs::$(AMDGPU.ROCArray)) =
            $me._Kval_Basic(itp_vals, dual_sd_IDs, base_sd_IDs, vals, sparse_IDs_by_el, sparse_ID_shift, K_val, itg_hostIDs, elIDs)
        _Res_Basic(itp_vals::$(AMDGPU.ROCArray), dual_sd_IDs, vals::$(AMDGPU.ROCArray), cpID_shift, el_g_cpIDs::$(AMDGPU.ROCArray),
                   residue::$(AMDGPU.ROCArray), itg_hostIDs::$(AMDGPU.ROCArray), elIDs::$(AMDGPU.ROCArray)) =
            $me._Res_Basic(itp_vals, dual_sd_IDs, vals, cpID_shift, el_g_cpIDs, residue, itg_hostIDs, elIDs)
    end
    return nothing
end

end # module
