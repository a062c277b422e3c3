"""metafem.jl_amd -- host-side mirror of MetaFEM.jl's assembly-and-solve interface over the
MI355X C ABI (libmetafem_mi355x.so).

The reference's host language is Julia, which this image does not have; the
Julia `ccall` shim a maintainer would add is in INTEGRATION.md.  This Python layer is the
runnable stand-in: same entry points, argument meaning and error behaviour as the reference
functions it names, with torch used only for device memory and streams.

  FEM_SpMat_CSR / mul_            src/misc/04_GPU_Utils.jl:120,131
  iterative_Solve                 src/solver/linear_solver/02_Preconditioner.jl:32-76
  Pr_Jacobi_ kernels              :103-148
  make_Brick + mesh_Classical     src/mesh/ref_geometry/201_Helper_TM.jl:36-51; unstructured_mesh/2_Interface.jl:7
  assemble_Global_Variables       src/solver/03_GlobalAssembly.jl:6-37
  update_OneStep                  src/solver/04_Time_Domain.jl:59-80

There is no CPU fallback: importing this package without the built HIP library raises, and
every call goes through the C ABI.
"""
from __future__ import annotations

import ctypes as C
import weakref
import math
from dataclasses import dataclass
from typing import Callable, Optional, Tuple

import torch

from . import _lib
from ._lib import (ElasticityParams, MetaFEMError, OpLayout, SolveOptions, SolveStats, ThermalParams, check, lib)

__all__ = ["Context", "FEM_SpMat_CSR", "mul_", "dot", "nrm2", "axpby_", "FEM_rand", "normalized_norm",
           "iterative_Solve", "Brick", "make_Brick", "ThermalDomain", "MetaFEMError", "SolveStats",
           "cg_", "bicgstabl_GS_", "idrs_", "cgs2_", "FACE_BITS"]

# solver / preconditioner selectors (the reference passes Julia functions: Sv_func! = idrs! ...)
cg_, bicgstabl_GS_, idrs_, cgs2_ = 0, 1, 2, 3
Identity, Pr_Jacobi_, Pr_Jacobi_colnorm_ = 0, 1, 2
Pl_Jacobi_, Pl_Jacobi_rownorm_ = 1, 2  # Pl_func selectors (02_Preconditioner.jl:155-168)

# reference local face ids (ref_geometry/002_Initialization.jl:8): 1 z=0, 2 y=0, 3 x=L, 4 y=L, 5 x=0, 6 z=L
FACE_BITS = {"z0": 1 << 0, "y0": 1 << 1, "x1": 1 << 2, "y1": 1 << 3, "x0": 1 << 4, "z1": 1 << 5}
ALL_FACES = 0x3F


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _need(t: torch.Tensor, dtype, name: str, n: Optional[int] = None) -> None:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise MetaFEMError(f"{name} must be a device (cuda) torch tensor")
    if t.dtype != dtype:
        raise MetaFEMError(f"{name} must have dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise MetaFEMError(f"{name} must be contiguous")
    if n is not None and t.numel() < n:
        raise MetaFEMError(f"{name} has {t.numel()} elements, needs {n}")


class Context:
    """One device + stream binding.  Uses torch's current stream so kernels order with torch ops."""

    def __init__(self, device: int = 0):
        if not torch.cuda.is_available():
            raise MetaFEMError("no HIP device visible: the MI355X backend has no CPU fallback")
        self.device = device
        torch.cuda.set_device(device)
        self._h = C.c_void_p()
        # handles created on this context (patterns, bricks, communicators): closed before the context itself, whatever the order in
        # which the interpreter drops the Python objects (a pattern destroyed after its context would touch freed memory)
        self._children = weakref.WeakSet()
        stream = torch.cuda.current_stream(device).cuda_stream
        check(lib.mfem_context_create(device, C.c_void_p(stream), C.byref(self._h)))

    def use_current_stream(self):
        check(lib.mfem_context_set_stream(self._h, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))

    def sync(self):
        check(lib.mfem_context_sync(self._h))

    def close(self):
        if self._h:
            for child in list(self._children):
                try:
                    child.close()
                except Exception:
                    pass
            lib.mfem_context_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx: Optional[Context] = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(torch.cuda.current_device())
    return _default_ctx


class FEM_SpMat_CSR:
    """CSR pattern handle: FEM_SpMat_CSR(K_J_ptr, K_J, K_vals, dims) (04_GPU_Utils.jl:120) without the
    values (they change every Newton step; pass them per call).  `rowptr` / `colidx` are BORROWED by the library and FROZEN for
    the lifetime of the handle: it keeps what it learnt from them at creation (row blocks, tiles that repeat one column-offset
    list, later the solver layouts).  After rewriting them in place call `replan()`; a different pattern needs a new handle."""

    def __init__(self, rowptr: torch.Tensor, colidx: torch.Tensor, n: int, index_base: int = 0,
                 ctx: Optional[Context] = None, _handle=None):
        self.ctx = ctx or default_context()
        self.rowptr, self.colidx = rowptr, colidx  # keep alive: the library borrows them
        self._owned = _handle is not None
        if _handle is not None:
            self._h = _handle
        else:
            if rowptr.dtype not in (torch.int32, torch.int64):
                raise MetaFEMError("rowptr must be int32 or int64")
            _need(rowptr, rowptr.dtype, "rowptr", n + 1)
            _need(colidx, torch.int32, "colidx")
            self._h = C.c_void_p()
            check(lib.mfem_csr_create(self.ctx._h, n, colidx.numel(), _ptr(rowptr), 64 if rowptr.dtype == torch.int64 else 32,
                                      _ptr(colidx), index_base, C.byref(self._h)))
        self.n = int(lib.mfem_csr_n(self._h))
        self.nnz = int(lib.mfem_csr_nnz(self._h))
        self.index_base = index_base
        self.ctx._children.add(self)

    @property
    def ncols(self) -> int:
        """Columns the pattern addresses = length of x and of a per-column vector (n + ghost entries for a slab pattern)."""
        return int(lib.mfem_csr_ncols(self._h))

    def replan(self):
        """Re-inspect the (rewritten in place) pattern arrays: mfem_csr_replan."""
        check(lib.mfem_csr_replan(self.ctx._h, self._h))

    def spmv_bytes(self):
        """(bytes one mul_ launch moves by design, column entries it reads): mfem_csr_spmv_bytes."""
        b, c = C.c_int64(), C.c_int64()
        check(lib.mfem_csr_spmv_bytes(self.ctx._h, self._h, C.byref(b), C.byref(c)))
        return b.value, c.value

    def close(self):
        if self._h:
            lib.mfem_csr_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def mul_(b: torch.Tensor, A: FEM_SpMat_CSR, vals: torch.Tensor, x: torch.Tensor, alpha: float = 1.0, beta: float = 0.0):
    """mul!(b, A, x, alpha, beta): b = alpha*A*x + beta*b (04_GPU_Utils.jl:131)."""
    _need(vals, torch.float64, "vals", A.nnz)
    _need(x, torch.float64, "x")
    _need(b, torch.float64, "b", A.n)
    check(lib.mfem_spmv_csr(A.ctx._h, A._h, _ptr(vals), _ptr(x), _ptr(b), alpha, beta))
    return b


def dot(x: torch.Tensor, y: torch.Tensor, ctx: Optional[Context] = None) -> float:
    ctx = ctx or default_context()
    _need(x, torch.float64, "x")
    _need(y, torch.float64, "y", x.numel())
    out = C.c_double()
    check(lib.mfem_dot(ctx._h, x.numel(), _ptr(x), _ptr(y), C.byref(out)))
    return out.value


def nrm2(x: torch.Tensor, ctx: Optional[Context] = None) -> float:
    ctx = ctx or default_context()
    _need(x, torch.float64, "x")
    out = C.c_double()
    check(lib.mfem_nrm2(ctx._h, x.numel(), _ptr(x), C.byref(out)))
    return out.value


def normalized_norm(x: torch.Tensor, ctx: Optional[Context] = None) -> float:
    """normalized_norm(x) = norm(x)/sqrt(length(x)) (04_Time_Domain.jl:51)."""
    return nrm2(x, ctx) / math.sqrt(x.numel())


def axpby_(a: float, x: torch.Tensor, b: float, y: torch.Tensor, ctx: Optional[Context] = None):
    ctx = ctx or default_context()
    _need(x, torch.float64, "x")
    _need(y, torch.float64, "y", x.numel())
    check(lib.mfem_axpby(ctx._h, x.numel(), a, _ptr(x), b, _ptr(y)))
    return y


def FEM_rand(n: int, seed: int = 0x5EED, stream_id: int = 0, ctx: Optional[Context] = None) -> torch.Tensor:
    """FEM_rand (04_GPU_Utils.jl:22) with an explicit seed (the reference's stream is unseeded, F9)."""
    ctx = ctx or default_context()
    x = torch.empty(n, dtype=torch.float64, device=f"cuda:{ctx.device}")
    check(lib.mfem_rand(ctx._h, n, seed, stream_id, _ptr(x)))
    return x


def jacobi_by_diagonal(A: FEM_SpMat_CSR, vals: torch.Tensor) -> torch.Tensor:
    """d has A.ncols entries (owned | ghost columns of a slab pattern, the ghost part left at 1: fill it with halo_)."""
    d = torch.ones(A.ncols, dtype=torch.float64, device=vals.device)
    check(lib.mfem_jacobi_by_diagonal(A.ctx._h, A._h, _ptr(vals), _ptr(d)))
    return d


def jacobi2_by_column(A: FEM_SpMat_CSR, vals: torch.Tensor) -> torch.Tensor:
    """Column 2-norms, A.ncols entries.  On a slab pattern with a communicator attached the ghost-column contributions go to
    their owners (mfem_halo_reduce) and the ghost entries come back as the owners' values: the GLOBAL column norms."""
    d = torch.zeros(A.ncols, dtype=torch.float64, device=vals.device)
    check(lib.mfem_jacobi2_by_column(A.ctx._h, A._h, _ptr(vals), _ptr(d)))
    return d


def jacobi_by_row(A: FEM_SpMat_CSR, vals: torch.Tensor) -> torch.Tensor:
    d = torch.zeros(A.n, dtype=torch.float64, device=vals.device)
    check(lib.mfem_jacobi_by_row(A.ctx._h, A._h, _ptr(vals), _ptr(d)))
    return d


def mat_div_jacobi_(A: FEM_SpMat_CSR, vals: torch.Tensor, d: torch.Tensor) -> torch.Tensor:
    check(lib.mfem_mat_div_jacobi(A.ctx._h, A._h, _ptr(vals), _ptr(d)))
    return vals


def iterative_Solve(A: FEM_SpMat_CSR, K_vals: torch.Tensor, residue: torch.Tensor, converge_tol: float, *,
                    Sv_func: int = idrs_, Pr_func: int = Pr_Jacobi_, Pl_func: int = Identity, max_pass: int = 4,
                    maxiter: int = 2000,
                    s: int = 0, seed: int = 0x5EED, check_every: int = 32, fixed_iterations: bool = False,
                    scale_in_place: bool = False, shadow: Optional[torch.Tensor] = None, cg_variant: int = 0
                    ) -> Tuple[torch.Tensor, SolveStats]:
    """iterative_Solve!(globalfield; Sv_func!, Pr_func!, Pl_func, max_pass, maxiter, s) (02_Preconditioner.jl:32-76).
    Pl_func: Identity, Pl_Jacobi_ (:155-168) or Pl_Jacobi_rownorm_ (normalized_by_row = true).
    cg_variant (cg_ only): 0 auto, 1 classic recurrence, 2 single reduction group per iteration (Chronopoulos-Gear), 3 classic
    recurrence carrying the preconditioned residual (one vector stream less per iteration), 4 plain CG on the symmetrically Jacobi-scaled
    matrix (one stream less again; on the mirrored-sweep layout, one rank: the auto choice there, else 3).

    Returns (delta_x, stats); delta_x is a NEW device vector like the reference's return value.
    """
    _need(K_vals, torch.float64, "K_vals", A.nnz)
    _need(residue, torch.float64, "residue", A.n)
    x = torch.empty(A.n, dtype=torch.float64, device=residue.device)
    o = SolveOptions(method=Sv_func, precond=Pr_func, l_or_s=s, maxiter=maxiter, max_pass=max_pass,
                     check_every=check_every, converge_tol=converge_tol, seed=seed,
                     fixed_iterations=1 if fixed_iterations else 0, scale_in_place=1 if scale_in_place else 0,
                     left_precond=Pl_func, cg_variant=cg_variant)
    st = SolveStats()
    if shadow is not None:
        _need(shadow, torch.float64, "shadow")
        check(lib.mfem_solve_set_shadow(A.ctx._h, _ptr(shadow), shadow.numel() // A.n))
    try:
        check(lib.mfem_solve(A.ctx._h, A._h, _ptr(K_vals), _ptr(residue), _ptr(x), C.byref(o), C.byref(st)))
    finally:
        if shadow is not None:
            lib.mfem_solve_set_shadow(A.ctx._h, None, 0)
    return x, st


def assemble_SparseID(controlpoint_IDs: torch.Tensor, ncp: int, n_fields: int = 1, index_base: int = 1,
                      with_slots: bool = True, ctx: Optional[Context] = None):
    """assemble_SparseID! (03_GlobalAssembly.jl:77-140) for unstructured connectivity.
    controlpoint_IDs: int32 device tensor of shape (nel, itp) in C order == [itp, nel] column-major.
    Returns (FEM_SpMat_CSR, sparse_IDs_by_el) with sparse_IDs_by_el of shape (n_fields^2, nel, itp_b, itp_a)."""
    ctx = ctx or default_context()
    _need(controlpoint_IDs, torch.int32, "controlpoint_IDs")
    nel, itp = controlpoint_IDs.shape
    slots = torch.empty((n_fields * n_fields, nel, itp, itp), dtype=torch.int32, device=controlpoint_IDs.device) if with_slots else None
    h = C.c_void_p()
    check(lib.mfem_pattern_build(ctx._h, itp, nel, ncp, _ptr(controlpoint_IDs), index_base, n_fields, C.byref(h), _ptr(slots)))
    A = FEM_SpMat_CSR.__new__(FEM_SpMat_CSR)
    A.ctx, A._h, A._owned, A.index_base = ctx, h, True, 0
    A.n, A.nnz = int(lib.mfem_csr_n(h)), int(lib.mfem_csr_nnz(h))
    A.rowptr = _tensor_from_ptr(lib.mfem_csr_rowptr64(h), A.n + 1, torch.int64, ctx.device, owner=A)
    A.colidx = _tensor_from_ptr(lib.mfem_csr_colidx(h), A.nnz, torch.int32, ctx.device, owner=A)
    ctx._children.add(A)
    return A, slots


def _op_layout(itg, itp, n_sd, n_host, index_base, colour_offsets):
    if colour_offsets is None:
        return OpLayout(itg, itp, n_sd, n_host, index_base, 0, None), None
    arr = (C.c_int64 * len(colour_offsets))(*[int(v) for v in colour_offsets])
    return OpLayout(itg, itp, n_sd, n_host, index_base, len(colour_offsets) - 1, arr), arr


def _Var_Basic(itp_vals, sd, cpID_shift, el_g_cpIDs, x, target, itg_hostIDs, elIDs, *, dims, index_base=1, ctx=None):
    """_Var_Basic(itp_vals, sd_IDs, cpID_shift, el_g_cpIDs, x_star, target, itg_hostIDs, elIDs) (06_FEM_Kernel.jl:1-13).
    dims = (itg, itp, n_sd, n_host); `sd` is the flat 0-based slot of the reference's sd_IDs tuple."""
    ctx = ctx or default_context()
    L, _keep = _op_layout(*dims, index_base, None)
    check(lib.mfem_op_var(ctx._h, C.byref(L), _ptr(itp_vals), sd, cpID_shift, _ptr(el_g_cpIDs), _ptr(x), _ptr(target),
                          _ptr(itg_hostIDs), _ptr(elIDs), elIDs.numel()))
    return target


def _Kval_Basic(itp_vals, dual_sd, base_sd, vals, sparse_IDs_by_el, sparse_ID_shift, K_val, itg_hostIDs, elIDs, *, dims,
                index_base=1, colour_offsets=None, ctx=None):
    """_Kval_Basic(...) (06_FEM_Kernel.jl:28-45); colour_offsets = None -> FP64 atomics like the reference."""
    ctx = ctx or default_context()
    L, _keep = _op_layout(*dims, index_base, colour_offsets)
    check(lib.mfem_op_kval(ctx._h, C.byref(L), _ptr(itp_vals), dual_sd, base_sd, _ptr(vals), _ptr(sparse_IDs_by_el),
                           sparse_ID_shift, _ptr(K_val), _ptr(itg_hostIDs), _ptr(elIDs), elIDs.numel()))
    return K_val


def _Res_Basic(itp_vals, dual_sd, vals, cpID_shift, el_g_cpIDs, residue, itg_hostIDs, elIDs, *, dims, index_base=1,
               colour_offsets=None, ctx=None):
    """_Res_Basic(...) (06_FEM_Kernel.jl:65-79)."""
    ctx = ctx or default_context()
    L, _keep = _op_layout(*dims, index_base, colour_offsets)
    check(lib.mfem_op_res(ctx._h, C.byref(L), _ptr(itp_vals), dual_sd, _ptr(vals), cpID_shift, _ptr(el_g_cpIDs),
                          _ptr(residue), _ptr(itg_hostIDs), _ptr(elIDs), elIDs.numel()))
    return residue


class Brick:
    """make_Brick(x, n, :CUBE) + mesh_Classical(itp_type=:Lagrange, itp_order, itg_order) + update_Mesh
    on device (201_Helper_TM.jl:36-51; 2_Interface.jl:7,98-108)."""

    def __init__(self, x: Tuple[float, float, float], n: Tuple[int, int, int], itp_order: int = 1, itg_order: int = 3,
                 ctx: Optional[Context] = None):
        self.ctx = ctx or default_context()
        self.x, self.n, self.itp_order, self.itg_order = tuple(x), tuple(n), itp_order, itg_order
        self._h = C.c_void_p()
        check(lib.mfem_brick_create(self.ctx._h, n[0], n[1], n[2], x[0], x[1], x[2], itp_order, itg_order, C.byref(self._h)))
        self.ncp = int(lib.mfem_brick_num_controlpoints(self._h))
        self.nel = int(lib.mfem_brick_num_elements(self._h))
        self.m = tuple(itp_order * ni + 1 for ni in n)
        self.slab = (0, self.m[0])
        self.ctx._children.add(self)

    def set_slab(self, plane_lo: int, plane_hi: int):
        check(lib.mfem_brick_set_slab(self._h, plane_lo, plane_hi))
        self.slab = (plane_lo, plane_hi)

    @property
    def n_owned(self) -> int:
        return (self.slab[1] - self.slab[0]) * self.m[1] * self.m[2]

    def coords_view(self, d: int) -> torch.Tensor:
        """Device view of controlpoints.x{d+1} (library-owned memory) as a torch tensor."""
        gw = self.itp_order  # ghost planes per side
        clo = max(self.slab[0] - gw, 0) if self.slab != (0, self.m[0]) else 0
        chi = min(self.slab[1] + gw, self.m[0]) if self.slab != (0, self.m[0]) else self.m[0]
        ncoord = (chi - clo) * self.m[1] * self.m[2]
        ptr = lib.mfem_brick_coords(self._h, d)
        return _tensor_from_ptr(ptr, ncoord, torch.float64, self.ctx.device, owner=self)

    def pattern(self, n_fields: int = 1) -> FEM_SpMat_CSR:
        """assemble_SparseID! (03_GlobalAssembly.jl:77-140) -> row-sorted CSR, K_val_ids = identity."""
        h = C.c_void_p()
        check(lib.mfem_brick_pattern(self.ctx._h, self._h, n_fields, C.byref(h)))
        n = int(lib.mfem_csr_n(h))
        nnz = int(lib.mfem_csr_nnz(h))
        A = FEM_SpMat_CSR.__new__(FEM_SpMat_CSR)
        A.ctx, A._h, A._owned, A.n, A.nnz, A.index_base = self.ctx, h, True, n, nnz, 0
        A.rowptr = _tensor_from_ptr(lib.mfem_csr_rowptr64(h), n + 1, torch.int64, self.ctx.device, owner=A)
        A.colidx = _tensor_from_ptr(lib.mfem_csr_colidx(h), nnz, torch.int32, self.ctx.device, owner=A)
        self.ctx._children.add(A)
        return A

    def assemble_thermal(self, A: FEM_SpMat_CSR, k: float, h: float = 0.0, Tenv: float = 0.0,
                         robin_faces: int = 0, out: Optional[torch.Tensor] = None, *, fixed_faces: int = 0,
                         h_penalty: float = 0.0, Tw: float = 0.0) -> torch.Tensor:
        """K_linear_func of -k*Bilinear(T{;i},T{;i}) + h*Bilinear(T, Tenv - T) on robin_faces
        (examples/thermal_conduction/3D_Script.jl:30-31) + the weakly imposed Dirichlet faces
        h_penalty*Bilinear(T, Tw - T) + k*Bilinear(T, n{i}*T{;i}) on fixed_faces (2D_Script.jl:58; K is nonsymmetric then)."""
        vals = out if out is not None else torch.empty(A.nnz, dtype=torch.float64, device=f"cuda:{self.ctx.device}")
        p = ThermalParams(k, h, Tenv, robin_faces, fixed_faces, h_penalty, Tw)
        check(lib.mfem_brick_assemble_thermal(self.ctx._h, self._h, A._h, C.byref(p), _ptr(vals)))
        return vals

    def residual_thermal(self, x_star: torch.Tensor, k: float, h: float = 0.0, Tenv: float = 0.0, robin_faces: int = 0,
                         s: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, *, fixed_faces: int = 0,
                         h_penalty: float = 0.0, Tw: float = 0.0) -> torch.Tensor:
        _need(x_star, torch.float64, "x_star", self.n_owned)
        res = out if out is not None else torch.empty(self.n_owned, dtype=torch.float64, device=x_star.device)
        p = ThermalParams(k, h, Tenv, robin_faces, fixed_faces, h_penalty, Tw)
        check(lib.mfem_brick_residual_thermal(self.ctx._h, self._h, C.byref(p), _ptr(x_star), _ptr(s), _ptr(res)))
        return res

    def assemble_elasticity(self, A: FEM_SpMat_CSR, lam: float, mu: float, tau: float = 0.0, penalty_faces: int = 0,
                            out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """K_linear_func of -Bilinear(eps{i,j}, sigma{i,j}) + tau*Bilinear(d{i}, dw{i} - d{i}), dw = 0
        (examples/linear_elasticity/cantilever/3D_Script.jl:52-60): the reference's 21 + 3 _Kval_Basic launches."""
        vals = out if out is not None else torch.empty(A.nnz, dtype=torch.float64, device=f"cuda:{self.ctx.device}")
        p = ElasticityParams(lam, mu, tau, penalty_faces, 0, (C.c_double * 6)(*([0.0] * 6)))
        check(lib.mfem_brick_assemble_elasticity(self.ctx._h, self._h, A._h, C.byref(p), _ptr(vals)))
        return vals

    def residual_elasticity(self, x_star: torch.Tensor, lam: float, mu: float, tau: float = 0.0, penalty_faces: int = 0,
                            traction_faces: int = 0, sig=(0.0,) * 6, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """sig = constant symmetric tensor (11, 22, 33, 23, 13, 12) of Bilinear(d{i}, sig{i,j}*n{j}) (:61)."""
        _need(x_star, torch.float64, "x_star", 3 * self.n_owned)
        res = out if out is not None else torch.empty(3 * self.n_owned, dtype=torch.float64, device=x_star.device)
        p = ElasticityParams(lam, mu, tau, penalty_faces, traction_faces, (C.c_double * 6)(*sig))
        check(lib.mfem_brick_residual_elasticity(self.ctx._h, self._h, C.byref(p), _ptr(x_star), _ptr(res)))
        return res

    def close(self):
        if self._h:
            lib.mfem_brick_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_Brick(x, n, itp_order: int = 1, itg_order: int = 3, ctx: Optional[Context] = None) -> Brick:
    return Brick(x, n, itp_order, itg_order, ctx)


class _PtrHolder:
    """__cuda_array_interface__ shim so torch can wrap library-owned device memory without copying."""

    def __init__(self, ptr: int, n: int, typestr: str, owner):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}
        self._owner = owner


def _tensor_from_ptr(ptr, n: int, dtype, device: int, owner=None) -> torch.Tensor:
    if n == 0 or not ptr:
        return torch.empty(0, dtype=dtype, device=f"cuda:{device}")
    typestr = {torch.float64: "<f8", torch.int64: "<i8", torch.int32: "<i4"}[dtype]
    return torch.as_tensor(_PtrHolder(int(ptr), n, typestr, owner), device=f"cuda:{device}")


@dataclass
class GeneralAlpha:
    """Static problems only here: max_time_level = 0, K_params = [1], beta = [1] (04_Time_Domain.jl:1-18)."""
    alpha_params: Tuple[float, ...] = (1.0, 1.0, 1.0)
    gamma_params: Tuple[float, ...] = (1.0, 1.0)


class ThermalDomain:
    """FEM_Domain + GlobalField for the thermal weak form of examples/thermal_conduction/3D_Script.jl:30-31
    on a structured brick: the two generated closures are the fused HIP kernels, update_OneStep is the
    reference's Newton driver (04_Time_Domain.jl:59-80)."""

    def __init__(self, brick: Brick, k: float, h: float, Tenv: float, robin_faces: int = ALL_FACES, *, fixed_faces: int = 0,
                 h_penalty: float = 0.0, Tw: float = 0.0):
        self.brick, self.k, self.h, self.Tenv, self.robin_faces = brick, k, h, Tenv, robin_faces
        # weakly imposed Dirichlet faces (fix_boundary of examples/thermal_conduction/2D_Script.jl:58)
        self.fixed = dict(fixed_faces=fixed_faces, h_penalty=h_penalty, Tw=Tw)
        dev = f"cuda:{brick.ctx.device}"
        self.A = brick.pattern(1)  # assemble_Global_Variables! -> assemble_SparseID!
        n = self.A.n
        self.basicfield_size = n
        self.x = torch.zeros(n, dtype=torch.float64, device=dev)
        self.dx = torch.zeros(n, dtype=torch.float64, device=dev)
        self.x_star = torch.zeros(n, dtype=torch.float64, device=dev)
        self.residue = torch.zeros(n, dtype=torch.float64, device=dev)
        self.K_linear = torch.zeros(self.A.nnz, dtype=torch.float64, device=dev)
        self.K_total = self.K_linear  # no nonlinear gradients in this form: K_total .= K_linear is an alias
        self.s = torch.zeros(n, dtype=torch.float64, device=dev)  # controlpoints.s
        self.converge_tol = 1e-6
        self.linear_solver: Callable = lambda gf: iterative_Solve(gf.A, gf.K_total, gf.residue, gf.converge_tol,
                                                                   Sv_func=idrs_, maxiter=2000, max_pass=10, s=8)[0]
        self.history = []

    def K_linear_func(self):
        self.brick.assemble_thermal(self.A, self.k, self.h, self.Tenv, self.robin_faces, out=self.K_linear, **self.fixed)

    def K_nonlinear_func(self):
        self.brick.residual_thermal(self.x_star, self.k, self.h, self.Tenv, self.robin_faces, s=self.s, out=self.residue, **self.fixed)

    def update_OneStep(self, max_iter: int = 4):
        self.dx.zero_()  # initialize_dx! with max_time_level = 0
        self.K_linear_func()
        counter = -1
        self.history = []
        while True:
            torch.add(self.x, self.dx, out=self.x_star)  # update_x_star!
            self.K_nonlinear_func()
            res = normalized_norm(self.residue, self.brick.ctx)
            counter += 1
            self.history.append(res)
            if res < self.converge_tol or counter > max_iter:
                break
            delta_x = self.linear_solver(self)
            self.dx -= delta_x  # update_dx!(-delta_x), beta = 1
        self.x += self.dx
        return self.history
