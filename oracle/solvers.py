"""Linear solvers (oracle; test infrastructure only).

Restates src/solver/linear_solver/:
  iterative_Solve!      02_Preconditioner.jl:32-76
  Pr_Jacobi! (+kernels) 02_Preconditioner.jl:103-148, _JacobiP :89-96
  Pl_Jacobi             02_Preconditioner.jl:155-177
  bicgstabl_GS!         03_BiCGstabl.jl:18-96
  idrs! / modify_Omega  04_IDRs.jl:1-8,26-95
  cgs2!                 07_CGS.jl:54-105
  solver_LU_CPU         01_Direct_Solver.jl:10-24  (== scipy spsolve)
  normalized_norm       ../04_Time_Domain.jl:51
  mul!                  src/misc/04_GPU_Utils.jl:131   (CUSPARSE mv!, b = alpha*A*x + beta*b)
plus a Jacobi-preconditioned CG which the reference does NOT have;
CG is only valid for the symmetric (Robin / penalty) problems.

The reference draws unseeded ``CUDA.Random.rand!`` shadow vectors (F9); here they come
from ``fem_rand`` -- a counter-based generator restated bit-for-bit by the HIP library
(metafem.jl_amd/csrc/rng.h) so iterates, not only converged answers, can be compared.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, Optional

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
from scipy.linalg import solve_triangular

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def fem_rand(seed: int, stream: int, n: int) -> np.ndarray:
    """U[0,1) doubles: splitmix64 finaliser of (seed, stream, index); mirrors csrc/rng.h."""
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        z = (np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * (idx + np.uint64(1))
             + np.uint64(0xD1B54A32D192ED03) * np.uint64(stream + 1))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def fem_sign(seed: int, k: int, n: int) -> np.ndarray:
    """+-1 doubles: bit k of the sign word of row i (csrc/rng.h::mfem_sign_word, bit for bit).  The shadow vectors P of idrs! are arbitrary random
    vectors in the reference (04_IDRs.jl:35: FEM_rand, unseeded); the product generates them as signs so that P' g needs no stored P (round 6), and this
    restatement draws the same ones so that iterates can be compared step by step."""
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        z = (np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * (idx + np.uint64(1))
             + np.uint64(0xD1B54A32D192ED03) * np.uint64(0x5149 + 1))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return np.where((z >> np.uint64(k)) & np.uint64(1), 1.0, -1.0)


def normalized_norm(x: np.ndarray) -> float:
    """04_Time_Domain.jl:51."""
    return float(np.linalg.norm(x) / math.sqrt(x.size))


def csr(rowptr, colidx, vals, n) -> sp.csr_matrix:
    return sp.csr_matrix((vals, colidx, rowptr), shape=(n, n))


def mul(b: np.ndarray, A: sp.csr_matrix, x: np.ndarray, alpha: float = 1.0, beta: float = 0.0) -> np.ndarray:
    """mul!(b, A, x, alpha, beta) (04_GPU_Utils.jl:131): b <- alpha*A*x + beta*b, in place."""
    y = A @ x
    if beta == 0.0:
        b[:] = alpha * y
    else:
        b[:] = alpha * y + beta * b
    return b


# -- preconditioners ---------------------------------------------------------------
class Identity:
    def __call__(self, b):
        return b


class JacobiP:
    """_JacobiP (02_Preconditioner.jl:89-96): b ./= jac_vec (in place)."""

    def __init__(self, jac_vec):
        self.jac_vec = jac_vec

    def __call__(self, b):
        b /= self.jac_vec
        return b


def jacobi_by_diagonal(A: sp.csr_matrix) -> np.ndarray:
    """Jacobi_By_Diagonal (:122-130): d_i = |K_ii|, rows without a stored diagonal keep 1."""
    d = np.ones(A.shape[0])
    rows = np.repeat(np.arange(A.shape[0]), np.diff(A.indptr))
    on = rows == A.indices
    d[rows[on]] = np.abs(A.data[on])
    return d


def pr_jacobi(A: sp.csr_matrix, normalized_by_column: bool = False) -> JacobiP:
    """Pr_Jacobi! (:103-120): scales the COLUMNS of A in place, returns x -> x ./ d."""
    if normalized_by_column:
        d = np.zeros(A.shape[1])
        np.add.at(d, A.indices, A.data ** 2)  # Jacobi2_By_Colomn :132-139
        d **= 0.5
    else:
        d = jacobi_by_diagonal(A)
    A.data /= d[A.indices]  # Mat_Div_Jacobi :141-148
    return JacobiP(d)


def pl_jacobi(A: sp.csr_matrix, normalized_by_row: bool = False) -> JacobiP:
    """Pl_Jacobi (:155-168)."""
    if normalized_by_row:
        d = np.sqrt(np.asarray(A.multiply(A).sum(axis=1)).ravel())  # Jacobi_By_Row :170-177
    else:
        d = jacobi_by_diagonal(A)
    return JacobiP(d)


# -- Krylov bodies --------------------------------------------------------------------
def bicgstabl_gs(x, A, b, r, *, Pl=Identity(), tol, maxiter, s=2, seed=0x5EED, shadow=None, **_):
    """bicgstabl_GS! (03_BiCGstabl.jl:18-96)."""
    mul(r, A, x, -1.0)
    r += b
    Pl(r)
    if normalized_norm(r) <= tol:
        return 0
    it = 1
    n = b.size
    g, gp, gpp, sig = np.zeros(s), np.zeros(s), np.zeros(s), np.zeros(s)
    tau = np.zeros((s, s))
    omega = rho0 = 1.0
    alpha = 0.0
    r_shadow = fem_rand(seed, 0, n) if shadow is None else shadow
    R = [r] + [np.zeros(n) for _ in range(s)]
    U = [np.zeros(n) for _ in range(s + 1)]
    while True:
        rho0 *= -omega
        for j in range(s):
            rho1 = float(r_shadow @ R[j])
            beta = alpha * rho1 / rho0
            rho0 = rho1
            for i in range(j + 1):
                U[i][:] = R[i] - beta * U[i]
            mul(U[j + 1], A, U[j])
            Pl(U[j + 1])
            alpha = rho0 / float(r_shadow @ U[j + 1])
            for i in range(j + 1):
                R[i] -= alpha * U[i + 1]
            mul(R[j + 1], A, R[j])
            Pl(R[j + 1])
            x += alpha * U[0]
        for j in range(s):
            for i in range(j):
                tau[i, j] = float(R[i + 1] @ R[j + 1]) / sig[i]
                R[j + 1] -= tau[i, j] * R[i + 1]
            sig[j] = float(R[j + 1] @ R[j + 1])
            gp[j] = float(R[0] @ R[j + 1]) / sig[j]
        g[s - 1] = gp[s - 1]
        omega = g[s - 1]
        for j in range(s - 2, -1, -1):
            g[j] = gp[j] - float(tau[j, j + 1:s] @ g[j + 1:s])
        for j in range(s - 1):
            gpp[j] = g[j + 1] + float(tau[j, j + 1:s - 1] @ g[j + 2:s])
        x += g[0] * R[0]
        R[0] -= gp[s - 1] * R[s]
        U[0] -= g[s - 1] * U[s]
        for j in range(s - 1):
            U[0] -= g[j] * U[j + 1]
            x += gpp[j] * R[j + 1]
            R[0] -= gp[j] * R[j + 1]
        it += s
        if normalized_norm(R[0]) <= tol or it >= maxiter:
            return it


def modify_omega(v1, v2):
    """04_IDRs.jl:1-8."""
    angle = math.sqrt(2.0) / 2
    n1, n2 = np.linalg.norm(v1), np.linalg.norm(v2)
    d = float(v1 @ v2)
    rho = abs(d / (n1 * n2))
    omega = d / (n1 * n1)
    return omega * angle / rho if rho < angle else omega


def idrs(x, A, b, r, *, Pl=Identity(), tol, maxiter, s=4, seed=0x5EED, shadow=None, shadow_kind="sign", **_):
    """idrs! (04_IDRs.jl:26-95).  P (:35, `FEM_rand`, unseeded): `shadow` if given, else seeded +-1 vectors (fem_sign; shadow_kind = "uniform": U[0,1)
    vectors from fem_rand, the reference's distribution)."""
    mul(r, A, x, -1.0)
    r += b
    Pl(r)
    if normalized_norm(r) <= tol:
        return 0
    it = 1
    n = b.size
    Ar = np.zeros(n)
    P = shadow if shadow is not None else [fem_sign(seed, k, n) if shadow_kind == "sign" else fem_rand(seed, k, n) for k in range(s)]
    U = [np.zeros(n) for _ in range(s)]
    G = [np.zeros(n) for _ in range(s)]
    Q, V = np.zeros(n), np.zeros(n)
    M, f = np.eye(s), np.zeros(s)
    omega = 1.0
    while True:
        for i in range(s):
            f[i] = float(P[i] @ r)
        for k in range(s):
            c = solve_triangular(M[k:, k:], f[k:], lower=True)
            V[:] = c[0] * G[k]
            Q[:] = c[0] * U[k]
            for i in range(k + 1, s):
                V += c[i - k] * G[i]
                Q += c[i - k] * U[i]
            V[:] = r - V
            U[k][:] = Q + omega * V
            mul(G[k], A, U[k])
            Pl(G[k])
            for i in range(k):
                alpha = float(P[i] @ G[k]) / M[i, i]
                G[k] -= alpha * G[i]
                U[k] -= alpha * U[i]
            for i in range(k, s):
                M[i, k] = float(P[i] @ G[k])
            beta = f[k] / M[k, k]
            x += beta * U[k]
            r -= beta * G[k]
            if normalized_norm(r) <= tol or it >= maxiter:
                return it
            f[k + 1:] -= beta * M[k + 1:, k]
            it += 1
        mul(Ar, A, r)
        Pl(Ar)
        omega = modify_omega(Ar, r)
        x += omega * r
        r -= omega * Ar
        if normalized_norm(r) <= tol or it >= maxiter:
            return it
        it += 1


def cg(x, A, b, r, *, Pl=Identity(), tol, maxiter, M_diag: Optional[np.ndarray] = None, **_):
    """Preconditioned CG (NOT in the reference, F5).  ``M_diag`` = |diag(A)| (Jacobi).

    Runs unchanged on a symmetric NEGATIVE definite A (the Robin thermal K): the
    iterates equal those of CG on (-A, -b).  Stop rule = the reference's:
    normalized_norm(r) <= tol or iter >= maxiter; returns the iteration count.
    """
    mul(r, A, x, -1.0)
    r += b
    if normalized_norm(r) <= tol:
        return 0
    dinv = None if M_diag is None else 1.0 / M_diag
    z = r.copy() if dinv is None else r * dinv
    p = z.copy()
    rz = float(r @ z)
    Ap = np.zeros_like(b)
    it = 0
    while True:
        mul(Ap, A, p)
        alpha = rz / float(p @ Ap)
        x += alpha * p
        r -= alpha * Ap
        it += 1
        if normalized_norm(r) <= tol or it >= maxiter:
            return it
        z = r if dinv is None else r * dinv
        rz_new = float(r @ z)
        beta = rz_new / rz
        rz = rz_new
        p[:] = z + beta * p


def cgs2(x, A, b, r, *, Pl=Identity(), tol, maxiter, seed=0x5EED, shadow=None, **_):
    """cgs2! (07_CGS.jl:54-105): CGS with two shadow vectors (Fokkema/Sleijpen/van der Vorst)."""
    from . import solvers_next

    return solvers_next.cgs2(x, A, b, r, Pl=Pl, tol=tol, maxiter=maxiter, seed=seed, shadow=shadow)


# -- wrapper ----------------------------------------------------------------------------
@dataclass
class SolveInfo:
    passes: int = 0
    iters: int = 0
    res: float = float("nan")


def iterative_solve(rowptr, colidx, K_vals, residue, converge_tol, *, Sv_func: Callable = idrs,
                    Pr_func: Optional[Callable] = pr_jacobi, Pl_func: Optional[Callable] = None,
                    max_pass: int = 4, info: Optional[SolveInfo] = None, **kwargs) -> np.ndarray:
    """iterative_Solve! (02_Preconditioner.jl:32-76).  ``K_vals`` is copied (the
    reference gathers ``K_total[K_val_ids]`` into a fresh array, :35)."""
    n = residue.size
    A = csr(rowptr, colidx, np.array(K_vals, dtype=np.float64, copy=True), n)
    Pr = Pr_func(A) if Pr_func is not None else Identity()
    Pl = Pl_func(A) if Pl_func is not None else Identity()
    b = residue
    r = b.copy()
    x = np.zeros(n)
    pass_number, tol_factor, total = 1, 1.0, 0
    while True:
        total += Sv_func(x, A, b, r, Pl=Pl, tol=tol_factor * converge_tol, **kwargs)
        mul(r, A, x, -1.0)
        r += b
        res = normalized_norm(r)
        if Pl_func is not None:
            pres = normalized_norm(Pl(r.copy()))
            tol_factor = min(pres / res, 1.0)
        if res < converge_tol or pass_number >= max_pass:
            break
        pass_number += 1
    if info is not None:
        info.passes, info.iters, info.res = pass_number, total, res
    return Pr(x)


def solver_lu_cpu(rowptr, colidx, K_vals, residue) -> np.ndarray:
    """solver_LU_CPU (01_Direct_Solver.jl:10-24): SparseArrays ``lu`` == scipy ``spsolve``."""
    A = csr(rowptr, colidx, np.asarray(K_vals, dtype=np.float64), residue.size).tocsc()
    return spla.spsolve(A, residue)


def solve_cg_jacobi(rowptr, colidx, K_vals, residue, converge_tol, maxiter, max_pass=1,
                    info: Optional[SolveInfo] = None) -> np.ndarray:
    """The added CG path: standard PCG with M = |diag K| on the UNSCALED matrix, wrapped in
    the reference's restart/true-residual loop (02_Preconditioner.jl:50-73)."""
    n = residue.size
    A = csr(rowptr, colidx, np.asarray(K_vals, dtype=np.float64), n)
    d = jacobi_by_diagonal(A)
    b = residue
    r = b.copy()
    x = np.zeros(n)
    pass_number, total = 1, 0
    while True:
        total += cg(x, A, b, r, tol=converge_tol, maxiter=maxiter, M_diag=d)
        mul(r, A, x, -1.0)
        r += b
        res = normalized_norm(r)
        if res < converge_tol or pass_number >= max_pass:
            break
        pass_number += 1
    if info is not None:
        info.passes, info.iters, info.res = pass_number, total, res
    return x
