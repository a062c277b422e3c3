"""Solvers of the "next" rows -- oracle; test infrastructure only.

cgs! / cgs2!  src/solver/linear_solver/07_CGS.jl:13-52,54-105
"""
from __future__ import annotations

import numpy as np


def cgs(x, A, b, r, *, Pl, tol, maxiter, **_):
    """cgs! (07_CGS.jl:13-52)."""
    from .solvers import mul, normalized_norm

    mul(r, A, x, -1.0)
    r += b
    Pl(r)
    if normalized_norm(r) <= tol:
        return 0
    it = 1
    r0 = r.copy()
    n = r0.size
    rho = 1.0
    u, p, s, v = (np.zeros(n) for _ in range(4))
    while True:
        rhobar = rho
        rho = float(r @ r0)
        beta = rho / rhobar
        s[:] = r + beta * p
        u[:] = s + beta * (p + beta * u)
        mul(v, A, u)
        Pl(v)
        alpha = rho / float(v @ r0)
        p[:] = s - alpha * v
        x += alpha * (p + s)
        mul(r, A, x, -1.0)
        r += b
        Pl(r)
        it += 1
        if normalized_norm(r) <= tol or it > maxiter:
            return it


def cgs2(x, A, b, r, *, Pl, tol, maxiter, seed=0x5EED, shadow=None, **_):
    """cgs2! (07_CGS.jl:54-105)."""
    from .solvers import fem_rand, mul, normalized_norm

    mul(r, A, x, -1.0)
    r += b
    Pl(r)
    if normalized_norm(r) <= tol:
        return 0
    it = 1
    r0 = r.copy()
    n = r0.size
    s0 = fem_rand(seed, 0, n) if shadow is None else shadow
    alpha = alphabar = sigma = sigmabar = 1.0
    u, w, s, v, t, c = (np.zeros(n) for _ in range(6))
    while True:
        rho = float(r @ r0)
        beta = 1 / alphabar * rho / sigma
        v[:] = r + beta * u
        rhobar = float(r @ s0)
        betabar = 1 / alpha * rhobar / sigmabar
        t[:] = r + betabar * s
        w[:] = t + beta * (u + betabar * w)
        mul(c, A, w)
        Pl(c)
        sigma = float(c @ r0)
        alpha = rho / sigma
        s[:] = t - alpha * c
        sigmabar = float(c @ s0)
        alphabar = rhobar / sigmabar
        u[:] = v - alphabar * c
        x += alpha * v + alphabar * s
        mul(r, A, x, -1.0)
        r += b
        Pl(r)
        it += 1
        if normalized_norm(r) <= tol or it > maxiter:
            return it
