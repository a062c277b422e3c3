"""Hand-derived AssembleWeakform term lists for the benchmark weak forms
(oracle; test infrastructure only).

These are what ``build_WeakForm`` + ``construct_AssembleWeakform`` produce
(src/symbolics/10_WeakForm.jl:72-124; src/solver/02_LocalAssembly.jl:30-58) for the
scripts cited per function -- derivation in SURVEY.md §3.4.  Field positions follow
``basic_vars`` sorted by symbol (02_LocalAssembly.jl:93-94).
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np

from .fem import AssembleWeakform, GradTerm, ResTerm


# -- thermal conduction ------------------------------------------------------------------
def thermal_domain(dim: int, k: float, alpha: float = 0.0, Tenv: float = 0.0, C: float = 0.0) -> AssembleWeakform:
    """heat_dissipation = -k*Bilinear(T{;i},T{;i}) + Bilinear(T, s + alpha*(Tenv - T))
    (examples/thermal_conduction/2D_Script.jl:56, 3D_Script.jl:30); C != 0 adds the transient term
    -C*Bilinear(T, T{;t}) of 3D_Script_Dynamics.jl:32 (needs max_time_level >= 1)."""
    wf = AssembleWeakform()
    if C != 0.0:
        wf.inner_vars.append(("T_t", 0, 0, 1))
        wf.residues.append(ResTerm(0, 0, lambda env: -C * env["T_t"]))
        wf.linear_gradients.append(GradTerm(0, 0, 0, 0, lambda env: -C, td_order=1))
    for d in range(dim):
        wf.inner_vars.append((f"T_{d}", 0, 1 + d, 0))
        wf.residues.append(ResTerm(0, 1 + d, lambda env, d=d: -k * env[f"T_{d}"]))
        wf.linear_gradients.append(GradTerm(0, 1 + d, 0, 1 + d, lambda env: -k))
    wf.cp_ext_vars.append(("s", "s", 0))
    if alpha != 0.0:  # eval_Constant! drops the term for alpha == 0 (10_WeakForm.jl:2-17)
        wf.inner_vars.append(("T", 0, 0, 0))
        wf.residues.append(ResTerm(0, 0, lambda env: env["s"] + alpha * (Tenv - env["T"])))
        wf.linear_gradients.append(GradTerm(0, 0, 0, 0, lambda env: -alpha))
    else:
        wf.residues.append(ResTerm(0, 0, lambda env: env["s"]))
    return wf


def thermal_convection(h: float, Tenv: float, em: float = 0.0, sigma_b: float = 0.0) -> AssembleWeakform:
    """conv_rad_boundary = h*Bilinear(T, Tenv - T) + em*sigma*Bilinear(T, Tenv^4 - T^4)
    (2D_Script.jl:57; 3D_Script.jl:31 is the em = 0 case)."""
    wf = AssembleWeakform(inner_vars=[("T", 0, 0, 0)])
    wf.residues.append(ResTerm(0, 0, lambda env: h * (Tenv - env["T"])))
    wf.linear_gradients.append(GradTerm(0, 0, 0, 0, lambda env: -h))
    if em != 0.0:
        c = em * sigma_b
        wf.residues.append(ResTerm(0, 0, lambda env: c * (Tenv ** 4 - env["T"] ** 4)))
        wf.nonlinear_gradients.append(GradTerm(0, 0, 0, 0, lambda env: -4.0 * c * env["T"] ** 3))
    return wf


def thermal_fixed(dim: int, h_penalty: float, Tw: float, k: float) -> AssembleWeakform:
    """fix_boundary = h_penalty*Bilinear(T, Tw - T) + k*Bilinear(T, n{i}*T{;i}) (2D_Script.jl:58)."""
    wf = AssembleWeakform(inner_vars=[("T", 0, 0, 0)])
    wf.residues.append(ResTerm(0, 0, lambda env: h_penalty * (Tw - env["T"])))
    wf.linear_gradients.append(GradTerm(0, 0, 0, 0, lambda env: -h_penalty))
    for d in range(dim):
        wf.inner_vars.append((f"T_{d}", 0, 1 + d, 0))
        wf.normals.append((f"n{d}", d))
        wf.residues.append(ResTerm(0, 0, lambda env, d=d: k * env[f"n{d}"] * env[f"T_{d}"]))
        wf.linear_gradients.append(GradTerm(0, 0, 0, 1 + d, lambda env, d=d: k * env[f"n{d}"]))
    return wf


# -- linear elasticity ---------------------------------------------------------------------
def elasticity_domain(dim: int, lam: float, mu: float) -> AssembleWeakform:
    """Elastrostatic_Domain = -Bilinear(eps{i,j}, sigma{i,j}), sigma = lam*delta*eps_mm + 2*mu*eps
    (examples/linear_elasticity/cantilever/3D_Script.jl:52-57).  dim^2 dual words d{i;j};
    3 gradient terms per diagonal dual, 2 per off-diagonal dual (21 launches in 3-D)."""
    wf = AssembleWeakform()
    for i in range(dim):
        for j in range(dim):
            wf.inner_vars.append((f"d{i}_{j}", i, 1 + j, 0))

    def sigma(env, i, j):
        s = mu * (env[f"d{i}_{j}"] + env[f"d{j}_{i}"])
        if i == j:
            s = s + lam * sum(env[f"d{m}_{m}"] for m in range(dim))
        return s

    for i in range(dim):
        for j in range(dim):
            wf.residues.append(ResTerm(i, 1 + j, lambda env, i=i, j=j: -sigma(env, i, j)))
            for kk in range(dim):
                for l in range(dim):
                    c = (lam if (i == j and kk == l) else 0.0) + mu * ((i == kk and j == l) + (i == l and j == kk))
                    if c != 0.0:
                        wf.linear_gradients.append(GradTerm(i, 1 + j, kk, 1 + l, lambda env, c=c: -c))
    return wf


def elasticity_inertia(dim: int, rho: float, c: float = 0.0) -> AssembleWeakform:
    """-Bilinear(d{i}, rho*(c*d{i;t} + d{i;t,t})): the damping + inertia term of
    examples/hypo_elastic_plasticity/J2Plasticity.jl:59 in the sign convention of Elastrostatic_Domain above
    (needs max_time_level = 2).  Returned separately; merge into a domain weak form with `merge`."""
    wf = AssembleWeakform()
    for i in range(dim):
        wf.inner_vars.append((f"d{i}_tt", i, 0, 2))
        wf.linear_gradients.append(GradTerm(i, 0, i, 0, lambda env: -rho, td_order=2))
        if c != 0.0:
            wf.inner_vars.append((f"d{i}_t", i, 0, 1))
            wf.linear_gradients.append(GradTerm(i, 0, i, 0, lambda env: -rho * c, td_order=1))
            wf.residues.append(ResTerm(i, 0, lambda env, i=i: -rho * (c * env[f"d{i}_t"] + env[f"d{i}_tt"])))
        else:
            wf.residues.append(ResTerm(i, 0, lambda env, i=i: -rho * env[f"d{i}_tt"]))
    return wf


def merge(*wfs: AssembleWeakform) -> AssembleWeakform:
    """Sum of weak forms on the same integration domain (Bilinear + Bilinear)."""
    out = AssembleWeakform()
    for wf in wfs:
        for a in ("inner_vars", "cp_ext_vars", "normals"):
            for item in getattr(wf, a):
                if item not in getattr(out, a):
                    getattr(out, a).append(item)
        out.residues += wf.residues
        out.linear_gradients += wf.linear_gradients
        out.nonlinear_gradients += wf.nonlinear_gradients
    return out


def elasticity_penalty(dim: int, tau: float, wall_syms: Optional[Sequence[str]] = None) -> AssembleWeakform:
    """WF_fixed_bdy = tau*Bilinear(d{i}, dw{i} - d{i}) (cantilever/3D_Script.jl:60)."""
    wf = AssembleWeakform()
    for i in range(dim):
        wf.inner_vars.append((f"d{i}", i, 0, 0))
        if wall_syms is not None:
            wf.cp_ext_vars.append((f"dw{i}", wall_syms[i], 0))
            wf.residues.append(ResTerm(i, 0, lambda env, i=i: tau * (env[f"dw{i}"] - env[f"d{i}"])))
        else:
            wf.residues.append(ResTerm(i, 0, lambda env, i=i: tau * (0.0 - env[f"d{i}"])))
        wf.linear_gradients.append(GradTerm(i, 0, i, 0, lambda env: -tau))
    return wf


def elasticity_traction(dim: int, sig) -> AssembleWeakform:
    """WF_right_bdy = Bilinear(d{i}, sigl{i,j}*n{j}) with a constant symmetric tensor sig[i][j]
    (cantilever/3D_Script.jl:61; the script stores sigl in control-point arrays filled with
    constants, :109-111 -- interpolating a constant field returns the constant)."""
    wf = AssembleWeakform()
    for j in range(dim):
        wf.normals.append((f"n{j}", j))
    for i in range(dim):
        if any(sig[i][j] != 0.0 for j in range(dim)):
            wf.residues.append(ResTerm(i, 0, lambda env, i=i: sum(sig[i][j] * env[f"n{j}"] for j in range(dim))))
    return wf


# -- incompressible flow, lid-driven cavity ---------------------------------------------------------
def cavity_weakforms(rho: float, mu: float, tau_b: float, dim: int = 2):
    """examples/incompressible_flow/lid_driven_cavity_flow/2D_Script.jl:47-72 -- SUPG/PSPG stabilised
    Navier-Stokes with weakly imposed (Nitsche-type) wall and lid conditions.  Fields sorted by symbol:
    p, u1, u2 (positions 0, 1, 2).  Second derivatives u{i;m,m} are dropped (explicit_max_sd_order = 1, :78).
    Returns (domain, boundary_fix, boundary_top) AssembleWeakforms."""
    from . import symform

    fields = ["p"] + [f"u{i + 1}" for i in range(dim)]
    R = range(dim)

    def build(kind):
        W = symform.Words(dim, fields)
        u = [W.val(f"u{i + 1}") for i in R]
        du = [[W.d(f"u{i + 1}", j) for j in R] for i in R]  # du[i][j] = u{i;j}
        p, dp = W.val("p"), [W.d("p", j) for j in R]
        B = []
        if kind == "domain":
            taum, tauc = W.ext("taum"), W.ext("tauc")
            Rc = sum(du[m][m] for m in R)                                                    # :51
            Rm = [sum(u[m] * du[i][m] for m in R) + dp[i] / rho for i in R]                  # :52 (u{i;m,m} dropped)
            for i in R:
                for j in R:
                    B.append((du[i][j], -rho * u[i] * u[j]))                                 # -rho Bilinear(u{i;j}, u{i} u{j})
                    B.append((du[i][j], mu * du[i][j]))                                      # mu Bilinear(u{i;j}, u{i;j})
                    B.append((du[i][j], taum * rho * Rm[i] * u[j]))                          # SUPG
                B.append((du[i][i], -p))                                                     # -Bilinear(u{i;i}, p)
                B.append((p, du[i][i]))                                                      # Bilinear(p, u{i;i})
                B.append((dp[i], taum * Rm[i]))                                              # PSPG
                B.append((du[i][i], tauc * rho * Rc))                                        # LSIC
        else:
            n = [W.n(j) for j in R]
            for i in R:                                                                      # NS_boundary_BASE :59
                B.append((u[i], rho * u[i] * sum(u[j] * n[j] for j in R)))
                B.append((u[i], p * n[i]))
                B.append((u[i], -mu * sum(du[i][j] * n[j] for j in R)))
            if kind == "top":                                                                # NS_boundary_DISP :61-62
                uw = [W.ext(f"uw{i + 1}") for i in R]
                for i in R:
                    B.append((u[i], rho * sum((uw[i] * uw[j] - u[i] * u[j]) * n[j] for j in R)))
                    B.append((p, (uw[i] - u[i]) * n[i]))
                    for j in R:
                        B.append((du[i][j], mu * (uw[i] - u[i]) * n[j]))
                    B.append((u[i], tau_b * rho * (u[i] - uw[i])))
            else:                                                                            # NS_boundary_FIX :64-65
                for i in R:
                    B.append((u[i], -rho * u[i] * sum(u[j] * n[j] for j in R)))
                    B.append((p, -u[i] * n[i]))
                    for j in R:
                        B.append((du[i][j], -mu * u[i] * n[j]))
                    B.append((u[i], tau_b * rho * u[i]))
        return symform.assemble(W, B)

    return build("domain"), build("fix"), build("top")
