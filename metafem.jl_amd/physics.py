"""Term lists of the weak forms the reference's example scripts define with @Sym/@Def (the output of its symbolic layer,
`initialize_LocalAssembly!`, for these forms), ready for `generic.GenericDomain`:

  thermal_domain / thermal_convection / thermal_fixed     examples/thermal_conduction/2D_Script.jl:54-58, 3D_Script.jl:29-32,
                                                          3D_Script_Dynamics.jl:32
  elasticity_domain / elasticity_inertia                  examples/linear_elasticity/cantilever/3D_Script.jl:52-57,
                                                          examples/hypo_elastic_plasticity/J2Plasticity.jl:59
  penalty / traction                                      cantilever/3D_Script.jl:60-62, stress_concentration/3D_Script.jl:48-51

A residual term is `ResTerm(dual field, dual word, f(env))`, a gradient term `GradTerm(dual field, dual word, base field,
base word, f(env), time-derivative order)`; word 0 is the value, 1 + j the derivative along x_j.  `env` maps inner-variable /
external names to [items, itg] device tensors.  Fields are numbered in symbol order (d1, d2, d3; p, u1, u2).
Nonlinear forms (e.g. the cavity's SUPG/PSPG Navier-Stokes) come from the host's symbolic layer; this module only carries
the closed-form linear ones.
"""
from __future__ import annotations

from typing import Optional, Sequence

from .generic import GradTerm, ResTerm, WeakForm

VOIGT = {2: ((1, 3), (3, 2)), 3: ((1, 6, 5), (6, 2, 4), (5, 4, 3))}  # symbolics/03_Word.jl:34-35 (1-based ids)


def thermal_domain(dim: int, k: float, alpha: float = 0.0, Tenv: float = 0.0, C: float = 0.0, source: str = "s") -> WeakForm:
    """-C Bilinear(T, T{;t}) - k Bilinear(T{;i}, T{;i}) + Bilinear(T, s + alpha (Tenv - T))."""
    wf = WeakForm()
    if C != 0.0:
        wf.inner_vars.append(("T_t", 0, 0, 1))
        wf.residues.append(ResTerm(0, 0, lambda env: -C * env["T_t"]))
        wf.linear_gradients.append(GradTerm(0, 0, 0, 0, lambda env: -C, 1))
    for d in range(dim):
        wf.inner_vars.append((f"T_{d}", 0, 1 + d, 0))
        wf.residues.append(ResTerm(0, 1 + d, lambda env, d=d: -k * env[f"T_{d}"]))
        wf.linear_gradients.append(GradTerm(0, 1 + d, 0, 1 + d, lambda env: -k))
    wf.cp_ext_vars.append((source, source, 0))
    if alpha != 0.0:
        wf.inner_vars.append(("T", 0, 0, 0))
        wf.residues.append(ResTerm(0, 0, lambda env: env[source] + alpha * (Tenv - env["T"])))
        wf.linear_gradients.append(GradTerm(0, 0, 0, 0, lambda env: -alpha))
    else:
        wf.residues.append(ResTerm(0, 0, lambda env: env[source]))
    return wf


def thermal_convection(h: float, Tenv: float, em: float = 0.0, sigma_b: float = 0.0) -> WeakForm:
    """h Bilinear(T, Tenv - T) + em sigma Bilinear(T, Tenv^4 - T^4): the radiative part is a nonlinear gradient."""
    wf = WeakForm(inner_vars=[("T", 0, 0, 0)])
    wf.residues.append(ResTerm(0, 0, lambda env: h * (Tenv - env["T"])))
    wf.linear_gradients.append(GradTerm(0, 0, 0, 0, lambda env: -h))
    if em != 0.0:
        c = em * sigma_b
        wf.residues.append(ResTerm(0, 0, lambda env: c * (Tenv ** 4 - env["T"] ** 4)))
        wf.nonlinear_gradients.append(GradTerm(0, 0, 0, 0, lambda env: -4.0 * c * env["T"] ** 3))
    return wf


def thermal_fixed(dim: int, h_penalty: float, Tw: float, k: float) -> WeakForm:
    """h_penalty Bilinear(T, Tw - T) + k Bilinear(T, n{i} T{;i})  (Nitsche-type wall, 2D_Script.jl:58)."""
    wf = WeakForm(inner_vars=[("T", 0, 0, 0)] + [(f"T_{d}", 0, 1 + d, 0) for d in range(dim)],
                  normals=[(f"n{d}", d) for d in range(dim)])
    wf.residues.append(ResTerm(0, 0, lambda env: h_penalty * (Tw - env["T"]) + k * sum(env[f"n{d}"] * env[f"T_{d}"] for d in range(dim))))
    wf.linear_gradients.append(GradTerm(0, 0, 0, 0, lambda env: -h_penalty))
    for d in range(dim):
        wf.linear_gradients.append(GradTerm(0, 0, 0, 1 + d, lambda env, d=d: k * env[f"n{d}"]))
    return wf


def elasticity_domain(dim: int, lam: float, mu: float) -> WeakForm:
    """-Bilinear(eps{i,j}, sigma{i,j}), sigma = lam delta eps{m,m} + 2 mu eps: dim^2 dual words, 21 gradient terms in 3-D."""
    wf = WeakForm()
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


def elasticity_inertia(dim: int, rho: float, c: float = 0.0) -> WeakForm:
    """-Bilinear(d{i}, rho (c d{i;t} + d{i;t,t}))  (needs max_time_level = 2)."""
    wf = WeakForm()
    for i in range(dim):
        wf.inner_vars.append((f"d{i}_tt", i, 0, 2))
        wf.linear_gradients.append(GradTerm(i, 0, i, 0, lambda env: -rho, 2))
        if c != 0.0:
            wf.inner_vars.append((f"d{i}_t", i, 0, 1))
            wf.linear_gradients.append(GradTerm(i, 0, i, 0, lambda env: -rho * c, 1))
            wf.residues.append(ResTerm(i, 0, lambda env, i=i: -rho * (c * env[f"d{i}_t"] + env[f"d{i}_tt"])))
        else:
            wf.residues.append(ResTerm(i, 0, lambda env, i=i: -rho * env[f"d{i}_tt"]))
    return wf


def penalty(components: Sequence[int], tau: float, wall_syms: Optional[Sequence[str]] = None) -> WeakForm:
    """tau Bilinear(d{i}, dw{i} - d{i}) for the listed components; dw = 0 unless nodal arrays are named."""
    wf = WeakForm()
    for n_, i in enumerate(components):
        wf.inner_vars.append((f"d{i}", i, 0, 0))
        if wall_syms is not None:
            sym = wall_syms[n_]
            wf.cp_ext_vars.append((f"dw{i}", sym, 0))
            wf.residues.append(ResTerm(i, 0, lambda env, i=i: tau * (env[f"dw{i}"] - env[f"d{i}"])))
        else:
            wf.residues.append(ResTerm(i, 0, lambda env, i=i: tau * (0.0 - env[f"d{i}"])))
        wf.linear_gradients.append(GradTerm(i, 0, i, 0, lambda env: -tau))
    return wf


def traction(dim: int, name: str, rows: Optional[Sequence[int]] = None) -> WeakForm:
    """Bilinear(d{i}, sig{i,j} n{j}) with sig a nodal SYMMETRIC_TENSOR external named <name><Voigt id>."""
    wf = WeakForm()
    V = VOIGT[dim]
    rows = list(range(dim)) if rows is None else list(rows)
    used = sorted({V[i][j] for i in rows for j in range(dim)})
    wf.cp_ext_vars = [(f"{name}{v}", f"{name}{v}", 0) for v in used]
    wf.normals = [(f"n{j}", j) for j in range(dim)]
    for i in rows:
        wf.residues.append(ResTerm(i, 0, lambda env, i=i: sum(env[f"{name}{V[i][j]}"] * env[f"n{j}"] for j in range(dim))))
    return wf


def merge(*wfs: WeakForm) -> WeakForm:
    out = WeakForm()
    for wf in wfs:
        for a in ("inner_vars", "cp_ext_vars", "normals"):
            for item in getattr(wf, a):
                if item not in getattr(out, a):
                    getattr(out, a).append(item)
        out.residues += wf.residues
        out.linear_gradients += wf.linear_gradients
        out.nonlinear_gradients += wf.nonlinear_gradients
    return out
