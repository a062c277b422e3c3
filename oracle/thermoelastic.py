"""Thermal bending of a clamped bar (oracle; test infrastructure only): examples/thermal_elasticity/themal_hypo_elasticity.jl -- make_Brick
(e 10 / 4) x e x e -> hex-20 serendipity, itg_order 5 (:9-14, :85); FOUR coupled fields d1, d2, d3, T with ONE time level
(heat capacity C T{;t}, viscous damping rho c d{i;t}: :65-66), thermal strain in the dual AND the base word of the elasticity form
(eps{i,j} = sym grad d - alpha T delta{i,j}, :62-66), convection to a nodal environment temperature Te on the front / back faces (:69), penalty-fixed
left face (:70); stepped with update_OneStep!(max_iter = 3), dt = 1, until max |d2_t| < 1e-4 and max |T_t| < 1e-2 (:107-126), solver
bicgstabl_GS!(s = 8) (:97).

The reference holds NO numbers for this example (the folder has a PNG and an mp4).  What can be checked: the steady state is the textbook thermal
bending of a free beam -- convection (h = k = 100, L = 1) leaves a linear temperature 200 -> 100 across the height (300 - T_front = k g / h = T_back,
T_front - T_back = g L  =>  g = 100), a stress-free curvature kappa = alpha g about the clamped end: tip deflection kappa l^2 / 2 = 0.25 and mean
elongation alpha T_mean l = 0.075.

Term lists (what src/symbolics + 02_LocalAssembly.jl make of :61-71; the dual word eps varies as sym grad(delta d) - alpha delta T delta{i,j}):
  dual d{i;j}:  sigma{i,j}                      gradients  lam delta delta + mu (delta delta + delta delta)  |  -alpha (3 lam + 2 mu) delta{i,j} wrt T
  dual T     :  -alpha sigma{m,m} + C T_t       gradients  -alpha (3 lam + 2 mu) delta{k,l} wrt d{k;l}  |  3 alpha^2 (3 lam + 2 mu) wrt T  |  C (td 1)
  dual T{;i} :  k T{;i}                         gradient   k
  dual d{i}  :  rho c d{i;t}                    gradient   rho c (td 1)
"""
from __future__ import annotations

import numpy as np

from . import fem, mesh as om, reference_element as re_, solvers
from .fem import AssembleWeakform, GradTerm, ResTerm

TPOS = 3  # field order d1, d2, d3, T (@Sym d T, :58)
INNER_INFOS = [("d1", 0, 0), ("d2", 1, 0), ("d3", 2, 0), ("T", 3, 0), ("d1_t", 0, 1), ("d2_t", 1, 1), ("d3_t", 2, 1), ("T_t", 3, 1)]


def domain_weakform(P: dict) -> AssembleWeakform:
    lam, mu, al, C, k, rho, c = P["lam"], P["mu"], P["alpha"], P["C"], P["k"], P["rho"], P["c"]
    b = 3 * lam + 2 * mu
    wf = AssembleWeakform()
    for i in range(3):
        for j in range(3):
            wf.inner_vars.append((f"d{i}_{j}", i, 1 + j, 0))
        wf.inner_vars.append((f"d{i}_t", i, 0, 1))
        wf.inner_vars.append((f"T_{i}", TPOS, 1 + i, 0))
    wf.inner_vars += [("T", TPOS, 0, 0), ("T_t", TPOS, 0, 1)]

    def sigma(env, i, j):
        s = mu * (env[f"d{i}_{j}"] + env[f"d{j}_{i}"])
        if i == j:
            s = s + lam * sum(env[f"d{m}_{m}"] for m in range(3)) - al * b * env["T"]
        return s

    for i in range(3):
        for j in range(3):
            wf.residues.append(ResTerm(i, 1 + j, lambda env, i=i, j=j: sigma(env, i, j)))
            for kk in range(3):
                for l in range(3):
                    cc = (lam if (i == j and kk == l) else 0.0) + mu * ((i == kk and j == l) + (i == l and j == kk))
                    if cc != 0.0:
                        wf.linear_gradients.append(GradTerm(i, 1 + j, kk, 1 + l, lambda env, cc=cc: cc))
        wf.linear_gradients.append(GradTerm(i, 1 + i, TPOS, 0, lambda env: -al * b))
        wf.residues.append(ResTerm(i, 0, lambda env, i=i: rho * c * env[f"d{i}_t"]))
        wf.linear_gradients.append(GradTerm(i, 0, i, 0, lambda env: rho * c, td_order=1))
        wf.residues.append(ResTerm(TPOS, 1 + i, lambda env, i=i: k * env[f"T_{i}"]))
        wf.linear_gradients.append(GradTerm(TPOS, 1 + i, TPOS, 1 + i, lambda env: k))
        wf.linear_gradients.append(GradTerm(TPOS, 0, i, 1 + i, lambda env: -al * b))
    wf.residues.append(ResTerm(TPOS, 0, lambda env: -al * sum(sigma(env, m, m) for m in range(3)) + C * env["T_t"]))
    wf.linear_gradients.append(GradTerm(TPOS, 0, TPOS, 0, lambda env: 3 * al * al * b))
    wf.linear_gradients.append(GradTerm(TPOS, 0, TPOS, 0, lambda env: C, td_order=1))
    return wf


def convection_weakform(P: dict) -> AssembleWeakform:
    """conv_bdy = h * Bilinear(T, T - Te), Te a CONTROLPOINT_VAR (:59, :69)."""
    h = P["h"]
    wf = AssembleWeakform(inner_vars=[("T", TPOS, 0, 0)], cp_ext_vars=[("Te", "Te", 0)])
    wf.residues.append(ResTerm(TPOS, 0, lambda env: h * (env["T"] - env["Te"])))
    wf.linear_gradients.append(GradTerm(TPOS, 0, TPOS, 0, lambda env: h))
    return wf


def fixed_weakform(P: dict) -> AssembleWeakform:
    """fixed_bdy = tau_b * Bilinear(d{i}, d{i}) (:70)."""
    tau = P["tau"]
    wf = AssembleWeakform(inner_vars=[(f"d{i}", i, 0, 0) for i in range(3)])
    for i in range(3):
        wf.residues.append(ResTerm(i, 0, lambda env, i=i: tau * env[f"d{i}"]))
        wf.linear_gradients.append(GradTerm(i, 0, i, 0, lambda env: tau))
    return wf


def parameters(L_box: float = 1.0) -> dict:
    E, nu = 210e3, 0.0  # :47-56
    return dict(lam=E * nu / ((1 + nu) * (1 - 2 * nu)), mu=E / (2 * (1 + nu)), tau=1000 * E / L_box, rho=1e3, c=0.01, h=100.0, C=1000.0, k=100.0,
                alpha=0.05e-3)


def build(e_number: int = 4, LW_ratio: int = 10, L_box: float = 1.0):
    """:8-105."""
    size = (L_box * LW_ratio, L_box, L_box)
    disc = re_.initialize_classical_element(3, "CUBE", 2, 1, 5, itp_type="Serendipity")
    vert, conn = om.make_brick(size, (int(e_number * LW_ratio / 4), e_number, e_number))
    msh = om.mesh_classical(vert, conn, disc)
    fac = om.boundary_facets(msh)
    err = L_box / e_number * 0.01
    cen = fac.centroid
    left = np.abs(cen[:, 0]) < err
    thermal = (np.abs(cen[:, 1]) < err) | (np.abs(cen[:, 1] - L_box) < err)  # front + back (:41)
    P = parameters(L_box)
    dom = fem.FEMDomain(msh, disc, 4, domain_weakform(P), [(fac.select(left), fixed_weakform(P)), (fac.select(thermal), convection_weakform(P))],
                        max_time_level=1)
    dom.converge_tol = 1e-6  # :99
    dom.dt = 1.0             # :107
    dx = L_box / e_number
    Te = np.zeros(msh.ncp)
    Te[(msh.coords[:, 1] > -0.05 * dx) & (msh.coords[:, 1] < 0.05 * dx * dx)] = 300.0  # :104-105 (the upper bound is err_scale * dx, as written)
    dom.controlpoints["Te"] = Te
    dom.params = dict(P, l=size[0], L=L_box, dx=dx)
    return dom


def lu(dom):
    return solvers.solver_lu_cpu(dom.pattern.rowptr, dom.pattern.colidx, dom.K_total, dom.residue)


def solver_of_the_script(dom):
    """:97."""
    return solvers.iterative_solve(dom.pattern.rowptr, dom.pattern.colidx, dom.K_total, dom.residue, dom.converge_tol,
                                   Sv_func=solvers.bicgstabl_gs, maxiter=2000, max_pass=20, s=8)


def run(dom, linear_solver=None, max_steps: int = 400, stop=True):
    """:107-126.  Returns per step (max |d2_t|, max |d2|, max |T_t|, max |T|) and the Newton histories."""
    dom.linear_solver = linear_solver or lu
    dom.x[:] = 0.0
    dom.t = 0.0
    n, N = dom.mesh.ncp, dom.basicfield_size
    log, hists = [], []
    for _ in range(max_steps):
        hists.append(dom.update_one_step(max_iter=3))
        d2, T = dom.x[n:2 * n], dom.x[3 * n:4 * n]
        d2t, Tt = dom.x[N + n:N + 2 * n], dom.x[N + 3 * n:N + 4 * n]
        log.append((np.abs(d2t).max(), np.abs(d2).max(), np.abs(Tt).max(), np.abs(T).max()))
        if stop and log[-1][0] < 1e-4 and log[-1][2] < 1e-2:
            break
    return np.array(log), hists
