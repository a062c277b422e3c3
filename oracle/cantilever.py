"""Cantilever driver (oracle; test infrastructure only): examples/linear_elasticity/cantilever/3D_Script.jl -- hex-20
serendipity, penalty-fixed left face, traction on the right face (sigl) and the back face (sig2), the two tractions
being CONTROLPOINT_VAR symmetric tensors named by Voigt id (symbolics/03_Word.jl:34-37,63-64: 3-D ids
[1 6 5; 6 2 4; 5 4 3])."""
from __future__ import annotations

import numpy as np

from . import fem, mesh as om, problems, reference_element as re_, solvers
from .fem import AssembleWeakform, ResTerm

VOIGT = {2: ((1, 3), (3, 2)), 3: ((1, 6, 5), (6, 2, 4), (5, 4, 3))}
INNER_INFOS = [("d1", 0, 0), ("d2", 1, 0), ("d3", 2, 0)]


def traction_field(dim: int, name: str, rows=None) -> AssembleWeakform:
    """Bilinear(d{i}, sig{i,j} * n{j}) with sig a nodal (CONTROLPOINT_VAR) symmetric tensor (3D_Script.jl:61-62);
    rows: the dual components that appear (stress_concentration/3D_Script.jl:51 keeps only d{2})."""
    wf = AssembleWeakform()
    V = VOIGT[dim]
    rows = range(dim) if rows is None else rows
    used = sorted({V[i][j] for i in rows for j in range(dim)})
    wf.cp_ext_vars = [(f"{name}{v}", f"{name}{v}", 0) for v in used]
    wf.normals = [(f"n{j}", j) for j in range(dim)]
    for i in rows:
        wf.residues.append(ResTerm(i, 0, lambda env, i=i: sum(env[f"{name}{V[i][j]}"] * env[f"n{j}"] for j in range(dim))))
    return wf


def build_cantilever(ne_x: int = 20, e_number: int = 4, L_box: float = 1.0, LW_ratio: float = 10.0, E: float = 210e9,
                     nu: float = 0.001):
    """:8-78.  The committed 3D_Cantilever.vtk has 1865 points / 320 hex-20 cells = a 20 x 4 x 4 mesh with
    E = 210e9 (its displacements are the current script's E = 1 result divided by 2.1e11 to 8 digits)."""
    size = (L_box * LW_ratio, L_box, L_box)
    disc = re_.initialize_classical_element(3, "CUBE", 2, 1, 5, itp_type="Serendipity")  # :80
    vert, conn = om.make_brick(size, (ne_x, e_number, e_number))
    msh = om.mesh_classical(vert, conn, disc)
    fac = om.boundary_facets(msh)
    err = L_box / e_number * 0.01  # :26
    c = fac.centroid
    left, right, back = np.abs(c[:, 0]) < err, np.abs(c[:, 0] - size[0]) < err, np.abs(c[:, 1] - L_box) < err
    lam = E * nu / ((1 + nu) * (1 - 2 * nu))
    mu = E / (2 * (1 + nu))
    tau = 1000 * E / L_box ** 2  # :50
    dom = fem.FEMDomain(msh, disc, 3, problems.elasticity_domain(3, lam, mu),
                        [(fac.select(left), problems.elasticity_penalty(3, tau)),
                         (fac.select(right), traction_field(3, "sl")), (fac.select(back), traction_field(3, "s2"))])
    dom.converge_tol = 1e-5  # :98
    for nm in ("sl", "s2"):
        for k in range(1, 7):
            dom.controlpoints[f"{nm}{k}"] = np.zeros(msh.ncp)
    dom.params = dict(L=L_box, l=size[0], E=E, dx=L_box / e_number)
    return dom


def set_load(dom, case: int, sigma: float = 1e6):
    """:109-111 concentrated (1), :124-125 uniform (2), :139 linearly distributed (3; the state 3D_Cantilever.vtk holds)."""
    cp = dom.controlpoints
    cp["sl6"][:] = sigma if case == 1 else 0.0
    if case == 2:
        cp["s22"][:] = sigma
    elif case == 3:
        cp["s22"][:] = sigma * (1.0 - dom.mesh.coords[:, 0] / dom.params["l"])
    else:
        cp["s22"][:] = 0.0


def beam_deflection(case: int, x: np.ndarray, L: float, l: float, E: float, sigma: float = 1e6) -> np.ndarray:
    """Euler-Bernoulli deflections the script plots against (:116,131,144)."""
    I = L ** 3 / 12.0
    if case == 1:
        return sigma * L / (6 * E * I) * (3 * l - x) * x ** 2
    if case == 2:
        return sigma / (24 * E * I) * (x ** 2 + 6 * l ** 2 - 4 * l * x) * x ** 2
    return sigma / (120 * l * E * I) * (10 * l ** 3 - 10 * l ** 2 * x + 5 * l * x ** 2 - x ** 3) * x ** 2


def midline(dom):
    """:104-105 -- control points on y = z = L/2, sorted by x."""
    P, c = dom.params, dom.mesh.coords
    m = (np.abs(c[:, 1] - P["L"] / 2) < 0.25 * P["dx"]) & (np.abs(c[:, 2] - P["L"] / 2) < 0.25 * P["dx"])
    ids = np.nonzero(m)[0]
    return ids[np.argsort(c[ids, 0])]


def lu(dom):
    return solvers.solver_lu_cpu(dom.pattern.rowptr, dom.pattern.colidx, dom.K_total, dom.residue)
