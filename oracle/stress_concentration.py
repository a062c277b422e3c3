"""Plate-with-hole driver (oracle; test infrastructure only): examples/linear_elasticity/stress_concentration/
2D_Script.jl / 3D_Script.jl -- Abaqus first-order CUBE meshes (.inp) upgraded to quad-8 / hex-20 serendipity,
symmetry planes imposed component-wise by penalty, unit traction sigl{2,2} on the y = L side."""
from __future__ import annotations

import numpy as np

from . import fem, mesh as om, problems, reference_element as re_, readers, solvers
from .cantilever import traction_field
from .fem import AssembleWeakform, GradTerm, ResTerm


def penalty_component(i: int, tau: float) -> AssembleWeakform:
    """tau * Bilinear(d{i}, dw{i} - d{i}) with dw = 0 (2D_Script.jl:45-46; 3D_Script.jl:48-50)."""
    wf = AssembleWeakform(inner_vars=[(f"d{i}", i, 0, 0)])
    wf.residues.append(ResTerm(i, 0, lambda env: tau * (0.0 - env[f"d{i}"])))
    wf.linear_gradients.append(GradTerm(i, 0, i, 0, lambda env: -tau))
    return wf


def build(vert: np.ndarray, conn: np.ndarray, E: float = 210e9, nu: float = 0.3, L: float = 5.0, err: float = 0.05,
          sigma: float = 1.0):
    """vert, conn = read_Mesh("2D_Mesh.inp" | "3D_Mesh.inp") (2D_Script.jl:6-7); constants :15,31-35,64-68."""
    dim = vert.shape[0]
    disc = re_.initialize_classical_element(dim, "CUBE", 2, 1, 5, itp_type="Serendipity")
    msh = om.mesh_classical(vert, conn, disc)
    fac = om.boundary_facets(msh)
    c = fac.centroid
    lam = E * nu / ((1 + nu) * (1 - 2 * nu))
    mu = E / (2 * (1 + nu))
    tau = 10000 * E / L ** 2
    lo = [np.abs(c[:, d]) < err for d in range(dim)]
    loaded = np.abs(c[:, 1] - L) < err
    bnd = [(fac.select(lo[d]), penalty_component(d, tau)) for d in range(dim)]
    bnd.append((fac.select(loaded), traction_field(dim, "sl", rows=[1])))
    dom = fem.FEMDomain(msh, disc, dim, problems.elasticity_domain(dim, lam, mu), bnd)
    dom.converge_tol = 1e-8
    for v in {1: (), 2: (2, 3), 3: (2, 4, 6)}[dim]:
        dom.controlpoints[f"sl{v}"] = np.zeros(msh.ncp)
    dom.controlpoints["sl2"][:] = sigma
    dom.params = dict(L=L, E=E)
    return dom


def lu(dom):
    return solvers.solver_lu_cpu(dom.pattern.rowptr, dom.pattern.colidx, dom.K_total, dom.residue)
