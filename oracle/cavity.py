"""Lid-driven cavity driver (oracle; test infrastructure only): the load-stepping loop of
examples/incompressible_flow/lid_driven_cavity_flow/2D_Script.jl:114-135 / :188-215 around update_OneStep!."""
from __future__ import annotations

import numpy as np

from . import fem, mesh as om, problems, reference_element as re_, solvers

INNER_INFOS = [("p", 0, 0), ("u1", 1, 0), ("u2", 2, 0)]  # local_innervar_infos, fields sorted by symbol


def build_cavity(ne: int, Cb: float = 128.0, rho: float = 1e3, mu: float = 1.0, L: float = 1.0):
    """:8-84 -- quad-8 serendipity, itg_order 5, walls (left, bottom, right) + lid (top)."""
    dx = L / ne
    tau_b = mu / rho * Cb / dx  # :45-46
    disc = re_.initialize_classical_element(2, "CUBE", 2, 1, 5, itp_type="Serendipity")
    vert, conn = om.make_square((L, L), (ne, ne))
    mesh = om.mesh_classical(vert, conn, disc)
    fac = om.boundary_facets(mesh)
    top = np.abs(fac.centroid[:, 1] - L) < dx * 0.01
    wd, wfix, wtop = problems.cavity_weakforms(rho, mu, tau_b)
    dom = fem.FEMDomain(mesh, disc, 3, wd, [(fac.select(~top), wfix), (fac.select(top), wtop)])
    dom.converge_tol = 1e-5  # :99
    dom.controlpoints["uw2"] = np.zeros(mesh.ncp)
    dom.params = dict(dx=dx, rho=rho, mu=mu, nu=mu / rho, L=L, dim=2)
    return dom


def set_step_parameters(dom, u_top: float):
    """:122-127 -- lid speed, dt, SUPG/PSPG parameters from the current velocity field."""
    P = dom.params
    dx, nu, dim = P["dx"], P["nu"], P["dim"]
    dt = dom.dt = 0.2 * dx / u_top
    ncp = dom.mesh.ncp
    dom.controlpoints["uw1"] = np.full(ncp, u_top)
    u1, u2 = dom.controlpoints["u1"], dom.controlpoints["u2"]
    taum = (4 / dt ** 2 + 9 * 16 * nu ** 2 * dim * dx ** (-4) + dx ** (-2) * (u1 ** 2.0 + u2 ** 2.0)) ** (-0.5)
    dom.controlpoints["taum"] = taum
    dom.controlpoints["tauc"] = (taum * (dim * dx ** (-2))) ** (-1.0)


def run_cavity(ne: int = 40, Re: float = 1000.0, Cb: float = 128.0, linear_solver=None, tmax=None, max_iter: int = 6):
    dom = build_cavity(ne, Cb)
    P = dom.params
    dom.linear_solver = linear_solver or (lambda d: solvers.solver_lu_cpu(d.pattern.rowptr, d.pattern.colidx, d.K_total, d.residue))
    u_st = Re / P["L"] * P["mu"] / P["rho"]
    if tmax is None:
        tmax = 10 if Re > 1000 else int(np.ceil(Re / 100))  # :119
    dom.x[:] = 0.0
    dom.t = 0.0
    dom.dessemble_x(INNER_INFOS)
    hists = []
    for i in range(1, tmax + 1):
        set_step_parameters(dom, u_st * (i / tmax))
        hists.append(dom.update_one_step(max_iter=max_iter))
        dom.dessemble_x(INNER_INFOS)
    return dom, hists
