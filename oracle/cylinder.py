"""3-D flow around a cylinder (oracle; test infrastructure only): examples/incompressible_flow/cylinder_flow/3D_MetaFEM_Script.jl --
COMSOL tetrahedral mesh (.mphtxt) -> tet-10 (SIMPLEX, itp_order 2, itg_order 6), SUPG/PSPG-stabilised Navier-Stokes with weakly
imposed inflow (parabolic profile), outflow (pressure penalty) and no-slip walls; one update_OneStep! with max_iter = 6 from u = p = 0,
linear solver idrs!(s = 8) with Pl_func = Pl_Jacobi (:90) -- the only shipped example that selects the left preconditioner.
Fields sorted by symbol: p, u1, u2, u3."""
from __future__ import annotations

import numpy as np

from . import fem, mesh as om, reference_element as re_, solvers

INNER_INFOS = [("p", 0, 0), ("u1", 1, 0), ("u2", 2, 0), ("u3", 3, 0)]


def weakforms(rho: float, mu: float, tau_b: float, tau_p: float, dim: int = 3):
    """:44-70.  Second derivatives u{i;m,m} are dropped (explicit_max_sd_order = 1, :77).  Returns (domain, inflow, outflow, fix)."""
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
            Rc = sum(du[m][m] for m in R)                                                    # :48
            Rm = [sum(u[m] * du[i][m] for m in R) + dp[i] / rho for i in R]                  # :49 (u{i;m,m} dropped)
            for i in R:
                for j in R:
                    B.append((du[i][j], -rho * u[i] * u[j]))                                 # :53
                    B.append((du[i][j], mu * du[i][j]))
                    B.append((du[i][j], taum * rho * Rm[i] * u[j]))                          # :54 SUPG
                B.append((du[i][i], -p))
                B.append((p, du[i][i]))
                B.append((dp[i], taum * Rm[i]))                                              # PSPG
                B.append((du[i][i], tauc * rho * Rc))                                        # LSIC
        else:
            n = [W.n(j) for j in R]
            for i in R:                                                                      # NS_boundary_BASE :56
                B.append((u[i], p * n[i]))
                B.append((u[i], -mu * sum(du[i][j] * n[j] for j in R)))
            if kind == "inflow":                                                             # :57
                uw = [W.ext(f"uw{i + 1}") for i in R]
                for i in R:
                    B.append((u[i], rho * uw[i] * sum(uw[j] * n[j] for j in R)))
                    B.append((p, (uw[i] - u[i]) * n[i]))
                    for j in R:
                        B.append((du[i][j], mu * (uw[i] - u[i]) * n[j]))
                    B.append((u[i], tau_b * rho * (u[i] - uw[i])))
            elif kind == "outflow":                                                          # :59
                for i in R:
                    B.append((u[i], rho * u[i] * sum(u[j] * n[j] for j in R)))
                B.append((p, tau_p * p))
            else:                                                                            # NS_boundary_FIX :60
                for i in R:
                    B.append((p, -u[i] * n[i]))
                    for j in R:
                        B.append((du[i][j], -mu * u[i] * n[j]))
                    B.append((u[i], tau_b * rho * u[i]))
        return symform.assemble(W, B)

    return build("domain"), build("inflow"), build("outflow"), build("fix")


def build(vert: np.ndarray, conn: np.ndarray, Cb: float = 128.0, rho: float = 1e3, mu: float = 1.0, dx: float = 0.02,
          L: float = 2.5, H: float = 0.41, Um: float = 0.45, itg_order: int = 6):
    """:8-42, 74-108."""
    dim = 3
    nu = mu / rho
    tau_b = mu / rho * Cb / dx   # :40
    tau_p = Cb * dx / mu         # :41
    disc = re_.initialize_classical_element(3, "SIMPLEX", 2, 1, itg_order, itp_type="Serendipity")  # :78
    msh = om.mesh_classical(vert, conn, disc)
    fac = om.boundary_facets(msh)
    c = fac.centroid
    err = 0.01
    left = (c[:, 0] < err) & (c[:, 0] > -err)            # :26
    right = (c[:, 0] < L + err) & (c[:, 0] > L - err)    # :27
    wd, w_in, w_out, w_fix = weakforms(rho, mu, tau_b, tau_p)
    dom = fem.FEMDomain(msh, disc, 4, wd, [(fac.select(~(left | right)), w_fix), (fac.select(left), w_in), (fac.select(right), w_out)])
    dom.converge_tol = 1e-6  # :91
    ys, zs = msh.coords[:, 1], msh.coords[:, 2]
    ncp = msh.ncp
    dom.controlpoints["uw1"] = (16 * Um / H ** 4) * (ys * zs * (H - ys) * (H - zs))  # :101
    dom.controlpoints["uw2"] = np.zeros(ncp)
    dom.controlpoints["uw3"] = np.zeros(ncp)
    dom.dt = 0.2 * dx / Um  # :103
    taum = (9 * 16 * nu ** 2 * dim * dx ** (-4)) ** (-0.5)  # :104
    dom.controlpoints["taum"] = np.full(ncp, taum)
    dom.controlpoints["tauc"] = np.full(ncp, (taum * (dim * dx ** (-2))) ** (-1.0))  # :105
    dom.params = dict(dx=dx, rho=rho, mu=mu, L=L, H=H, Um=Um)
    return dom


def solver_of_the_script(dom):
    """:90 -- iterative_Solve!(x; Sv_func! = idrs!, Pl_func = Pl_Jacobi, maxiter = 2000, max_pass = 10, s = 8)."""
    return solvers.iterative_solve(dom.pattern.rowptr, dom.pattern.colidx, dom.K_total, dom.residue, dom.converge_tol,
                                   Sv_func=solvers.idrs, Pl_func=solvers.pl_jacobi, maxiter=2000, max_pass=10, s=8)


def run(vert, conn, linear_solver=None, max_iter: int = 6, **kw):
    dom = build(vert, conn, **kw)
    dom.linear_solver = linear_solver or solver_of_the_script
    dom.x[:] = 0.0
    dom.t = 0.0
    dom.dessemble_x(INNER_INFOS)
    hist = dom.update_one_step(max_iter=max_iter)  # :106
    dom.dessemble_x(INNER_INFOS)
    return dom, hist
