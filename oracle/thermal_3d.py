"""3-D heat conduction on an external tetrahedral mesh (oracle; test infrastructure only):
examples/thermal_conduction/3D_Script.jl -- COMSOL mesh scaled by 1/100, tet-10 (SIMPLEX, order 2, itg_order 5), convection
on the whole boundary, uniform source; 3D_Script_Dynamics.jl adds -C*Bilinear(T, T{;t})."""
from __future__ import annotations

import numpy as np

from . import fem, mesh as om, problems, reference_element as re_, solvers


def build(vert: np.ndarray, conn: np.ndarray, k: float = 0.6, h: float = 25.0, T0: float = 293.15, s: float = 1600.0,
          scale: float = 0.01, C: float = 0.0):
    disc = re_.initialize_classical_element(3, "SIMPLEX", 2, 1, 5, itp_type="Serendipity")  # 3D_Script.jl:38
    msh = om.mesh_classical(vert * scale, conn, disc)  # :9 vert ./ 100
    fac = om.boundary_facets(msh)
    dom = fem.FEMDomain(msh, disc, 1, problems.thermal_domain(3, k, C=C), [(fac, problems.thermal_convection(h, T0))],
                        max_time_level=1 if C != 0.0 else 0)
    dom.controlpoints["s"] = np.full(msh.ncp, s)  # :54
    dom.controlpoints["T"] = np.full(msh.ncp, T0)  # :53 (never assembled into x in the static script: x0 = 0)
    dom.converge_tol = 1e-6  # :48
    return dom


def lu(dom):
    return solvers.solver_lu_cpu(dom.pattern.rowptr, dom.pattern.colidx, dom.K_total, dom.residue)
