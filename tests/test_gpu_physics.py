"""The product on its own (no oracle in the loop): mesh generation, control points, boundary selection, weak-form term lists,
device geometry / pattern / operators / Krylov -- against the reference's committed results for two of its example scripts."""
import os

import numpy as np
import pytest
from scipy.spatial import cKDTree

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_ceramic_strip_with_product_modules_only(mf):
    """examples/thermal_conduction/2D_Script.jl (h_penalty of the committed VTK, see tests/test_oracle_golden.py)."""
    import torch
    from metafem_jl_amd import element, generic as G, mesh as pm, physics as ph

    L1, L2, nx, ny = 0.02, 0.01, 40, 20
    space = element.classical_space(2, "Serendipity", 2, 5)
    vert, conn = pm.make_Square((L1, L2), (nx, ny))
    msh = pm.mesh_Classical(vert, conn, space)
    fac = pm.get_BoundaryMesh(msh)
    err = (L1 / nx) * 0.01
    c = fac.centroid
    lr = fac.select((np.abs(c[:, 0]) < err) | (np.abs(c[:, 0] - L1) < err))
    top = fac.select(np.abs(c[:, 1] - L2) < err)
    gd = G.GenericDomain(mf.default_context(), space, msh.coords, msh.cp_ids, 1, ph.thermal_domain(2, 3.0),
                         [(lr.element_ID, lr.element_eindex, ph.thermal_fixed(2, 1.0e5, 1173.15, 3.0)),
                          (top.element_ID, top.element_eindex, ph.thermal_convection(50.0, 323.15, 0.7, 5.669e-8))],
                         element_colours="auto")
    gd.controlpoints["s"] = torch.zeros(msh.ncp, dtype=torch.float64, device="cuda")
    gd.converge_tol = 1e-6
    gd.linear_solver = lambda g: mf.iterative_Solve(g.A, g.K_total, g.residue, 1e-3 * g.converge_tol, Sv_func=mf.idrs_, maxiter=2000,
                                                    max_pass=10, s=8)[0]
    hist = gd.update_OneStep()
    assert hist[-1] < 1e-6 and len(hist) <= 6
    z = np.load(os.path.join(GOLD, "ceramic_strip_T.npz"))
    d, idx = cKDTree(msh.coords).query(z["xy"])
    assert d.max() < 2e-9
    T = gd.x.cpu().numpy()
    assert (np.abs(T[idx] - z["T"]) / np.abs(z["T"])).max() < 1e-5


def test_cantilever_with_product_modules_only(mf, tmp_path):
    """examples/linear_elasticity/cantilever/3D_Script.jl, last load case, + write_VTK of the result read back."""
    import torch
    from metafem_jl_amd import element, generic as G, mesh as pm, physics as ph, vtk as pv

    L, e, R, E, nu = 1.0, 4, 10.0, 210e9, 0.001
    space = element.classical_space(3, "Serendipity", 2, 5)
    vert, conn = pm.make_Brick((L * R, L, L), (20, e, e))
    msh = pm.mesh_Classical(vert, conn, space)
    fac = pm.get_BoundaryMesh(msh)
    c, err = fac.centroid, L / e * 0.01
    left, right, back = fac.select(np.abs(c[:, 0]) < err), fac.select(np.abs(c[:, 0] - L * R) < err), fac.select(np.abs(c[:, 1] - L) < err)
    lam, mu, tau = E * nu / ((1 + nu) * (1 - 2 * nu)), E / (2 * (1 + nu)), 1000 * E / L ** 2
    gd = G.GenericDomain(mf.default_context(), space, msh.coords, msh.cp_ids, 3, ph.elasticity_domain(3, lam, mu),
                         [(left.element_ID, left.element_eindex, ph.penalty((0, 1, 2), tau)),
                          (right.element_ID, right.element_eindex, ph.traction(3, "sl")),
                          (back.element_ID, back.element_eindex, ph.traction(3, "s2"))])
    zero = torch.zeros(msh.ncp, dtype=torch.float64, device="cuda")
    for nm in ("sl", "s2"):
        for k in range(1, 7):
            gd.controlpoints[f"{nm}{k}"] = zero.clone()
    gd.controlpoints["s22"] = torch.tensor(1e6 * (1.0 - msh.coords[:, 0] / (L * R)), device="cuda")
    gd.converge_tol = 1e-5
    gd.linear_solver = lambda g: mf.iterative_Solve(g.A, g.K_total, g.residue, 0.5 * g.converge_tol, Sv_func=mf.idrs_, maxiter=2000,
                                                    max_pass=20, s=8)[0]
    hist = gd.update_OneStep()
    assert hist[-1] < gd.converge_tol
    got = gd.x.cpu().numpy()
    z = np.load(os.path.join(GOLD, "cantilever_hex20.npz"))
    d, idx = cKDTree(msh.coords).query(z["xyz"])
    n, scale = msh.ncp, np.abs(z["d2"]).max()
    for f, nm in enumerate(("d1", "d2", "d3")):
        assert np.abs(got[f * n:(f + 1) * n][idx] - z[nm]).max() < 1e-6 * scale, nm
    path = str(tmp_path / "cantilever.vtk")
    pv.write_VTK(path, msh.coords, msh.cp_ids, space, {"d2": got[n:2 * n], "d3": got[2 * n:], "d1": got[:n]})
    from oracle import vtk as ov  # reader only

    pts, sc = ov.read_vtk_points_scalars(path)
    assert list(sc) == ["d2", "d3", "d1"] and np.array_equal(sc["d2"], got[n:2 * n]) and np.array_equal(pts, msh.coords)
