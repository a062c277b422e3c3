"""GPU end-to-end on UNSTRUCTURED meshes from the reference's example folders, built with the product's own host entry
points (mesh.read_Mesh-format arrays -> mesh_Classical -> get_BoundaryMesh -> device geometry + pattern + S3 operators +
Krylov) and compared with the reference's committed results:
  examples/linear_elasticity/stress_concentration/{2D,3D}_Script.jl -> {2D,3D}_MetaFEM.vtk   (quad-8 / hex-20 from .inp)
  examples/thermal_conduction/3D_Script.jl                           -> 3D_MetaFEM_Result.vtk (tet-10 from .mphtxt)
"""
import os

import numpy as np
import pytest
from scipy.spatial import cKDTree

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _wf(wf):
    from metafem_jl_amd import generic as G

    return G.WeakForm(inner_vars=list(wf.inner_vars), cp_ext_vars=list(wf.cp_ext_vars), normals=list(wf.normals),
                      residues=[G.ResTerm(r.dual_pos, r.dual_s, r.fn) for r in wf.residues],
                      linear_gradients=[G.GradTerm(g.dual_pos, g.dual_s, g.base_pos, g.base_s, g.fn, g.td_order) for g in wf.linear_gradients],
                      nonlinear_gradients=[G.GradTerm(g.dual_pos, g.dual_s, g.base_pos, g.base_s, g.fn, g.td_order) for g in wf.nonlinear_gradients])


def _sampler(msh, shape, dim):
    """oracle.sampling.Sampler over the PRODUCT's mesh arrays (same node order as the oracle's discretization: tests/test_product_tables.py)."""
    import types

    from oracle import reference_element as re_
    from oracle.sampling import Sampler

    disc = re_.initialize_classical_element(dim, shape, 2, 1, 5, itp_type="Serendipity")
    return Sampler(types.SimpleNamespace(coords=np.asarray(msh.coords), cp_ids=np.asarray(msh.cp_ids), nel=msh.cp_ids.shape[1]), disc)


@pytest.mark.parametrize("dim", [2, 3])
def test_stress_concentration_on_gpu_reproduces_reference_vtk(mf, dim):
    import torch
    from metafem_jl_amd import element, generic as G, mesh as pm
    from oracle import problems, stress_concentration as scn
    from oracle.cantilever import traction_field

    z = np.load(os.path.join(GOLD, f"stress_concentration_{dim}d.npz"))
    space = element.classical_space(dim, "Serendipity", 2, 5)
    msh = pm.mesh_Classical(z["vert"], z["conn"].astype(np.int64), space)
    fac = pm.get_BoundaryMesh(msh)
    E, nu, L, err = 210e9, 0.3, 5.0, 0.05  # 2D_Script.jl:15,31-35
    lam, mu, tau = E * nu / ((1 + nu) * (1 - 2 * nu)), E / (2 * (1 + nu)), 10000 * E / L ** 2
    c = fac.centroid
    bnd = []
    for d in range(dim):
        f = fac.select(np.abs(c[:, d]) < err)
        bnd.append((f.element_ID, f.element_eindex, _wf(scn.penalty_component(d, tau))))
    f = fac.select(np.abs(c[:, 1] - L) < err)
    bnd.append((f.element_ID, f.element_eindex, _wf(traction_field(dim, "sl", rows=[1]))))
    gd = G.GenericDomain(mf.default_context(), space, msh.coords, msh.cp_ids, dim, _wf(problems.elasticity_domain(dim, lam, mu)), bnd)
    for v in {2: (2, 3), 3: (2, 4, 6)}[dim]:
        gd.controlpoints[f"sl{v}"] = torch.full((msh.ncp,), 1.0 if v == 2 else 0.0, dtype=torch.float64, device="cuda")
    gd.converge_tol = 1e-8
    stats = []

    def solver(g):  # 3D_Script.jl:69
        dx, st = mf.iterative_Solve(g.A, g.K_total, g.residue, 0.5 * g.converge_tol, Sv_func=mf.idrs_, maxiter=2000, max_pass=20, s=20)
        stats.append(st)
        return dx

    gd.linear_solver = solver
    hist = gd.update_OneStep()
    assert hist[-1] < gd.converge_tol and all(s.converged for s in stats)
    got = gd.x.cpu().numpy()
    d, idx = cKDTree(msh.coords).query(z["xyz"])
    assert d.max() < 1e-7 and msh.ncp == z["d1"].size
    n, scale = msh.ncp, np.abs(z["d2"]).max()
    for fld in range(dim):
        assert np.abs(got[fld * n:(fld + 1) * n][idx] - z[f"d{fld + 1}"]).max() < 1e-5 * scale
    # the reference's committed line samples of this result ({2D,3D}_MetaFEM_{x,y}.csv, 3D_Script.jl:93-94), from the GPU field
    zl = np.load(os.path.join(GOLD, "line_samples.npz"))
    S = _sampler(msh, "CUBE", dim)
    fields = {f"d{i + 1}": got[i * n:(i + 1) * n] for i in range(dim)}
    for tag in ("x", "y"):
        pts, mask = zl[f"stress{dim}d_{tag}_pts"], zl[f"stress{dim}d_{tag}_mask"].astype(bool)
        smp, valid = S.sample(fields, pts[:, :dim], tol=1e-5)
        inside = mask & valid
        assert inside.sum() >= mask.sum() - 1
        sc = max(np.nanmax(np.abs(zl[f"stress{dim}d_{tag}_d{i + 1}"][mask])) for i in range(dim))
        for i in range(dim):
            assert np.abs(smp[f"d{i + 1}"][inside] - zl[f"stress{dim}d_{tag}_d{i + 1}"][inside]).max() < 2e-4 * sc


def test_tet10_thermal_on_gpu_reproduces_reference_vtk(mf):
    import torch
    from metafem_jl_amd import element, generic as G, mesh as pm
    from oracle import problems

    z = np.load(os.path.join(GOLD, "pikachu_tet10.npz"))
    space = element.classical_space(3, "Serendipity", 2, 5, shape="SIMPLEX")
    msh = pm.mesh_Classical(z["vert"] / 100.0, z["conn"].astype(np.int64), space)
    fac = pm.get_BoundaryMesh(msh)
    assert msh.ncp == 23703 and len(fac) == 3120
    T0 = 293.15
    gd = G.GenericDomain(mf.default_context(), space, msh.coords, msh.cp_ids, 1, _wf(problems.thermal_domain(3, 0.6)),
                         [(fac.element_ID, fac.element_eindex, _wf(problems.thermal_convection(25.0, T0)))])
    gd.controlpoints["s"] = torch.full((msh.ncp,), 1600.0, dtype=torch.float64, device="cuda")
    gd.converge_tol = 1e-6
    stats = []

    def solver(g):  # 3D_Script.jl:47
        dx, st = mf.iterative_Solve(g.A, g.K_total, g.residue, 1e-2 * g.converge_tol, Sv_func=mf.idrs_, maxiter=2000, max_pass=10, s=8)
        stats.append(st)
        return dx

    gd.linear_solver = solver
    hist = gd.update_OneStep()
    assert hist[-1] < gd.converge_tol and all(s.converged for s in stats)
    T = gd.x.cpu().numpy()
    d, idx = cKDTree(msh.coords * 100.0).query(z["xyz"])
    assert d.max() < 1e-4
    assert (np.abs(T[idx] - z["T"]) / z["T"]).max() < 1e-5
    # the reference's committed line samples of this result (MetaFEM_a.csv, MetaFEM_b.csv, 3D_Script.jl:73-74), from the GPU field
    zl = np.load(os.path.join(GOLD, "line_samples.npz"))
    S = _sampler(msh, "SIMPLEX", 3)
    for tag in ("a", "b"):
        pts, mask = zl[f"thermal_{tag}_pts"], zl[f"thermal_{tag}_mask"].astype(bool)
        smp, valid = S.sample({"T": T}, pts / 100.0, tol=1e-5)
        inside = mask & valid
        assert inside.sum() >= mask.sum() - 2
        assert np.abs(smp["T"][inside] - zl[f"thermal_{tag}_T"][inside]).max() < 0.02
    # transient variant (3D_Script_Dynamics.jl): three backward-Euler steps heat the body monotonically towards that state
    gt = G.GenericDomain(mf.default_context(), space, msh.coords, msh.cp_ids, 1, _wf(problems.thermal_domain(3, 0.6, C=4.184e3)),
                         [(fac.element_ID, fac.element_eindex, _wf(problems.thermal_convection(25.0, T0)))], max_time_level=1)
    gt.controlpoints["s"] = gd.controlpoints["s"]
    gt.controlpoints["T"] = torch.full((msh.ncp,), T0, dtype=torch.float64, device="cuda")
    gt.assemble_X([("T", 0, 0)])
    gt.dt, gt.converge_tol, gt.linear_solver = 1.0, 1e-6, solver
    prev = gt.x[:msh.ncp].clone()
    for _ in range(3):
        gt.update_OneStep()
        cur = gt.x[:msh.ncp]
        assert torch.all(cur >= prev - 1e-9) and float(cur.max()) < float(gd.x.max())
        prev = cur.clone()
    assert float(prev.max()) > T0 + 0.5


def test_auto_colouring_gives_bitwise_reproducible_assembly_equal_to_atomics(mf):
    """mesh.colour_Elements on the tetrahedral example mesh (50 colours): coloured accumulation is identical run to run and
    agrees with the FP64-atomics accumulation of the same operators to round-off."""
    import torch
    from metafem_jl_amd import element, generic as G, mesh as pm
    from oracle import problems

    z = np.load(os.path.join(GOLD, "pikachu_tet10.npz"))
    space = element.classical_space(3, "Serendipity", 2, 5, shape="SIMPLEX")
    msh = pm.mesh_Classical(z["vert"] / 100.0, z["conn"].astype(np.int64), space)
    fac = pm.get_BoundaryMesh(msh)
    out = {}
    for mode in ("auto", "auto", None):
        gd = G.GenericDomain(mf.default_context(), space, msh.coords, msh.cp_ids, 1, _wf(problems.thermal_domain(3, 0.6, alpha=3.0, Tenv=290.0)),
                             [(fac.element_ID, fac.element_eindex, _wf(problems.thermal_convection(25.0, 293.15, 0.7, 5.669e-8)))],
                             element_colours=mode)
        gd.controlpoints["s"] = torch.full((msh.ncp,), 1600.0, dtype=torch.float64, device="cuda")
        gd.x.copy_(torch.tensor(300.0 + 5.0 * np.sin(40.0 * msh.coords[:, 0]), device="cuda"))
        gd.update_Time(); gd.initialize_dx(); gd.K_linear_func(); gd.update_x_star(); gd.K_nonlinear_func()
        out.setdefault(mode, []).append((gd.K_total.cpu().numpy(), gd.residue.cpu().numpy()))
        if mode == "auto":
            assert gd.groups[0].colour_offsets is not None and gd.groups[1].colour_offsets is not None
    (K1, R1), (K2, R2) = out["auto"]
    assert np.array_equal(K1, K2) and np.array_equal(R1, R2)
    Ka, Ra = out[None][0]
    assert np.abs(K1 - Ka).max() <= 1e-13 * np.abs(Ka).max() and np.abs(R1 - Ra).max() <= 1e-12 * np.abs(Ra).max()
