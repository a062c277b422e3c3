"""examples/thermal_elasticity/themal_hypo_elasticity.jl on the GPU through the product's generic path: hex-20 bar 10 x 4 x 4, FOUR coupled fields
(d1, d2, d3, T; 16 sparse blocks of which 12 are populated), one time level (C T{;t}, rho c d{i;t}), thermal strain in the dual and the base word of
the elasticity form, convection to a nodal environment temperature on two faces, penalty-fixed end, update_OneStep(max_iter = 3) with dt = 1 until
max |d2_t| < 1e-4 and max |T_t| < 1e-2, the script's solver bicgstabl_GS!(s = 8, maxiter = 2000, max_pass = 20) (:97).  The reference holds no numbers
for this example; checked against the oracle's run of the same script (first steps) and the textbook steady state (thermal bending: tip deflection 0.25,
elongation 0.075, temperatures 200 / 100)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_thermoelastic_bending_on_gpu(mf):
    import torch
    from metafem_jl_amd import element, generic as G, mesh as pm
    from oracle import mesh as om, thermoelastic as te
    from test_gpu_unstructured import _wf

    L_box, e_number, LW = 1.0, 4, 10
    size = (L_box * LW, L_box, L_box)
    space = element.classical_space(3, "Serendipity", 2, 5)
    vert, conn = om.make_brick(size, (int(e_number * LW / 4), e_number, e_number))
    msh = pm.mesh_Classical(vert, conn, space)
    fac = pm.get_BoundaryMesh(msh)
    err = L_box / e_number * 0.01
    c = fac.centroid
    left = fac.select(np.abs(c[:, 0]) < err)
    thermal = fac.select((np.abs(c[:, 1]) < err) | (np.abs(c[:, 1] - L_box) < err))
    P = te.parameters(L_box)
    gd = G.GenericDomain(mf.default_context(), space, msh.coords, msh.cp_ids, 4, _wf(te.domain_weakform(P)),
                         [(left.element_ID, left.element_eindex, _wf(te.fixed_weakform(P))),
                          (thermal.element_ID, thermal.element_eindex, _wf(te.convection_weakform(P)))], max_time_level=1)
    n = msh.ncp
    N = 4 * n
    assert n == 965 and gd.A.n == N
    gd.converge_tol, gd.dt = 1e-6, 1.0
    dx = L_box / e_number
    Te = np.zeros(n)
    Te[(msh.coords[:, 1] > -0.05 * dx) & (msh.coords[:, 1] < 0.05 * dx * dx)] = 300.0
    gd.controlpoints["Te"] = torch.tensor(Te, device="cuda")
    stats = []

    def solver(g):
        dxv, st = mf.iterative_Solve(g.A, g.K_total, g.residue, 1e-10, Sv_func=mf.bicgstabl_GS_, maxiter=2000, max_pass=20, s=8)
        stats.append(st)
        return dxv

    gd.linear_solver = solver
    # the oracle's run of the same script (LU solves), first three steps
    od = te.build(e_number=e_number)
    olog, _ = te.run(od, max_steps=3, stop=False)
    log = []
    for step in range(300):
        hist = gd.update_OneStep(max_iter=3)
        assert hist[-1] < gd.converge_tol, (step, hist)
        x = gd.x
        log.append((float(x[N + n:N + 2 * n].abs().max()), float(x[n:2 * n].abs().max()), float(x[N + 3 * n:N + 4 * n].abs().max()),
                    float(x[3 * n:4 * n].abs().max())))
        if step == 2:
            assert np.abs(np.array(log) - olog).max() <= 1e-7 * np.abs(olog).max(), (log, olog)
            # field by field at the same control points (the product numbers the edge nodes of the serendipity mesh in another order: match by coordinates)
            po, pg = np.lexsort(np.round(od.mesh.coords, 9).T), np.lexsort(np.round(msh.coords, 9).T)
            assert np.abs(od.mesh.coords[po] - msh.coords[pg]).max() < 1e-12
            xg = x[:N].cpu().numpy()
            dmax = np.abs(od.x[:3 * n]).max()  # (a displacement component is weighed by the largest one: the linear solves stop at 1e-10 of the WHOLE residual, d3 is the smallest component)
            for f in range(4):
                fo, fg = od.x[f * n:(f + 1) * n][po], xg[f * n:(f + 1) * n][pg]
                assert np.abs(fo - fg).max() <= 1e-7 * (dmax if f < 3 else np.abs(fo).max()), f
        if log[-1][0] < 1e-4 and log[-1][2] < 1e-2:  # :125
            break
    assert len(log) < 150
    xs = gd.x.cpu().numpy()
    cc = msh.coords
    tip = np.abs(cc[:, 0] - size[0]) < 1e-9
    assert abs(xs[n:2 * n][tip].mean() - 0.25) < 2e-3 and abs(xs[:n][tip].mean() - 0.075) < 1e-3
    mid = np.abs(cc[:, 0] - 5.0) < 0.3
    T = xs[3 * n:4 * n]
    assert abs(T[mid & (np.abs(cc[:, 1]) < 1e-9)].mean() - 200.0) < 0.2 and abs(T[mid & (np.abs(cc[:, 1] - 1.0) < 1e-9)].mean() - 100.0) < 0.2
    assert sum(1 for st in stats if st.converged) == len(stats)
