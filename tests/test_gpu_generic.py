"""GPU end-to-end on the GENERIC path (device geometry + pattern + S3 operators + Krylov), checked against the oracle AND
against the reference's own committed outputs:
  C1  examples/thermal_conduction/2D_Script.jl            -> 2D_Ceramic_Strip.vtk
  C5  examples/incompressible_flow/lid_driven_cavity_flow  -> 2D_Cavity_Flow.vtk (Re = 1000)
"""
import os

import numpy as np
import pytest
from scipy.spatial import cKDTree

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _to_product_wf(mf, wf):
    from metafem_jl_amd import generic as G

    return G.WeakForm(inner_vars=list(wf.inner_vars), cp_ext_vars=list(wf.cp_ext_vars), normals=list(wf.normals),
                      residues=[G.ResTerm(r.dual_pos, r.dual_s, r.fn) for r in wf.residues],
                      linear_gradients=[G.GradTerm(g.dual_pos, g.dual_s, g.base_pos, g.base_s, g.fn, g.td_order) for g in wf.linear_gradients],
                      nonlinear_gradients=[G.GradTerm(g.dual_pos, g.dual_s, g.base_pos, g.base_s, g.fn, g.td_order) for g in wf.nonlinear_gradients])


def _gpu_domain(mf, od, itp_type, order, itg, colours=None):
    """Build the product's GenericDomain from an oracle FEMDomain's mesh + weak forms (the weak forms are data)."""
    from metafem_jl_amd import element, generic as G

    space = element.classical_space(od.disc.dim, itp_type, order, itg)
    bnd = [(f.element_ID, f.element_eindex, _to_product_wf(mf, wf)) for f, wf in od.boundaries]
    return G.GenericDomain(mf.default_context(), space, od.mesh.coords, od.mesh.cp_ids, od.n_fields, _to_product_wf(mf, od.domain_wf),
                           bnd, element_colours=colours, max_time_level=od.max_time_level,
                           dissipative=od.time.gamma_params[0] == 1.0)


@pytest.mark.parametrize("kind", ["quad8", "hex8", "hex27", "hex20"])
def test_device_geometry_update_matches_oracle(mf, kind):
    import torch
    from oracle import fem, mesh as om, problems, reference_element as re_

    if kind == "quad8":
        disc = re_.initialize_classical_element(2, "CUBE", 2, 1, 5, itp_type="Serendipity")
        vert, conn = om.make_square((2.0, 1.0), (5, 4))
        args = ("Serendipity", 2, 5)
    else:
        order, t = {"hex8": (1, "Lagrange"), "hex27": (2, "Lagrange"), "hex20": (2, "Serendipity")}[kind]
        disc = re_.initialize_classical_element(3, "CUBE", order, 1, 5, itp_type=t)
        vert, conn = om.make_brick((1.0, 2.0, 1.5), (3, 2, 2))
        args = (t, order, 5)
    msh = om.mesh_classical(vert, conn, disc)
    c = msh.coords
    msh.coords = c + 0.03 * np.sin(2.0 * c[:, ::-1] + 0.3)
    fac = om.boundary_facets(msh)
    dim = disc.dim
    od = fem.FEMDomain(msh, disc, 1, problems.thermal_domain(dim, 1.0), [(fac, problems.thermal_convection(1.0, 0.0))])
    gd = _gpu_domain(mf, od, *args)
    g0 = gd.groups[0]
    nel, itp, itg = msh.nel, disc.itp_func_num, disc.itg_func_num
    v = g0.vals.cpu().numpy().reshape(nel, 1 + dim, itp, itg).transpose(3, 2, 1, 0)
    assert np.allclose(v, od.elgeo.integral_vals, rtol=0, atol=1e-12 * np.abs(od.elgeo.integral_vals).max())
    assert np.allclose(g0.weights.cpu().numpy().reshape(nel, itg).T, od.elgeo.integral_weights, rtol=1e-13)
    g1, fg = gd.groups[1], od.fgeo[0]
    nf, itb = len(fac), disc.bdy_itg_func_num
    fv = g1.vals.cpu().numpy().reshape(nf, 1 + dim, itp, itb).transpose(3, 2, 1, 0)
    assert np.allclose(fv, fg.integral_vals, rtol=0, atol=1e-12 * np.abs(fg.integral_vals).max())
    assert np.allclose(g1.weights.cpu().numpy().reshape(nf, itb).T, fg.integral_weights, rtol=1e-13)
    assert np.allclose(g1.normals.cpu().numpy().transpose(2, 1, 0), fg.normal_directions, rtol=0, atol=1e-13)


def test_c1_thermal_strip_on_gpu_reproduces_reference_vtk(mf):
    """The reference example, generic GPU path, the reference's own solver choice (idrs!, s = 8)."""
    import torch
    from oracle import fem, mesh as om, problems, reference_element as re_, solvers

    L1, L2, nx, ny = 0.02, 0.01, 40, 20
    disc = re_.initialize_classical_element(2, "CUBE", 2, 1, 5, itp_type="Serendipity")
    vert, conn = om.make_square((L1, L2), (nx, ny))
    mesh = om.mesh_classical(vert, conn, disc)
    fac = om.boundary_facets(mesh)
    err = (L1 / nx) * 0.01
    lr = (np.abs(fac.centroid[:, 0]) < err) | (np.abs(fac.centroid[:, 0] - L1) < err)
    top = np.abs(fac.centroid[:, 1] - L2) < err
    od = fem.FEMDomain(mesh, disc, 1, problems.thermal_domain(2, 3),
                       [(fac.select(lr), problems.thermal_fixed(2, 1.0e5, 1173.15, 3)),  # h_penalty of the VTK, see test_oracle_golden
                        (fac.select(top), problems.thermal_convection(50, 323.15, 0.7, 5.669e-8))])
    od.controlpoints["s"] = np.zeros(mesh.ncp)
    od.converge_tol = 1e-6
    od.linear_solver = lambda d: solvers.solver_lu_cpu(d.pattern.rowptr, d.pattern.colidx, d.K_total, d.residue)
    od.update_one_step()

    gd = _gpu_domain(mf, od, "Serendipity", 2, 5)
    gd.controlpoints["s"] = torch.zeros(mesh.ncp, dtype=torch.float64, device="cuda")
    gd.converge_tol = 1e-6
    gd.linear_solver = lambda g: mf.iterative_Solve(g.A, g.K_total, g.residue, 1e-3 * g.converge_tol, Sv_func=mf.idrs_, maxiter=2000,
                                                    max_pass=10, s=8)[0]
    hist = gd.update_OneStep()
    assert hist[-1] < 1e-6 and len(hist) <= 6
    T = gd.x.cpu().numpy()
    assert np.abs(T - od.x).max() <= 1e-7 * np.abs(od.x).max()          # vs the oracle (same Newton path)
    z = np.load(os.path.join(GOLD, "ceramic_strip_T.npz"))
    d, idx = cKDTree(mesh.coords).query(z["xy"])
    assert (np.abs(T[idx] - z["T"]) / np.abs(z["T"])).max() < 1e-5       # vs the reference's committed result


def test_c5_cavity_newton_steps_match_oracle(mf):
    """Two load steps on a 10 x 10 cavity: 3 fields, 9 sparse blocks, SUPG/PSPG nonlinear gradients, Nitsche walls."""
    import torch
    from oracle import cavity, solvers

    od = cavity.build_cavity(10, Cb=128.0)
    od.linear_solver = lambda d: solvers.solver_lu_cpu(d.pattern.rowptr, d.pattern.colidx, d.K_total, d.residue)
    gd = _gpu_domain(mf, od, "Serendipity", 2, 5)
    gd.converge_tol = od.converge_tol = 1e-8
    for sv, s in ((mf.bicgstabl_GS_, 4), (mf.idrs_, 8)):
        od.x[:] = 0.0
        od.dessemble_x(cavity.INNER_INFOS)
        gd.x.zero_()
        # tolerance relative to the right-hand side: an absolute 1e-12 is below what FP64 can reach on this matrix
        gd.linear_solver = lambda g, sv=sv, s=s: mf.iterative_Solve(g.A, g.K_total, g.residue, 1e-10 * mf.normalized_norm(g.residue),
                                                                     Sv_func=sv, maxiter=4000, max_pass=20, s=s)[0]
        for step in (1, 2):
            cavity.set_step_parameters(od, 0.05 * step)
            for k in ("uw1", "uw2", "taum", "tauc"):
                gd.controlpoints[k] = torch.tensor(od.controlpoints[k], device="cuda")
            gd.dt = od.dt
            ho = od.update_one_step(max_iter=6)
            od.dessemble_x(cavity.INNER_INFOS)
            hg = gd.update_OneStep(max_iter=6)
            assert len(hg) == len(ho)
            assert np.allclose(hg[:2], ho[:2], rtol=1e-8)
            got = gd.x.cpu().numpy()
            n = od.mesh.ncp
            for f in (1, 2):  # velocities
                assert np.abs(got[f * n:(f + 1) * n] - od.x[f * n:(f + 1) * n]).max() <= 1e-7 * 0.1
            pg, po = got[:n], od.x[:n]  # pressure: up to the unpinned constant
            assert np.abs((pg - pg.mean()) - (po - po.mean())).max() <= 1e-6 * (po.max() - po.min() + 1e-30)


def test_c5_cavity_re1000_on_gpu_reproduces_reference_vtk(mf):
    """configs[4] at the reference's size: 40 x 40 quad-8, 14 883 DOF, Re = 1000, ten load steps, with the script's own
    solver choice cgs2! (2D_Script.jl:97: maxiter = 5000, max_pass = 20)."""
    import torch
    from oracle import cavity

    z = np.load(os.path.join(GOLD, "cavity_flow_Re1000.npz"))
    od = cavity.build_cavity(40, Cb=8.0)  # only used for mesh + term lists + the parameter formulas of the script
    gd = _gpu_domain(mf, od, "Serendipity", 2, 5)
    gd.converge_tol = 1e-5
    stats = []

    def solver(g):
        dx, st = mf.iterative_Solve(g.A, g.K_total, g.residue, 1e-3 * g.converge_tol, Sv_func=mf.cgs2_, maxiter=5000, max_pass=20)
        stats.append(st)
        return dx

    gd.linear_solver = solver
    P = od.params
    n = od.mesh.ncp
    u_st = 1000.0 / P["L"] * P["mu"] / P["rho"]
    od.controlpoints["u1"], od.controlpoints["u2"] = np.zeros(n), np.zeros(n)
    for i in range(1, 11):
        cavity.set_step_parameters(od, u_st * i / 10)  # host-side parameter formulas (2D_Script.jl:122-127)
        for k in ("uw1", "uw2", "taum", "tauc"):
            gd.controlpoints[k] = torch.tensor(od.controlpoints[k], device="cuda")
        gd.dt = od.dt
        hist = gd.update_OneStep(max_iter=6)
        assert hist[-1] < 1e-5, (i, hist)
        x = gd.x.cpu().numpy()
        od.controlpoints["u1"], od.controlpoints["u2"] = x[n:2 * n], x[2 * n:3 * n]  # dessemble_X!
    assert all(s.converged for s in stats)
    d, idx = cKDTree(od.mesh.coords).query(z["xy"])
    assert np.abs(x[n:2 * n][idx] - z["u1"]).max() < 5e-4
    assert np.abs(x[2 * n:][idx] - z["u2"]).max() < 5e-4
    pa, pb = x[:n][idx], z["p"]
    assert np.abs((pa - pa.mean()) - (pb - pb.mean())).max() < 5e-4 * (pb.max() - pb.min())


def test_cantilever_hex20_on_gpu_reproduces_reference_vtk(mf):
    """examples/linear_elasticity/cantilever/3D_Script.jl on the generic GPU path: hex-20 serendipity, 3 fields (9 sparse
    blocks, 21 linear gradient launches), nodal symmetric-tensor externals, facet normals, the script's idrs!(s = 8)."""
    import torch
    from oracle import cantilever as cl

    z = np.load(os.path.join(GOLD, "cantilever_hex20.npz"))
    od = cl.build_cantilever()
    od.linear_solver = cl.lu
    gd = _gpu_domain(mf, od, "Serendipity", 2, 5)
    gd.converge_tol = od.converge_tol
    stats = []

    def solver(g):
        # globalfield.converge_tol itself, like the reference (02_Preconditioner.jl:33): with K ~ 1e11 the FP64 floor of
        # ||r||/sqrt(n) is ~1e-7, so a tighter absolute tolerance would be unreachable
        dx, st = mf.iterative_Solve(g.A, g.K_total, g.residue, 0.5 * g.converge_tol, Sv_func=mf.idrs_, maxiter=2000, max_pass=20, s=8)
        stats.append(st)
        return dx

    gd.linear_solver = solver
    n = od.mesh.ncp
    for case in (1, 2, 3):
        cl.set_load(od, case)
        for k, v in od.controlpoints.items():
            gd.controlpoints[k] = torch.tensor(v, device="cuda")
        ho = od.update_one_step()
        hg = gd.update_OneStep()
        assert hg[-1] < gd.converge_tol and np.isclose(hg[0], ho[0], rtol=1e-9)
        got = gd.x.cpu().numpy()
        assert np.abs(got - od.x).max() <= 1e-7 * np.abs(od.x).max()
    assert all(s.converged for s in stats)
    d, idx = cKDTree(od.mesh.coords).query(z["xyz"])
    scale = np.abs(z["d2"]).max()
    for f, nm in enumerate(("d1", "d2", "d3")):
        assert np.abs(got[f * n:(f + 1) * n][idx] - z[nm]).max() < 1e-6 * scale, nm


@pytest.mark.parametrize("dissipative", [True, False])
def test_transient_thermal_generalised_alpha_matches_oracle(mf, dissipative):
    """3D_Script_Dynamics.jl's weak form (-C Bilinear(T, T{;t}) + conduction + source, convective boundary) on hex-8,
    max_time_level = 1: K_params, predictor, x*/dx updates of 04_Time_Domain.jl:9-49 over several steps."""
    import torch
    from oracle import fem, mesh as om, problems, reference_element as re_, solvers

    disc = re_.initialize_classical_element(3, "CUBE", 1, 1, 3)
    msh = om.lattice_mesh((1.0, 0.8, 0.6), (5, 4, 3), disc)
    fac = om.boundary_facets_structured((1.0, 0.8, 0.6), (5, 4, 3), 3)
    T0 = 293.15
    od = fem.FEMDomain(msh, disc, 1, problems.thermal_domain(3, 0.6, C=4.184e3), [(fac, problems.thermal_convection(25.0, T0))],
                       max_time_level=1, dissipative=dissipative)
    od.controlpoints["s"] = 1600.0 * (1.0 + msh.coords[:, 0])
    od.controlpoints["T"] = np.full(msh.ncp, T0)
    od.assemble_x([("T", 0, 0)])
    od.dt = 40.0
    od.converge_tol = 1e-8
    od.linear_solver = lambda d: solvers.solver_lu_cpu(d.pattern.rowptr, d.pattern.colidx, d.K_total, d.residue)
    gd = _gpu_domain(mf, od, "Lagrange", 1, 3)
    gd.controlpoints["s"] = torch.tensor(od.controlpoints["s"], device="cuda")
    gd.controlpoints["T"] = torch.tensor(od.controlpoints["T"], device="cuda")
    gd.assemble_X([("T", 0, 0)])
    gd.dt, gd.converge_tol = od.dt, od.converge_tol
    gd.linear_solver = lambda g: mf.iterative_Solve(g.A, g.K_total, g.residue, 1e-4 * g.converge_tol, Sv_func=mf.idrs_, maxiter=500,
                                                    max_pass=10, s=8)[0]
    for step in range(4):
        ho, hg = od.update_one_step(), gd.update_OneStep()
        assert len(ho) == len(hg) and np.isclose(hg[0], ho[0], rtol=1e-9)
        assert np.allclose(gd.K_params, od.time.K_params)
        got = gd.x.cpu().numpy()
        n = msh.ncp
        assert np.abs(got[:n] - od.x[:n]).max() <= 1e-9 * np.abs(od.x[:n]).max()
        assert np.abs(got[n:] - od.x[n:]).max() <= 1e-8 * np.abs(od.x[n:]).max()
    assert od.x[:n].max() > T0 + 1.0  # it did heat up


def test_elastodynamics_two_time_levels_matches_oracle(mf):
    """max_time_level = 2: -Bilinear(eps, sigma) - Bilinear(d{i}, rho (c d{i;t} + d{i;t,t})) (J2Plasticity.jl:59 without the
    plasticity), penalty-fixed end, suddenly applied traction: three implicit steps against the oracle."""
    import torch
    from oracle import fem, mesh as om, problems, reference_element as re_, solvers

    size, nel = (2.0, 0.5, 0.5), (6, 2, 2)
    disc = re_.initialize_classical_element(3, "CUBE", 1, 1, 3)
    msh = om.lattice_mesh(size, nel, disc)
    fac = om.boundary_facets_structured(size, nel, 3)
    c = fac.centroid
    left, right = np.abs(c[:, 0]) < 1e-9, np.abs(c[:, 0] - size[0]) < 1e-9
    E, nu, rho = 100.0, 0.3, 2.0
    lam, mu = E * nu / ((1 + nu) * (1 - 2 * nu)), E / (2 * (1 + nu))
    sig = [[0.0, 0.3, 0.0], [0.3, 0.0, 0.0], [0.0, 0.0, 0.0]]
    od = fem.FEMDomain(msh, disc, 3, problems.merge(problems.elasticity_domain(3, lam, mu), problems.elasticity_inertia(3, rho, 0.4)),
                       [(fac.select(left), problems.elasticity_penalty(3, 1000.0 * E)), (fac.select(right), problems.elasticity_traction(3, sig))],
                       max_time_level=2)
    od.dt = 0.05
    od.converge_tol = 1e-9
    od.linear_solver = lambda d: solvers.solver_lu_cpu(d.pattern.rowptr, d.pattern.colidx, d.K_total, d.residue)
    gd = _gpu_domain(mf, od, "Lagrange", 1, 3)
    gd.dt, gd.converge_tol = od.dt, od.converge_tol
    gd.linear_solver = lambda g: mf.iterative_Solve(g.A, g.K_total, g.residue, 1e-3 * g.converge_tol, Sv_func=mf.bicgstabl_GS_,
                                                    maxiter=2000, max_pass=10, s=4)[0]
    for step in range(3):
        ho, hg = od.update_one_step(), gd.update_OneStep()
        assert np.isclose(hg[0], ho[0], rtol=1e-9) and hg[-1] < gd.converge_tol
        got = gd.x.cpu().numpy()
        n = od.basicfield_size
        for lvl in range(3):
            ref = od.x[lvl * n:(lvl + 1) * n]
            assert np.abs(got[lvl * n:(lvl + 1) * n] - ref).max() <= 1e-7 * np.abs(ref).max(), (step, lvl)
    assert np.abs(od.x[:n]).max() > 1e-6


@pytest.mark.parametrize("case", ["cavity", "cantilever", "thermal_hex8_coloured"])
def test_batched_operators_equal_the_per_term_call_sequence(mf, case):
    """mfem_op_{var,kval,res}_batch (one launch per integration domain) against the reference's one-launch-per-term
    sequence through mfem_op_{var,kval,res}: same K_linear, K_total and residue to round-off (the order in which the terms
    of a block are summed differs)."""
    import torch
    from oracle import cantilever as cl, cavity, fem, mesh as om, problems, reference_element as re_

    colours = None
    if case == "cavity":
        od = cavity.build_cavity(8, Cb=8.0)
        od.controlpoints["u1"], od.controlpoints["u2"] = np.zeros(od.mesh.ncp), np.zeros(od.mesh.ncp)
        cavity.set_step_parameters(od, 0.3)
        args = ("Serendipity", 2, 5)
    elif case == "cantilever":
        od = cl.build_cantilever(ne_x=6, e_number=2)
        cl.set_load(od, 3)
        args = ("Serendipity", 2, 5)
    else:
        disc = re_.initialize_classical_element(3, "CUBE", 1, 1, 3)
        n = (5, 4, 3)
        msh = om.lattice_mesh((1.0, 0.8, 0.6), n, disc)
        fac = om.boundary_facets_structured((1.0, 0.8, 0.6), n, 3)
        od = fem.FEMDomain(msh, disc, 1, problems.thermal_domain(3, 0.6, alpha=2.0, Tenv=300.0), [(fac, problems.thermal_convection(25.0, 293.15, 0.7, 5.669e-8))])
        od.controlpoints["s"] = 1600.0 * (1.0 + msh.coords[:, 1])
        eg = np.meshgrid(*[np.arange(k) for k in n], indexing="ij")
        colours = ((eg[0] & 1) + 2 * (eg[1] & 1) + 4 * (eg[2] & 1)).ravel()
        args = ("Lagrange", 1, 3)
    doms = []
    rng = np.random.default_rng(5)
    x0 = rng.standard_normal(od.x.size) * 0.1 + (300.0 if case == "thermal_hex8_coloured" else 0.0)
    for batched in (True, False):
        from metafem_jl_amd import element, generic as G

        space = element.classical_space(od.disc.dim, *args)
        bnd = [(f.element_ID, f.element_eindex, _to_product_wf(mf, wf)) for f, wf in od.boundaries]
        gd = G.GenericDomain(mf.default_context(), space, od.mesh.coords, od.mesh.cp_ids, od.n_fields, _to_product_wf(mf, od.domain_wf), bnd,
                             element_colours=colours, batched=batched)
        for k, v in od.controlpoints.items():
            gd.controlpoints[k] = torch.tensor(np.asarray(v, dtype=np.float64), device="cuda")
        gd.dt = od.dt
        gd.x.copy_(torch.tensor(x0, device="cuda"))
        gd.update_Time()
        gd.initialize_dx()
        gd.K_linear_func()
        gd.update_x_star()
        gd.K_nonlinear_func()
        doms.append(gd)
    a, b = doms
    for name in ("K_linear", "K_total", "residue"):
        va, vb = getattr(a, name).cpu().numpy(), getattr(b, name).cpu().numpy()
        assert np.abs(va).max() > 0
        assert np.abs(va - vb).max() <= 1e-12 * np.abs(vb).max(), name


@pytest.mark.parametrize("coloured", [False, True, "rows"])
@pytest.mark.parametrize("case", ["cavity", "cantilever", "thermal_hex8", "thermal_hex27", "tet10"])
def test_fused_mesh_assembly_equals_the_operator_path(mf, case, coloured):
    """mfem_mesh_assemble_elements / _facets (geometry on the fly + all constant-coefficient terms of a domain in one launch)
    against the stored-table operator path (mfem_update_basic_* + mfem_op_kval_batch): same K_linear to round-off, with
    atomics, with colour batches and in the row-owner form (mfem_mesh_assemble_elements_rows, run twice: bitwise equal); quad-8 (2-D, 3 fields), hex-20 (3 fields, penalty facets), hex-8 (mass + gradient
    words, convection facets), tet-10 (slanted facets)."""
    import torch
    from metafem_jl_amd import element, generic as G, mesh as pm
    from oracle import cantilever as cl, cavity, fem, mesh as om, problems, reference_element as re_

    shape = "CUBE"
    if case == "cavity":
        od = cavity.build_cavity(8, Cb=8.0)
        od.controlpoints["u1"], od.controlpoints["u2"] = np.zeros(od.mesh.ncp), np.zeros(od.mesh.ncp)
        cavity.set_step_parameters(od, 0.3)
        args = ("Serendipity", 2, 5)
    elif case == "cantilever":
        od = cl.build_cantilever(ne_x=6, e_number=2)
        cl.set_load(od, 3)
        args = ("Serendipity", 2, 5)
    elif case in ("thermal_hex8", "thermal_hex27"):
        order = 1 if case == "thermal_hex8" else 2  # (hex-27: 27 nodes x 27 Gauss points -- the staged form of the row-owner kernel with two waves per workgroup)
        disc = re_.initialize_classical_element(3, "CUBE", order, 1, 3 if order == 1 else 5)
        n = (5, 4, 3) if order == 1 else (3, 2, 2)
        msh = om.lattice_mesh((1.0, 0.8, 0.6), n, disc)
        msh.coords[:, 0] += 0.05 * np.sin(3.0 * msh.coords[:, 1]) * msh.coords[:, 2]  # non-affine elements
        fac = om.boundary_facets_structured((1.0, 0.8, 0.6), n, 3)
        od = fem.FEMDomain(msh, disc, 1, problems.thermal_domain(3, 0.6, alpha=2.0, Tenv=300.0), [(fac, problems.thermal_convection(25.0, 293.15))])
        od.controlpoints["s"] = 1600.0 * (1.0 + msh.coords[:, 1])
        args = ("Lagrange", order, 3 if order == 1 else 5)
    else:
        shape = "SIMPLEX"
        space_h = element.classical_space(3, "Serendipity", 2, 5, shape="SIMPLEX")
        vert, conn = pm.make_Brick((1.0, 0.7, 0.5), (3, 2, 2), shape="SIMPLEX")
        vert = vert + 0.03 * np.sin(5.0 * vert[[1, 2, 0], :])
        m = pm.mesh_Classical(vert, conn, space_h)
        f = pm.get_BoundaryMesh(m)
    doms = []
    rows = coloured == "rows"
    coloured = coloured is True
    for fused in ((True, True, False) if rows else (True, False)):
        if case == "tet10":
            from metafem_jl_amd import physics

            space = space_h
            gd = G.GenericDomain(mf.default_context(), space, m.coords, m.cp_ids, 1, physics.thermal_domain(3, 0.6),
                                 [(f.element_ID, f.element_eindex, physics.thermal_convection(25.0, 293.15))],
                                 element_colours="auto" if coloured else None, fused=fused, row_owner=rows)
            gd.controlpoints["s"] = torch.full((m.ncp,), 1600.0, dtype=torch.float64, device="cuda")
        else:
            space = element.classical_space(od.disc.dim, *args)
            bnd = [(ff.element_ID, ff.element_eindex, _to_product_wf(mf, wf)) for ff, wf in od.boundaries]
            gd = G.GenericDomain(mf.default_context(), space, od.mesh.coords, od.mesh.cp_ids, od.n_fields, _to_product_wf(mf, od.domain_wf), bnd,
                                 element_colours="auto" if coloured else None, fused=fused, row_owner=rows)
            for k, v in od.controlpoints.items():
                gd.controlpoints[k] = torch.tensor(np.asarray(v, dtype=np.float64), device="cuda")
            gd.dt = od.dt
        gd.update_Time()
        gd.K_linear_func()
        doms.append(gd.K_linear.cpu().numpy())
        if len(doms) == 1:
            gd_rows = gd
    a, b = doms[0], doms[-1]
    assert np.abs(b).max() > 0
    assert np.abs(a - b).max() <= 1e-12 * np.abs(b).max()
    if rows and case != "cavity":  # fixed summation order (the facet terms of these cases hit disjoint entries per launch... or are coloured)
        # the gather by node (round 6: several nodes per wave, a node's field rows together) against the gather by row
        from metafem_jl_amd import _lib

        _lib.lib.mfem_debug_set_mesh_gather_rows(1)
        try:
            gd_rows.K_linear_func()
            by_row = gd_rows.K_linear.cpu().numpy()
        finally:
            _lib.lib.mfem_debug_set_mesh_gather_rows(0)
        assert np.abs(by_row - doms[0]).max() <= 1e-14 * np.abs(doms[0]).max()  # (the facets add with atomics: bitwise is tests/test_gpu_u20.py, elements alone)


@pytest.mark.parametrize("case", ["cantilever", "hex27_thermal"])
def test_wave_forms_of_the_batched_operators_on_small_meshes(mf, case):
    """mfem_op_var_batch / _res_batch in their wave-per-item forms (csrc/ops.hip, round 6; default from 256 items) forced on meshes of a few elements
    (knob value 2): hex-20 elasticity with penalty facets (the cantilever pin's domain) and a hex-27 Lagrange thermal domain (27 nodes, 27 Gauss points:
    two lane groups) -- the residual of K_nonlinear_func at a random x* against the oracle's term-by-term one."""
    import torch
    from metafem_jl_amd import _lib
    from oracle import cantilever as cl, fem, mesh as om, problems, reference_element as re_

    if case == "cantilever":
        od = cl.build_cantilever(ne_x=6, e_number=2)
        cl.set_load(od, 3)
        args = ("Serendipity", 2, 5)
    else:
        disc = re_.initialize_classical_element(3, "CUBE", 2, 1, 5)
        n = (3, 2, 2)
        msh = om.lattice_mesh((1.0, 0.8, 0.6), n, disc)
        msh.coords[:, 0] += 0.05 * np.sin(3.0 * msh.coords[:, 1]) * msh.coords[:, 2]
        fac = om.boundary_facets_structured((1.0, 0.8, 0.6), n, 3)
        od = fem.FEMDomain(msh, disc, 1, problems.thermal_domain(3, 0.6), [(fac, problems.thermal_convection(25.0, 293.15))])
        od.controlpoints["s"] = 1600.0 * (1.0 + msh.coords[:, 1])
        args = ("Lagrange", 2, 5)
    rng = np.random.default_rng(5)
    xs = rng.standard_normal(od.n_fields * od.mesh.ncp)
    od.update_time()
    od.K_linear_func()
    od.x_star[:xs.size] = xs
    od.K_nonlinear_func()
    try:
        _lib.lib.mfem_debug_set_op_wave_forms(2)
        gd = _gpu_domain(mf, od, *args)
        for k, v in od.controlpoints.items():
            gd.controlpoints[k] = torch.tensor(np.asarray(v, dtype=np.float64), device="cuda")
        gd.dt = od.dt
        gd.update_Time()
        gd.K_linear_func()
        gd.x_star[:xs.size] = torch.tensor(xs, device="cuda")
        gd.K_nonlinear_func()
        got = gd.residue.cpu().numpy()
    finally:
        _lib.lib.mfem_debug_set_op_wave_forms(1)
    assert np.abs(od.residue).max() > 0
    assert np.abs(got - od.residue).max() <= 1e-11 * np.abs(od.residue).max()

