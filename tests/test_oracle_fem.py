"""Oracle self-consistency: patch tests, symmetry, manufactured solutions, solver agreement (CPU)."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import fem, mesh as om, problems, reference_element as re_, solvers

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _thermal(n, x=(1.0, 1.0, 1.0), order=1, itg=3, h=25.0, k=0.6, distort=False, src=1600.0):
    disc = re_.initialize_classical_element(3, "CUBE", order, 1, itg)
    msh = om.lattice_mesh(x, n, disc)
    if distort:
        c = msh.coords
        msh.coords = c + 0.02 * np.stack([np.sin(3 * c[:, 1]), np.sin(2 * c[:, 2]), c[:, 0] * c[:, 1]], axis=1)
    fac = om.boundary_facets_structured(x, n, 3)
    od = fem.FEMDomain(msh, disc, 1, problems.thermal_domain(3, k), [(fac, problems.thermal_convection(h, 293.15))])
    od.controlpoints["s"] = np.full(msh.ncp, src)
    od.update_time()
    od.K_linear_func()
    return od


@pytest.mark.parametrize("order,itg", [(1, 3), (2, 5)])
def test_thermal_K_is_symmetric_negative_definite(order, itg):
    od = _thermal((3, 2, 2), order=order, itg=itg, distort=True)
    A = solvers.csr(od.pattern.rowptr, od.pattern.colidx, od.K_linear, od.pattern.n).toarray()
    assert np.allclose(A, A.T, atol=1e-12 * np.abs(A).max())
    assert np.linalg.eigvalsh(A).max() < 0


def test_K_times_one_is_only_the_boundary_part():
    od = _thermal((3, 3, 3), h=0.0)
    A = solvers.csr(od.pattern.rowptr, od.pattern.colidx, od.K_linear, od.pattern.n)
    assert np.abs(A @ np.ones(od.pattern.n)).max() < 1e-12
    od = _thermal((3, 3, 3), h=25.0)
    A = solvers.csr(od.pattern.rowptr, od.pattern.colidx, od.K_linear, od.pattern.n)
    assert abs((A @ np.ones(od.pattern.n)).sum() + 25.0 * 6.0) < 1e-10  # -h * area


def test_lattice_mesh_equals_unstructured_mesh_up_to_numbering():
    disc = re_.initialize_classical_element(3, "CUBE", 2, 1, 5)
    a = om.lattice_mesh((1.0, 2.0, 1.5), (2, 2, 1), disc)
    vert, conn = om.make_brick((1.0, 2.0, 1.5), (2, 2, 1))
    b = om.mesh_classical(vert, conn, disc)
    assert a.ncp == b.ncp == 5 * 5 * 3
    # same element -> same node coordinates in basis order
    assert np.allclose(a.coords[a.cp_ids], b.coords[b.cp_ids], atol=1e-14)


def test_make_brick_ordering():
    vert, conn = om.make_brick((1.0, 1.0, 1.0), (2, 3, 4))
    assert vert.shape == (3, 3 * 4 * 5) and conn.shape == (8, 24)
    # node id = i*(n2+1)*(n3+1) + j*(n3+1) + k, k fastest (201_Helper_TM.jl:38-41)
    assert np.allclose(vert[:, 1], [0, 0, 0.25])
    # first element, counter-clockwise bottom then top (:43-51)
    assert list(conn[:, 0]) == [0, 20, 25, 5, 1, 21, 26, 6]


def test_linear_field_patch_test():
    """k lap(T) = 0 with T = a.x: the residual of the exact linear field vanishes at interior nodes."""
    od = _thermal((3, 3, 3), h=0.0, distort=True, src=0.0)
    c = od.mesh.coords
    od.x_star[:] = 2.0 + 3.0 * c[:, 0] - 1.5 * c[:, 1] + 0.5 * c[:, 2]
    od.K_nonlinear_func()
    p = 3
    interior = np.ones((p + 1,) * 3, bool)
    interior[[0, -1], :, :] = interior[:, [0, -1], :] = interior[:, :, [0, -1]] = False
    assert np.abs(od.residue[interior.ravel()]).max() < 1e-12


def test_newton_converges_in_one_solve_for_linear_problem():
    od = _thermal((4, 3, 3))
    od.t = 0.0
    od.converge_tol = 1e-9
    od.linear_solver = lambda d: solvers.solver_lu_cpu(d.pattern.rowptr, d.pattern.colidx, d.K_total, d.residue)
    hist = od.update_one_step()
    assert len(hist) == 2 and hist[1] < 1e-9
    assert od.x.min() > 293.15  # heated body above ambient


@pytest.mark.parametrize("name,kw", [("bicgstabl_gs", dict(s=2)), ("bicgstabl_gs", dict(s=4)), ("idrs", dict(s=4)),
                                     ("idrs", dict(s=8)), ("cgs2", {})])
def test_reference_krylov_bodies_reach_the_direct_solution(name, kw):
    od = _thermal((5, 4, 3), distort=True)
    od.update_x_star()
    od.K_nonlinear_func()
    ref = solvers.solver_lu_cpu(od.pattern.rowptr, od.pattern.colidx, od.K_total, od.residue)
    info = solvers.SolveInfo()
    x = solvers.iterative_solve(od.pattern.rowptr, od.pattern.colidx, od.K_total, od.residue, 1e-11,
                                Sv_func=getattr(solvers, name), maxiter=500, max_pass=10, info=info, **kw)
    assert info.res < 1e-11
    assert np.abs(x - ref).max() <= 1e-9 * np.abs(ref).max()


def test_cg_equals_direct_and_runs_on_negative_definite_K():
    od = _thermal((5, 4, 3))
    od.update_x_star()
    od.K_nonlinear_func()
    ref = solvers.solver_lu_cpu(od.pattern.rowptr, od.pattern.colidx, od.K_total, od.residue)
    x = solvers.solve_cg_jacobi(od.pattern.rowptr, od.pattern.colidx, od.K_total, od.residue, 1e-12, 1000)
    assert np.abs(x - ref).max() <= 1e-10 * np.abs(ref).max()


def test_right_jacobi_scales_columns_and_unscales_solution():
    rng = np.random.default_rng(0)
    A = sp.random(40, 40, 0.2, random_state=1, format="csr") + sp.diags(rng.uniform(2, 5, 40))
    A = A.tocsr()
    A.sort_indices()
    B = A.copy()
    P = solvers.pr_jacobi(B)
    assert np.allclose(B.toarray(), A.toarray() / np.abs(A.diagonal())[None, :])
    assert np.allclose(P(np.ones(40)), 1.0 / np.abs(A.diagonal()))


def test_elasticity_rigid_body_modes_and_symmetry():
    E, nu = 1.0, 0.3
    lam, mu = E * nu / ((1 + nu) * (1 - 2 * nu)), E / (2 * (1 + nu))
    disc = re_.initialize_classical_element(3, "CUBE", 1, 1, 3)
    msh = om.lattice_mesh((2.0, 1.0, 1.0), (3, 2, 2), disc)
    od = fem.FEMDomain(msh, disc, 3, problems.elasticity_domain(3, lam, mu), [])
    od.update_time()
    od.K_linear_func()
    assert len(od.domain_wf.linear_gradients) == 21
    assert od.pattern.blocks == [(i, j) for i in range(3) for j in range(3)]
    A = solvers.csr(od.pattern.rowptr, od.pattern.colidx, od.K_linear, od.pattern.n)
    D = A.toarray()
    assert np.allclose(D, D.T, atol=1e-13)
    n = msh.ncp
    c = msh.coords
    for mode in range(6):
        u = np.zeros(3 * n)
        if mode < 3:
            u[mode * n:(mode + 1) * n] = 1.0
        else:  # rotation about axis mode-3
            a, b = [(1, 2), (2, 0), (0, 1)][mode - 3]
            u[a * n:(a + 1) * n] = -c[:, b]
            u[b * n:(b + 1) * n] = c[:, a]
        assert np.abs(A @ u).max() < 1e-12  # field-major DOF layout (A9)


def test_committed_oracle_fixtures_are_reproducible():
    z = np.load(os.path.join(GOLD, "oracle_thermal_hex8_4x4x4.npz"))
    od = _thermal(tuple(z["n"]), x=tuple(z["x"]))
    assert np.array_equal(od.pattern.rowptr, z["rowptr"]) and np.array_equal(od.pattern.colidx, z["colidx"])
    assert np.allclose(od.K_linear, z["K"], rtol=0, atol=1e-14 * np.abs(z["K"]).max())


def test_generalised_alpha_static_defaults():
    od = _thermal((2, 2, 2))
    assert list(od.time.K_params) == [1.0] and list(od.time.beta_params) == [1.0]  # A11: max_time_level = 0


def test_fem_rand_range_and_determinism():
    a = solvers.fem_rand(0x5EED, 0, 1000)
    assert np.array_equal(a, solvers.fem_rand(0x5EED, 0, 1000))
    assert a.min() >= 0 and a.max() < 1 and abs(a.mean() - 0.5) < 0.05
    assert not np.array_equal(a, solvers.fem_rand(0x5EED, 1, 1000))


# ---- generalised-alpha transient path (04_Time_Domain.jl:9-49), the 3D_Script_Dynamics.jl weak form ----
def test_transient_lumped_heating_is_exact_under_backward_euler():
    """No boundary terms, uniform source: C dT/dt = s  =>  T(t) = T0 + s t / C, which backward Euler (gamma = alpha = 1)
    integrates exactly; the mass term also makes the otherwise singular conduction matrix invertible."""
    disc = re_.initialize_classical_element(3, "CUBE", 1, 1, 3)
    msh = om.lattice_mesh((1.0, 0.5, 0.25), (3, 2, 2), disc)
    Cv, s0, T0 = 4.184e3, 1600.0, 293.15
    dom = fem.FEMDomain(msh, disc, 1, problems.thermal_domain(3, 0.6, C=Cv), [], max_time_level=1)
    dom.controlpoints["s"] = np.full(msh.ncp, s0)
    dom.controlpoints["T"] = np.full(msh.ncp, T0)
    dom.assemble_x([("T", 0, 0)])
    dom.dt = 0.5
    dom.linear_solver = lambda d: solvers.solver_lu_cpu(d.pattern.rowptr, d.pattern.colidx, d.K_total, d.residue)
    for step in range(1, 4):
        hist = dom.update_one_step()
        assert hist[-1] < dom.converge_tol
        n = msh.ncp
        assert np.allclose(dom.x[:n], T0 + s0 * dom.t / Cv, rtol=1e-12)
        assert np.allclose(dom.x[n:], s0 / Cv, rtol=1e-9)  # the rate level
    assert np.isclose(dom.t, 1.5)
    assert np.allclose(dom.time.K_params, [1.0, 1.0 / dom.dt])


def test_transient_conduction_mode_decays_at_the_backward_euler_rate():
    """Insulated bar, T0 = cos(pi x): the consistent-mass hex-8 semi-discretisation keeps the nodal cosine an
    eigenvector, so one backward-Euler step multiplies it by 1 / (1 + dt*lam_h) with the 1-D linear-element
    eigenvalue lam_h = (k/C) * (6/h^2) * (1 - c) / (2 + c), c = cos(pi h)."""
    nx, k, Cv, dt = 16, 0.6, 50.0, 0.3
    disc = re_.initialize_classical_element(3, "CUBE", 1, 1, 3)
    msh = om.lattice_mesh((1.0, 0.2, 0.2), (nx, 1, 1), disc)
    dom = fem.FEMDomain(msh, disc, 1, problems.thermal_domain(3, k, C=Cv), [], max_time_level=1)
    dom.controlpoints["s"] = np.zeros(msh.ncp)
    mode = np.cos(np.pi * msh.coords[:, 0])
    dom.controlpoints["T"] = mode.copy()
    dom.assemble_x([("T", 0, 0)])
    dom.dt = dt
    dom.converge_tol = 1e-12
    dom.linear_solver = lambda d: solvers.solver_lu_cpu(d.pattern.rowptr, d.pattern.colidx, d.K_total, d.residue)
    h = 1.0 / nx
    c = np.cos(np.pi * h)
    lam = (k / Cv) * (6.0 / h ** 2) * (1 - c) / (2 + c)
    amp = 1.0
    for _ in range(3):
        dom.update_one_step()
        amp /= 1.0 + dt * lam
        assert np.allclose(dom.x[:msh.ncp], amp * mode, atol=1e-10)


def test_thermoelastic_example_reaches_the_textbook_thermal_bending():
    """examples/thermal_elasticity/themal_hypo_elasticity.jl (oracle/thermoelastic.py): four coupled fields, one time level, thermal strain in the dual
    and the base word of the elasticity form, convection to a nodal environment temperature.  The reference holds no numbers for it; the steady state
    is the stress-free thermal bending of a beam: linear temperature 200 -> 100 across the height, tip deflection alpha g l^2 / 2 = 0.25, mean elongation
    alpha T_mean l = 0.075.  Coarse mesh (5 x 2 x 2 hex-20), the script's stopping rule."""
    from oracle import thermoelastic as te

    dom = te.build(e_number=2)
    assert dom.n_fields == 4 and dom.max_time_level == 1 and (3, 0) in dom.pattern.blocks and (0, 3) in dom.pattern.blocks  # the coupling blocks
    log, hists = te.run(dom)
    assert len(log) < 120 and log[-1][0] < 1e-4 and log[-1][2] < 1e-2   # :125
    assert all(h[-1] < 1e-6 for h in hists)                              # every step's Newton loop reached the script's tolerance (the form is linear: 2 evaluations)
    n, c = dom.mesh.ncp, dom.mesh.coords
    tip = np.abs(c[:, 0] - 10.0) < 1e-9
    assert abs(dom.x[n:2 * n][tip].mean() - 0.25) < 2e-3 and abs(dom.x[:n][tip].mean() - 0.075) < 1e-3
    mid = np.abs(c[:, 0] - 5.0) < 0.3
    T = dom.x[3 * n:4 * n]
    assert abs(T[mid & (np.abs(c[:, 1]) < 1e-9)].mean() - 200.0) < 0.2 and abs(T[mid & (np.abs(c[:, 1] - 1.0) < 1e-9)].mean() - 100.0) < 0.2
    # K is not symmetric (the T row couples to grad d through -alpha sigma_mm, the d rows to T through the thermal strain: same constant, but the
    # viscous / capacity terms and the penalty sit on different blocks) -- the script's solver is bicgstabl_GS!(s = 8): one step with it
    dom2 = te.build(e_number=2)
    log2, _ = te.run(dom2, linear_solver=te.solver_of_the_script, max_steps=2, stop=False)
    assert np.abs(log2 - log[:2]).max() < 1e-6 * np.abs(log[:2]).max()
