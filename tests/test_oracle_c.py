"""The C/OpenMP restatement (cpu_baseline) agrees with the numpy oracle (CPU)."""
import numpy as np
import pytest

from oracle import cport, fem, mesh as om, problems, reference_element as re_, solvers


@pytest.fixture(scope="module")
def pair():
    n, x = (6, 5, 4), (1.0, 1.0, 1.0)
    c = cport.CThermal(n, x=x).setup()
    fac = om.boundary_facets_structured(x, n, 3)
    od = fem.FEMDomain(c.mesh, c.disc, 1, problems.thermal_domain(3, 0.6), [(fac, problems.thermal_convection(25.0, 293.15))])
    od.controlpoints["s"] = np.full(c.mesh.ncp, 1600.0)
    od.update_time()
    return c, od


def test_pattern_and_slots(pair):
    c, od = pair
    assert np.array_equal(c.rowptr, od.pattern.rowptr) and np.array_equal(c.colidx, od.pattern.colidx)
    slots = c.slots.reshape(c.mesh.nel, 8, 8).transpose(2, 1, 0)  # [a, b, e]
    assert np.array_equal(slots, od.pattern.sparse_ids_by_el((0, 0)))


def test_geometry_tables(pair):
    c, od = pair
    iv = c.ivals.reshape(c.mesh.nel, 4, 8, 8).transpose(3, 2, 1, 0)  # [q, a, s, e]
    assert np.allclose(iv, od.elgeo.integral_vals, rtol=0, atol=1e-14)
    assert np.allclose(c.w.reshape(c.mesh.nel, 8).T, od.elgeo.integral_weights, rtol=1e-15)


def test_operators_K_and_R(pair):
    c, od = pair
    c.K_linear_func()
    od.K_linear_func()
    assert np.abs(c.K - od.K_linear).max() <= 1e-14 * np.abs(od.K_linear).max()
    c.xstar[:] = 300 + np.random.default_rng(0).standard_normal(c.mesh.ncp)
    od.x_star[:] = c.xstar
    c.K_nonlinear_func()
    od.K_nonlinear_func()
    assert np.abs(c.residue - od.residue).max() <= 1e-13 * np.abs(od.residue).max()


def test_cg_iterates(pair):
    c, od = pair
    c.K_linear_func(); od.K_linear_func()
    c.xstar[:] = 0.0; od.x_star[:] = 0.0
    c.K_nonlinear_func(); od.K_nonlinear_func()
    x, it, res = c.solve_cg(1e-10, 1000)
    info = solvers.SolveInfo()
    ref = solvers.solve_cg_jacobi(od.pattern.rowptr, od.pattern.colidx, od.K_total, od.residue, 1e-10, 1000, info=info)
    assert it == info.iters and res <= 1e-10
    assert np.abs(x - ref).max() <= 1e-12 * np.abs(ref).max()
    _, itf, _ = c.solve_cg(0.0, 7, fixed=True)
    assert itf == 7
