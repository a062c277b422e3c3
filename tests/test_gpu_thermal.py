"""GPU parity: structured hex-8 thermal path (pattern, K, residual, CG solve, Newton step) vs the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

K_COND, H, TENV, SRC = 0.6, 25.0, 293.15, 1600.0  # examples/thermal_conduction/3D_Script.jl:21-25,56


def _oracle_domain(x, n, itg_order=3, distort=None, faces=None):
    from oracle import fem, mesh as om, problems, reference_element as re_

    disc = re_.initialize_classical_element(3, "CUBE", 1, 1, itg_order)
    msh = om.lattice_mesh(x, n, disc)
    if distort is not None:
        msh.coords[:] = distort(msh.coords)
    fac = om.boundary_facets_structured(x, n, 3)
    if faces is not None:
        fac = fac.select(np.isin(fac.element_eindex, faces))
    od = fem.FEMDomain(msh, disc, 1, problems.thermal_domain(3, K_COND), [(fac, problems.thermal_convection(H, TENV))])
    od.controlpoints["s"] = np.full(msh.ncp, SRC)
    return od


def _distort(c):
    out = c.copy()
    out[:, 0] += 0.03 * np.sin(3 * c[:, 1]) * np.cos(2 * c[:, 2])
    out[:, 1] += 0.02 * np.sin(2 * c[:, 0] + c[:, 2])
    out[:, 2] += 0.025 * c[:, 0] * c[:, 1]
    return out


@pytest.mark.parametrize("n", [(1, 1, 1), (2, 1, 3), (4, 4, 4), (7, 5, 3)])
def test_pattern_is_the_oracle_csr(mf, n):
    x = (1.0, 2.0, 0.5)
    od = _oracle_domain(x, n)
    A = mf.make_Brick(x, n).pattern(1)
    assert A.n == od.pattern.n and A.nnz == od.pattern.nnz
    assert np.array_equal(A.rowptr.cpu().numpy(), od.pattern.rowptr)
    assert np.array_equal(A.colidx.cpu().numpy(), od.pattern.colidx)


@pytest.mark.parametrize("n,itg,distorted,faces", [((1, 1, 1), 3, False, None), ((4, 4, 4), 3, False, None),
                                                   ((5, 3, 6), 3, True, None), ((3, 4, 2), 5, True, [0, 2, 4]),
                                                   ((6, 6, 6), 1, False, [5]), ((9, 8, 7), 7, True, None)])
def test_thermal_matrix_and_residual(mf, n, itg, distorted, faces):
    import torch

    x = (1.0, 1.5, 0.75)
    od = _oracle_domain(x, n, itg, _distort if distorted else None, faces)
    od.update_time()
    od.K_linear_func()
    rng = np.random.default_rng(1)
    od.x_star[:] = 300.0 + 20.0 * rng.standard_normal(od.basicfield_size)
    od.K_nonlinear_func()

    brick = mf.make_Brick(x, n, 1, itg)
    if distorted:
        for d in range(3):
            brick.coords_view(d).copy_(torch.tensor(od.mesh.coords[:, d], device="cuda"))
    bits = 0x3F if faces is None else sum(1 << f for f in faces)
    A = brick.pattern(1)
    K = brick.assemble_thermal(A, K_COND, H, TENV, bits).cpu().numpy()
    assert np.max(np.abs(K - od.K_linear)) <= 1e-13 * np.max(np.abs(od.K_linear))
    s = torch.full((brick.ncp,), SRC, dtype=torch.float64, device="cuda")
    R = brick.residual_thermal(torch.tensor(od.x_star, device="cuda"), K_COND, H, TENV, bits, s=s).cpu().numpy()
    assert np.max(np.abs(R - od.residue)) <= 1e-12 * np.max(np.abs(od.residue))


def test_cg_solve_matches_direct_solve(mf):
    """Linear problem: 1e-10 relative solution parity (BASELINE.json north_star) vs solver_LU_CPU == spsolve."""
    import torch
    from oracle import solvers

    x, n = (1.0, 1.0, 1.0), (12, 10, 8)
    od = _oracle_domain(x, n)
    od.converge_tol = 1e-9
    od.linear_solver = lambda d: solvers.solver_lu_cpu(d.pattern.rowptr, d.pattern.colidx, d.K_total, d.residue)
    od.update_one_step()

    dom = mf.ThermalDomain(mf.make_Brick(x, n), K_COND, H, TENV)
    dom.s.fill_(SRC)
    dom.converge_tol = 1e-9
    stats = []

    def solver(gf):
        dx, st = mf.iterative_Solve(gf.A, gf.K_total, gf.residue, 1e-13, Sv_func=mf.cg_, maxiter=2000, max_pass=3)
        stats.append(st)
        return dx

    dom.linear_solver = solver
    hist = dom.update_OneStep()
    got = dom.x.cpu().numpy()
    assert len(hist) == 2 and hist[1] < 1e-9  # linear: one solve, two residual evaluations
    assert stats[0].converged == 1
    assert np.max(np.abs(got - od.x)) <= 1e-10 * np.max(np.abs(od.x))


def test_cg_iterates_match_oracle_cg(mf):
    """Same algorithm, same arithmetic order up to reductions: iteration counts and iterates agree."""
    import torch
    from oracle import solvers

    od = _oracle_domain((1.0, 1.0, 1.0), (8, 8, 8))
    od.update_time()
    od.K_linear_func()
    od.update_x_star()
    od.K_nonlinear_func()
    info = solvers.SolveInfo()
    ref = solvers.solve_cg_jacobi(od.pattern.rowptr, od.pattern.colidx, od.K_total, od.residue, 1e-8, 1000, info=info)
    A = mf.FEM_SpMat_CSR(torch.tensor(od.pattern.rowptr, device="cuda"), torch.tensor(od.pattern.colidx, device="cuda"),
                         od.pattern.n)
    dx, st = mf.iterative_Solve(A, torch.tensor(od.K_total, device="cuda"), torch.tensor(od.residue, device="cuda"), 1e-8,
                                Sv_func=mf.cg_, maxiter=1000, max_pass=1, check_every=7)
    assert st.iterations == info.iters
    assert np.max(np.abs(dx.cpu().numpy() - ref)) <= 1e-9 * np.max(np.abs(ref))


def test_cg_fixed_iterations_mode(mf):
    import torch

    brick = mf.make_Brick((1.0, 1.0, 1.0), (16, 16, 16))
    A = brick.pattern(1)
    K = brick.assemble_thermal(A, K_COND, H, TENV, 0x3F)
    b = torch.ones(A.n, dtype=torch.float64, device="cuda")
    _, st = mf.iterative_Solve(A, K, b, 1e-30, Sv_func=mf.cg_, maxiter=37, max_pass=1, fixed_iterations=True)
    # 37 products of the iterations + the true residual after the pass; the first pass's r = b - A x0 needs none (x0 = 0: 02_Preconditioner.jl:45)
    assert st.iterations == 37 and st.passes == 1 and st.spmv_count == 37 + 1


@pytest.mark.parametrize("n,itg", [((9, 33, 17), 3), ((4, 16, 31), 3), ((3, 17, 5), 5)])
def test_sweep_kernels_equal_tile_kernels_and_keep_K_bitwise_symmetric(mf, n, itg):
    """The plane-sweep kernels (sum-factorised element integration, several (j, k) tiles and i segments) against the 4 x 4 x 8
    tile kernels with the stored-table arithmetic on a distorted brick; K must stay bitwise symmetric (the symmetric-sweep
    SpMV mirrors its lower diagonals only after checking exactly that)."""
    import torch
    import scipy.sparse as sp
    from metafem_jl_amd import _lib

    x = (1.0, 2.0, 1.5)
    brick = mf.make_Brick(x, n, 1, itg)
    rng = np.random.default_rng(3)
    for d in range(3):
        c = brick.coords_view(d)
        c.add_(torch.tensor(0.2 * min(x[i] / n[i] for i in range(3)) * (rng.random(c.numel()) - 0.5), device="cuda"))
    A = brick.pattern(1)
    xs = torch.tensor(300.0 + 20.0 * rng.standard_normal(A.n), device="cuda")
    s = torch.tensor(SRC * (1.0 + 0.1 * rng.standard_normal(A.n)), device="cuda")
    out = {}
    try:
        for variant in (1, 0):
            _lib.lib.mfem_debug_set_hex8_thermal(variant)
            K = brick.assemble_thermal(A, K_COND, H, TENV, 0x3F).cpu().numpy()
            R = brick.residual_thermal(xs, K_COND, H, TENV, 0x3F, s=s).cpu().numpy()
            R0 = brick.residual_thermal(xs, K_COND, H, TENV, 0x3F).cpu().numpy()  # no source vector
            out[variant] = (K, R, R0)
    finally:
        _lib.lib.mfem_debug_set_hex8_thermal(0)
    for got, ref in zip(out[0], out[1]):
        assert np.max(np.abs(got - ref)) <= 2e-13 * np.max(np.abs(ref))
    M = sp.csr_matrix((out[0][0], A.colidx.cpu().numpy(), A.rowptr.cpu().numpy()), shape=(A.n, A.n))
    D = (M - M.T).tocoo()
    assert D.nnz == 0 or np.max(np.abs(D.data)) == 0.0


@pytest.mark.parametrize("case", ["uniform", "sheared", "half_distorted"])
@pytest.mark.parametrize("itg", [3, 5])
def test_affine_element_shortcut_of_the_matrix_sweep(mf, case, itg):
    """Round 4: on an affine element (all eight nodes on the affine image of the reference cube to 16 ulp of the coordinates' magnitude) the sweep kernel takes
    Ke = sum_t g0[t] K6[t] -- one adjugate and the quadrature's reference integrals -- instead of the sum-factorised integration.  Against the oracle and
    against the general path (bit 2 of mfem_debug_set_hex8_thermal), with the rows staged through LDS and written per thread (bit 1): uniform brick,
    sheared brick (affine, full J), brick distorted where x > 0.5 (both paths inside one workgroup)."""
    import torch
    from metafem_jl_amd import _lib

    x, n = (1.0, 2.0, 0.5), (20, 17, 19)  # more than one 15 x 15 tile in j and k, tiles cut by the lattice
    od = _oracle_domain(x, n, itg_order=itg, faces=[])
    c = od.mesh.coords.copy()
    if case == "sheared":
        c = c @ np.array([[1.0, 0.3, -0.2], [0.1, 0.9, 0.25], [-0.15, 0.2, 1.1]]).T + np.array([0.3, -0.2, 0.1])
    elif case == "half_distorted":
        c = np.where(c[:, :1] > 0.5, _distort(c), c)
    od = _oracle_domain(x, n, itg_order=itg, distort=lambda _c: c, faces=[])
    od.update_time()
    od.K_linear_func()
    brick = mf.make_Brick(x, n, 1, itg)
    for d in range(3):
        brick.coords_view(d).copy_(torch.tensor(c[:, d], device="cuda"))
    A = brick.pattern(1)
    rng = np.random.default_rng(4)
    od.controlpoints["s"] = 100.0 * rng.standard_normal(od.mesh.ncp)
    od.x_star[:] = 300.0 + 10.0 * rng.standard_normal(od.basicfield_size)
    od.K_nonlinear_func()
    xs, ss = torch.tensor(od.x_star, device="cuda"), torch.tensor(od.controlpoints["s"], device="cuda")
    Ks, Rs = {}, {}
    try:
        for knob in (0, 4, 2, 6):
            _lib.lib.mfem_debug_set_hex8_thermal(knob)
            Ks[knob] = brick.assemble_thermal(A, K_COND, 0.0, TENV, 0).cpu().numpy()
            Rs[knob] = brick.residual_thermal(xs, K_COND, 0.0, TENV, 0, s=ss).cpu().numpy()  # the matrix-free residual sweep takes the shortcut too
    finally:
        _lib.lib.mfem_debug_set_hex8_thermal(0)
    rscale = np.abs(od.residue).max()
    for knob, R in Rs.items():
        assert np.abs(R - od.residue).max() <= 1e-12 * rscale, knob
    assert np.abs(Rs[0] - Rs[4]).max() <= 1e-13 * rscale
    scale = np.abs(od.K_linear).max()
    for knob, K in Ks.items():
        assert np.abs(K - od.K_linear).max() <= 2e-13 * scale, knob
    assert np.array_equal(Ks[0], Ks[2]) and np.array_equal(Ks[4], Ks[6])  # the write-out does not touch the values
    assert np.abs(Ks[0] - Ks[4]).max() <= 1e-13 * scale
    if case != "half_distorted":
        assert not np.array_equal(Ks[0], Ks[4])  # (another code path: other rounding)
    M = sp_csr(Ks[0], A)
    assert abs(M - M.T).max() <= 1e-13 * scale  # symmetric to round-off on either path


def sp_csr(K, A):
    import scipy.sparse as sp

    return sp.csr_matrix((K, A.colidx.cpu().numpy(), A.rowptr.cpu().numpy()), shape=(A.n, A.n))
