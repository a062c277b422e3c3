"""GPU parity: fused hex-8 linear elasticity (3 DOF/node, field-major) vs the oracle's 21-term assembly."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")

E_MOD, NU = 1.0, 0.3
LAM, MU = E_MOD * NU / ((1 + NU) * (1 - 2 * NU)), E_MOD / (2 * (1 + NU))
TAU = 1000.0 * E_MOD
SIG = [[0.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 0.0]]  # sigma_22 = 1 on the y = L face


def _oracle(x, n, distort=False, itg=3):
    from oracle import fem, mesh as om, problems, reference_element as re_

    disc = re_.initialize_classical_element(3, "CUBE", 1, 1, itg)
    msh = om.lattice_mesh(x, n, disc)
    if distort:
        c = msh.coords
        msh.coords = c + 0.03 * np.stack([np.sin(2 * c[:, 1]), np.cos(2 * c[:, 2]) - 1, c[:, 0] * c[:, 1] / x[0]], axis=1)
    fac = om.boundary_facets_structured(x, n, 3)
    fixed = fac.select(fac.element_eindex == 4)  # x = 0  (face id 5)
    load = fac.select(fac.element_eindex == 3)   # y = L  (face id 4)
    od = fem.FEMDomain(msh, disc, 3, problems.elasticity_domain(3, LAM, MU),
                       [(fixed, problems.elasticity_penalty(3, TAU)), (load, problems.elasticity_traction(3, SIG))])
    od.update_time()
    od.K_linear_func()
    return od


@pytest.mark.parametrize("n,distort,itg", [((1, 1, 1), False, 3), ((3, 2, 2), False, 3), ((4, 3, 5), True, 3), ((3, 3, 2), True, 5),
                                           ((5, 6, 7), True, 3)])  # (336 control points: blocks of 32 that wrap lines and planes)
def test_elasticity_pattern_matrix_residual(mf, n, distort, itg):
    import torch

    x = (3.0, 1.0, 1.0)
    od = _oracle(x, n, distort, itg)
    rng = np.random.default_rng(2)
    od.x_star[:] = 0.01 * rng.standard_normal(od.basicfield_size)
    od.K_nonlinear_func()
    brick = mf.make_Brick(x, n, 1, itg)
    if distort:
        for d in range(3):
            brick.coords_view(d).copy_(torch.tensor(od.mesh.coords[:, d], device="cuda"))
    A = brick.pattern(3)
    assert np.array_equal(A.rowptr.cpu().numpy(), od.pattern.rowptr)
    assert np.array_equal(A.colidx.cpu().numpy(), od.pattern.colidx)
    K = brick.assemble_elasticity(A, LAM, MU, TAU, mf.FACE_BITS["x0"]).cpu().numpy()
    assert np.max(np.abs(K - od.K_linear)) <= 2e-13 * np.max(np.abs(od.K_linear))
    # fixed order of the accumulation steps, no atomics: a second assembly gives the same bits
    assert np.array_equal(K, brick.assemble_elasticity(A, LAM, MU, TAU, mf.FACE_BITS["x0"]).cpu().numpy())
    R = brick.residual_elasticity(torch.tensor(od.x_star, device="cuda"), LAM, MU, TAU, mf.FACE_BITS["x0"],
                                  mf.FACE_BITS["y1"], (0.0, 1.0, 0.0, 0.0, 0.0, 0.0)).cpu().numpy()
    assert np.max(np.abs(R - od.residue)) <= 1e-12 * np.max(np.abs(od.residue))


def test_against_committed_golden_fixture(mf):
    import torch

    z = np.load(os.path.join(GOLD, "oracle_elasticity_hex8_3x2x2.npz"))
    brick = mf.make_Brick(tuple(z["x"]), tuple(int(v) for v in z["n"]))
    A = brick.pattern(3)
    K = brick.assemble_elasticity(A, float(z["lam"]), float(z["mu"]), float(z["tau"]), mf.FACE_BITS["x0"])
    assert np.max(np.abs(K.cpu().numpy() - z["K"])) <= 2e-13 * np.abs(z["K"]).max()
    R0 = brick.residual_elasticity(torch.zeros(A.n, dtype=torch.float64, device="cuda"), float(z["lam"]), float(z["mu"]),
                                   float(z["tau"]), mf.FACE_BITS["x0"], mf.FACE_BITS["y1"], (0.0, 1.0, 0.0, 0.0, 0.0, 0.0))
    assert np.max(np.abs(R0.cpu().numpy() - z["R0"])) <= 1e-13 * np.abs(z["R0"]).max()
    # Newton step: solve K delta = R, d = -delta (04_Time_Domain.jl:76-79); linear => 1e-10 parity
    for sv, s in ((mf.idrs_, 8), (mf.bicgstabl_GS_, 2), (mf.cg_, 0)):
        dx, st = mf.iterative_Solve(A, K, R0, 1e-13 * float(np.linalg.norm(z["R0"])) / np.sqrt(A.n), Sv_func=sv, maxiter=3000,
                                    max_pass=10, s=s)
        assert st.converged == 1
        assert np.max(np.abs(-dx.cpu().numpy() - z["d"])) <= 1e-9 * np.abs(z["d"]).max(), sv


def test_sweep_residual_equals_the_per_point_kernel(mf):
    """The plane-sweep residual kernel (one sum-factorised integration per element) against the kernel that integrates an element once
    per adjacent control point, on a distorted mesh spanning several (j, k) tiles and plane segments: equal to round-off."""
    import torch
    from metafem_jl_amd import _lib

    x, n = (2.0, 1.0, 1.5), (37, 17, 33)
    brick = mf.make_Brick(x, n, 1, 3)
    cs = [brick.coords_view(d).clone() for d in range(3)]
    brick.coords_view(0).add_(0.004 * torch.sin(7 * cs[1]) * torch.cos(5 * cs[2]))
    brick.coords_view(1).add_(0.003 * torch.cos(6 * cs[0] + cs[2]))
    brick.coords_view(2).add_(0.005 * cs[0] * cs[1])
    A = brick.pattern(3)
    xs = 0.01 * (mf.FEM_rand(A.n, 11, 0) - 0.5)
    args = (xs, LAM, MU, TAU, mf.FACE_BITS["x0"] | mf.FACE_BITS["z1"], mf.FACE_BITS["y1"] | mf.FACE_BITS["x1"], (0.3, 1.0, -0.2, 0.1, 0.05, -0.4))
    R_sweep = brick.residual_elasticity(*args).clone()
    _lib.lib.mfem_debug_set_elasticity(2)
    try:
        R_point = brick.residual_elasticity(*args).clone()
    finally:
        _lib.lib.mfem_debug_set_elasticity(0)
    scale = float(R_point.abs().max())
    assert scale > 0 and float((R_sweep - R_point).abs().max()) <= 2e-13 * scale
    # a second call gives bitwise the same result (fixed summation order, no atomics)
    assert torch.equal(R_sweep, brick.residual_elasticity(*args))


def test_matrix_kernel_equals_the_row_owner_kernel(mf):
    """The default matrix kernel (thread per (control point, element), conflict-free accumulation steps in LDS, trilinear-map Jacobian) against
    the row-owner kernel that accumulates in global memory (table form of the Jacobian, other summation order), on a distorted mesh of 2 400
    blocks of 32 control points that wrap lines and planes, with penalty faces on two sides: equal to round-off."""
    import torch
    from metafem_jl_amd import _lib

    x, n = (2.0, 1.0, 1.5), (37, 43, 47)
    brick = mf.make_Brick(x, n, 1, 3)
    cs = [brick.coords_view(d).clone() for d in range(3)]
    brick.coords_view(0).add_(0.004 * torch.sin(7 * cs[1]) * torch.cos(5 * cs[2]))
    brick.coords_view(1).add_(0.003 * torch.cos(6 * cs[0] + cs[2]))
    brick.coords_view(2).add_(0.005 * cs[0] * cs[1])
    A = brick.pattern(3)
    faces = mf.FACE_BITS["x0"] | mf.FACE_BITS["z1"]
    K = brick.assemble_elasticity(A, LAM, MU, TAU, faces).clone()
    _lib.lib.mfem_debug_set_elasticity(1)
    try:
        K_row = brick.assemble_elasticity(A, LAM, MU, TAU, faces).clone()
    finally:
        _lib.lib.mfem_debug_set_elasticity(0)
    scale = float(K_row.abs().max())
    assert scale > 0 and float((K - K_row).abs().max()) <= 2e-13 * scale
    assert torch.equal(K, brick.assemble_elasticity(A, LAM, MU, TAU, faces))


@pytest.mark.parametrize("case", ["uniform", "sheared", "half_distorted"])
def test_affine_element_shortcut_of_the_matrix_kernel(mf, case):
    """Round 4: on an affine element (parallelepiped: the mixed coefficients of its trilinear map vanish to 16 ulp of the coordinates' magnitude) the matrix
    kernel takes G = det Jinv^T M Jinv from the quadrature's reference integrals instead of integrating at 8 Gauss points.  Against the oracle (2e-13) and
    against the general path (bit 5 of mfem_debug_set_elasticity) on the uniform brick, a sheared brick (affine, full J) and a brick distorted only where
    x > 1.5 (both paths inside one launch, inside one wave)."""
    import torch
    from metafem_jl_amd import _lib
    from oracle import fem, mesh as om, problems, reference_element as re_

    x, n = (3.0, 1.0, 1.0), (6, 3, 4)
    disc = re_.initialize_classical_element(3, "CUBE", 1, 1, 3)
    msh = om.lattice_mesh(x, n, disc)
    c = msh.coords.copy()
    if case == "sheared":
        c = c @ np.array([[1.0, 0.3, -0.2], [0.1, 0.9, 0.25], [-0.15, 0.2, 1.1]]).T + np.array([0.3, -0.2, 0.1])
    elif case == "half_distorted":
        c = c + 0.03 * np.stack([np.sin(2 * c[:, 1]), np.cos(2 * c[:, 2]) - 1, c[:, 0] * c[:, 1] / x[0]], axis=1) * (c[:, :1] > 1.5)
    msh.coords = c
    od = fem.FEMDomain(msh, disc, 3, problems.elasticity_domain(3, LAM, MU), [])
    od.update_time()
    od.K_linear_func()
    brick = mf.make_Brick(x, n, 1, 3)
    for d in range(3):
        brick.coords_view(d).copy_(torch.tensor(c[:, d], device="cuda"))
    A = brick.pattern(3)
    Ks = {}
    try:
        for knob in (0, 1 << 5):
            _lib.lib.mfem_debug_set_elasticity(knob)
            Ks[knob] = brick.assemble_elasticity(A, LAM, MU, 0.0, 0).cpu().numpy()
    finally:
        _lib.lib.mfem_debug_set_elasticity(0)
    scale = np.abs(od.K_linear).max()
    assert np.abs(Ks[0] - od.K_linear).max() <= 2e-13 * scale and np.abs(Ks[1 << 5] - od.K_linear).max() <= 2e-13 * scale
    assert np.abs(Ks[0] - Ks[1 << 5]).max() <= 1e-13 * scale
    if case != "half_distorted":
        assert not np.array_equal(Ks[0], Ks[1 << 5])  # (the two paths round differently: the knob does select another code path)
