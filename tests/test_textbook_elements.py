"""Known answers for the element types NO reference artefact pins (hex-8 / hex-27 Lagrange cubes appear in no shipped example): closed forms from
the textbooks, independent of the oracle's own code path --
  * the hex-8 Laplacian of the unit cube: K[a][a] = 1/3, 0 between edge neighbours, -1/12 between face- and body-diagonal neighbours;
  * energies int |grad u|^2 of polynomial fields the basis holds exactly (hex-8: trilinear; hex-27: u = x^2 + y z -> 4/3 + 1/3 + 1/3 = 2);
  * linear elasticity: uniaxial strain u = (x, 0, 0) has energy (lam + 2 mu) V, a shear u = (y, 0, 0) mu V, rigid motions none;
  * a Robin face adds h int N_a N_b: the whole boundary block sums to h x area.
CPU: the oracle; GPU (-m gpu): the product's fused kernels through the C ABI.  The backend's sign convention is the reference's (-Bilinear forms): K is
NEGATIVE definite, so energies appear with a minus sign."""
import numpy as np
import pytest
import scipy.sparse as sp

LAM, MU = 0.7, 1.3


def _hamming(a, b):
    return bin(a ^ b).count("1")


def _hex8_unit_cube_laplacian():
    K = np.zeros((8, 8))
    for a in range(8):
        for b in range(8):
            K[a, b] = {0: 1.0 / 3.0, 1: 0.0, 2: -1.0 / 12.0, 3: -1.0 / 12.0}[_hamming(a, b)]
    return K


def _oracle_K(order, n, x, form, itg):
    from oracle import fem, mesh as om, problems, reference_element as re_

    disc = re_.initialize_classical_element(3, "CUBE", order, 1, itg)
    msh = om.lattice_mesh(x, n, disc)
    if form == "thermal":
        od = fem.FEMDomain(msh, disc, 1, problems.thermal_domain(3, 1.0), [])
        od.controlpoints["s"] = np.zeros(msh.ncp)
    else:
        od = fem.FEMDomain(msh, disc, 3, problems.elasticity_domain(3, LAM, MU), [])
    od.update_time()
    od.K_linear_func()
    M = sp.csr_matrix((od.K_linear, od.pattern.colidx, od.pattern.rowptr), shape=(od.pattern.n, od.pattern.n))
    return M, msh.coords


def _check_thermal(M, c, order):
    n = c.shape[0]
    if order == 1 and n == 8:
        # lattice numbering: id = (i * 2 + j) * 2 + k, i.e. bit 2 = x, bit 1 = y, bit 0 = z: Hamming distances are what the closed form needs
        assert np.abs(-M.toarray() - _hex8_unit_cube_laplacian()).max() < 1e-14
    x, y, z = c.T
    for u, energy in ((x + 2 * y - 3 * z + 0.5, 14.0), (x * y * z, 1.0 / 3.0)) + (((x * x + y * z, 2.0),) if order == 2 else ()):
        # unit cube: int |grad(x + 2y - 3z)|^2 = 14; |grad(xyz)|^2 = y^2 z^2 + x^2 z^2 + x^2 y^2 -> 3 * 1/9
        assert abs(-(u @ (M @ u)) - energy) < 1e-12 * max(energy, 1.0)
    assert np.abs(M @ np.ones(n)).max() < 1e-13  # constants carry no energy


def _check_elasticity(M, c):
    n = c.shape[0]
    x, y, z = c.T
    zero = np.zeros(n)
    vol = 1.0
    uni = np.concatenate([x, zero, zero])       # eps_xx = 1: sigma_xx = lam + 2 mu
    shear = np.concatenate([y, zero, zero])     # eps_xy = 1/2: sigma_xy = mu, energy 2 * mu / 2 * ... = mu
    assert abs(-(uni @ (M @ uni)) - (LAM + 2 * MU) * vol) < 1e-12
    assert abs(-(shear @ (M @ shear)) - MU * vol) < 1e-12
    for rigid in (np.concatenate([np.ones(n), zero, zero]), np.concatenate([-y, x, zero]), np.concatenate([zero, -z, y])):
        assert np.abs(M @ rigid).max() < 1e-12


@pytest.mark.parametrize("order,n,itg", [(1, (1, 1, 1), 3), (1, (2, 3, 2), 3), (2, (1, 1, 1), 5), (2, (2, 2, 1), 5)])
def test_oracle_thermal_elements_reproduce_the_textbook(order, n, itg):
    M, c = _oracle_K(order, n, (1.0, 1.0, 1.0), "thermal", itg)
    _check_thermal(M, c, order)


def test_oracle_hex8_elasticity_reproduces_the_textbook():
    M, c = _oracle_K(1, (2, 1, 2), (1.0, 1.0, 1.0), "elasticity", 3)
    _check_elasticity(M, c)


@pytest.mark.gpu
@pytest.mark.parametrize("order,n,itg", [(1, (1, 1, 1), 3), (1, (2, 3, 2), 3), (1, (5, 4, 6), 3), (2, (1, 1, 1), 5), (2, (2, 2, 1), 5), (2, (3, 2, 3), 5)])
def test_product_thermal_kernels_reproduce_the_textbook(mf, order, n, itg):
    """mfem_brick_assemble_thermal (hex-8: plane-sweep sum-factorised kernel; hex-27: FP64-MFMA Ke, affine shortcut and general path)."""
    import torch
    from metafem_jl_amd import _lib

    for knob in ((0, 1 << 8) if order == 2 else (0,)):
        _lib.lib.mfem_debug_set_hex27(knob)
        try:
            b = mf.make_Brick((1.0, 1.0, 1.0), n, order, itg)
            A = b.pattern(1)
            K = b.assemble_thermal(A, 1.0, 0.0, 0.0, 0).cpu().numpy()
        finally:
            _lib.lib.mfem_debug_set_hex27(0)
        M = sp.csr_matrix((K, A.colidx.cpu().numpy(), A.rowptr.cpu().numpy()), shape=(A.n, A.n))
        c = np.stack([b.coords_view(d).cpu().numpy() for d in range(3)], axis=1)
        _check_thermal(M, c, order)
    # a Robin face adds -h int N_a N_b: the boundary blocks sum to -h x area (here the two x faces: area 2)
    KR = b.assemble_thermal(A, 1.0, 7.0, 0.0, mf.FACE_BITS["x0"] | mf.FACE_BITS["x1"]).cpu().numpy()
    assert abs((KR - K).sum() + 7.0 * 2.0) < 1e-11


@pytest.mark.gpu
def test_product_hex8_elasticity_kernel_reproduces_the_textbook(mf):
    b = mf.make_Brick((1.0, 1.0, 1.0), (3, 2, 4))
    A = b.pattern(3)
    K = b.assemble_elasticity(A, LAM, MU, 0.0, 0).cpu().numpy()
    M = sp.csr_matrix((K, A.colidx.cpu().numpy(), A.rowptr.cpu().numpy()), shape=(A.n, A.n))
    c = np.stack([b.coords_view(d).cpu().numpy() for d in range(3)], axis=1)
    _check_elasticity(M, c)
