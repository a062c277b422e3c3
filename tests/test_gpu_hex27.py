"""GPU parity: hex-27 (Lagrange-2) thermal path -- FP64 MFMA Ke = B^T D B with colour-partitioned scatter."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
K_COND, H, TENV, SRC = 0.6, 25.0, 293.15, 1600.0


def _oracle(x, n, itg=5, distort=False, faces=None):
    from oracle import fem, mesh as om, problems, reference_element as re_

    disc = re_.initialize_classical_element(3, "CUBE", 2, 1, itg)
    msh = om.lattice_mesh(x, n, disc)
    if distort:
        c = msh.coords
        msh.coords = c + 0.02 * np.stack([np.sin(3 * c[:, 1]) * np.cos(c[:, 2]), np.sin(2 * c[:, 0] + c[:, 2]), c[:, 0] * c[:, 1]], axis=1)
    fac = om.boundary_facets_structured(x, n, 3)
    if faces is not None:
        fac = fac.select(np.isin(fac.element_eindex, faces))
    od = fem.FEMDomain(msh, disc, 1, problems.thermal_domain(3, K_COND), [(fac, problems.thermal_convection(H, TENV))])
    od.controlpoints["s"] = np.full(msh.ncp, SRC)
    od.update_time()
    od.K_linear_func()
    return od


@pytest.fixture(params=[0, 1 << 11, 1 | (1 << 16), 1 | (3 << 16), 2, 3],
                ids=["default", "two_pass_gather", "two_pass_ring_1_plane", "two_pass_ring_3_planes", "fp64_atomics", "colour_scatter"])
def variant(request):
    """All matrix-assembly variants: the default choice (all-affine meshes: rows from G0; distorted meshes with 3 Gauss points per direction: rows from G_q,
    k_hex27_rows_gq), MFMA Ke -> scratch ring + LDS row-building gather (bit 11: the row-owner kernel of general elements off), FP64 atomics,
    colour-partitioned RMW scatter."""
    from metafem_jl_amd import _lib

    _lib.lib.mfem_debug_set_hex27(request.param)
    yield request.param
    _lib.lib.mfem_debug_set_hex27(0)


@pytest.mark.parametrize("n,itg,distort,faces", [((1, 1, 1), 5, False, None), ((2, 2, 2), 5, False, None), ((3, 2, 4), 5, True, None),
                                                 ((2, 3, 1), 4, True, [0, 3, 5]), ((3, 3, 3), 3, True, None), ((2, 1, 2), 7, True, None)])
def test_hex27_pattern_matrix_residual(mf, variant, n, itg, distort, faces):
    import torch

    x = (1.0, 1.5, 0.75)
    od = _oracle(x, n, itg, distort, faces)
    rng = np.random.default_rng(3)
    od.x_star[:] = 300.0 + 10.0 * rng.standard_normal(od.basicfield_size)
    od.K_nonlinear_func()
    brick = mf.make_Brick(x, n, 2, itg)
    if distort:
        for d in range(3):
            brick.coords_view(d).copy_(torch.tensor(od.mesh.coords[:, d], device="cuda"))
    A = brick.pattern(1)
    assert np.array_equal(A.rowptr.cpu().numpy(), od.pattern.rowptr)
    assert np.array_equal(A.colidx.cpu().numpy(), od.pattern.colidx)
    bits = 0x3F if faces is None else sum(1 << f for f in faces)
    K = brick.assemble_thermal(A, K_COND, H, TENV, bits).cpu().numpy()
    assert np.max(np.abs(K - od.K_linear)) <= 1e-12 * np.max(np.abs(od.K_linear))
    s = torch.full((brick.ncp,), SRC, dtype=torch.float64, device="cuda")
    R = brick.residual_thermal(torch.tensor(od.x_star, device="cuda"), K_COND, H, TENV, bits, s=s).cpu().numpy()
    assert np.max(np.abs(R - od.residue)) <= 1e-11 * np.max(np.abs(od.residue))


def test_hex27_golden_fixture_and_solve(mf, variant):
    import torch

    z = np.load(os.path.join(GOLD, "oracle_thermal_hex27_2x2x2.npz"))
    brick = mf.make_Brick(tuple(z["x"]), tuple(int(v) for v in z["n"]), 2, 5)
    dom = mf.ThermalDomain(brick, K_COND, H, TENV)
    dom.s.fill_(SRC)
    dom.converge_tol = 1e-9
    dom.linear_solver = lambda gf: mf.iterative_Solve(gf.A, gf.K_total, gf.residue, 1e-13, Sv_func=mf.cg_, maxiter=2000, max_pass=3)[0]
    hist = dom.update_OneStep()
    assert np.max(np.abs(dom.K_linear.cpu().numpy() - z["K"])) <= 1e-12 * np.abs(z["K"]).max()
    assert len(hist) == 2 and hist[1] < 1e-9
    assert np.max(np.abs(dom.x.cpu().numpy() - z["T"])) <= 1e-10 * np.abs(z["T"]).max()


def test_hex27_matrix_is_symmetric(mf, variant):
    import scipy.sparse as sp

    brick = mf.make_Brick((1.0, 1.0, 1.0), (4, 3, 3), 2, 5)
    A = brick.pattern(1)
    K = brick.assemble_thermal(A, K_COND, H, TENV, 0x3F).cpu().numpy()
    M = sp.csr_matrix((K, A.colidx.cpu().numpy(), A.rowptr.cpu().numpy()), shape=(A.n, A.n))
    assert abs(M - M.T).max() <= 1e-13 * abs(M).max()


@pytest.mark.parametrize("case", ["uniform", "sheared", "half_distorted", "offset_origin"])
def test_hex27_affine_element_shortcut(mf, case):
    """Round 4: elements whose 27 nodes are an affine image of the reference nodes take a constant-Jacobian shortcut in the MFMA kernel (G_q = w_q G0).
    Against the oracle (<= 1e-12) and against the general path (bit 8 of mfem_debug_set_hex27, <= 1e-13) on: the uniform brick; a SHEARED brick (affine,
    J full, not symmetric); a brick whose x > 0.5 half is distorted (both paths inside one launch); a brick far from the origin (the affine test is
    made to 16 ulp of the coordinates' magnitude, where mid-node coordinates differ from the corner average by round-off)."""
    import torch
    from metafem_jl_amd import _lib
    from oracle import fem, mesh as om, problems, reference_element as re_

    x, n = (1.0, 1.5, 0.75), (4, 3, 5)
    disc = re_.initialize_classical_element(3, "CUBE", 2, 1, 5)
    msh = om.lattice_mesh(x, n, disc)
    c = msh.coords.copy()
    if case == "sheared":
        M = np.array([[1.0, 0.3, -0.2], [0.1, 0.9, 0.25], [-0.15, 0.2, 1.1]])
        c = c @ M.T + np.array([0.3, -0.2, 0.1])
    elif case == "half_distorted":
        bump = 0.02 * np.stack([np.sin(3 * c[:, 1]) * np.cos(c[:, 2]), np.sin(2 * c[:, 0] + c[:, 2]), c[:, 0] * c[:, 1]], axis=1)
        c = c + bump * (c[:, :1] > 0.5)
    elif case == "offset_origin":
        c = c + np.array([1000.0, -333.0, 77.7])
    msh.coords = c
    od = fem.FEMDomain(msh, disc, 1, problems.thermal_domain(3, K_COND), [])
    od.controlpoints["s"] = np.zeros(msh.ncp)
    od.update_time()
    od.K_linear_func()
    brick = mf.make_Brick(x, n, 2, 5)
    for d in range(3):
        brick.coords_view(d).copy_(torch.tensor(c[:, d], device="cuda"))
    A = brick.pattern(1)
    rng = np.random.default_rng(4)
    od.controlpoints["s"] = 100.0 * rng.standard_normal(msh.ncp)
    od.x_star[:] = 300.0 + 10.0 * rng.standard_normal(od.basicfield_size)
    od.K_nonlinear_func()
    xs, ss = torch.tensor(od.x_star, device="cuda"), torch.tensor(od.controlpoints["s"], device="cuda")
    Ks, Rs = {}, {}
    try:
        for knob in (0, 1 << 8, 1 << 9):
            _lib.lib.mfem_debug_set_hex27(knob)
            before = _lib.lib.mfem_debug_hex27_direct_count()
            Ks[knob] = brick.assemble_thermal(A, K_COND, 0.0, TENV, 0).cpu().numpy()
            # a mesh whose elements are ALL affine is assembled without Ke being stored (round 4: per-element G0 + the 1-D reference integrals); one
            # distorted element sends the whole mesh through the two-pass MFMA path
            assert (_lib.lib.mfem_debug_hex27_direct_count() > before) == (knob == 0 and case != "half_distorted")
            if knob == 1 << 9:
                continue
            Rs[knob] = brick.residual_thermal(xs, K_COND, 0.0, TENV, 0, s=ss).cpu().numpy()  # the matrix-free residual takes the shortcut too
    finally:
        _lib.lib.mfem_debug_set_hex27(0)
    rscale = np.abs(od.residue).max()
    rtol = 1e-11 if case != "offset_origin" else 1e-8
    assert np.abs(Rs[0] - od.residue).max() <= rtol * rscale and np.abs(Rs[1 << 8] - od.residue).max() <= rtol * rscale
    assert np.abs(Rs[0] - Rs[1 << 8]).max() <= 0.1 * rtol * rscale
    scale = np.abs(od.K_linear).max()
    # (far from the origin both paths -- and the oracle -- compute J from differences of coordinates of magnitude 1e3: 1e-16 x 1e3 / h per entry)
    tol = 1e-12 if case != "offset_origin" else 1e-9
    assert np.abs(Ks[0] - od.K_linear).max() <= tol * scale
    assert np.abs(Ks[1 << 8] - od.K_linear).max() <= tol * scale
    assert np.abs(Ks[0] - Ks[1 << 8]).max() <= 0.1 * tol * scale
    assert np.abs(Ks[1 << 9] - od.K_linear).max() <= tol * scale and np.abs(Ks[0] - Ks[1 << 9]).max() <= 0.1 * tol * scale


def test_hex27_scratch_free_path_on_a_long_lattice(mf):
    """k_hex27_direct keeps the row-box tables (lo, c, P per direction) in LDS up to 1024 lattice planes + lines + points and reads them from memory beyond: a
    brick of 520 x 3 x 2 elements (1041 + 7 + 5 lattice coordinates) takes the second way.  Against the two-pass MFMA path (bit 9 of mfem_debug_set_hex27)."""
    import torch
    from metafem_jl_amd import _lib

    brick = mf.make_Brick((26.0, 0.3, 0.2), (520, 3, 2), 2, 5)
    A = brick.pattern(1)
    try:
        before = _lib.lib.mfem_debug_hex27_direct_count()
        Kd = brick.assemble_thermal(A, K_COND, H, TENV, 0x3F)
        assert _lib.lib.mfem_debug_hex27_direct_count() == before + 1
        _lib.lib.mfem_debug_set_hex27(1 << 9)
        Kt = brick.assemble_thermal(A, K_COND, H, TENV, 0x3F)
        assert _lib.lib.mfem_debug_hex27_direct_count() == before + 1
    finally:
        _lib.lib.mfem_debug_set_hex27(0)
    assert float((Kd - Kt).abs().max()) <= 1e-13 * float(Kt.abs().max())
    assert bool(torch.isfinite(Kd).all())


def _centre_nodes(n):
    """lattice ids of the elements' centre nodes (odd, odd, odd): a node that belongs to ONE element -- moving it makes exactly that element non-affine"""
    m = [2 * v + 1 for v in n]
    I, J, K = np.meshgrid(np.arange(n[0]), np.arange(n[1]), np.arange(n[2]), indexing="ij")
    return (((2 * I + 1) * m[1] + (2 * J + 1)) * m[2] + (2 * K + 1)).ravel()


@pytest.mark.parametrize("percent", [1, 30, 75, 100])
def test_hex27_per_element_choice_on_mixed_meshes(mf, percent):
    """Round 5: no all-or-nothing switch.  A mesh with `percent` % of its elements distorted (their centre node moved: exactly those
    elements become non-affine) is assembled with the affine elements computed in place and the others through pass 1 into a scratch that holds only them
    (k_hex27<true, true> in list mode + the streamed runs of k_hex27_direct).  Against the oracle <= 1e-12, against the two-pass MFMA path <= 1e-13; which path
    ran is asserted: below 30 % stored elements the per-element choice, from 30 % on the row-owner kernel of general elements (k_hex27_rows_gq: rows from
    per-element G_q, no Ke stored); with that kernel off (bit 11) the choice up to 80 %, beyond it the plain two-pass path -- or the choice when forced, bits 24-30."""
    import torch
    from metafem_jl_amd import _lib
    from oracle import fem, mesh as om, problems, reference_element as re_

    x, n = (1.0, 1.5, 0.75), (5, 4, 5)
    disc = re_.initialize_classical_element(3, "CUBE", 2, 1, 5)
    msh = om.lattice_mesh(x, n, disc)
    cn = _centre_nodes(n)
    rng = np.random.default_rng(percent)
    pick = cn if percent == 100 else rng.choice(cn, size=max(1, (len(cn) * percent) // 100), replace=False)
    msh.coords[pick] += 0.03 * rng.standard_normal((len(pick), 3)) * np.array(x) / np.array(n)
    od = fem.FEMDomain(msh, disc, 1, problems.thermal_domain(3, K_COND), [])
    od.controlpoints["s"] = np.zeros(msh.ncp)
    od.update_time()
    od.K_linear_func()
    brick = mf.make_Brick(x, n, 2, 5)
    for d in range(3):
        brick.coords_view(d).copy_(torch.tensor(msh.coords[:, d], device="cuda"))
    A = brick.pattern(1)
    lib = _lib.lib
    scale = np.abs(od.K_linear).max()
    try:
        m0, d0, r0 = lib.mfem_debug_hex27_mixed_count(), lib.mfem_debug_hex27_direct_count(), lib.mfem_debug_hex27_rows_count()
        K = brick.assemble_thermal(A, K_COND, 0.0, TENV, 0).cpu().numpy()
        assert lib.mfem_debug_hex27_direct_count() == d0                            # not the all-affine path ...
        assert (lib.mfem_debug_hex27_mixed_count() > m0) == (percent < 30)         # ... the per-element choice below 30 % of the elements,
        assert (lib.mfem_debug_hex27_rows_count() > r0) == (percent >= 30)          # from there on the row-owner kernel of general elements
        lib.mfem_debug_set_hex27(1 << 11)                                           # that kernel off: the per-element choice up to 80 %, the two-pass path beyond
        m2 = lib.mfem_debug_hex27_mixed_count()
        K1 = brick.assemble_thermal(A, K_COND, 0.0, TENV, 0).cpu().numpy()
        assert (lib.mfem_debug_hex27_mixed_count() > m2) == (percent <= 80) and lib.mfem_debug_hex27_rows_count() == r0 + (percent >= 30)
        lib.mfem_debug_set_hex27((1 << 10) | (1 << 11))                             # the choice off too: the two-pass path whole (round 4)
        m1 = lib.mfem_debug_hex27_mixed_count()
        K2 = brick.assemble_thermal(A, K_COND, 0.0, TENV, 0).cpu().numpy()
        assert lib.mfem_debug_hex27_mixed_count() == m1
        lib.mfem_debug_set_hex27((100 << 24) | (1 << 11))                           # forced for any fraction
        Kf = brick.assemble_thermal(A, K_COND, 0.0, TENV, 0).cpu().numpy()
        assert lib.mfem_debug_hex27_mixed_count() > m1
        lib.mfem_debug_set_hex27(1 << 2)                                            # the row-owner kernel from 1 % on
        r1 = lib.mfem_debug_hex27_rows_count()
        Kr = brick.assemble_thermal(A, K_COND, 0.0, TENV, 0).cpu().numpy()
        assert lib.mfem_debug_hex27_rows_count() == r1 + 1
    finally:
        lib.mfem_debug_set_hex27(0)
    for got in (K, K1, K2, Kf, Kr):
        assert np.abs(got - od.K_linear).max() <= 1e-12 * scale
    assert np.abs(K1 - K2).max() <= 1e-13 * scale and np.abs(Kf - K2).max() <= 1e-13 * scale
    assert np.abs(K - K2).max() <= 2e-13 * scale and np.abs(Kr - K2).max() <= 2e-13 * scale  # (rows from G_q: another summation order than the MFMA tiles)
    # with faces and a second assembly on the same workspace (the tables of the first call are reused in place)
    Ka = brick.assemble_thermal(A, K_COND, H, TENV, 0x3F)
    Kb = brick.assemble_thermal(A, K_COND, H, TENV, 0x3F)
    assert float((Ka - Kb).abs().max()) == 0.0


def test_hex27_assembly_time_is_monotone_in_the_distorted_fraction(mf):
    """... and costs what its parts cost: at 48^3 elements the assembly with 0 / 1 / 25 / 50 % distorted elements takes increasing time, 1 % within 1.6x
    of the all-affine mesh allowing for timing noise (measured 1.11x; round 4: 2.6x -- the whole mesh took the two-pass path), and never more than the two-pass path forced on the same mesh."""
    import torch
    from metafem_jl_amd import _lib

    n = (48, 48, 48)
    brick = mf.make_Brick((1.0, 1.0, 1.0), n, 2, 5)
    A = brick.pattern(1)
    K = torch.empty(A.nnz, dtype=torch.float64, device="cuda")
    base = [brick.coords_view(d).clone() for d in range(3)]
    cn = torch.tensor(_centre_nodes(n), device="cuda")
    perm = cn[torch.randperm(cn.numel(), device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))]

    def timed(reps=5):
        brick.assemble_thermal(A, K_COND, 0.0, TENV, 0, out=K)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            brick.assemble_thermal(A, K_COND, 0.0, TENV, 0, out=K)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    t = {}
    for pct in (0, 1, 25, 50):
        for d in range(3):
            brick.coords_view(d).copy_(base[d])
        k = (cn.numel() * pct) // 100
        if k:
            brick.coords_view(0)[perm[:k]] += 0.003
        t[pct] = timed()
    try:
        _lib.lib.mfem_debug_set_hex27((1 << 10) | (1 << 11))
        t["two_pass"] = timed()
    finally:
        _lib.lib.mfem_debug_set_hex27(0)
    print("hex-27 48^3 assembly ms by distorted fraction:", {k: round(v, 3) for k, v in t.items()})
    # (margins for timing noise on a shared pool: the measured sequence is 0.23 / 0.26 / 0.40 / 0.51 ms against 0.65 for the two-pass path)
    assert t[0] <= 1.25 * t[1] and t[1] <= 1.25 * t[25] and t[25] <= 1.25 * t[50]
    assert t[1] <= 1.6 * t[0]
    assert t[50] <= 1.15 * t["two_pass"]


@pytest.mark.parametrize("n,x,itg", [((1, 1, 1), (1.0, 1.0, 1.0), 5), ((2, 3, 2), (1.0, 1.5, 0.75), 5), ((5, 4, 7), (1.0, 1.5, 0.75), 5), ((9, 2, 3), (2.0, 0.5, 0.75), 4),
                                     ((4, 4, 4), (1.0, 1.0, 1.0), 5)])
def test_hex27_rows_from_gq_on_distorted_meshes(mf, n, x, itg):
    """Round 5: the row-owner kernel of general elements (k_hex27_rows_gq -- G_q of every element stored, 1296 bytes each; every (row, element) run computed
    in place by sum factorisation; no Ke, no scatter).  Fully distorted meshes whose sizes leave partial 4 x 4 x 4 tiles in every direction, with all six Robin
    faces: against the oracle <= 1e-12, against the two-pass MFMA path <= 2e-13, twice bitwise the same (the additions into a row come in program order), and
    the matrix stays symmetric to round-off.  A mesh with four Gauss points per direction does not take it (its tables are compile-time constants for three)."""
    import scipy.sparse as sp
    from metafem_jl_amd import _lib

    import torch

    od = _oracle(x, n, itg, True)
    brick = mf.make_Brick(x, n, 2, itg)
    for d in range(3):
        brick.coords_view(d).copy_(torch.tensor(od.mesh.coords[:, d], device="cuda"))
    A = brick.pattern(1)
    lib = _lib.lib
    r0 = lib.mfem_debug_hex27_rows_count()
    K = brick.assemble_thermal(A, K_COND, H, TENV, 0x3F)
    assert lib.mfem_debug_hex27_rows_count() == r0 + 1
    Kagain = brick.assemble_thermal(A, K_COND, H, TENV, 0x3F)
    assert float((K - Kagain).abs().max()) == 0.0
    try:
        lib.mfem_debug_set_hex27(1 << 11)
        Kt = brick.assemble_thermal(A, K_COND, H, TENV, 0x3F)
        assert lib.mfem_debug_hex27_rows_count() == r0 + 2
    finally:
        lib.mfem_debug_set_hex27(0)
    scale = np.abs(od.K_linear).max()
    assert np.abs(K.cpu().numpy() - od.K_linear).max() <= 1e-12 * scale
    assert float((K - Kt).abs().max()) <= 2e-13 * scale
    M = sp.csr_matrix((K.cpu().numpy(), A.colidx.cpu().numpy(), A.rowptr.cpu().numpy()), shape=(A.n, A.n))
    assert abs(M - M.T).max() <= 1e-13 * scale


def test_hex27_rows_from_gq_needs_three_gauss_points(mf):
    import torch
    from metafem_jl_amd import _lib

    od = _oracle((1.0, 1.5, 0.75), (2, 1, 2), 7, True)
    brick = mf.make_Brick((1.0, 1.5, 0.75), (2, 1, 2), 2, 7)
    for d in range(3):
        brick.coords_view(d).copy_(torch.tensor(od.mesh.coords[:, d], device="cuda"))
    A = brick.pattern(1)
    r0 = _lib.lib.mfem_debug_hex27_rows_count()
    K = brick.assemble_thermal(A, K_COND, H, TENV, 0x3F).cpu().numpy()
    assert _lib.lib.mfem_debug_hex27_rows_count() == r0
    assert np.abs(K - od.K_linear).max() <= 1e-12 * np.abs(od.K_linear).max()
