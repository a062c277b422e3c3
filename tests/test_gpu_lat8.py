"""Solver layout mode 5 (csrc/spmv_lat8.hip): the 3-field 27-point lattice matrix (hex-8 elasticity, cantilever/3D_Script.jl's system) stored as
symmetric lattice tiles.  The caller's contract stays CSR (mul!, misc/04_GPU_Utils.jl:131; iterative_Solve!, 02_Preconditioner.jl:32-76): every check
is against the CSR kernel / the diagonal-slotted or sliced layout on the same values.  Tolerances: 1e-13 relative for one SpMV (other summation
order), 1e-8 for converged solutions (tolerance of the solves 1e-11)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LAM, MU, TAU = 0.5769230769230769, 0.38461538461538464, 1000.0


def _mode(b, A):
    from metafem_jl_amd import _lib

    mode = C.c_int32()
    _lib.check(_lib.lib.mfem_csr_solver_layout(b.ctx._h, A._h, C.byref(mode), None, None, None))
    return mode.value


@pytest.fixture()
def small_layouts():
    from metafem_jl_amd import _lib

    _lib.lib.mfem_debug_set_layout_min_rows(0, 0)
    yield _lib
    _lib.lib.mfem_debug_set_layout_min_rows(262144, 1000000)
    _lib.lib.mfem_debug_set_lat8(1)
    _lib.lib.mfem_debug_set_remainder(1)


@pytest.mark.parametrize("dims", [(3, 3, 3), (4, 4, 4), (3, 4, 5), (9, 5, 17), (5, 16, 4), (17, 9, 12), (1, 40, 3), (20, 20, 20), (8, 8, 16), (7, 7, 15)])
def test_spmv_equals_the_csr_kernel(mf, small_layouts, dims):
    """Tiles (8 x 8 x 16 nodes) and units (4 x 4 x 4) cut by the lattice in every direction, exact fits included; alpha / beta."""
    import torch

    _lib = small_layouts
    b = mf.make_Brick((1.0, 0.7, 1.3), dims, 1, 3)
    A = b.pattern(3)
    if A.n < 128:
        pytest.skip("less than one block: CSR kernel")
    K = b.assemble_elasticity(A, LAM, MU, TAU, mf.FACE_BITS["x0"])
    assert _mode(b, A) == 5
    x = mf.FEM_rand(A.n, 3, 0) - 0.5
    y0 = torch.zeros(A.n, dtype=torch.float64, device="cuda")
    mf.mul_(y0, A, K, x)
    # pass 2 by the kernel that walks the covering blocks (default) and by the staged gather (bit 2: all loads of a tile in flight at once): the same sums
    for knob in (1, 1 | 4):
        _lib.lib.mfem_debug_set_lat8(knob)
        for alpha, beta in ((1.0, 0.0), (-2.5, 0.75)):
            y1 = torch.full((A.n,), 7.0, dtype=torch.float64, device="cuda")
            c0 = int(_lib.lib.mfem_debug_lat8_spmv_count())
            _lib.check(_lib.lib.mfem_spmv_solver_layout(b.ctx._h, A._h, K.data_ptr(), x.data_ptr(), y1.data_ptr(), alpha, beta))
            assert int(_lib.lib.mfem_debug_lat8_spmv_count()) == c0 + 1
            assert _lib.lib.mfem_debug_lat8_asymmetry(A._h) <= 1e-13  # per-row measure |y_layout - y_csr|_r / |a_rr| of the bind's probe (gate: 4e-13)
            want = alpha * y0 + beta * 7.0
            assert float((want - y1).abs().max()) <= 1e-13 * float(y0.abs().max())


def test_values_that_are_not_symmetric_take_the_other_layout(mf, small_layouts):
    import torch

    _lib = small_layouts
    b = mf.make_Brick((1.0, 1.0, 1.0), (6, 5, 7), 1, 3)
    A = b.pattern(3)
    K = b.assemble_elasticity(A, LAM, MU, TAU, mf.FACE_BITS["x0"])
    x = mf.FEM_rand(A.n, 3, 0) - 0.5
    for eps, served in ((1e-9, 0), (1e-15, 1), (float("nan"), 0)):
        K2 = K.clone()
        pos = int(A.nnz // 2 + 5)
        K2[pos] = K2[pos] * (1.0 + eps) if eps == eps else float("nan")
        y0 = torch.zeros(A.n, dtype=torch.float64, device="cuda")
        y1 = torch.zeros_like(y0)
        mf.mul_(y0, A, K2, x)
        c0 = int(_lib.lib.mfem_debug_lat8_spmv_count())
        _lib.check(_lib.lib.mfem_spmv_solver_layout(b.ctx._h, A._h, K2.data_ptr(), x.data_ptr(), y1.data_ptr(), 1.0, 0.0))
        assert int(_lib.lib.mfem_debug_lat8_spmv_count()) - c0 == served, eps
        if eps == eps:
            assert float((y0 - y1).abs().max()) <= 1e-13 * float(y0.abs().max())
        else:
            assert bool(torch.isnan(y1).any())


def test_solvers(mf, small_layouts):
    """bicgstabl_GS! / idrs! / cgs2! with the right Jacobi scaling (applied to x while it is staged), cg!, and the cases that must not take the
    layout (scale_in_place, a left preconditioner); every solution equals the solve on the other layouts."""
    _lib = small_layouts
    b = mf.make_Brick((1.0, 1.0, 1.0), (10, 9, 8), 1, 3)
    A = b.pattern(3)
    K = b.assemble_elasticity(A, LAM, MU, TAU, mf.FACE_BITS["x0"])
    rhs = mf.FEM_rand(A.n, 5, 0) - 0.5
    cases = [
        ("cg jacobi", dict(Sv_func=mf.cg_), True),
        ("bicgstab jacobi", dict(Sv_func=mf.bicgstabl_GS_, s=2), True),
        ("bicgstab colnorm", dict(Sv_func=mf.bicgstabl_GS_, s=2, Pr_func=mf.Pr_Jacobi_colnorm_), True),
        ("bicgstab plain", dict(Sv_func=mf.bicgstabl_GS_, s=2, Pr_func=mf.Identity), True),
        ("bicgstab in place", dict(Sv_func=mf.bicgstabl_GS_, s=2, scale_in_place=True), False),
        ("idrs jacobi", dict(Sv_func=mf.idrs_, s=4), True),
        ("idrs left", dict(Sv_func=mf.idrs_, s=4, Pl_func=mf.Pl_Jacobi_), False),
        ("cgs2 jacobi", dict(Sv_func=mf.cgs2_), True),
    ]
    for name, kw, expect in cases:
        sol = {}
        for lat in (1, 0):
            _lib.lib.mfem_debug_set_lat8(lat)
            c0 = int(_lib.lib.mfem_debug_lat8_spmv_count())
            x, st = mf.iterative_Solve(A, K.clone(), rhs, 1e-11, maxiter=6000, max_pass=4, **kw)
            used = int(_lib.lib.mfem_debug_lat8_spmv_count()) > c0
            assert used == (expect and lat == 1), (name, lat)
            assert st.converged, name
            sol[lat] = x.cpu().numpy()
        assert np.abs(sol[1] - sol[0]).max() <= 1e-8 * np.abs(sol[0]).max(), name


def test_a_solve_whose_values_are_refused_starts_over_on_the_other_layouts(mf, small_layouts):
    _lib = small_layouts
    _lib.lib.mfem_debug_set_remainder(0)  # (round 5: a single asymmetric entry would be repaired by a remainder and keep the tiles -- tests/test_gpu_remainder.py; this test is about the refusal path)
    b = mf.make_Brick((1.0, 1.0, 1.0), (9, 7, 6), 1, 3)
    A = b.pattern(3)
    K = b.assemble_elasticity(A, LAM, MU, TAU, mf.FACE_BITS["x0"])
    rhs = mf.FEM_rand(A.n, 5, 0) - 0.5
    K2 = K.clone()
    w = K[A.nnz // 2:A.nnz // 2 + 60].abs()
    K2[int(A.nnz // 2 + int(w.argsort(descending=True)[1]))] *= 1.0 + 1e-6  # the second largest entry of a middle row: off-diagonal, not small
    sols = []
    for vals, expect in ((K, True), (K2, False), (K, True)):
        c0 = int(_lib.lib.mfem_debug_lat8_spmv_count())
        x, st = mf.iterative_Solve(A, vals, rhs, 1e-11, Sv_func=mf.bicgstabl_GS_, s=2, maxiter=6000, max_pass=4)
        assert st.converged and (int(_lib.lib.mfem_debug_lat8_spmv_count()) > c0) == expect
        sols.append(x.cpu().numpy())
    assert np.abs(sols[0] - sols[2]).max() <= 1e-8 * np.abs(sols[0]).max()
    assert np.abs(sols[0] - sols[1]).max() <= 1e-2 * np.abs(sols[0]).max()


@pytest.mark.parametrize("lo,hi", [(0, 7), (7, 15), (15, 21), (9, 10), (3, 12)])
def test_spmv_on_slabs_equals_the_csr_kernel(mf, small_layouts, lo, hi):
    """A slab of a 20 x 9 x 6 brick (first, middle, last, one plane thick, not tile-aligned): x carries a low and a high ghost block per field.  The
    entries towards the upper ghost plane are stored entries (their mirrored products fall on cells nobody gathers), those towards the lower ghost plane
    are taken from the CSR values in the second pass; with and without the right Jacobi scaling the solvers hand over."""
    import torch
    from metafem_jl_amd import parallel as par

    _lib = small_layouts
    n = (20, 9, 6)
    sb = mf.make_Brick((2.0, 1.0, 1.0), n, 1, 3)
    sb.set_slab(lo, hi)
    A = sb.pattern(3)
    K = sb.assemble_elasticity(A, LAM, MU, TAU, mf.FACE_BITS["x0"])
    assert _mode(sb, A) == 5
    nloc = par.local_vector_length(lo, hi, n[1] + 1, n[2] + 1, 3, order=1)
    assert nloc == A.ncols
    x = mf.FEM_rand(nloc, 3, 0) - 0.5
    y0 = torch.zeros(A.n, dtype=torch.float64, device="cuda")
    mf.mul_(y0, A, K, x)
    y1 = torch.full((A.n,), 7.0, dtype=torch.float64, device="cuda")
    c0 = int(_lib.lib.mfem_debug_lat8_spmv_count())
    _lib.check(_lib.lib.mfem_spmv_solver_layout(sb.ctx._h, A._h, K.data_ptr(), x.data_ptr(), y1.data_ptr(), 1.0, 0.0))
    assert int(_lib.lib.mfem_debug_lat8_spmv_count()) == c0 + 1
    assert float((y0 - y1).abs().max()) <= 1e-13 * float(y0.abs().max())


def _symmetric_values(A):
    """Values that are a function of the unordered (row, column) pair: symmetric bit for bit, diagonally dominant."""
    import torch

    rp, ci = A.rowptr.to(torch.int64), A.colidx.to(torch.int64)
    rows = torch.repeat_interleave(torch.arange(A.n, device="cuda"), rp[1:] - rp[:-1])
    lo, hi = torch.minimum(rows, ci), torch.maximum(rows, ci)
    w = -1.0 + 0.2 * (((lo * 2654435761 + hi * 40503) % 1000).to(torch.float64) / 1000.0)
    return torch.where(rows == ci, torch.full_like(w, 90.0), w)


@pytest.mark.parametrize("F", [1, 2])
@pytest.mark.parametrize("dims", [(3, 4, 5), (9, 5, 17), (8, 8, 16), (20, 7, 15)])
def test_one_and_two_fields(mf, small_layouts, F, dims):
    """The same construction for F = 1 and F = 2 fields (14 / 55 steps per unit; F = 2 has an odd number of chunks: the register buffers swap roles
    between a wave's two units).  One field: the layout query answers for cg!, which keeps the bitwise patch sweep -- bit 1 of mfem_debug_set_lat8 makes
    the query and the diagnostic SpMV entry take the tiles."""
    import torch

    _lib = small_layouts
    b = mf.make_Brick((1.0, 0.7, 1.3), dims, 1, 3)
    A = b.pattern(F)
    K = _symmetric_values(A)
    if F == 1:
        assert _mode(b, A) != 5
        _lib.lib.mfem_debug_set_lat8(3)
    assert _mode(b, A) == 5
    x = mf.FEM_rand(A.n, 3, 0) - 0.5
    y0 = torch.zeros(A.n, dtype=torch.float64, device="cuda")
    mf.mul_(y0, A, K, x)
    for alpha, beta in ((1.0, 0.0), (-2.5, 0.75)):
        y1 = torch.full((A.n,), 7.0, dtype=torch.float64, device="cuda")
        c0 = int(_lib.lib.mfem_debug_lat8_spmv_count())
        _lib.check(_lib.lib.mfem_spmv_solver_layout(b.ctx._h, A._h, K.data_ptr(), x.data_ptr(), y1.data_ptr(), alpha, beta))
        assert int(_lib.lib.mfem_debug_lat8_spmv_count()) == c0 + 1
        assert float((alpha * y0 + beta * 7.0 - y1).abs().max()) <= 1e-13 * float(y0.abs().max())


def test_one_field_solvers(mf, small_layouts):
    """hex-8 thermal (one field): idrs! / bicgstabl_GS! with Pr_Jacobi! run on the tiles (their A D^-1 cannot take the symmetric patch sweep),
    cg! keeps the patch sweep; on a slab too; every solution equals the solve without the tiles."""
    _lib = small_layouts
    b = mf.make_Brick((1.0, 1.0, 1.0), (12, 9, 10), 1, 3)
    A = b.pattern(1)
    K = b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
    rhs = mf.FEM_rand(A.n, 5, 0) - 0.5
    for name, kw, expect in (("idrs", dict(Sv_func=mf.idrs_, s=8), True), ("bicgstab", dict(Sv_func=mf.bicgstabl_GS_, s=2), True),
                             ("cgs2", dict(Sv_func=mf.cgs2_), True), ("cg", dict(Sv_func=mf.cg_), False)):
        sol = {}
        for lat in (1, 0):
            _lib.lib.mfem_debug_set_lat8(lat)
            c0 = int(_lib.lib.mfem_debug_lat8_spmv_count())
            x, st = mf.iterative_Solve(A, K, rhs, 1e-11, maxiter=3000, max_pass=4, **kw)
            assert st.converged and (int(_lib.lib.mfem_debug_lat8_spmv_count()) > c0) == (expect and lat == 1), (name, lat)
            sol[lat] = x.cpu().numpy()
        assert np.abs(sol[1] - sol[0]).max() <= 1e-8 * np.abs(sol[0]).max(), name


def test_other_patterns_are_refused(mf, small_layouts):
    """hex-27 with three fields keeps its layout; so does a one-field pattern as far as the query (= cg!) is concerned."""
    _lib = small_layouts
    b = mf.make_Brick((1.0, 1.0, 1.0), (7, 7, 7))
    assert _mode(b, b.pattern(1)) != 5 and _mode(b, b.pattern(2)) == 5
    b27 = mf.make_Brick((1.0, 1.0, 1.0), (5, 4, 4), 2, 5)
    assert _mode(b27, b27.pattern(3)) != 5


def test_caller_supplied_csr_takes_the_lattice_tiles(mf, small_layouts):
    """1-based Int32 CSR of the 3-field matrix through mfem_csr_create (the reference's K_J_ptr / K_J): the lattice is read off row 0."""
    import torch

    _lib = small_layouts
    b = mf.make_Brick((1.0, 1.0, 1.0), (9, 6, 7), 1, 3)
    A0 = b.pattern(3)
    K = b.assemble_elasticity(A0, LAM, MU, TAU, mf.FACE_BITS["x0"])
    rp = (A0.rowptr.clone() + 1).to(torch.int32)
    ci = (A0.colidx.clone() + 1).to(torch.int32)
    A = mf.FEM_SpMat_CSR(rp, ci, A0.n, 1, ctx=b.ctx)
    assert _mode(b, A) == 5
    rhs = mf.FEM_rand(A.n, 5, 0) - 0.5
    c0 = int(_lib.lib.mfem_debug_lat8_spmv_count())
    x1, st = mf.iterative_Solve(A, K, rhs, 1e-11, Sv_func=mf.bicgstabl_GS_, s=2, maxiter=6000, max_pass=4)
    assert st.converged and int(_lib.lib.mfem_debug_lat8_spmv_count()) > c0
    x0, _ = mf.iterative_Solve(A0, K, rhs, 1e-11, Sv_func=mf.bicgstabl_GS_, s=2, maxiter=6000, max_pass=4)
    assert float((x1 - x0).abs().max()) <= 1e-8 * float(x0.abs().max())


@pytest.mark.parametrize("fields,dims", [(3, (20, 20, 20)), (3, (9, 5, 17)), (1, (33, 31, 29)), (1, (8, 8, 16))])
def test_tiles_are_bitwise_reproducible(mf, small_layouts, fields, dims):
    """Round 6: pass 1 of the tiles runs its steps phase-major -- a phase = the (dj, dk) of the steps' offset, barriers between
    phases -- so every LDS cell receives its mirrored products from ONE wave per phase in program order: y is bitwise the same from run to run (until
    round 5 the waves of a workgroup added concurrently: ~1e-16 relative).  20 products of the layout are identical bit for bit, and so are two solves
    (iterates AND iteration counts) with the reference's two solvers, which run on A D^-1 through these tiles."""
    import torch

    _lib = small_layouts
    b = mf.make_Brick((1.0, 0.7, 1.3), dims, 1, 3)
    A = b.pattern(fields)
    if fields == 1:
        K = b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
        _lib.lib.mfem_debug_set_lat8(3)  # (one field: the layout query / diagnostic product answer for cg! unless asked)
    else:
        K = b.assemble_elasticity(A, LAM, MU, TAU, mf.FACE_BITS["x0"])
    assert _mode(b, A) == 5
    x = mf.FEM_rand(A.n, 3, 0) - 0.5
    ys = []
    for _ in range(20):
        y = torch.full((A.n,), 7.0, dtype=torch.float64, device="cuda")
        _lib.check(_lib.lib.mfem_spmv_solver_layout(b.ctx._h, A._h, K.data_ptr(), x.data_ptr(), y.data_ptr(), 1.0, 0.0))
        ys.append(y)
    assert all(torch.equal(ys[0], y) for y in ys[1:])
    y0 = torch.zeros(A.n, dtype=torch.float64, device="cuda")
    mf.mul_(y0, A, K, x)
    assert float((y0 - ys[0]).abs().max()) <= 1e-13 * float(y0.abs().max())
    rhs = mf.FEM_rand(A.n, 5, 0) - 0.5
    for kw in (dict(Sv_func=mf.idrs_, s=8), dict(Sv_func=mf.bicgstabl_GS_, s=2)):
        runs = []
        for _ in range(3):
            c0 = int(_lib.lib.mfem_debug_lat8_spmv_count())
            xs, st = mf.iterative_Solve(A, K, rhs, 1e-10, maxiter=3000, max_pass=4, **kw)
            assert st.converged == 1 and int(_lib.lib.mfem_debug_lat8_spmv_count()) > c0
            runs.append((xs.clone(), st.iterations, st.spmv_count, st.final_res))
        for r in runs[1:]:
            assert torch.equal(r[0], runs[0][0]) and r[1:] == runs[0][1:]
