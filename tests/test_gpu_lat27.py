"""Solver layout mode 4 (csrc/spmv_lat27.hip): the hex-27 lattice matrix stored as symmetric lattice tiles.  The caller's contract stays
CSR (mul!, misc/04_GPU_Utils.jl:131; iterative_Solve!, solver/03_Iterative_Solvers.jl:31-49): every check is against the CSR kernel /
the sliced layout on the same values.  Tolerances: 1e-13 relative for one SpMV (other summation order), 1e-9 for converged solutions."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

K_COND, H, TENV = 0.6, 25.0, 293.15


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
    _lib.lib.mfem_debug_set_lat27(1)
    _lib.lib.mfem_debug_set_remainder(1)


@pytest.mark.parametrize("dims", [(3, 3, 3), (4, 4, 4), (3, 4, 5), (9, 5, 17), (5, 16, 4), (17, 9, 12), (1, 40, 3), (20, 20, 20)])
def test_spmv_equals_the_csr_kernel(mf, small_layouts, dims):
    """Tiles and units cut by the lattice in every direction (lattice points 2 n + 1: never a multiple of the 8 x 8 x 32 tile), alpha / beta."""
    import torch

    _lib = small_layouts
    b = mf.make_Brick((1.0, 0.7, 1.3), dims, 2, 5)
    A = b.pattern(1)
    if A.n < 128:
        pytest.skip("less than one block: CSR kernel")
    K = b.assemble_thermal(A, K_COND, H, TENV, 0x3F)
    assert _mode(b, A) == 4
    x = mf.FEM_rand(A.n, 3, 0) - 0.5
    y0 = torch.zeros(A.n, dtype=torch.float64, device="cuda")
    mf.mul_(y0, A, K, x)
    # pass 2 by the staged gather (default: all loads of a tile in flight at once) and by the kernel that walks the covering blocks (bit 1): the same sums
    for knob in (1, 1 | 2):
        _lib.lib.mfem_debug_set_lat27(knob)
        for alpha, beta in ((1.0, 0.0), (-2.5, 0.75)):
            y1 = torch.full((A.n,), 7.0, dtype=torch.float64, device="cuda")
            c0 = int(_lib.lib.mfem_debug_lat27_spmv_count())
            _lib.check(_lib.lib.mfem_spmv_solver_layout(b.ctx._h, A._h, K.data_ptr(), x.data_ptr(), y1.data_ptr(), alpha, beta))
            assert int(_lib.lib.mfem_debug_lat27_spmv_count()) == c0 + 1
            want = alpha * y0 + beta * 7.0
            assert float((want - y1).abs().max()) <= 1e-13 * float(y0.abs().max())


def test_values_that_are_not_symmetric_take_the_sliced_layout(mf, small_layouts):
    """The layout pass measures max |A[r][c] - A[c][r]|: one perturbed entry (relative 1e-9) and the bind leaves the solve to mode 3; a perturbation
    at round-off level (1e-15) keeps mode 4.  Either way y equals the CSR kernel's on the values given."""
    import torch

    _lib = small_layouts
    b = mf.make_Brick((1.0, 1.0, 1.0), (6, 5, 7), 2, 5)
    A = b.pattern(1)
    K = b.assemble_thermal(A, K_COND, H, TENV, 0x3F)
    x = mf.FEM_rand(A.n, 3, 0) - 0.5
    for eps, served in ((1e-9, 0), (1e-15, 1), (float("nan"), 0)):
        K2 = K.clone()
        pos = int(A.nnz // 2 + 1)  # an off-diagonal entry somewhere in the middle
        K2[pos] = K2[pos] * (1.0 + eps) if eps == eps else float("nan")
        y0 = torch.zeros(A.n, dtype=torch.float64, device="cuda")
        y1 = torch.zeros_like(y0)
        mf.mul_(y0, A, K2, x)
        c0 = int(_lib.lib.mfem_debug_lat27_spmv_count())
        _lib.check(_lib.lib.mfem_spmv_solver_layout(b.ctx._h, A._h, K2.data_ptr(), x.data_ptr(), y1.data_ptr(), 1.0, 0.0))
        assert int(_lib.lib.mfem_debug_lat27_spmv_count()) - c0 == served, eps
        if eps == eps:
            assert float((y0 - y1).abs().max()) <= 1e-13 * float(y0.abs().max())
        else:
            assert bool(torch.isnan(y1).any())


def test_solvers(mf, small_layouts):
    """Every solver runs on the layout: cg! on A, the others on A D^-1 with the right Jacobi scaling applied to x while it is staged (the stored
    matrix stays the symmetric A); not with scale_in_place (the caller's array then holds A D^-1, which is not symmetric) and not with a left
    preconditioner.  Every solution equals the sliced-layout solve."""
    _lib = small_layouts
    b = mf.make_Brick((1.0, 1.0, 1.0), (10, 9, 8), 2, 5)
    A = b.pattern(1)
    K = b.assemble_thermal(A, K_COND, H, TENV, 0x3F)
    rhs = mf.FEM_rand(A.n, 5, 0) - 0.5
    cases = [
        ("cg jacobi", dict(Sv_func=mf.cg_), True),
        ("cg plain", dict(Sv_func=mf.cg_, Pr_func=mf.Identity), True),
        ("bicgstab jacobi", dict(Sv_func=mf.bicgstabl_GS_, s=2), True),
        ("bicgstab colnorm", dict(Sv_func=mf.bicgstabl_GS_, s=2, Pr_func=mf.Pr_Jacobi_colnorm_), True),
        ("bicgstab in place", dict(Sv_func=mf.bicgstabl_GS_, s=2, scale_in_place=True), False),
        ("idrs left", dict(Sv_func=mf.idrs_, s=4, Pl_func=mf.Pl_Jacobi_), False),
        ("cgs2 jacobi", dict(Sv_func=mf.cgs2_), True),
        ("bicgstab plain", dict(Sv_func=mf.bicgstabl_GS_, s=2, Pr_func=mf.Identity), True),
        ("idrs jacobi", dict(Sv_func=mf.idrs_, s=4), True),
    ]
    for name, kw, expect in cases:
        sol = {}
        for lat in (1, 0):
            _lib.lib.mfem_debug_set_lat27(lat)
            c0 = int(_lib.lib.mfem_debug_lat27_spmv_count())
            x, st = mf.iterative_Solve(A, K.clone(), rhs, 1e-11, maxiter=3000, max_pass=4, **kw)
            used = int(_lib.lib.mfem_debug_lat27_spmv_count()) > c0
            assert used == (expect and lat == 1), (name, lat)
            assert st.converged, name
            sol[lat] = x.cpu().numpy()
        assert np.abs(sol[1] - sol[0]).max() <= 1e-9 * np.abs(sol[0]).max(), name


@pytest.mark.parametrize("variant", [1, 2, 3, 4])
def test_fused_cg_iteration_equals_spmv_plus_update(mf, small_layouts, variant):
    """One rank, CG on the lattice tiles: pass 2 runs inside the residual update (k_lat27_gather_cg: A p is never stored, p . A p comes from pass 1).
    Bit 2 of mfem_debug_set_lat27 restores SpMV (pass 1 + pass 2) + k_cg_update: the same iterates to round-off for every recurrence (classic,
    z-carrying, scaled), with and without the Jacobi preconditioner, after a fixed number of iterations and at convergence."""
    _lib = small_layouts
    b = mf.make_Brick((1.0, 0.8, 1.2), (9, 10, 17), 2, 5)   # tiles cut by the lattice in every direction, two tile layers in i
    A = b.pattern(1)
    K = b.assemble_thermal(A, K_COND, H, TENV, 0x3F)
    rhs = mf.FEM_rand(A.n, 5, 0) - 0.5
    for kw in (dict(), dict(Pr_func=mf.Identity)):
        for fixed in (True, False):
            sol = {}
            for knob in (1, 1 | 4):
                _lib.lib.mfem_debug_set_lat27(knob)
                c0 = int(_lib.lib.mfem_debug_lat27_spmv_count())
                x, st = mf.iterative_Solve(A, K, rhs, 1e-30 if fixed else 1e-11, Sv_func=mf.cg_, maxiter=25 if fixed else 3000, max_pass=1 if fixed else 4,
                                           fixed_iterations=fixed, cg_variant=variant, **kw)
                assert int(_lib.lib.mfem_debug_lat27_spmv_count()) > c0
                assert fixed or st.converged
                sol[knob] = x.cpu().numpy()
            assert np.abs(sol[1] - sol[1 | 4]).max() <= 1e-11 * np.abs(sol[1 | 4]).max(), (variant, kw, fixed)


def test_a_solve_whose_values_are_refused_starts_over_on_the_other_layouts(mf, small_layouts):
    """The sliced layout is planned only after the lattice tiles have refused values of the pattern once (mfem_solve then starts over); later
    symmetric solves on the same pattern take the tiles again."""
    _lib = small_layouts
    _lib.lib.mfem_debug_set_remainder(0)  # (round 5: a single asymmetric entry would be repaired by a remainder and keep the tiles -- tests/test_gpu_remainder.py; this test is about the refusal path)
    b = mf.make_Brick((1.0, 1.0, 1.0), (8, 7, 6), 2, 5)
    A = b.pattern(1)
    K = b.assemble_thermal(A, K_COND, H, TENV, 0x3F)
    rhs = mf.FEM_rand(A.n, 5, 0) - 0.5
    K2 = K.clone()
    w = K[A.nnz // 2:A.nnz // 2 + 60].abs()
    K2[int(A.nnz // 2 + int(w.argsort(descending=True)[1]))] *= 1.0 + 1e-6  # the second largest entry of a middle row: off-diagonal, not small
    sols = []
    for vals, expect in ((K, True), (K2, False), (K, True), (K2, False)):
        c0 = int(_lib.lib.mfem_debug_lat27_spmv_count())
        x, st = mf.iterative_Solve(A, vals, rhs, 1e-11, Sv_func=mf.bicgstabl_GS_, s=2, maxiter=3000, max_pass=4)
        assert st.converged and (int(_lib.lib.mfem_debug_lat27_spmv_count()) > c0) == expect
        sols.append(x.cpu().numpy())
    assert np.abs(sols[0] - sols[2]).max() <= 1e-9 * np.abs(sols[0]).max()
    assert np.abs(sols[1] - sols[3]).max() <= 1e-9 * np.abs(sols[1]).max()
    assert np.abs(sols[0] - sols[1]).max() <= 1e-3 * np.abs(sols[0]).max()   # (a 1e-6 perturbation of one entry)


@pytest.mark.parametrize("lo,hi", [(0, 6), (6, 14), (14, 21), (8, 10), (2, 12)])
def test_spmv_on_slabs_equals_the_csr_kernel(mf, small_layouts, lo, hi):
    """Slabs of a 10 x 6 x 5 hex-27 brick (21 lattice planes; slabs start on element boundaries = even planes): x carries a low and a high block of
    two ghost planes.  Entries towards the upper ghost planes are stored entries, those of the first owned plane towards the two lower ghost planes
    come from the CSR values in the second pass."""
    import torch
    from metafem_jl_amd import parallel as par

    _lib = small_layouts
    n = (10, 6, 5)
    sb = mf.make_Brick((2.0, 1.0, 1.0), n, 2, 5)
    sb.set_slab(lo, hi)
    A = sb.pattern(1)
    K = sb.assemble_thermal(A, K_COND, H, TENV, 0x3F)
    assert _mode(sb, A) == 4
    nloc = par.local_vector_length(lo, hi, 2 * n[1] + 1, 2 * n[2] + 1, 1, order=2)
    assert nloc == A.ncols
    x = mf.FEM_rand(nloc, 3, 0) - 0.5
    y0 = torch.zeros(A.n, dtype=torch.float64, device="cuda")
    mf.mul_(y0, A, K, x)
    y1 = torch.full((A.n,), 7.0, dtype=torch.float64, device="cuda")
    c0 = int(_lib.lib.mfem_debug_lat27_spmv_count())
    _lib.check(_lib.lib.mfem_spmv_solver_layout(sb.ctx._h, A._h, K.data_ptr(), x.data_ptr(), y1.data_ptr(), 1.0, 0.0))
    assert int(_lib.lib.mfem_debug_lat27_spmv_count()) == c0 + 1
    assert float((y0 - y1).abs().max()) <= 1e-13 * float(y0.abs().max())


def test_patterns_that_only_look_like_the_lattice_are_refused(mf, small_layouts):
    """The lattice hint of mfem_brick_pattern only proposes the layout: hex-8 (order 1) and the 3-field pattern keep their layouts."""
    _lib = small_layouts
    b8 = mf.make_Brick((1.0, 1.0, 1.0), (7, 7, 7))  # 8^3 points: even counts
    assert _mode(b8, b8.pattern(1)) != 4
    b27 = mf.make_Brick((1.0, 1.0, 1.0), (5, 4, 4), 2, 5)
    assert _mode(b27, b27.pattern(3)) != 4
    # order-1 lattice with odd point counts (the hint alone cannot tell it from order 2): refused by the entry-by-entry check
    b1 = mf.make_Brick((1.0, 1.0, 1.0), (6, 6, 6))
    assert _mode(b1, b1.pattern(1)) != 4


@pytest.mark.parametrize("rp_dtype,base", [("int32", 1), ("int64", 0)])
def test_caller_supplied_csr_takes_the_lattice_tiles(mf, small_layouts, rp_dtype, base):
    """The reference hands over its own K_J_ptr / K_J (1-based Int32, columns sorted; mfem_csr_create): no lattice hint comes with them.  The plan
    reads the lattice off row 0, checks every entry, and the solve runs on the tiles; a pattern with one column moved does not."""
    import torch

    _lib = small_layouts
    b = mf.make_Brick((1.0, 1.0, 1.0), (7, 5, 6), 2, 5)
    A0 = b.pattern(1)
    K = b.assemble_thermal(A0, K_COND, H, TENV, 0x3F)
    rp = (A0.rowptr.clone() + base).to(getattr(torch, rp_dtype))
    ci = (A0.colidx.clone() + base).to(torch.int32)
    A = mf.FEM_SpMat_CSR(rp, ci, A0.n, base, ctx=b.ctx)
    assert _mode(b, A) == 4
    rhs = mf.FEM_rand(A.n, 5, 0) - 0.5
    c0 = int(_lib.lib.mfem_debug_lat27_spmv_count())
    x1, st = mf.iterative_Solve(A, K, rhs, 1e-11, Sv_func=mf.cg_, maxiter=3000, max_pass=2)
    assert st.converged and int(_lib.lib.mfem_debug_lat27_spmv_count()) > c0
    x0, st0 = mf.iterative_Solve(A0, K, rhs, 1e-11, Sv_func=mf.cg_, maxiter=3000, max_pass=2)
    assert float((x1 - x0).abs().max()) <= 1e-9 * float(x0.abs().max())
    ci2 = ci.clone()
    j = int(A0.nnz // 2)
    ci2[j], ci2[j + 1] = ci[j + 1].item(), ci[j].item()  # two neighbouring columns of a middle row swapped: not the stencil order any more
    A2 = mf.FEM_SpMat_CSR(rp, ci2, A0.n, base, ctx=b.ctx)
    assert _mode(b, A2) != 4


@pytest.mark.parametrize("dims", [(9, 5, 17), (20, 20, 20), (4, 4, 40)])
def test_tiles_are_bitwise_reproducible(mf, small_layouts, dims):
    """Round 6: pass 1 of the hex-27 tiles in its deterministic form -- lane = row, the two waves of a cube split the node types by the
    parity of their (j, k) column, steps phase-major by the (dj, dk) of their offset with barriers between the phases -- gives a y that is the same bit for
    bit from run to run (20 products), equals the CSR kernel's to round-off and the four-lanes-per-row kernel's (bit 3 of the "lat27" knob: ds_add_f64
    across waves, ~1e-16) to round-off; CG, idrs!(8) and bicgstabl_GS!(2) solves repeat bit for bit, iteration counts included."""
    import torch

    _lib = small_layouts
    b = mf.make_Brick((1.0, 0.7, 1.3), dims, 2, 5)
    A = b.pattern(1)
    K = b.assemble_thermal(A, K_COND, H, TENV, 0x3F)
    assert _mode(b, A) == 4
    x = mf.FEM_rand(A.n, 3, 0) - 0.5
    y0 = torch.zeros(A.n, dtype=torch.float64, device="cuda")
    mf.mul_(y0, A, K, x)
    ys = []
    for _ in range(20):
        y = torch.full((A.n,), 7.0, dtype=torch.float64, device="cuda")
        c0 = int(_lib.lib.mfem_debug_lat27_spmv_count())
        _lib.check(_lib.lib.mfem_spmv_solver_layout(b.ctx._h, A._h, K.data_ptr(), x.data_ptr(), y.data_ptr(), 1.0, 0.0))
        assert int(_lib.lib.mfem_debug_lat27_spmv_count()) == c0 + 1
        ys.append(y)
    assert all(torch.equal(ys[0], y) for y in ys[1:])
    assert float((y0 - ys[0]).abs().max()) <= 1e-13 * float(y0.abs().max())
    _lib.lib.mfem_debug_set_lat27(1 | 8)
    yq = torch.zeros(A.n, dtype=torch.float64, device="cuda")
    _lib.check(_lib.lib.mfem_spmv_solver_layout(b.ctx._h, A._h, K.data_ptr(), x.data_ptr(), yq.data_ptr(), 1.0, 0.0))
    assert float((yq - ys[0]).abs().max()) <= 1e-13 * float(y0.abs().max())
    _lib.lib.mfem_debug_set_lat27(1)
    rhs = mf.FEM_rand(A.n, 5, 0) - 0.5
    for kw in (dict(Sv_func=mf.cg_), dict(Sv_func=mf.idrs_, s=8), dict(Sv_func=mf.bicgstabl_GS_, s=2)):
        runs = []
        for _ in range(3):
            c0 = int(_lib.lib.mfem_debug_lat27_spmv_count())
            xs, st = mf.iterative_Solve(A, K, rhs, 1e-10, maxiter=3000, max_pass=4, **kw)
            assert st.converged == 1 and int(_lib.lib.mfem_debug_lat27_spmv_count()) > c0
            runs.append((xs.clone(), st.iterations, st.spmv_count, st.final_res))
        for r in runs[1:]:
            assert torch.equal(r[0], runs[0][0]) and r[1:] == runs[0][1:]
