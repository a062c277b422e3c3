"""GPU tests of round 4's boundary hardening: no C++ exception crosses the C ABI (injected std::bad_alloc -> MFEM_ERR_ALLOC + mfem_last_error),
the cycle-graph cache separates solves that differ only in the lattice tiles' column scaling, a scaled working copy is never bound as a lattice-tile
layout, and the symmetry gate of the lattice tiles weighs the probe's difference per row."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _remainder_back_on():
    """(tests that switch the skew remainder of the lattice tiles off must not leak that into later tests)"""
    yield
    from metafem_jl_amd import _lib

    _lib.lib.mfem_debug_set_remainder(1)

K_COND, H, TENV = 0.6, 25.0, 293.15
LAM, MU = 0.5769230769230769, 0.38461538461538464
MFEM_ERR_ALLOC = -6


def test_injected_host_allocation_failure_comes_back_as_a_status(mf):
    """include/metafem_mi355x.h, error convention ('never throws/aborts across the boundary'): the handle structs and planning
    vectors are host allocations; when one throws std::bad_alloc the entry point returns MFEM_ERR_ALLOC with mfem_last_error() set, handles
    passed in stay usable, and the next call works."""
    import torch
    from metafem_jl_amd import _lib

    lib = _lib.lib
    ctx = mf.default_context()
    try:
        # mfem_context_create
        lib.mfem_debug_fail_host_alloc(1)
        h = C.c_void_p()
        rc = lib.mfem_context_create(torch.cuda.current_device(), None, C.byref(h))
        assert rc == MFEM_ERR_ALLOC and not h.value
        msg = lib.mfem_last_error().decode()
        assert "mfem_context_create" in msg and "bad_alloc" in msg
        # mfem_brick_create
        for nth in (1, 2):  # the handle struct; the host vectors of the first per-dimension table (the half-built handle is released)
            lib.mfem_debug_fail_host_alloc(nth)
            with pytest.raises(mf.MetaFEMError, match="rc=-6.*mfem_brick_create"):
                mf.make_Brick((1.0, 1.0, 1.0), (4, 4, 4))
        b = mf.make_Brick((1.0, 1.0, 1.0), (4, 4, 4))  # disarmed by the throw: works again
        lib.mfem_debug_fail_host_alloc(1)
        with pytest.raises(mf.MetaFEMError, match="rc=-6.*mfem_brick_pattern"):
            b.pattern(1)
        A = b.pattern(1)
        K = b.assemble_thermal(A, K_COND, H, TENV, mf.ALL_FACES)
        # mfem_csr_create on caller arrays
        rp, ci = A.rowptr.clone(), A.colidx.clone()
        lib.mfem_debug_fail_host_alloc(1)
        with pytest.raises(mf.MetaFEMError, match="rc=-6.*mfem_csr_create"):
            mf.FEM_SpMat_CSR(rp, ci, A.n)
        A2 = mf.FEM_SpMat_CSR(rp, ci, A.n)
        x = mf.FEM_rand(A.n, 3, 0)
        y, y2 = torch.empty_like(x), torch.empty_like(x)
        mf.mul_(y, A, K, x)
        mf.mul_(y2, A2, K, x)
        assert torch.equal(y, y2)
        # an armed countdown that no call reaches does nothing
        lib.mfem_debug_fail_host_alloc(1000)
        mf.mul_(y2, A2, K, x)
        assert torch.equal(y, y2)
    finally:
        lib.mfem_debug_fail_host_alloc(0)
    assert ctx is mf.default_context()


def test_cached_cycle_graph_is_not_shared_between_jacobi_and_identity_on_the_lattice_tiles(mf):
    """the lattice-tile kernels take the right Jacobi scaling as a kernel argument (applied to x while it is staged).  Two solves on
    one pattern, one value array and one workspace -- one with Pr_Jacobi!, one with Identity -- must not replay each other's captured cycle
    (n <= 4 M: graphs on; 3 x 45^3 = 273 375 rows >= the tiles' threshold).  Each solution is checked against its own residual and against the
    same solve with graph replay off."""
    import torch
    from metafem_jl_amd import _lib

    b = mf.make_Brick((1.0, 1.0, 1.0), (44, 44, 44))
    A = b.pattern(3)
    assert 262144 <= A.n <= 4000000
    K = b.assemble_elasticity(A, LAM, MU, 1000.0, mf.FACE_BITS["x0"])
    rhs = mf.FEM_rand(A.n, 7, 0) - 0.5
    kw = dict(Sv_func=mf.bicgstabl_GS_, s=2, maxiter=4000, max_pass=4)
    ref = {}
    _lib.lib.mfem_debug_set_graphs(0, 0)
    try:
        for pr in (mf.Pr_Jacobi_, mf.Identity):
            ref[pr], st = mf.iterative_Solve(A, K, rhs, 1e-10, Pr_func=pr, **kw)
            assert st.converged == 1
    finally:
        _lib.lib.mfem_debug_set_graphs(1, 0)
    r = torch.empty_like(rhs)
    c0 = int(_lib.lib.mfem_debug_lat8_spmv_count())
    for pr in (mf.Pr_Jacobi_, mf.Identity, mf.Pr_Jacobi_, mf.Identity):
        x, st = mf.iterative_Solve(A, K, rhs, 1e-10, Pr_func=pr, **kw)
        assert st.converged == 1, pr
        mf.mul_(r, A, K, x)
        assert mf.normalized_norm(r - rhs) <= 2e-10, pr
        assert float((x - ref[pr]).abs().max()) <= 1e-6 * float(ref[pr].abs().max()), pr
    assert int(_lib.lib.mfem_debug_lat8_spmv_count()) > c0  # (the tiles did serve these solves)


def test_refused_lattice_tiles_with_no_other_layout_do_not_bind_the_unfilled_scaled_copy(mf):
    """hex-27 one-field lattice, 65^3 points (tiles planned from 180 000 rows, sliced layout only from 1 000 000: default thresholds),
    NONSYMMETRIC values, bicgstabl_GS! with right Jacobi.  The first solve sees the tiles refuse the values and starts over; from then on the
    working values are a scaled copy in the workspace, filled after the point where the tiles used to be bound from it.  Every solve must return
    the solution of the caller's matrix."""
    import torch
    from metafem_jl_amd import _lib

    _lib.lib.mfem_debug_set_remainder(0)  # (round 5: these scattered asymmetries touch ~6 % of the rows -- a remainder would repair them and keep the tiles; the refusal path is what is under test)
    b = mf.make_Brick((1.0, 1.0, 1.0), (32, 32, 32), 2, 5)
    A = b.pattern(1)
    assert 180000 <= A.n < 1000000
    K = b.assemble_thermal(A, K_COND, H, TENV, mf.ALL_FACES)
    K2 = K.clone()
    idx = torch.arange(1, A.nnz, 997, device="cuda")
    K2[idx] *= 1.0 + 1e-3  # scattered entries, diagonal and off-diagonal: no longer symmetric
    rhs = mf.FEM_rand(A.n, 9, 0) - 0.5
    r = torch.empty_like(rhs)
    sols = []
    c0 = int(_lib.lib.mfem_debug_lat27_spmv_count())
    for k in range(3):
        x, st = mf.iterative_Solve(A, K2, rhs, 1e-10, Sv_func=mf.bicgstabl_GS_, s=2, Pr_func=mf.Pr_Jacobi_, maxiter=4000, max_pass=4)
        assert st.converged == 1, k
        mf.mul_(r, A, K2, x)
        assert mf.normalized_norm(r - rhs) <= 2e-10, k
        sols.append(x)
    assert int(_lib.lib.mfem_debug_lat27_spmv_count()) == c0  # never served by the tiles
    assert float((sols[1] - sols[0]).abs().max()) <= 1e-7 * float(sols[0].abs().max())
    assert float((sols[2] - sols[0]).abs().max()) <= 1e-7 * float(sols[0].abs().max())
    # and the symmetric values on the same pattern afterwards take the tiles again
    x, st = mf.iterative_Solve(A, K, rhs, 1e-10, Sv_func=mf.bicgstabl_GS_, s=2, Pr_func=mf.Pr_Jacobi_, maxiter=4000, max_pass=4)
    assert st.converged == 1 and int(_lib.lib.mfem_debug_lat27_spmv_count()) > c0
    mf.mul_(r, A, K, x)
    assert mf.normalized_norm(r - rhs) <= 2e-10
    # round 5: with the remainder on, the same nonsymmetric values keep the tiles at DEFAULT thresholds and give the same solution
    _lib.lib.mfem_debug_set_remainder(1)
    c1, r1 = int(_lib.lib.mfem_debug_lat27_spmv_count()), int(_lib.lib.mfem_debug_rem_spmv_count())
    x, st = mf.iterative_Solve(A, K2, rhs, 1e-10, Sv_func=mf.bicgstabl_GS_, s=2, Pr_func=mf.Pr_Jacobi_, maxiter=4000, max_pass=4)
    assert st.converged == 1 and int(_lib.lib.mfem_debug_lat27_spmv_count()) > c1 and int(_lib.lib.mfem_debug_rem_spmv_count()) > r1
    mf.mul_(r, A, K2, x)
    assert mf.normalized_norm(r - rhs) <= 2e-10
    assert float((x - sols[0]).abs().max()) <= 1e-7 * float(sols[0].abs().max())


def test_symmetry_gate_weighs_the_probe_per_row(mf):
    """on a badly scaled matrix (penalty rows 1e8 times the interior rows) an asymmetric pair in an interior row is far below
    4e-13 x the GLOBAL max |a|, but not below 4e-13 x its own row's diagonal: the tiles must refuse it (the solve then runs on the caller's exact
    values); a skew perturbation with zero row sums (convection-like) must be seen too -- the probe's entries carry random signs."""
    import torch
    from metafem_jl_amd import _lib

    lib = _lib.lib
    lib.mfem_debug_set_layout_min_rows(0, 0)
    try:
        b = mf.make_Brick((1.0, 1.0, 1.0), (12, 11, 10))
        A = b.pattern(3)
        K = b.assemble_elasticity(A, LAM, MU, 1e8, mf.FACE_BITS["x0"])
        x = mf.FEM_rand(A.n, 3, 0) - 0.5
        y = torch.empty_like(x)

        def served(vals):
            c0 = int(lib.mfem_debug_lat8_spmv_count())
            _lib.check(lib.mfem_spmv_solver_layout(b.ctx._h, A._h, vals.data_ptr(), x.data_ptr(), y.data_ptr(), 1.0, 0.0))
            return int(lib.mfem_debug_lat8_spmv_count()) - c0

        assert served(K) == 1
        amax = float(K.abs().max())
        # an interior row (far from the penalised face x = 0): its second-largest entry is an off-diagonal of ordinary size
        nn = 13 * 12 * 11
        r = (8 * 12 * 11 + 5 * 11 + 5)
        lo, hi = int(A.rowptr[r]), int(A.rowptr[r + 1])
        w = K[lo:hi].abs()
        pos = lo + int(w.argsort(descending=True)[1])
        delta = 1e-9 * float(K[pos].abs())
        assert delta < 1e-3 * 4e-13 * amax  # invisible to a gate relative to the global max |a| ...
        K2 = K.clone()
        K2[pos] += delta
        assert served(K2) == 0                # ... and refused by the per-row gate
        assert lib.mfem_debug_lat8_asymmetry(A._h) > 4e-13
        # skew part with zero row sums inside one row: +d on one off-diagonal, -d on another (their mirrors untouched)
        pos2 = lo + int(w.argsort(descending=True)[2])
        K3 = K.clone()
        K3[pos] += 1e-7 * float(K[pos].abs())
        K3[pos2] -= 1e-7 * float(K[pos].abs())
        assert served(K3) == 0
    finally:
        lib.mfem_debug_set_layout_min_rows(262144, 1000000)


@pytest.mark.parametrize("order,fields", [(1, 3), (2, 1)])
def test_residual_reported_by_a_lattice_tile_solve_comes_from_the_callers_csr_values(mf, order, fields):
    """modes 4 / 5 run the Krylov loop on a half-stored copy that passed a symmetry gate; the residual the caller is told (and `converged`) is
    recomputed at the end with the CSR kernel on the caller's own values -- here: equal to ||b - A x|| / sqrt(n) from mul!, for values with an asymmetry
    just BELOW the gate (the loop then ran on the symmetrised matrix) as for symmetric ones."""
    import torch
    from metafem_jl_amd import _lib

    lib = _lib.lib
    lib.mfem_debug_set_layout_min_rows(0, 0)
    try:
        b = mf.make_Brick((1.0, 1.0, 1.0), (9, 8, 7), order, 5 if order == 2 else 3)
        A = b.pattern(fields)
        K = b.assemble_thermal(A, K_COND, H, TENV, mf.ALL_FACES) if fields == 1 else b.assemble_elasticity(A, LAM, MU, 1000.0, mf.FACE_BITS["x0"])
        count = lib.mfem_debug_lat27_spmv_count if order == 2 else lib.mfem_debug_lat8_spmv_count
        rhs = mf.FEM_rand(A.n, 7, 0) - 0.5
        r = torch.empty_like(rhs)
        for eps in (0.0, 1e-13):
            K2 = K.clone()
            if eps:
                row = A.n // 2
                lo = int(A.rowptr[row])
                K2[lo] += eps * float(K2[lo:int(A.rowptr[row + 1])].abs().max())  # first entry of a middle row: an off-diagonal one
            c0 = int(count())
            x, st = mf.iterative_Solve(A, K2, rhs, 1e-9, Sv_func=mf.bicgstabl_GS_, s=2, maxiter=4000, max_pass=4)
            assert int(count()) > c0 and st.converged == 1  # the tiles served the solve
            mf.mul_(r, A, K2, x)
            true_res = mf.normalized_norm(r - rhs)
            assert st.final_res == pytest.approx(true_res, rel=1e-6, abs=1e-16)
    finally:
        lib.mfem_debug_set_layout_min_rows(262144, 1000000)
