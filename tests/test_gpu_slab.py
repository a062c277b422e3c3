"""GPU parity for the slab (multi-GPU) building blocks that one GPU can exercise: the slab pattern / assembly /
residual equal the corresponding rows of the global problem, and the RCCL-attached solve path (world = 1)
reproduces the plain solve.  The 2-rank protocol itself is covered on CPU by tests/test_dist_gloo.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

K_COND, H, TENV, SRC = 0.6, 25.0, 293.15, 1600.0


@pytest.mark.parametrize("n,lo,hi", [((6, 3, 4), 0, 3), ((6, 3, 4), 3, 7), ((7, 2, 2), 2, 5), ((5, 4, 3), 1, 2)])
def test_slab_rows_equal_global_rows(mf, n, lo, hi):
    import torch
    from metafem_jl_amd import parallel as par

    x = (2.0, 1.0, 1.5)
    m1, m2 = n[1] + 1, n[2] + 1
    pl = m1 * m2
    gb = mf.make_Brick(x, n)
    gA = gb.pattern(1)
    gK = gb.assemble_thermal(gA, K_COND, H, TENV, 0x3F).cpu().numpy()
    grp, gcol = gA.rowptr.cpu().numpy(), gA.colidx.cpu().numpy()
    rng = np.random.default_rng(0)
    gx = 300.0 + rng.standard_normal(gA.n)
    gs = torch.full((gA.n,), SRC, dtype=torch.float64, device="cuda")
    gR = gb.residual_thermal(torch.tensor(gx, device="cuda"), K_COND, H, TENV, 0x3F, s=gs).cpu().numpy()

    sb = mf.make_Brick(x, n)
    sb.set_slab(lo, hi)
    sA = sb.pattern(1)
    n_owned = (hi - lo) * pl
    assert sA.n == n_owned
    r0, r1 = lo * pl, hi * pl
    rp = sA.rowptr.cpu().numpy()
    assert np.array_equal(rp, grp[r0:r1 + 1] - grp[r0])
    cols = gcol[grp[r0]:grp[r1]]
    expect = par.slab_local_index(cols // pl, (cols % pl) // m2, cols % m2, 0, lo, hi, m1, m2, 1)
    assert np.array_equal(sA.colidx.cpu().numpy(), expect)
    sK = sb.assemble_thermal(sA, K_COND, H, TENV, 0x3F).cpu().numpy()
    assert np.array_equal(sK, gK[grp[r0]:grp[r1]])  # same kernel, same arithmetic: bitwise
    # local x with ghosts filled from the global vector
    nloc = par.local_vector_length(lo, hi, m1, m2, 1)
    xl = np.zeros(nloc)
    xl[:n_owned] = gx[r0:r1]
    if lo > 0:
        xl[n_owned:n_owned + pl] = gx[r0 - pl:r0]
    if hi < n[0] + 1:
        xl[n_owned + pl:n_owned + 2 * pl] = gx[r1:r1 + pl]
    sl = torch.full((nloc,), SRC, dtype=torch.float64, device="cuda")
    sR = sb.residual_thermal(torch.tensor(xl, device="cuda"), K_COND, H, TENV, 0x3F, s=sl).cpu().numpy()
    assert np.array_equal(sR, gR[r0:r1])
    # SpMV on the slab with ghosts == rows of the global product
    y = torch.zeros(n_owned, dtype=torch.float64, device="cuda")
    mf.mul_(y, sA, torch.tensor(sK, device="cuda"), torch.tensor(xl, device="cuda"))
    gy = torch.zeros(gA.n, dtype=torch.float64, device="cuda")
    mf.mul_(gy, gA, torch.tensor(gK, device="cuda"), torch.tensor(gx, device="cuda"))
    assert np.allclose(y.cpu().numpy(), gy.cpu().numpy()[r0:r1], rtol=1e-14, atol=1e-10)


@pytest.mark.parametrize("lo,hi", [(0, 2), (2, 5), (3, 6)])
def test_elasticity_slab_rows_equal_global_rows(mf, lo, hi):
    """3 fields, field-major: slab rows (f, owned node) and ghost columns behind all owned entries."""
    import torch
    from metafem_jl_amd import parallel as par

    x, n = (2.0, 1.0, 1.0), (5, 2, 3)
    lam, mu, tau = 0.5769, 0.3846, 1000.0
    m1, m2 = n[1] + 1, n[2] + 1
    pl = m1 * m2
    ncp = (n[0] + 1) * pl
    gb = mf.make_Brick(x, n)
    gA = gb.pattern(3)
    gK = gb.assemble_elasticity(gA, lam, mu, tau, mf.FACE_BITS["x0"]).cpu().numpy()
    grp, gcol = gA.rowptr.cpu().numpy(), gA.colidx.cpu().numpy()
    gx = 0.01 * np.random.default_rng(1).standard_normal(3 * ncp)
    sig = (0.0, 1.0, 0.0, 0.0, 0.0, 0.3)
    gR = gb.residual_elasticity(torch.tensor(gx, device="cuda"), lam, mu, tau, mf.FACE_BITS["x0"], mf.FACE_BITS["y1"], sig).cpu().numpy()
    sb = mf.make_Brick(x, n)
    sb.set_slab(lo, hi)
    sA = sb.pattern(3)
    n_owned = (hi - lo) * pl
    assert sA.n == 3 * n_owned
    sK = sb.assemble_elasticity(sA, lam, mu, tau, mf.FACE_BITS["x0"]).cpu().numpy()
    srp, scol = sA.rowptr.cpu().numpy(), sA.colidx.cpu().numpy()
    nloc = par.local_vector_length(lo, hi, m1, m2, 3)
    xl = np.zeros(nloc)
    for f in range(3):
        rows_g = f * ncp + np.arange(lo * pl, hi * pl)
        for r_loc, r_g in zip(f * n_owned + np.arange(n_owned), rows_g):
            a, b = grp[r_g], grp[r_g + 1]
            assert srp[r_loc + 1] - srp[r_loc] == b - a
            cg = gcol[a:b]
            g, node = cg // ncp, cg % ncp
            expect = par.slab_local_index(node // pl, (node % pl) // m2, node % m2, g, lo, hi, m1, m2, 3)
            assert np.array_equal(scol[srp[r_loc]:srp[r_loc + 1]], expect)
            assert np.array_equal(sK[srp[r_loc]:srp[r_loc + 1]], gK[a:b])
        for i in range(max(lo - 1, 0), min(hi + 1, n[0] + 1)):
            jj, kk = np.meshgrid(np.arange(m1), np.arange(m2), indexing="ij")
            li = par.slab_local_index(np.full(jj.size, i), jj.ravel(), kk.ravel(), f, lo, hi, m1, m2, 3)
            xl[li] = gx[f * ncp + i * pl + jj.ravel() * m2 + kk.ravel()]
    sR = sb.residual_elasticity(torch.tensor(xl, device="cuda"), lam, mu, tau, mf.FACE_BITS["x0"], mf.FACE_BITS["y1"], sig).cpu().numpy()
    for f in range(3):
        assert np.array_equal(sR[f * n_owned:(f + 1) * n_owned], gR[f * ncp + lo * pl:f * ncp + hi * pl])


@pytest.mark.parametrize("order", [1, 2])
def test_rccl_world1_solve_equals_plain_solve(mf, order):
    import torch
    from metafem_jl_amd import parallel as par

    brick = mf.make_Brick((1.0, 1.0, 1.0), (10, 9, 8) if order == 1 else (5, 4, 4), order, 3 if order == 1 else 5)
    A = brick.pattern(1)
    K = brick.assemble_thermal(A, K_COND, H, TENV, 0x3F)
    b = mf.FEM_rand(A.n, 3, 0)
    ref, st0 = mf.iterative_Solve(A, K, b, 1e-11, Sv_func=mf.cg_, maxiter=500, max_pass=2)
    comm = par.SlabComm(brick.ctx, brick, 0, 1, n_fields=1)
    try:
        t = torch.tensor([1.5, 2.5], dtype=torch.float64, device="cuda")
        assert comm.allreduce_(t).cpu().tolist() == [1.5, 2.5]
        for sv in (mf.cg_, mf.bicgstabl_GS_, mf.idrs_):
            x, st = mf.iterative_Solve(A, K, b, 1e-11, Sv_func=sv, maxiter=500, max_pass=3)
            assert st.converged == 1
            assert np.abs((x - ref).cpu().numpy()).max() <= 1e-8 * ref.abs().max().item()
        x, st = mf.iterative_Solve(A, K, b, 1e-11, Sv_func=mf.cg_, maxiter=500, max_pass=2)
        assert st.iterations == st0.iterations
        assert torch.equal(x, ref)
        # round 6: the Krylov cycles as hipGraphs WITH the communicator's calls recorded (mfem_debug_set_graphs bit 1 / MFEM_GRAPH_COMM=1; one rank: the
        # all-reduces -- the exchange has no neighbour): the same iterates and iteration counts as the direct launches, and cycles were captured
        from metafem_jl_amd import _lib

        g0 = int(_lib.lib.mfem_debug_graph_comm_count())
        try:
            for sv, kw in ((mf.cg_, dict()), (mf.cg_, dict(cg_variant=1)), (mf.bicgstabl_GS_, dict(s=2)), (mf.idrs_, dict(s=8))):
                _lib.lib.mfem_debug_set_graphs(1, 0)
                x0, s0 = mf.iterative_Solve(A, K, b, 1e-11, Sv_func=sv, maxiter=500, max_pass=3, **kw)
                _lib.lib.mfem_debug_set_graphs(3, 0)
                x1, s1 = mf.iterative_Solve(A, K, b, 1e-11, Sv_func=sv, maxiter=500, max_pass=3, **kw)
                assert s1.converged == 1 and s1.iterations == s0.iterations and torch.equal(x0, x1)
            assert int(_lib.lib.mfem_debug_graph_comm_count()) > g0
        finally:
            _lib.lib.mfem_debug_set_graphs(1, 0)
    finally:
        comm.close()


@pytest.mark.parametrize("lo,hi", [(0, 9), (9, 21), (21, 33)])
def test_solver_layout_spmv_on_slabs_with_ghost_columns(mf, lo, hi):
    """The layout mfem_solve runs its SpMVs on (diagonal-slotted blocks + explicit-column blocks for the planes that touch
    ghost columns) against the CSR kernel on a first, a middle and a last slab of a 32 x 24 x 24 brick: the middle slab
    has ghost planes on both sides, the layout every rank except the two end ranks of a multi-GPU run uses."""
    import ctypes as C

    import torch
    from metafem_jl_amd import _lib, parallel as par

    n = (32, 24, 24)
    m1, m2 = n[1] + 1, n[2] + 1
    sb = mf.make_Brick((2.0, 1.0, 1.0), n)
    sb.set_slab(lo, hi)
    A = sb.pattern(1)
    K = sb.assemble_thermal(A, K_COND, H, TENV, 0x3F)
    nloc = par.local_vector_length(lo, hi, m1, m2, 1)
    x = mf.FEM_rand(nloc, 3, 0) - 0.5
    mode, reg = C.c_int32(), C.c_int64()
    _lib.lib.mfem_debug_set_layout_min_rows(0, 0)
    _lib.check(_lib.lib.mfem_csr_solver_layout(sb.ctx._h, A._h, C.byref(mode), None, None, C.byref(reg)))
    assert mode.value == 2 and 0 < reg.value < A.n  # the planes next to a ghost plane stay on explicit columns
    y0 = torch.zeros(A.n, dtype=torch.float64, device="cuda")
    y1 = torch.full((A.n,), 7.0, dtype=torch.float64, device="cuda")
    mf.mul_(y0, A, K, x)
    _lib.check(_lib.lib.mfem_spmv_solver_layout(sb.ctx._h, A._h, K.data_ptr(), x.data_ptr(), y1.data_ptr(), 1.0, 0.0))
    assert float((y0 - y1).abs().max()) <= 1e-13 * float(y0.abs().max())
    # alpha / beta form
    y2 = y0.clone()
    _lib.check(_lib.lib.mfem_spmv_solver_layout(sb.ctx._h, A._h, K.data_ptr(), x.data_ptr(), y2.data_ptr(), -0.5, 2.0))
    _lib.lib.mfem_debug_set_layout_min_rows(262144, 1000000)
    assert float((y2 - 1.5 * y0).abs().max()) <= 1e-12 * float(y0.abs().max())


@pytest.mark.parametrize("n,lo,hi,itg,chunk", [((4, 2, 3), 0, 4, 5, 0), ((4, 2, 3), 4, 9, 5, 0), ((5, 2, 2), 2, 6, 5, 1), ((5, 2, 2), 6, 11, 3, 2),
                                             ((3, 3, 2), 2, 4, 5, 1)])
def test_hex27_slab_rows_equal_global_rows(mf, n, lo, hi, itg, chunk):
    """Order-2 lattice: slabs start / end on element boundaries and carry two ghost planes per side.  Pattern, two-pass MFMA
    assembly (also with the scratch ring), residual and SpMV of a slab equal the corresponding rows of the global problem."""
    import torch
    from metafem_jl_amd import _lib, parallel as par

    x = (2.0, 1.0, 1.5)
    m0, m1, m2 = 2 * n[0] + 1, 2 * n[1] + 1, 2 * n[2] + 1
    pl = m1 * m2
    _lib.lib.mfem_debug_set_hex27(1 | (chunk << 16))
    try:
        gb = mf.make_Brick(x, n, 2, itg)
        gA = gb.pattern(1)
        gK = gb.assemble_thermal(gA, K_COND, H, TENV, 0x3F).cpu().numpy()
        grp, gcol = gA.rowptr.cpu().numpy(), gA.colidx.cpu().numpy()
        rng = np.random.default_rng(0)
        gx = 300.0 + rng.standard_normal(gA.n)
        gs = torch.full((gA.n,), SRC, dtype=torch.float64, device="cuda")
        gR = gb.residual_thermal(torch.tensor(gx, device="cuda"), K_COND, H, TENV, 0x3F, s=gs).cpu().numpy()

        sb = mf.make_Brick(x, n, 2, itg)
        sb.set_slab(lo, hi)
        sA = sb.pattern(1)
        n_owned = (hi - lo) * pl
        assert sA.n == n_owned
        r0, r1 = lo * pl, hi * pl
        assert np.array_equal(sA.rowptr.cpu().numpy(), grp[r0:r1 + 1] - grp[r0])
        cols = gcol[grp[r0]:grp[r1]]
        expect = par.slab_local_index(cols // pl, (cols % pl) // m2, cols % m2, 0, lo, hi, m1, m2, 1, order=2)
        assert np.array_equal(sA.colidx.cpu().numpy(), expect)
        sK = sb.assemble_thermal(sA, K_COND, H, TENV, 0x3F).cpu().numpy()
        assert np.array_equal(sK, gK[grp[r0]:grp[r1]])  # same kernels, same summation order: bitwise
        nloc = par.local_vector_length(lo, hi, m1, m2, 1, order=2)
        xl = np.zeros(nloc)
        for i in range(max(lo - 2, 0), min(hi + 2, m0)):
            jj, kk = np.meshgrid(np.arange(m1), np.arange(m2), indexing="ij")
            li = par.slab_local_index(np.full(jj.size, i), jj.ravel(), kk.ravel(), 0, lo, hi, m1, m2, 1, order=2)
            xl[li] = gx[i * pl + jj.ravel() * m2 + kk.ravel()]
        sl = torch.full((nloc,), SRC, dtype=torch.float64, device="cuda")
        sR = sb.residual_thermal(torch.tensor(xl, device="cuda"), K_COND, H, TENV, 0x3F, s=sl).cpu().numpy()
        assert np.array_equal(sR, gR[r0:r1])
        y = torch.zeros(n_owned, dtype=torch.float64, device="cuda")
        mf.mul_(y, sA, torch.tensor(sK, device="cuda"), torch.tensor(xl, device="cuda"))
        gy = torch.zeros(gA.n, dtype=torch.float64, device="cuda")
        mf.mul_(gy, gA, torch.tensor(gK, device="cuda"), torch.tensor(gx, device="cuda"))
        assert np.allclose(y.cpu().numpy(), gy.cpu().numpy()[r0:r1], rtol=1e-14, atol=1e-10)
    finally:
        _lib.lib.mfem_debug_set_hex27(0)


@pytest.mark.parametrize("n,lo,hi", [((6, 3, 5), 0, 4), ((6, 3, 5), 4, 8), ((6, 3, 5), 8, 13), ((5, 2, 2), 2, 6), ((7, 3, 3), 6, 10)])
@pytest.mark.parametrize("knob,name", [(0, "rows_from_gq"), (1 << 11, "two_pass"), ((1 << 11) | (100 << 24), "per_element_choice")])
def test_hex27_distorted_slab_rows_equal_global_rows(mf, n, lo, hi, knob, name):
    """Round 5: DISTORTED order-2 meshes on slabs -- every assembly path of general elements (rows from per-element G_q: tiles of 4 x 4 x 4 lattice points that
    start in front of the slab's first plane and end behind its last one; the two-pass MFMA path; the per-element choice) gives a slab the rows the global
    assembly gives, bitwise (same kernels, same summation order per row), and which path ran is asserted."""
    import torch
    from metafem_jl_amd import _lib

    x = (2.0, 1.0, 1.5)
    m1, m2 = 2 * n[1] + 1, 2 * n[2] + 1
    pl = m1 * m2
    lib = _lib.lib
    lib.mfem_debug_set_hex27(knob)
    try:
        gb = mf.make_Brick(x, n, 2, 5)
        c = [gb.coords_view(d).clone() for d in range(3)]
        bump = [0.02 * torch.sin(3 * c[1]) * torch.cos(c[2]), 0.02 * torch.sin(2 * c[0] + c[2]), 0.02 * c[0] * c[1]]
        for d in range(3):
            gb.coords_view(d).copy_(c[d] + bump[d])
        gA = gb.pattern(1)
        r_before = lib.mfem_debug_hex27_rows_count()
        gK = gb.assemble_thermal(gA, K_COND, H, TENV, 0x3F)
        assert (lib.mfem_debug_hex27_rows_count() > r_before) == (knob == 0)
        grp = gA.rowptr.cpu().numpy()
        sb = mf.make_Brick(x, n, 2, 5)
        sb.set_slab(lo, hi)
        clo, chi = max(lo - 2, 0), min(hi + 2, 2 * n[0] + 1)
        for d in range(3):
            sb.coords_view(d).copy_((c[d] + bump[d])[clo * pl:chi * pl])
        sA = sb.pattern(1)
        r_before = lib.mfem_debug_hex27_rows_count()
        sK = sb.assemble_thermal(sA, K_COND, H, TENV, 0x3F)
        assert (lib.mfem_debug_hex27_rows_count() > r_before) == (knob == 0)
        r0, r1 = lo * pl, hi * pl
        assert torch.equal(sK, gK[int(grp[r0]):int(grp[r1])])
    finally:
        lib.mfem_debug_set_hex27(0)


def test_hex27_slab_needs_element_boundaries(mf):
    sb = mf.make_Brick((1.0, 1.0, 1.0), (4, 2, 2), 2, 5)
    with pytest.raises(Exception):
        sb.set_slab(3, 7)
    sb.set_slab(2, 6)
    from metafem_jl_amd import _lib
    A = sb.pattern(1)
    _lib.lib.mfem_debug_set_hex27(3)
    try:
        with pytest.raises(Exception):
            sb.assemble_thermal(A, K_COND, H, TENV, 0x3F)  # the colour-scatter variant refuses a slab
    finally:
        _lib.lib.mfem_debug_set_hex27(0)


@pytest.mark.parametrize("lo,hi", [(0, 6), (6, 14), (14, 21)])
def test_solver_layout_spmv_on_hex27_slabs(mf, lo, hi):
    """Rows of uneven length (27 .. 125 entries) with ghost columns up to two planes away: the row-sorted sliced-ELL layout
    of the solver against the CSR kernel on a first, a middle and a last slab of a 10 x 6 x 6 hex-27 brick."""
    import ctypes as C

    import torch
    from metafem_jl_amd import _lib, parallel as par

    n = (10, 6, 6)
    m1, m2 = 2 * n[1] + 1, 2 * n[2] + 1
    sb = mf.make_Brick((2.0, 1.0, 1.0), n, 2, 5)
    sb.set_slab(lo, hi)
    A = sb.pattern(1)
    K = sb.assemble_thermal(A, K_COND, H, TENV, 0x3F)
    nloc = par.local_vector_length(lo, hi, m1, m2, 1, order=2)
    x = mf.FEM_rand(nloc, 3, 0) - 0.5
    mode = C.c_int32()
    _lib.lib.mfem_debug_set_layout_min_rows(0, 0)
    _lib.lib.mfem_debug_set_lat27(0)  # (slabs on element boundaries would take the lattice tiles: tests/test_gpu_lat27.py)
    try:
        _lib.check(_lib.lib.mfem_csr_solver_layout(sb.ctx._h, A._h, C.byref(mode), None, None, None))
        assert mode.value == 3
        y0 = torch.zeros(A.n, dtype=torch.float64, device="cuda")
        y1 = torch.full((A.n,), 7.0, dtype=torch.float64, device="cuda")
        mf.mul_(y0, A, K, x)
        _lib.check(_lib.lib.mfem_spmv_solver_layout(sb.ctx._h, A._h, K.data_ptr(), x.data_ptr(), y1.data_ptr(), 1.0, 0.0))
        assert float((y0 - y1).abs().max()) <= 1e-13 * float(y0.abs().max())
    finally:
        _lib.lib.mfem_debug_set_lat27(1)
        _lib.lib.mfem_debug_set_layout_min_rows(262144, 1000000)
