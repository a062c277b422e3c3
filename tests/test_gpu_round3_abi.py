"""GPU tests of the entry points added in round 3: mfem_csr_replan, mfem_csr_spmv_bytes, mfem_debug_comm_selftest, and the
row-owner assembly's refusal of elements that list a control point twice."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _lattice_csr(m):
    """27-point stencil pattern on an m[0] x m[1] x m[2] lattice (last index fastest)."""
    import scipy.sparse as sp

    if isinstance(m, int):
        m = (m, m, m)
    m0, m1, m2 = m
    idx = np.arange(m0 * m1 * m2).reshape(m0, m1, m2)
    rows, cols = [], []
    for di in (-1, 0, 1):
        for dj in (-1, 0, 1):
            for dk in (-1, 0, 1):
                src = idx[max(0, -di):m0 - max(0, di), max(0, -dj):m1 - max(0, dj), max(0, -dk):m2 - max(0, dk)]
                dst = idx[max(0, di):m0 - max(0, -di), max(0, dj):m1 - max(0, -dj), max(0, dk):m2 - max(0, -dk)]
                rows.append(src.ravel())
                cols.append(dst.ravel())
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    M = sp.csr_matrix((np.ones(rows.size), (rows, cols)), shape=(idx.size, idx.size))
    M.sort_indices()
    return M.indptr.astype(np.int64), M.indices.astype(np.int32)


def test_spmv_bytes_counts_the_columns_the_kernel_reads(mf):
    """Interior tiles of a lattice stencil repeat one column-offset list: the CSR kernel reads fewer columns than nnz there, and
    mfem_csr_spmv_bytes says how many (bench.py prices the launch with it)."""
    import torch

    rowptr, cols = _lattice_csr((9, 10, 300))  # 300-point lattice lines: 64-row tiles inside a line repeat one offset list
    n, nnz = 9 * 10 * 300, cols.size
    A = mf.FEM_SpMat_CSR(torch.tensor(rowptr, device="cuda"), torch.tensor(cols, device="cuda"), n)
    byts, cread = A.spmv_bytes()
    assert 0 < cread < nnz  # some tiles elided ...
    assert cread > nnz // 10  # ... but boundary tiles and every tile's first row are still read
    assert byts >= nnz * 8 + cread * 4 + n * 16 + (n + 1) * 8  # + the tile table
    assert byts < nnz * 12 + n * 16 + (n + 1) * 8
    # a random pattern has nothing to elide
    rng = np.random.default_rng(5)
    lens = rng.integers(20, 30, size=4000)
    rp = np.zeros(4001, dtype=np.int64)
    rp[1:] = np.cumsum(lens)
    cc = np.concatenate([np.sort(rng.choice(4000, size=l, replace=False)) for l in lens]).astype(np.int32)
    B = mf.FEM_SpMat_CSR(torch.tensor(rp, device="cuda"), torch.tensor(cc, device="cuda"), 4000)
    b2, c2 = B.spmv_bytes()
    assert c2 == cc.size and b2 >= cc.size * 12 + 4000 * 16 + 4001 * 8


def test_replan_after_rewriting_the_pattern_in_place(mf):
    """The handle caches which tiles repeat one column-offset list.  Rewriting colidx IN PLACE to a pattern where they do not (same n,
    nnz, row lengths) silently breaks mul! -- shown here -- until mfem_csr_replan re-inspects."""
    import scipy.sparse as sp
    import torch

    m = 24
    rowptr, cols = _lattice_csr(m)
    n, nnz = m ** 3, cols.size
    rng = np.random.default_rng(3)
    vals = rng.standard_normal(nnz)
    x = rng.standard_normal(n)
    rp_t, ci_t = torch.tensor(rowptr, device="cuda"), torch.tensor(cols, device="cuda")
    A = mf.FEM_SpMat_CSR(rp_t, ci_t, n)
    v_t, x_t = torch.tensor(vals, device="cuda"), torch.tensor(x, device="cuda")
    y = torch.empty(n, dtype=torch.float64, device="cuda")
    mf.mul_(y, A, v_t, x_t)
    ref = sp.csr_matrix((vals, cols, rowptr), shape=(n, n)) @ x
    assert np.abs(y.cpu().numpy() - ref).max() <= 1e-13 * np.abs(ref).max()
    # same row lengths, columns of every odd row reversed-and-reflected inside its own row range: sorted per row no longer needed by
    # the kernel, but the rows no longer share one offset list
    new_cols = cols.copy()
    for r in range(1, n, 2):
        lo, hi = rowptr[r], rowptr[r + 1]
        new_cols[lo:hi] = np.sort((cols[lo:hi].astype(np.int64) * 7 + r) % n).astype(np.int32)
    ci_t.copy_(torch.tensor(new_cols, device="cuda"))
    torch.cuda.synchronize()
    ref2 = np.zeros(n)
    np.add.at(ref2, np.repeat(np.arange(n), np.diff(rowptr)), vals * x[new_cols])
    A.replan()
    mf.mul_(y, A, v_t, x_t)
    assert np.abs(y.cpu().numpy() - ref2).max() <= 1e-13 * np.abs(ref2).max()
    _, cread = A.spmv_bytes()
    assert cread > 0


@pytest.mark.parametrize("count,rounds", [(1, 1), (66049, 3), (1 << 20, 2)])
def test_rccl_choreography_selftest_world1(mf, count, rounds):
    """The RCCL call sequence of one overlapped SpMV + reduction group (grouped send / recv on the halo stream between two events,
    all-reduce on the context stream, one communicator) on a one-rank ring: every RCCL entry point the N > 1 run uses is executed."""
    from metafem_jl_amd import _lib, parallel as par

    brick = mf.make_Brick((1.0, 1.0, 1.0), (6, 5, 4), 1, 3)
    comm = par.SlabComm(brick.ctx, brick, 0, 1, n_fields=1)
    try:
        _lib.check(_lib.lib.mfem_debug_comm_selftest(brick.ctx._h, count, rounds))
    finally:
        comm.close()
    # without a communicator the call is refused, not crashed
    assert _lib.lib.mfem_debug_comm_selftest(brick.ctx._h, 8, 1) == -1


def test_row_owner_assembly_refuses_collapsed_elements_and_the_host_falls_back(mf):
    """An element that lists one control point twice (a collapsed quad) would make two lanes of the row-owner gather add at the
    same position.  mfem_mesh_row_ranks reports it (MFEM_ERR_UNSUPPORTED) and GenericDomain takes the scatter form: K equals the
    oracle-style dense accumulation."""
    import torch
    from metafem_jl_amd import element, generic, physics

    ctx = mf.default_context()
    space = element.classical_space(2, "Lagrange", 1, 3)
    # 2 x 1 quads; the second quad is collapsed to a triangle: its nodes 2 and 3 (basis order) are the same control point
    coords = np.array([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0], [1.0, 1.0], [2.0, 0.5]])
    cp = np.array([[0, 1, 2, 3], [1, 4, 3, 4]]).T  # [itp, nel], basis (tensor) order
    wf = physics.thermal_domain(2, 0.6)
    zero_s = lambda d: d.controlpoints.__setitem__("s", torch.zeros(5, dtype=torch.float64, device="cuda"))
    dom = generic.GenericDomain(ctx, space, coords, cp, 1, wf, [], row_owner=True)
    zero_s(dom)
    dom.K_linear_func()
    K_rows = dom.K_linear.cpu().numpy().copy()
    assert dom.row_owner is False  # fell back
    dom2 = generic.GenericDomain(ctx, space, coords, cp, 1, wf, [], row_owner=False)
    zero_s(dom2)
    dom2.K_linear_func()
    assert np.allclose(K_rows, dom2.K_linear.cpu().numpy(), rtol=1e-13, atol=1e-15)
    dom3 = generic.GenericDomain(ctx, space, coords, cp, 1, wf, [], fused=False)
    zero_s(dom3)
    dom3.K_linear_func()
    assert np.allclose(K_rows, dom3.K_linear.cpu().numpy(), rtol=1e-12, atol=1e-14)
    assert np.abs(K_rows).max() > 0


@pytest.mark.parametrize("knob", [1 | (4 << 4), 1 | 4 | (2 << 4), 1 | 4])
def test_sliced_layout_region_sort_and_xcd_walk_equal_the_csr_kernel(mf, knob):
    """mfem_debug_set_sell bits 4-7 (rows sorted inside lattice regions of (8 x value)^3 points) and bit 2 (XCD-contiguous block walks): only the
    order of the work changes; y equals the CSR kernel's on a hex-27 brick, also on a slab with ghost columns."""
    import ctypes as C

    import torch
    from metafem_jl_amd import _lib, parallel as par

    _lib.lib.mfem_debug_set_layout_min_rows(0, 0)
    _lib.lib.mfem_debug_set_sell(knob)
    _lib.lib.mfem_debug_set_lat27(0)  # (the full brick would take the lattice-tile layout)
    try:
        for slab in (None, (6, 14)):
            b = mf.make_Brick((2.0, 1.0, 1.0), (10, 9, 8), 2, 5)
            if slab:
                b.set_slab(*slab)
            A = b.pattern(1)
            K = b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
            nloc = par.local_vector_length(slab[0], slab[1], b.m[1], b.m[2], 1, order=2) if slab else A.n
            x = mf.FEM_rand(nloc, 3, 0) - 0.5
            mode = C.c_int32()
            _lib.check(_lib.lib.mfem_csr_solver_layout(b.ctx._h, A._h, C.byref(mode), None, None, None))
            assert mode.value == 3
            y0 = torch.zeros(A.n, dtype=torch.float64, device="cuda")
            y1 = torch.full((A.n,), 7.0, dtype=torch.float64, device="cuda")
            mf.mul_(y0, A, K, x)
            _lib.check(_lib.lib.mfem_spmv_solver_layout(b.ctx._h, A._h, K.data_ptr(), x.data_ptr(), y1.data_ptr(), 1.0, 0.0))
            assert float((y0 - y1).abs().max()) <= 1e-13 * float(y0.abs().max())
    finally:
        _lib.lib.mfem_debug_set_sell(1)
        _lib.lib.mfem_debug_set_lat27(1)
        _lib.lib.mfem_debug_set_layout_min_rows(262144, 1000000)
