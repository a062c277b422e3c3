"""GPU parity: S1 primitives (SpMV / dot / axpby / rand / Jacobi kernels) against the oracle."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def _random_csr(n, avg, rng, ragged=True, empty_rows=True):
    lens = rng.integers(0 if empty_rows else 1, 2 * avg, size=n) if ragged else np.full(n, avg)
    lens = np.minimum(lens, n)
    rowptr = np.zeros(n + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(lens)
    cols = np.concatenate([np.sort(rng.choice(n, size=l, replace=False)) for l in lens]) if rowptr[-1] else np.zeros(0, int)
    vals = rng.standard_normal(rowptr[-1])
    return rowptr, cols.astype(np.int32), vals


@pytest.mark.parametrize("n,avg,bits,base", [(1, 1, 32, 0), (37, 3, 32, 1), (1000, 27, 64, 0), (5000, 27, 32, 1),
                                             (3000, 81, 64, 1), (257, 130, 32, 0)])
def test_spmv_matches_oracle(mf, n, avg, bits, base):
    import torch
    from oracle import solvers

    rng = np.random.default_rng(n * 7 + avg)
    rowptr, cols, vals = _random_csr(n, avg, rng)
    x = rng.standard_normal(n)
    y0 = rng.standard_normal(n)
    A_or = solvers.csr(rowptr, cols, vals, n)
    rp_t = torch.tensor(rowptr + base, dtype=torch.int64 if bits == 64 else torch.int32, device="cuda")
    A = mf.FEM_SpMat_CSR(rp_t, torch.tensor(cols + base, dtype=torch.int32, device="cuda"), n, index_base=base)
    v_t, x_t = torch.tensor(vals, device="cuda"), torch.tensor(x, device="cuda")
    scale = np.abs(A_or).dot(np.abs(x)) + np.abs(y0) + 1e-300
    for alpha, beta in [(1.0, 0.0), (-1.0, 0.0), (0.5, 2.0)]:
        y_t = torch.tensor(y0, device="cuda")
        mf.mul_(y_t, A, v_t, x_t, alpha, beta)
        ref = solvers.mul(y0.copy(), A_or, x, alpha, beta)
        assert np.max(np.abs(y_t.cpu().numpy() - ref) / scale) < 1e-14


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 6, 7, 2048, 2054, 1027])  # + 2048 (bit 27 of the knob): without the 2688-entry wave tile; + 1024 (bit 26): row-block tiles round-robin
@pytest.mark.parametrize("kind", ["hex8", "hex8x3", "hex27", "quad8", "banded_shift"])
def test_spmv_csr_kernel_variants_equal_oracle(mf, kind, variant):
    """Every tile variant of the CSR kernel behind mul! (mfem_debug_set_spmv: 0 = row-transposing tile kernel (default), 1-3 =
    product-tile kernel with different tile sizes) against the oracle, incl. 1-based int32 row pointers, empty rows, odd tile
    starts and an x longer than n."""
    import torch
    from metafem_jl_amd import _lib
    from oracle import mesh as om, operators as oo, reference_element as re_, solvers

    rng = np.random.default_rng(7)
    if kind == "banded_shift":  # 1-based, int32 rowptr, odd tile starts, empty rows
        n = 5000
        lens = rng.integers(0, 40, size=n)
        rowptr = np.zeros(n + 1, dtype=np.int64)
        rowptr[1:] = np.cumsum(lens)
        cols = np.concatenate([np.sort(rng.choice(np.arange(max(0, r - 60), min(n, r + 60)), size=l, replace=False))
                               for r, l in enumerate(lens)]).astype(np.int32)
        base, bits = 1, 32
    else:
        F = 3 if kind == "hex8x3" else 1
        if kind == "quad8":
            disc = re_.initialize_classical_element(2, "CUBE", 2, 1, 5, itp_type="Serendipity")
            vert, conn = om.make_square((2.0, 1.0), (40, 25))
            m = om.mesh_classical(vert, conn, disc)
        else:
            order = 2 if kind == "hex27" else 1
            disc = re_.initialize_classical_element(3, "CUBE", order, 1, 3)
            m = om.lattice_mesh((1.0, 1.0, 1.0), (9, 7, 11) if order == 1 else (4, 3, 5), disc)
        pat = oo.assemble_sparse_id(m.cp_ids, m.ncp, [(i, j) for i in range(F) for j in range(F)])
        rowptr, cols, n = pat.rowptr, pat.colidx, pat.n
        base, bits = 0, 64
    vals = rng.standard_normal(rowptr[-1])
    x = rng.standard_normal(n + 5)  # x longer than n
    ref = solvers.csr(rowptr, cols, vals, n) @ x[:n]
    scale = np.abs(solvers.csr(rowptr, cols, np.abs(vals), n) @ np.abs(x[:n])) + 1e-300
    _lib.lib.mfem_debug_set_spmv(variant << 16, 0)
    try:
        A = mf.FEM_SpMat_CSR(torch.tensor(rowptr + base, dtype=torch.int64 if bits == 64 else torch.int32, device="cuda"),
                             torch.tensor(cols + base, dtype=torch.int32, device="cuda"), n, index_base=base)
        y = torch.full((n,), 3.0, dtype=torch.float64, device="cuda")
        mf.mul_(y, A, torch.tensor(vals, device="cuda"), torch.tensor(x, device="cuda"), 2.0, -1.0)
        got = y.cpu().numpy()
    finally:
        _lib.lib.mfem_debug_set_spmv(0, 0)  # library default
    assert np.max(np.abs(got - (2.0 * ref - 3.0)) / (2 * scale + 3.0)) < 1e-14


def test_spmv_long_rows_fallback(mf):
    """Rows longer than the LDS tile take the wave-per-row path."""
    import torch
    from oracle import solvers

    rng = np.random.default_rng(5)
    n = 6000
    lens = np.full(n, 3)
    lens[17] = 5000
    lens[n - 1] = 4500
    rowptr = np.zeros(n + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(lens)
    cols = np.concatenate([np.sort(rng.choice(n, size=l, replace=False)) for l in lens]).astype(np.int32)
    vals = rng.standard_normal(rowptr[-1])
    x = rng.standard_normal(n)
    A = mf.FEM_SpMat_CSR(torch.tensor(rowptr, device="cuda"), torch.tensor(cols, device="cuda"), n)
    y = torch.zeros(n, dtype=torch.float64, device="cuda")
    mf.mul_(y, A, torch.tensor(vals, device="cuda"), torch.tensor(x, device="cuda"))
    ref = solvers.csr(rowptr, cols, vals, n) @ x
    scale = np.abs(solvers.csr(rowptr, cols, np.abs(vals), n) @ np.abs(x)) + 1e-300
    assert np.max(np.abs(y.cpu().numpy() - ref) / scale) < 1e-14


@pytest.mark.parametrize("bits,base", [(64, 0), (32, 1)])
def test_spmv_row_block_kernel_on_ragged_rows(mf, bits, base):
    """Rows of very uneven length take wave tiles cut by nonzeros (k_spmv_csr_rb): rows near the longest the plan admits, runs of
    several hundred empty rows (more rows in a tile than row pointers staged in LDS), single-entry rows, trailing empty rows, alpha / beta
    and the fused dot product's y."""
    import torch
    from oracle import solvers

    rng = np.random.default_rng(11)
    lens = []
    while len(lens) < 6000:
        kind = rng.integers(0, 5)
        if kind == 0: lens += [0] * int(rng.integers(130, 400))            # a tile of > 128 rows
        elif kind == 1: lens += [int(rng.integers(300, 384))] * int(rng.integers(1, 40))
        elif kind == 2: lens += [1] * int(rng.integers(1, 300))
        elif kind == 3: lens += list(rng.integers(20, 130, size=int(rng.integers(10, 200))))
        else: lens += [125, 75] * int(rng.integers(5, 60))                 # alternating stencils, as on an order-2 lattice line
    lens += [0] * 37                                                        # trailing empty rows
    lens = np.array(lens)
    n = len(lens)
    rowptr = np.zeros(n + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(lens)
    assert rowptr[-1] >= 16 * n and lens.max() <= 384                       # what the plan asks for
    cols = np.concatenate([np.sort(rng.choice(n, size=l, replace=False)) for l in lens if l > 0]).astype(np.int32)
    vals = rng.standard_normal(rowptr[-1])
    x = rng.standard_normal(n)
    ref = solvers.csr(rowptr, cols, vals, n) @ x
    scale = np.abs(solvers.csr(rowptr, cols, np.abs(vals), n) @ np.abs(x)) + 1e-300
    A = mf.FEM_SpMat_CSR(torch.tensor(rowptr + base, dtype=torch.int64 if bits == 64 else torch.int32, device="cuda"),
                         torch.tensor(cols + base, dtype=torch.int32, device="cuda"), n, index_base=base)
    y = torch.full((n,), 3.0, dtype=torch.float64, device="cuda")
    mf.mul_(y, A, torch.tensor(vals, device="cuda"), torch.tensor(x, device="cuda"), 2.0, -1.0)
    assert np.max(np.abs(y.cpu().numpy() - (2.0 * ref - 3.0)) / (2 * scale + 3.0)) < 1e-14


def test_spmv_row_block_kernel_column_elision(mf):
    """Tiles whose rows of equal parity repeat the column offsets of the tile's first two rows read only those two rows' columns
    (inspected once per pattern).  Alternating stencils as on an order-2 lattice line, clipped at both ends, with single rows whose
    offsets deviate in the middle of otherwise regular stretches: the result equals the oracle's and is bitwise the one computed with the
    inspection switched off."""
    import torch
    from metafem_jl_amd import _lib
    from oracle import solvers

    rng = np.random.default_rng(3)
    n = 40000
    even = np.array([-700, -699, -350, -2, -1, 0, 1, 2, 350, 699, 700] + list(range(-40, -10)) + list(range(10, 40)))
    odd = np.array([-350, -1, 0, 1, 350] + list(range(-25, -5)) + list(range(5, 25)))
    rows, cols = [], []
    odd_one_out = set(rng.choice(np.arange(2000, n - 2000), size=40, replace=False).tolist())
    for r in range(n):
        off = np.sort(even if r % 2 == 0 else odd)
        if r in odd_one_out:
            off = np.sort(np.concatenate([off[off != 1], [3]]))  # same length, one offset moved
        c = r + off
        c = c[(c >= 0) & (c < n)]
        rows.append(len(c))
        cols.append(c)
    rowptr = np.zeros(n + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(rows)
    cols = np.concatenate(cols).astype(np.int32)
    assert rowptr[-1] >= 16 * n
    vals = rng.standard_normal(rowptr[-1])
    x = rng.standard_normal(n)
    ref = solvers.csr(rowptr, cols, vals, n) @ x
    scale = np.abs(solvers.csr(rowptr, cols, np.abs(vals), n) @ np.abs(x)) + 1e-300
    ys = []
    try:
        for knob in (3 << 16, (3 << 16) | (1 << 25)):  # row blocks with / without the column inspection (set before the pattern exists)
            _lib.lib.mfem_debug_set_spmv(knob, 0)
            A = mf.FEM_SpMat_CSR(torch.tensor(rowptr, device="cuda"), torch.tensor(cols, device="cuda"), n)
            y = torch.zeros(n, dtype=torch.float64, device="cuda")
            mf.mul_(y, A, torch.tensor(vals, device="cuda"), torch.tensor(x, device="cuda"))
            ys.append(y)
    finally:
        _lib.lib.mfem_debug_set_spmv(0, 0)
    assert np.max(np.abs(ys[0].cpu().numpy() - ref) / scale) < 1e-14
    assert torch.equal(ys[0], ys[1])


def test_spmv_wave_tile_kernel_column_elision(mf):
    """The same for the tiles of a fixed row count (rows of one 27-entry stencil, as of the hex-8 scalar operator): a tile whose 64 rows
    all repeat the offsets of its first row reads that row's columns only; clipped ends and rows with a moved offset keep their tiles on
    the full column stream."""
    import torch
    from metafem_jl_amd import _lib
    from oracle import solvers

    rng = np.random.default_rng(4)
    n = 30000
    sten = np.sort(np.array([a * 900 + b * 30 + c for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1)]))
    moved = set(rng.choice(np.arange(1000, n - 1000), size=25, replace=False).tolist())
    cols, lens = [], []
    for r in range(n):
        off = sten if r not in moved else np.sort(np.concatenate([sten[sten != 31], [33]]))
        c = r + off
        c = c[(c >= 0) & (c < n)]
        cols.append(c)
        lens.append(len(c))
    rowptr = np.zeros(n + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(lens)
    cols = np.concatenate(cols).astype(np.int32)
    vals = rng.standard_normal(rowptr[-1])
    x = rng.standard_normal(n)
    ref = solvers.csr(rowptr, cols, vals, n) @ x
    scale = np.abs(solvers.csr(rowptr, cols, np.abs(vals), n) @ np.abs(x)) + 1e-300
    ys = []
    try:
        for knob in (0, 1 << 25):  # default kernel for these rows = wave tiles of 64 rows; with / without the column inspection
            _lib.lib.mfem_debug_set_spmv(knob, 0)
            A = mf.FEM_SpMat_CSR(torch.tensor(rowptr + 1, dtype=torch.int32, device="cuda"), torch.tensor(cols + 1, device="cuda"), n, index_base=1)
            y = torch.full((n,), 2.0, dtype=torch.float64, device="cuda")
            mf.mul_(y, A, torch.tensor(vals, device="cuda"), torch.tensor(x, device="cuda"), -1.5, 0.5)
            ys.append(y)
    finally:
        _lib.lib.mfem_debug_set_spmv(0, 0)
    assert np.max(np.abs(ys[0].cpu().numpy() - (-1.5 * ref + 1.0)) / (1.5 * scale + 1.0)) < 1e-14
    assert torch.equal(ys[0], ys[1])


def test_spmv_unaligned_values_pointer(mf):
    """A values array that is not 16-byte aligned must take the scalar path, not fault."""
    import torch
    from oracle import solvers

    rng = np.random.default_rng(11)
    rowptr, cols, vals = _random_csr(2000, 27, rng)
    x = rng.standard_normal(2000)
    buf = torch.zeros(vals.size + 1, dtype=torch.float64, device="cuda")
    buf[1:] = torch.tensor(vals, device="cuda")
    A = mf.FEM_SpMat_CSR(torch.tensor(rowptr, device="cuda"), torch.tensor(cols, device="cuda"), 2000)
    y = torch.zeros(2000, dtype=torch.float64, device="cuda")
    mf.mul_(y, A, buf[1:], torch.tensor(x, device="cuda"))
    ref = solvers.csr(rowptr, cols, vals, 2000) @ x
    assert np.allclose(y.cpu().numpy(), ref, rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 1000, 1 << 20, (1 << 20) + 3])
def test_dot_nrm2_axpby(mf, n):
    import torch

    rng = np.random.default_rng(n)
    x, y = rng.standard_normal(n), rng.standard_normal(n)
    xt, yt = torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda")
    assert abs(mf.dot(xt, yt) - x @ y) <= 1e-13 * (np.abs(x) @ np.abs(y))
    assert abs(mf.nrm2(xt) - np.linalg.norm(x)) <= 1e-13 * np.linalg.norm(x)
    mf.axpby_(0.3, xt, -1.7, yt)
    assert np.allclose(yt.cpu().numpy(), 0.3 * x - 1.7 * y, rtol=1e-15, atol=1e-15)
    # odd-offset (8-byte aligned only) views
    if n > 2:
        a, b = xt[1:], yt[1:]
        ref = x[1:] @ (0.3 * x - 1.7 * y)[1:]
        assert abs(mf.dot(a, b) - ref) <= 1e-12 * max(1.0, abs(ref))


def test_rand_is_bitwise_the_oracle_generator(mf):
    from oracle import solvers

    for seed, stream, n in [(0x5EED, 0, 1000), (1, 3, 4097), (2 ** 63 + 5, 7, 64)]:
        got = mf.FEM_rand(n, seed, stream).cpu().numpy()
        assert np.array_equal(got, solvers.fem_rand(seed, stream, n))
        assert got.min() >= 0.0 and got.max() < 1.0


def test_jacobi_kernels(mf):
    import torch
    from oracle import solvers

    rng = np.random.default_rng(3)
    n = 3000
    rowptr, cols, vals = _random_csr(n, 20, rng)
    # make sure most rows have a diagonal
    M = solvers.csr(rowptr, cols, vals, n) + sp.diags(rng.standard_normal(n) + 3.0)
    M = M.tocsr()
    M.sort_indices()
    # knock the diagonal out of a few rows: those keep d = 1 (02_Preconditioner.jl:110,122-130)
    A = mf.FEM_SpMat_CSR(torch.tensor(M.indptr.astype(np.int32), device="cuda"),
                         torch.tensor(M.indices.astype(np.int32), device="cuda"), n)
    v = torch.tensor(M.data, device="cuda")
    assert np.array_equal(mf.jacobi_by_diagonal(A, v).cpu().numpy(), solvers.jacobi_by_diagonal(M))
    col = np.sqrt(np.asarray(M.multiply(M).sum(axis=0)).ravel())
    row = np.sqrt(np.asarray(M.multiply(M).sum(axis=1)).ravel())
    assert np.allclose(mf.jacobi2_by_column(A, v).cpu().numpy(), col, rtol=1e-13)
    assert np.allclose(mf.jacobi_by_row(A, v).cpu().numpy(), row, rtol=1e-13)
    d = solvers.jacobi_by_diagonal(M)
    mf.mat_div_jacobi_(A, v, torch.tensor(d, device="cuda"))
    assert np.array_equal(v.cpu().numpy(), M.data / d[M.indices])


def test_errors_are_reported_not_thrown(mf):
    import torch

    with pytest.raises(mf.MetaFEMError):
        mf.FEM_SpMat_CSR(torch.zeros(3, dtype=torch.int64, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda"),
                         2, index_base=5)
    with pytest.raises(mf.MetaFEMError):
        mf.dot(torch.zeros(4, device="cuda"), torch.zeros(4, device="cuda"))  # float32 tensors


@pytest.mark.parametrize("dims,order,fields,slab", [
    ((37, 29, 53), 1, 1, None), ((37, 29, 53), 1, 1, (5, 21)), ((19, 23, 17), 1, 3, None), ((19, 23, 17), 1, 3, (7, 14)),
    ((40, 9, 11), 1, 1, (0, 17)), ((11, 13, 9), 2, 1, None), ((64, 5, 5), 1, 1, (30, 65)), ((5, 5, 260), 1, 1, None),
    ((128, 3, 3), 1, 3, None)])
def test_solver_layouts_agree_with_the_csr_kernel_on_odd_shapes(mf, dims, order, fields, slab):
    """Every solver layout the inspector can pick (diagonal slots incl. several lists, explicit columns, row-sorted sliced ELL
    with and without block offset lists) against the CSR tile kernel on bricks with awkward extents, 1 and 3 fields, slabs with
    ghost columns, hex-8 and hex-27 patterns; random values and a random x that includes the ghost entries."""
    import torch
    from metafem_jl_amd import _lib, parallel as par

    b = mf.make_Brick((1.0, 1.0, 1.0), dims, order, 3 if order == 1 else 5)
    m1, m2 = order * dims[1] + 1, order * dims[2] + 1
    if slab is not None:
        b.set_slab(*slab)
    A = b.pattern(fields)
    nloc = A.n if slab is None else par.local_vector_length(slab[0], slab[1], m1, m2, fields)
    K = mf.FEM_rand(A.nnz, 11, 0) - 0.5
    x = mf.FEM_rand(nloc, 12, 0) - 0.5
    y0 = torch.zeros(A.n, dtype=torch.float64, device="cuda")
    mf.mul_(y0, A, K, x)
    scale = float(y0.abs().max())
    _lib.lib.mfem_debug_set_layout_min_rows(0, 0)
    try:
        for ell, sell in ((1, 1), (3, 1), (0, 1), (0, 3)):
            _lib.lib.mfem_debug_set_ell(ell | (6 << 4))
            _lib.lib.mfem_debug_set_sell(sell)
            y1 = torch.full((A.n,), 3.0, dtype=torch.float64, device="cuda")
            _lib.check(_lib.lib.mfem_spmv_solver_layout(b.ctx._h, A._h, K.data_ptr(), x.data_ptr(), y1.data_ptr(), 1.0, 0.0))
            assert float((y0 - y1).abs().max()) <= 1e-13 * scale, (ell, sell)
    finally:
        _lib.lib.mfem_debug_set_ell(1 | (6 << 4))
        _lib.lib.mfem_debug_set_sell(1)
        _lib.lib.mfem_debug_set_layout_min_rows(262144, 1000000)


def test_solver_layouts_on_degenerate_matrices(mf):
    """Empty rows, a matrix smaller than one block, a single dense row among short ones, n = 0: the inspector must fall back
    (or pad) without touching memory it does not own, and the result must equal the CSR kernel."""
    import ctypes as C

    import scipy.sparse as sp
    import torch
    from metafem_jl_amd import _lib

    rng = np.random.default_rng(4)
    mats = []
    n = 700
    M = sp.random(n, n, density=0.01, random_state=1, format="lil")
    M[5, :] = 0.0                      # an empty row
    M[9, :] = rng.standard_normal(n)   # one dense row (700 entries) among rows of ~7
    mats.append(M.tocsr())
    mats.append(sp.random(40, 40, density=0.2, random_state=2, format="csr"))       # less than one 128-row block
    mats.append(sp.csr_matrix((300, 300)))                                             # no entries at all
    mats.append(sp.diags([1.0, -2.0, 1.0], [-1, 0, 1], shape=(1000, 1000)).tocsr())  # tridiagonal: 3 diagonals, short rows at the ends
    _lib.lib.mfem_debug_set_layout_min_rows(0, 0)
    try:
        for M in mats:
            M.sort_indices()
            nn = M.shape[0]
            rp = torch.tensor(M.indptr, dtype=torch.int64, device="cuda")
            ci = torch.tensor(M.indices if M.nnz else np.zeros(1, dtype=np.int32), dtype=torch.int32, device="cuda")
            vals = torch.tensor(M.data if M.nnz else np.zeros(1), dtype=torch.float64, device="cuda")
            A = mf.FEM_SpMat_CSR(rp, ci[:M.nnz] if M.nnz else ci[:0], nn)
            x = torch.tensor(rng.standard_normal(nn), device="cuda")
            y0 = torch.full((nn,), 5.0, dtype=torch.float64, device="cuda")
            y1 = y0.clone()
            mf.mul_(y0, A, vals[:M.nnz], x, 2.0, -1.0)
            _lib.check(_lib.lib.mfem_spmv_solver_layout(A.ctx._h, A._h, vals.data_ptr(), x.data_ptr(), y1.data_ptr(), 2.0, -1.0))
            ref = 2.0 * (M @ x.cpu().numpy()) - 5.0
            assert np.allclose(y0.cpu().numpy(), ref, rtol=1e-13, atol=1e-12)
            assert np.allclose(y1.cpu().numpy(), ref, rtol=1e-13, atol=1e-12)
    finally:
        _lib.lib.mfem_debug_set_layout_min_rows(262144, 1000000)


def test_closing_a_context_closes_the_handles_created_on_it(mf):
    """A pattern / brick destroyed after its context would touch freed memory (mfem_csr_destroy invalidates the context's cached cycle
    graphs): Context.close() closes its children first, so the order in which Python drops the objects does not matter."""
    import torch

    ctx = mf.Context(torch.cuda.current_device())
    brick = mf.Brick((1.0, 1.0, 1.0), (3, 2, 2), 1, 3, ctx=ctx)
    A = brick.pattern(1)
    rp = torch.tensor([0, 1, 2], dtype=torch.int32, device="cuda")
    ci = torch.tensor([0, 1], dtype=torch.int32, device="cuda")
    B = mf.FEM_SpMat_CSR(rp, ci, 2, ctx=ctx)
    ctx.close()
    assert not A._h and not B._h and not brick._h
    A.close()      # closing again is a no-op
    B.close()
    brick.close()
    ctx.close()
    # the C ABI itself tolerates the wrong order for patterns: destroyed after its context a pattern only frees its own memory
    import ctypes as C
    from metafem_jl_amd import _lib
    h_ctx, h_csr = C.c_void_p(), C.c_void_p()
    _lib.check(_lib.lib.mfem_context_create(torch.cuda.current_device(), C.c_void_p(torch.cuda.current_stream().cuda_stream), C.byref(h_ctx)))
    _lib.check(_lib.lib.mfem_csr_create(h_ctx, 2, 2, C.c_void_p(rp.data_ptr()), 32, C.c_void_p(ci.data_ptr()), 0, C.byref(h_csr)))
    _lib.check(_lib.lib.mfem_context_destroy(h_ctx))
    _lib.check(_lib.lib.mfem_csr_destroy(h_csr))
