"""The unstructured serendipity hex-20 legs of bench.py (`u20_thermal_96`, `u20_elasticity_96`): what a user of
mesh_Classical(...; itp_type = :Serendipity, itp_order = 2) + idrs!(s = 8) gets (examples/thermal_conduction/3D_Script.jl:39,49,
examples/linear_elasticity/cantilever/3D_Script.jl:75,88).  The mesh is a brick handed over as UNSTRUCTURED connectivity: control points numbered by
mesh_Classical, element order shuffled in blocks.  Small: K and R against the oracle's term-by-term assembly on the same arrays (oracle/fem.py restates
05_CodeGenerator.jl:52-154 + 06_FEM_Kernel.jl), idrs!(8) against spsolve.  Full size (96^3, 884 736 elements): size-independent properties --
symmetry as a bilinear form, K 1 = the boundary integral, rigid-body modes in the null space of the elasticity operator, the solver layout's product
equal to mul!'s, which kernels ran."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def B():
    import bench
    import bench_legs as L

    return L.Bench(bench.parse_args([]))


def _to_oracle_wf(wf):
    """The term lists of metafem.jl_amd/physics.py are plain data + arithmetic closures: the oracle's FEMDomain (oracle/fem.py) evaluates them on numpy
    arrays as they are."""
    return wf


@pytest.mark.parametrize("fields", [1, 3])
def test_small_mesh_against_the_oracle(mf, B, fields):
    """4^3 elements in shuffled blocks of 8: K_linear_func (mfem_mesh_assemble_elements_rows + _facets on the pattern of mfem_pattern_build) and
    K_nonlinear_func (S3 operators) against the oracle's assembly of the SAME mesh arrays and weak forms; one Newton step with idrs!(8) against LU."""
    import torch

    import bench_legs as L
    from metafem_jl_amd import generic as G, physics
    from oracle import fem, mesh as om, reference_element as re_, solvers

    B._umesh_key = None
    space, msh, fac = B.unstructured_mesh(4, block=8)
    disc = re_.initialize_classical_element(3, "CUBE", 2, 1, 5, itp_type="Serendipity")
    # the oracle on the PRODUCT's arrays (its own mesh_Classical may number the edge nodes in another order: irrelevant here)
    omesh = om.ClassicalMesh(3, np.asarray(msh.coords), np.asarray(msh.cp_ids), np.asarray(msh.vert_conn), msh.n_vertices)
    ofac = om.boundary_facets(omesh)
    assert np.array_equal(ofac.element_ID, fac.element_ID) and np.array_equal(ofac.element_eindex, fac.element_eindex)
    if fields == 1:
        wf = physics.thermal_domain(3, L.K_COND)
        bnd = [(fac, physics.thermal_convection(L.H, L.TENV))]
        obnd = [(ofac, _to_oracle_wf(bnd[0][1]))]
    else:
        wf = physics.elasticity_domain(3, L.LAM, L.MU)
        c, oc = fac.centroid, ofac.centroid
        bnd = [(fac.select(np.abs(c[:, 0]) < 1e-9), physics.penalty([0, 1, 2], L.TAU)), (fac.select(np.abs(c[:, 1] - 1.0) < 1e-9), physics.traction(3, "sl", rows=[1]))]
        obnd = [(ofac.select(np.abs(oc[:, 0]) < 1e-9), _to_oracle_wf(bnd[0][1])), (ofac.select(np.abs(oc[:, 1] - 1.0) < 1e-9), _to_oracle_wf(bnd[1][1]))]
    gd = G.GenericDomain(B.ctx, space, msh.coords, msh.cp_ids, fields, wf, [(f.element_ID, f.element_eindex, w) for f, w in bnd])
    od = fem.FEMDomain(omesh, disc, fields, _to_oracle_wf(wf), obnd)
    ext = {"s": np.full(msh.ncp, L.SRC)} if fields == 1 else {f"sl{v}": np.full(msh.ncp, 1.0 if v == 2 else 0.0) for v in (2, 4, 6)}
    for k, v in ext.items():
        gd.controlpoints[k] = torch.tensor(v, device="cuda")
        od.controlpoints[k] = v
    gd.K_linear_func()
    gd.K_nonlinear_func()
    od.update_time(); od.K_linear_func(); od.update_x_star(); od.K_nonlinear_func()
    Kg, Ko = gd.K_total.cpu().numpy(), od.K_total
    assert np.abs(Kg - Ko).max() <= 1e-12 * np.abs(Ko).max()
    Rg, Ro = gd.residue.cpu().numpy(), od.residue
    assert np.abs(Rg - Ro).max() <= 1e-11 * np.abs(Ro).max()
    dx, st = mf.iterative_Solve(gd.A, gd.K_total, gd.residue, 1e-10 * solvers.normalized_norm(Ro), Sv_func=mf.idrs_, Pr_func=mf.Pr_Jacobi_, maxiter=3000, max_pass=10, s=8)
    ref = solvers.solver_lu_cpu(od.pattern.rowptr, od.pattern.colidx, Ko, Ro)
    assert st.converged == 1 and np.abs(dx.cpu().numpy() - ref).max() <= 1e-7 * np.abs(ref).max()


@pytest.mark.parametrize("fields", [1, 2, 3])
def test_gather_by_node_adds_in_the_order_of_the_gather_by_row(mf, B, fields):
    """Elements alone (no facet launches: those add with atomics), 6^3 hex-20 in shuffled blocks: the row-owner assembly twice and once with the round-5
    gather (one wave per row) -- the same bits (k_mesh_gather_nodes: three nodes per wave with one field, a node's field rows together with more)."""
    import bench_legs as L
    from metafem_jl_amd import _lib, generic as G, physics

    B._umesh_key = None
    space, msh, fac = B.unstructured_mesh(6, block=8)
    if fields == 1:
        wf = physics.thermal_domain(3, L.K_COND)
    elif fields == 3:
        wf = physics.elasticity_domain(3, L.LAM, L.MU)
    else:
        wf = physics.elasticity_domain(2, L.LAM, L.MU)  # two fields coupled through d/dx, d/dy on the 3-D mesh
    gd = G.GenericDomain(B.ctx, space, msh.coords, msh.cp_ids, fields, wf, [])
    rows0 = int(_lib.lib.mfem_debug_mesh_rows_count())
    gd.K_linear_func()
    assert int(_lib.lib.mfem_debug_mesh_rows_count()) > rows0
    a = gd.K_linear.cpu().numpy().copy()
    gd.K_linear_func()
    assert np.array_equal(a, gd.K_linear.cpu().numpy())
    _lib.lib.mfem_debug_set_mesh_gather_rows(1)
    try:
        gd.K_linear_func()
        by_row = gd.K_linear.cpu().numpy()
    finally:
        _lib.lib.mfem_debug_set_mesh_gather_rows(0)
    assert np.abs(a).max() > 0 and np.array_equal(a, by_row)
    B._umesh_key = None


@pytest.mark.parametrize("fields", [1, 3])
@pytest.mark.parametrize("shape", ["CUBE", "SIMPLEX"])
def test_full_size_properties(mf, B, fields, shape):
    """96^3 hex-20 elements (3.6 M control points; 1.9e9 nonzeros with three fields) and the tet-10 leg's mesh (a 64^3 brick cut into 1.31 M 10-node
    tetrahedra), unstructured: properties that hold at any size."""
    import torch

    import bench_legs as L
    from metafem_jl_amd import _lib, generic as G, physics

    n = 96 if shape == "CUBE" else 64
    space, msh, fac = B.unstructured_mesh(n, shape=shape)
    ncp = msh.ncp
    if fields == 1:
        gd = G.GenericDomain(B.ctx, space, msh.coords, msh.cp_ids, 1, physics.thermal_domain(3, L.K_COND),
                             [(fac.element_ID, fac.element_eindex, physics.thermal_convection(L.H, L.TENV))])
    else:  # the operator alone: its null space is the six rigid-body modes
        gd = G.GenericDomain(B.ctx, space, msh.coords, msh.cp_ids, 3, physics.elasticity_domain(3, L.LAM, L.MU), [])
    rows0 = int(_lib.lib.mfem_debug_mesh_rows_count())
    gd.K_linear_func()
    assert int(_lib.lib.mfem_debug_mesh_rows_count()) > rows0  # the row-owner form ran (scratch within its budget)
    A, K = gd.A, gd.K_linear
    N = fields * ncp
    assert A.n == N
    f64 = dict(dtype=torch.float64, device="cuda")
    kmax = float(K.abs().max())
    y = torch.empty(N, **f64)
    if fields == 1:
        # K 1 = -h * (boundary integral of the shape functions): its sum is -h * area, interior rows vanish (the gradients of a constant)
        mf.mul_(y, A, K, torch.ones(N, **f64))
        assert abs(float(y.sum()) + L.H * 6.0) <= 1e-9 * L.H * 6.0
        c = torch.tensor(msh.coords, device="cuda")
        interior = ((c > 1.5 / n) & (c < 1.0 - 1.5 / n)).all(dim=1)
        assert float(y[interior].abs().max()) <= 1e-11 * kmax
        # K x = -k * (Laplace form) - h * (boundary mass): for x = x_1 (linear field, exactly represented) the interior rows vanish too
        mf.mul_(y, A, K, c[:, 0].contiguous())
        assert float(y[interior].abs().max()) <= 1e-11 * kmax
    else:
        c = torch.tensor(msh.coords, device="cuda")
        zero = torch.zeros(ncp, **f64)
        modes = [torch.cat([torch.ones(ncp, **f64) if d == k else zero for d in range(3)]) for k in range(3)]
        modes += [torch.cat([zero, -c[:, 2], c[:, 1]]), torch.cat([c[:, 2], zero, -c[:, 0]]), torch.cat([-c[:, 1], c[:, 0], zero])]
        for m in modes:
            mf.mul_(y, A, K, m.contiguous())
            assert float(y.abs().max()) <= 1e-10 * kmax
    # symmetric as a bilinear form: u . K v = v . K u
    u, v = mf.FEM_rand(N, 1, 0) - 0.5, mf.FEM_rand(N, 2, 0) - 0.5
    Kv, Ku = torch.empty(N, **f64), torch.empty(N, **f64)
    mf.mul_(Kv, A, K, v)
    mf.mul_(Ku, A, K, u)
    a, b = float(u @ Kv), float(v @ Ku)
    assert abs(a - b) <= 1e-10 * (float(u.abs() @ Kv.abs()) + 1e-300)
    # the layout the Krylov loop runs on (mode 3: rows sorted by length IN MESH ORDER -- round 6 --, SELL-128) gives mul!'s product
    mode = C.c_int32()
    _lib.check(_lib.lib.mfem_csr_solver_layout(B.ctx._h, A._h, C.byref(mode), None, None, None))
    assert mode.value == 3
    if fields == 3:  # the three rows of a node share the node's coupling list: the node-blocked form (one column index per 3 x 3 values)
        assert int(_lib.lib.mfem_debug_bsell_fields(A._h)) == 3
    yl = torch.empty(N, **f64)
    _lib.check(_lib.lib.mfem_spmv_solver_layout(B.ctx._h, A._h, K.data_ptr(), v.data_ptr(), yl.data_ptr(), 1.0, 0.0))
    assert float((yl - Kv).abs().max()) <= 1e-13 * float(Kv.abs().max())
    del gd
    torch.cuda.empty_cache()


def test_unstructured_patterns_keep_mesh_order_in_the_sliced_layout(mf, B):
    """Round 6: the row sort of the sliced layout (csrc/spmv_sell.hip) used the hash of a row's diagonal list as its low key -- on a lattice that groups
    the rows of one node type, on an unstructured pattern it SHUFFLES the rows of one length, and the x gathers of a 128-row block come from all over the
    vector (the hex-20 meshes of every shipped example: 0.18 of HBM, 3 x slower than the CSR kernel).  Signatures that do not repeat stay out of the key:
    the layout's product must then be at least as fast as mul! on the same matrix."""
    import torch

    import bench_legs as L
    from metafem_jl_amd import generic as G, physics

    space, msh, fac = B.unstructured_mesh(48)
    gd = G.GenericDomain(B.ctx, space, msh.coords, msh.cp_ids, 1, physics.thermal_domain(3, L.K_COND),
                         [(fac.element_ID, fac.element_eindex, physics.thermal_convection(L.H, L.TENV))])
    gd.K_linear_func()
    gd.controlpoints["s"] = torch.full((msh.ncp,), L.SRC, dtype=torch.float64, device="cuda")
    gd.K_nonlinear_func()
    read = B.spmv_timer()
    mf.iterative_Solve(gd.A, gd.K_total, gd.residue, 1e-300, Sv_func=mf.idrs_, Pr_func=mf.Pr_Jacobi_, maxiter=200, max_pass=1, s=8, fixed_iterations=True)
    layout_ms, n_l = read()
    csr = B.csr_kernel_roofline(gd.A, gd.K_total, "u20_1_48")
    assert n_l >= 200 and layout_ms <= 1.15 * csr["avg_launch_ms"], (layout_ms, csr["avg_launch_ms"])


@pytest.mark.parametrize("n,fields", [(6, 3), (10, 3)])
def test_field_periodic_blocks_of_the_sliced_layout(mf, B, n, fields):
    """Round 6: a field-major multi-field matrix lists, in row (g, i), the nodes coupled to i once per column field -- col[f P + t] = col[t] + f * (rows per
    field).  Full 128-row blocks of the sliced layout with that form (flag 2, found by k_sell_block_periodic) read ONE column slot per node for the F fields:
    a third of the column stream of a three-field matrix.  The product equals mul!'s to round-off and the one with the whole column stream (bit 3 of the
    "sell" knob); it is the same from run to run bit for bit; a one-field matrix has no such blocks."""
    import torch

    import bench_legs as L
    from metafem_jl_amd import _lib, generic as G, physics

    B._umesh_key = None
    space, msh, fac = B.unstructured_mesh(n, block=16)
    wf = physics.elasticity_domain(3, L.LAM, L.MU)
    _lib.lib.mfem_debug_set_layout_min_rows(0, 0)
    try:
        gd = G.GenericDomain(B.ctx, space, msh.coords, msh.cp_ids, fields, wf, [])
        gd.K_linear_func()
        A, K = gd.A, gd.K_linear
        x = mf.FEM_rand(A.n, 3, 0) - 0.5
        mode = C.c_int32()
        _lib.lib.mfem_debug_set_bsell(0)  # (the row-sorted form: the node-blocked one is test_node_blocked_layout)
        _lib.check(_lib.lib.mfem_csr_solver_layout(B.ctx._h, A._h, C.byref(mode), None, None, None))
        assert mode.value == 3 and int(_lib.lib.mfem_debug_bsell_fields(A._h)) == 0
        nper = int(_lib.lib.mfem_debug_sell_periodic_blocks(A._h))
        assert nper >= (A.n // 128) // 2, nper  # (the blocks where the row length changes and the last one are not)
        y0 = torch.empty(A.n, dtype=torch.float64, device="cuda")
        mf.mul_(y0, A, K, x)
        ys = []
        for knob in (1, 1, 1 | 8):
            _lib.lib.mfem_debug_set_sell(knob)
            y = torch.empty(A.n, dtype=torch.float64, device="cuda")
            _lib.check(_lib.lib.mfem_spmv_solver_layout(B.ctx._h, A._h, K.data_ptr(), x.data_ptr(), y.data_ptr(), 1.0, 0.0))
            ys.append(y)
        assert torch.equal(ys[0], ys[1])
        for y in ys:
            assert float((y - y0).abs().max()) <= 1e-13 * float(y0.abs().max())
        # one field: nothing periodic
        g1 = G.GenericDomain(B.ctx, space, msh.coords, msh.cp_ids, 1, physics.thermal_domain(3, L.K_COND), [])
        g1.K_linear_func()
        _lib.check(_lib.lib.mfem_csr_solver_layout(B.ctx._h, g1.A._h, C.byref(mode), None, None, None))
        assert int(_lib.lib.mfem_debug_sell_periodic_blocks(g1.A._h)) == 0
    finally:
        _lib.lib.mfem_debug_set_sell(1)
        _lib.lib.mfem_debug_set_bsell(1)
        _lib.lib.mfem_debug_set_layout_min_rows(262144, 1000000)


@pytest.mark.parametrize("F", [2, 4, 6])
def test_field_periodic_blocks_two_four_six_fields(mf, B, F):
    """The field-periodic form of the sliced layout for other field counts (the cylinder-flow example has four: p, u1, u2, u3): a field-major F-field
    matrix built as kron(ones(F, F), P1) on the pattern P1 of a one-field hex-20 mesh, random values.  F = 6 is served as three periods of two node lists."""
    import scipy.sparse as sp
    import torch

    import bench_legs as L
    from metafem_jl_amd import _lib, generic as G, physics

    B._umesh_key = None
    space, msh, fac = B.unstructured_mesh(7, block=16)
    g1 = G.GenericDomain(B.ctx, space, msh.coords, msh.cp_ids, 1, physics.thermal_domain(3, L.K_COND), [])
    rp = g1.A.rowptr.cpu().numpy().astype(np.int64) - g1.A.index_base
    ci = g1.A.colidx.cpu().numpy().astype(np.int64) - g1.A.index_base
    n1 = g1.A.n
    P1 = sp.csr_matrix((np.ones(ci.size), ci, rp), shape=(n1, n1))
    M = sp.kron(sp.csr_matrix(np.ones((F, F))), P1, format="csr")
    M.sort_indices()
    rng = np.random.default_rng(F)
    M.data = rng.standard_normal(M.nnz)
    n = F * n1
    _lib.lib.mfem_debug_set_layout_min_rows(0, 0)
    try:
        A = mf.FEM_SpMat_CSR(torch.tensor(M.indptr, dtype=torch.int64, device="cuda"), torch.tensor(M.indices, dtype=torch.int32, device="cuda"), n, index_base=0, ctx=B.ctx)
        K = torch.tensor(M.data, device="cuda")
        mode = C.c_int32()
        _lib.lib.mfem_debug_set_bsell(0)
        _lib.check(_lib.lib.mfem_csr_solver_layout(B.ctx._h, A._h, C.byref(mode), None, None, None))
        assert mode.value == 3 and int(_lib.lib.mfem_debug_bsell_fields(A._h)) == 0
        assert int(_lib.lib.mfem_debug_sell_periodic_blocks(A._h)) >= (n // 128) // 2
        x = mf.FEM_rand(n, 3, 0) - 0.5
        want = torch.tensor(M @ x.cpu().numpy(), device="cuda")
        for knob in (1, 1 | 8):
            _lib.lib.mfem_debug_set_sell(knob)
            y = torch.empty(n, dtype=torch.float64, device="cuda")
            _lib.check(_lib.lib.mfem_spmv_solver_layout(B.ctx._h, A._h, K.data_ptr(), x.data_ptr(), y.data_ptr(), 1.0, 0.0))
            assert float((y - want).abs().max()) <= 1e-13 * float(want.abs().max())
    finally:
        _lib.lib.mfem_debug_set_sell(1)
        _lib.lib.mfem_debug_set_bsell(1)
        _lib.lib.mfem_debug_set_layout_min_rows(262144, 1000000)


@pytest.mark.parametrize("F", [2, 3, 4, 6])
def test_node_blocked_layout(mf, B, F):
    """Mode 3 on a field-major multi-field matrix of an unstructured mesh takes the node-blocked form (csrc/spmv_sell.hip "BSELL": a lane owns a node, one
    column index and F gathers of x per F x F values): kron(ones(F, F), P1) on the pattern P1 of a one-field hex-20 mesh with random values, against scipy;
    y = alpha A x + beta y; and the same values through a Jacobi-preconditioned idrs!(8) solve (the layout copy with the column scaling folded in) against
    the row-sorted form.  F = 6 is served as three super-fields of two."""
    import scipy.sparse as sp
    import torch

    import bench_legs as L
    from metafem_jl_amd import _lib, generic as G, physics

    B._umesh_key = None
    space, msh, fac = B.unstructured_mesh(7, block=16)
    g1 = G.GenericDomain(B.ctx, space, msh.coords, msh.cp_ids, 1, physics.thermal_domain(3, L.K_COND), [])
    rp = g1.A.rowptr.cpu().numpy().astype(np.int64) - g1.A.index_base
    ci = g1.A.colidx.cpu().numpy().astype(np.int64) - g1.A.index_base
    n1 = g1.A.n
    P1 = sp.csr_matrix((np.ones(ci.size), ci, rp), shape=(n1, n1))
    M = sp.kron(sp.csr_matrix(np.ones((F, F))), P1, format="csr")
    M.sort_indices()
    rng = np.random.default_rng(10 + F)
    M.data = rng.standard_normal(M.nnz)
    M = (M + sp.diags(np.asarray(abs(M).sum(axis=1)).ravel() + 1.0)).tocsr()  # diagonally dominant: the solve below converges
    M.sort_indices()
    n = F * n1
    _lib.lib.mfem_debug_set_layout_min_rows(0, 0)
    try:
        def pattern():
            return mf.FEM_SpMat_CSR(torch.tensor(M.indptr, dtype=torch.int64, device="cuda"), torch.tensor(M.indices, dtype=torch.int32, device="cuda"), n, index_base=0, ctx=B.ctx)

        A = pattern()
        K = torch.tensor(M.data, device="cuda")
        mode = C.c_int32()
        _lib.check(_lib.lib.mfem_csr_solver_layout(B.ctx._h, A._h, C.byref(mode), None, None, None))
        assert mode.value == 3
        assert int(_lib.lib.mfem_debug_bsell_fields(A._h)) == {2: 2, 3: 3, 4: 4, 6: 3}[F]
        x = mf.FEM_rand(n, 3, 0) - 0.5
        xs = x.cpu().numpy()
        c0 = int(_lib.lib.mfem_debug_bsell_spmv_count())
        y = torch.empty(n, dtype=torch.float64, device="cuda")
        _lib.check(_lib.lib.mfem_spmv_solver_layout(B.ctx._h, A._h, K.data_ptr(), x.data_ptr(), y.data_ptr(), 1.0, 0.0))
        want = M @ xs
        assert int(_lib.lib.mfem_debug_bsell_spmv_count()) > c0
        assert np.abs(y.cpu().numpy() - want).max() <= 1e-13 * np.abs(want).max()
        y2 = y.clone()
        _lib.check(_lib.lib.mfem_spmv_solver_layout(B.ctx._h, A._h, K.data_ptr(), x.data_ptr(), y2.data_ptr(), -0.5, 2.0))
        want2 = -0.5 * want + 2.0 * y.cpu().numpy()
        assert np.abs(y2.cpu().numpy() - want2).max() <= 1e-13 * np.abs(want2).max()
        # a solve (right Jacobi: the layout copy divides by the column's diagonal entry) on both forms
        b = torch.tensor(M @ np.ones(n), device="cuda")
        sols = []
        for on in (1, 0):
            _lib.lib.mfem_debug_set_bsell(on)
            Ab = pattern()
            dx, st = mf.iterative_Solve(Ab, K, b, 1e-11, Sv_func=mf.idrs_, Pr_func=mf.Pr_Jacobi_, maxiter=500, max_pass=4, s=8)
            assert st.converged == 1
            assert int(_lib.lib.mfem_debug_bsell_fields(Ab._h)) == (int(_lib.lib.mfem_debug_bsell_fields(A._h)) if on else 0)
            sols.append(dx.cpu().numpy())
        for sol in sols:
            assert np.abs(sol - 1.0).max() <= 1e-8
    finally:
        _lib.lib.mfem_debug_set_sell(1)
        _lib.lib.mfem_debug_set_bsell(1)
        _lib.lib.mfem_debug_set_layout_min_rows(262144, 1000000)


@pytest.mark.parametrize("coloured", [False, True])
@pytest.mark.parametrize("fields", [1, 3])
def test_wave_forms_of_the_batched_operators(mf, B, fields, coloured):
    """mfem_op_var_batch / mfem_op_res_batch on hex-20 (a persistent wave per item, table slabs in LDS: csrc/ops.hip k_op_*_batch_wave) against the
    sub-wave forms of rounds 2-5 on the same 8^3 shuffled mesh: the residual of K_nonlinear_func at a random x* (inner variables through var, residues
    through res; FP64 atomics or colour batches), and the oracle-checked small case of test_small_mesh_against_the_oracle stays on the old forms (< 256 items)."""
    import torch

    import bench_legs as L
    from metafem_jl_amd import _lib, generic as G, physics

    B._umesh_key = None
    space, msh, fac = B.unstructured_mesh(8, block=8)
    if fields == 1:
        wf = physics.thermal_domain(3, L.K_COND)
        bnd = [(fac.element_ID, fac.element_eindex, physics.thermal_convection(L.H, L.TENV))]
    else:
        wf = physics.elasticity_domain(3, L.LAM, L.MU)
        c = fac.centroid
        wall = fac.select(np.abs(c[:, 0]) < 1e-9)
        bnd = [(wall.element_ID, wall.element_eindex, physics.penalty([0, 1, 2], L.TAU))]
    res = []
    try:
        for on in (1, 0):
            _lib.lib.mfem_debug_set_op_wave_forms(on)
            gd = G.GenericDomain(B.ctx, space, msh.coords, msh.cp_ids, fields, wf, bnd, element_colours="auto" if coloured else None)
            if fields == 1:
                gd.controlpoints["s"] = torch.full((msh.ncp,), L.SRC, dtype=torch.float64, device="cuda")
            rng = np.random.default_rng(3)
            gd.x_star[:fields * msh.ncp] = torch.tensor(rng.standard_normal(fields * msh.ncp), device="cuda")
            gd.K_linear_func()
            gd.K_nonlinear_func()
            res.append(gd.residue.cpu().numpy().copy())
    finally:
        _lib.lib.mfem_debug_set_op_wave_forms(1)
    assert np.abs(res[0]).max() > 0
    assert np.abs(res[0] - res[1]).max() <= 1e-13 * np.abs(res[1]).max()
    B._umesh_key = None
