"""GPU parity: iterative_Solve! with bicgstabl_GS! / idrs! against the oracle restatement and the direct solve."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _thermal_system(n=(7, 6, 5), distort=True):
    from oracle import fem, mesh as om, problems, reference_element as re_

    x = (1.0, 1.0, 1.0)
    disc = re_.initialize_classical_element(3, "CUBE", 1, 1, 3)
    msh = om.lattice_mesh(x, n, disc)
    if distort:
        c = msh.coords
        msh.coords = c + 0.02 * np.stack([np.sin(3 * c[:, 1]), np.sin(2 * c[:, 2]), c[:, 0] * c[:, 1]], axis=1)
    fac = om.boundary_facets_structured(x, n, 3)
    od = fem.FEMDomain(msh, disc, 1, problems.thermal_domain(3, 0.6), [(fac, problems.thermal_convection(25.0, 293.15))])
    od.controlpoints["s"] = np.full(msh.ncp, 1600.0)
    od.update_time(); od.K_linear_func(); od.update_x_star(); od.K_nonlinear_func()
    return od.pattern.rowptr, od.pattern.colidx, od.K_total.copy(), od.residue.copy()


def _nonsymmetric_system():
    """C1's first Newton system: quad-8 serendipity, Nitsche Dirichlet => nonsymmetric K."""
    from oracle import fem, mesh as om, problems, reference_element as re_

    L1, L2, nx, ny = 0.02, 0.01, 16, 8
    disc = re_.initialize_classical_element(2, "CUBE", 2, 1, 5, itp_type="Serendipity")
    vert, conn = om.make_square((L1, L2), (nx, ny))
    mesh = om.mesh_classical(vert, conn, disc)
    fac = om.boundary_facets(mesh)
    err = (L1 / nx) * 0.01
    lr = (np.abs(fac.centroid[:, 0]) < err) | (np.abs(fac.centroid[:, 0] - L1) < err)
    top = np.abs(fac.centroid[:, 1] - L2) < err
    dom = fem.FEMDomain(mesh, disc, 1, problems.thermal_domain(2, 3),
                        [(fac.select(lr), problems.thermal_fixed(2, 1000.0, 1173.15, 3)),
                         (fac.select(top), problems.thermal_convection(50, 323.15, 0.7, 5.669e-8))])
    dom.controlpoints["s"] = np.zeros(mesh.ncp)
    dom.update_time(); dom.K_linear_func(); dom.update_x_star(); dom.K_nonlinear_func()
    return dom.pattern.rowptr, dom.pattern.colidx, dom.K_total.copy(), dom.residue.copy()


def _gpu_solve(mf, sysm, **kw):
    import torch

    rowptr, col, K, b = sysm
    A = mf.FEM_SpMat_CSR(torch.tensor(rowptr, device="cuda"), torch.tensor(col, device="cuda"), b.size)
    Kt = torch.tensor(K, device="cuda")
    dx, st = mf.iterative_Solve(A, Kt, torch.tensor(b, device="cuda"), **kw)
    return dx.cpu().numpy(), st, Kt.cpu().numpy()


@pytest.mark.parametrize("system", ["thermal", "nonsym"])
@pytest.mark.parametrize("method,s", [("bicgstabl_gs", 2), ("bicgstabl_gs", 4), ("idrs", 4), ("idrs", 8), ("idrs", 11),
                                      ("cgs2", 0)])
def test_solver_matches_oracle_and_direct(mf, system, method, s):
    from oracle import solvers

    sysm = _thermal_system() if system == "thermal" else _nonsymmetric_system()
    rowptr, col, K, b = sysm
    ref = solvers.solver_lu_cpu(rowptr, col, K, b)
    tol = 1e-10 * solvers.normalized_norm(b)
    info = solvers.SolveInfo()
    xo = solvers.iterative_solve(rowptr, col, K, b, tol, Sv_func=getattr(solvers, method), maxiter=600, max_pass=6, s=s,
                                 seed=0x5EED, info=info)
    sv = {"bicgstabl_gs": mf.bicgstabl_GS_, "idrs": mf.idrs_, "cgs2": mf.cgs2_}[method]
    x, st, K_after = _gpu_solve(mf, sysm, converge_tol=tol, Sv_func=sv, maxiter=600, max_pass=6, s=s, seed=0x5EED, check_every=5)
    assert st.converged == 1 and st.final_res < tol
    assert np.array_equal(K_after, K)  # scale_in_place = False: the caller's K_total is untouched
    scale = np.abs(ref).max()
    assert np.abs(x - ref).max() <= 1e-8 * scale
    assert np.abs(x - xo).max() <= 1e-8 * scale
    # same algorithm + same shadow vectors: iteration counts agree up to round-off at the stop test
    # (the recurrence residual can cross `tol` one step earlier/later, which may cost or save a restart pass)
    assert abs(st.passes - info.passes) <= 1
    if st.passes == info.passes:
        assert abs(st.iterations - info.iters) <= max(4, s + 1), (st.iterations, info.iters)


def test_first_sweep_iterates_are_identical_to_the_oracle(mf):
    """One BiCGStab(2) sweep / a few IDR steps from x0 = 0 with the same shadow vectors: agreement to round-off."""
    from oracle import solvers

    sysm = _thermal_system((5, 5, 5), distort=False)
    rowptr, col, K, b = sysm
    for method, sv, s, maxiter in [("bicgstabl_gs", mf.bicgstabl_GS_, 2, 3), ("idrs", mf.idrs_, 4, 4), ("cgs2", mf.cgs2_, 0, 3)]:
        xo = solvers.iterative_solve(rowptr, col, K, b, 1e-300, Sv_func=getattr(solvers, method), maxiter=maxiter, max_pass=1,
                                     s=s, seed=7)
        x, st, _ = _gpu_solve(mf, sysm, converge_tol=1e-300, Sv_func=sv, maxiter=maxiter, max_pass=1, s=s, seed=7)
        assert np.abs(x - xo).max() <= 1e-11 * np.abs(xo).max(), method


def test_user_shadow_vectors_and_scale_in_place(mf):
    import torch
    from oracle import solvers

    sysm = _thermal_system((5, 4, 4))
    rowptr, col, K, b = sysm
    rng = np.random.default_rng(9)
    shadow = rng.random(b.size)
    tol = 1e-9 * solvers.normalized_norm(b)
    xo = solvers.iterative_solve(rowptr, col, K, b, tol, Sv_func=solvers.bicgstabl_gs, maxiter=400, max_pass=4, s=2, shadow=shadow)
    x, st, K_after = _gpu_solve(mf, sysm, converge_tol=tol, Sv_func=mf.bicgstabl_GS_, maxiter=400, max_pass=4, s=2,
                                shadow=torch.tensor(shadow, device="cuda"), scale_in_place=True)
    assert np.abs(x - xo).max() <= 1e-8 * np.abs(xo).max()
    # Pr_Jacobi! semantics: the matrix handed in is column-scaled in place (02_Preconditioner.jl:118,141-148)
    A = solvers.csr(rowptr, col, K.copy(), b.size)
    solvers.pr_jacobi(A)
    assert np.allclose(K_after, A.data, rtol=1e-15)


@pytest.mark.parametrize("precond", ["none", "colnorm"])
def test_other_right_preconditioners(mf, precond):
    from oracle import solvers

    sysm = _thermal_system((4, 4, 4))
    rowptr, col, K, b = sysm
    ref = solvers.solver_lu_cpu(rowptr, col, K, b)
    tol = 1e-10 * solvers.normalized_norm(b)
    pr = {"none": mf.Identity, "colnorm": mf.Pr_Jacobi_colnorm_}[precond]
    x, st, _ = _gpu_solve(mf, sysm, converge_tol=tol, Sv_func=mf.idrs_, Pr_func=pr, maxiter=800, max_pass=6, s=8)
    assert st.converged == 1
    assert np.abs(x - ref).max() <= 1e-8 * np.abs(ref).max()


@pytest.mark.parametrize("pr", ["none", "diag"])
@pytest.mark.parametrize("pl", ["diag", "rownorm"])
@pytest.mark.parametrize("method,s", [("idrs", 8), ("bicgstabl_gs", 2), ("cgs2", 0)])
def test_left_jacobi_matches_oracle(mf, pr, pl, method, s):
    """Pl_func = Pl_Jacobi (02_Preconditioner.jl:155-168; the cylinder-flow script's choice) with and without Pr_Jacobi!."""
    from functools import partial
    from oracle import solvers

    sysm = _nonsymmetric_system()
    rowptr, col, K, b = sysm
    ref = solvers.solver_lu_cpu(rowptr, col, K, b)
    tol = 1e-9 * solvers.normalized_norm(b)
    info = solvers.SolveInfo()
    xo = solvers.iterative_solve(rowptr, col, K, b, tol, Sv_func=getattr(solvers, method), maxiter=600, max_pass=6, s=s, seed=3,
                                 Pr_func=solvers.pr_jacobi if pr == "diag" else None,
                                 Pl_func=partial(solvers.pl_jacobi, normalized_by_row=(pl == "rownorm")), info=info)
    sv = {"bicgstabl_gs": mf.bicgstabl_GS_, "idrs": mf.idrs_, "cgs2": mf.cgs2_}[method]
    x, st, K_after = _gpu_solve(mf, sysm, converge_tol=tol, Sv_func=sv, maxiter=600, max_pass=6, s=s, seed=3, check_every=5,
                                Pr_func=mf.Pr_Jacobi_ if pr == "diag" else mf.Identity,
                                Pl_func=mf.Pl_Jacobi_ if pl == "diag" else mf.Pl_Jacobi_rownorm_)
    assert st.converged == 1 and st.final_res < tol
    assert np.array_equal(K_after, K)
    scale = np.abs(ref).max()
    assert np.abs(x - ref).max() <= 1e-7 * scale
    assert np.abs(x - xo).max() <= 1e-7 * scale
    assert abs(st.passes - info.passes) <= 1


def test_left_jacobi_first_iterates_and_cg_refusal(mf):
    from oracle import solvers

    sysm = _nonsymmetric_system()
    rowptr, col, K, b = sysm
    xo = solvers.iterative_solve(rowptr, col, K, b, 1e-300, Sv_func=solvers.idrs, maxiter=5, max_pass=1, s=4, seed=7,
                                 Pl_func=solvers.pl_jacobi)
    x, st, _ = _gpu_solve(mf, sysm, converge_tol=1e-300, Sv_func=mf.idrs_, maxiter=5, max_pass=1, s=4, seed=7, Pl_func=mf.Pl_Jacobi_)
    assert np.abs(x - xo).max() <= 1e-10 * np.abs(xo).max()
    with pytest.raises(mf.MetaFEMError):
        _gpu_solve(mf, sysm, converge_tol=1e-9, Sv_func=mf.cg_, maxiter=5, max_pass=1, Pl_func=mf.Pl_Jacobi_)


def test_zero_rhs_returns_zero_iterations(mf):
    import torch

    rowptr, col, K, b = _thermal_system((3, 3, 3))
    for sv in (mf.cg_, mf.bicgstabl_GS_, mf.idrs_, mf.cgs2_):
        x, st, _ = _gpu_solve(mf, (rowptr, col, K, np.zeros_like(b)), converge_tol=1e-12, Sv_func=sv, maxiter=50, max_pass=2)
        assert st.iterations == 0 and st.passes == 1 and np.all(x == 0.0)


def test_maxiter_and_max_pass_are_honoured(mf):
    sysm = _thermal_system((6, 6, 6))
    x, st, _ = _gpu_solve(mf, sysm, converge_tol=1e-30, Sv_func=mf.idrs_, maxiter=10, max_pass=3, s=4)
    assert st.passes == 3 and st.converged == 0
    assert 3 * 10 <= st.iterations <= 3 * 11  # idrs! returns at iter >= maxiter (04_IDRs.jl:79,92)
    x, st, _ = _gpu_solve(mf, sysm, converge_tol=1e-30, Sv_func=mf.bicgstabl_GS_, maxiter=9, max_pass=2, s=2)
    assert st.passes == 2 and st.iterations == 2 * 9  # iter = 1, 3, 5, 7, 9 (03_BiCGstabl.jl:93-94)


@pytest.mark.parametrize("case", ["thermal_odd_n", "elasticity_3_fields", "hex27_not_eligible"])
def test_slot_major_copy_gives_the_same_solves_as_the_csr_kernel(mf, case):
    """mfem_solve transposes the working values into a slot-major padded copy when the rows are near-uniform (spmv_ell.hip);
    same Krylov iterates as with the CSR tile kernel up to the summation order inside a row."""
    import ctypes as C

    import torch
    from metafem_jl_amd import _lib

    if case == "thermal_odd_n":
        brick = mf.make_Brick((1.0, 1.0, 1.0), (24, 24, 24))  # n = 25^3 = 15625: odd, exercises the pad row of the 2-rows-per-lane kernels
        A = brick.pattern(1)
        K = brick.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
    elif case == "elasticity_3_fields":
        brick = mf.make_Brick((1.0, 1.0, 1.0), (5, 4, 3))
        A = brick.pattern(3)
        K = brick.assemble_elasticity(A, 0.5769, 0.3846, 1000.0, 0x10)
    else:
        brick = mf.make_Brick((1.0, 1.0, 1.0), (3, 3, 3), 2, 5)
        A = brick.pattern(1)
        K = brick.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
    b = mf.FEM_rand(A.n, 3, 1) - 0.5
    out = {}
    # 1: slot-major copy, diagonal-slotted when the pattern allows (default kernel); 1 | 8 << 16: the same without the shared x
    # loads of consecutive diagonals; 3: explicit columns only; 4: no uniform layouts -> row-sorted sliced ELL; 0: CSR tile kernel
    _lib.lib.mfem_debug_set_layout_min_rows(0, 0)  # the layouts are reserved for large systems by default
    for ell in (1, 1 | (8 << 16), 3, 4, 12, 0):  # 12: sliced ELL reading explicit columns even in single-list blocks
        _lib.lib.mfem_debug_set_ell((ell & ~12) | (6 << 4))
        _lib.lib.mfem_debug_set_sell(0 if ell == 0 else (3 if ell == 12 else 1))
        try:
            for sv, s in ((mf.idrs_, 4), (mf.bicgstabl_GS_, 2), (mf.cgs2_, 0)):
                x, st = mf.iterative_Solve(A, K, b, 1e-300, Sv_func=sv, maxiter=6, max_pass=1, s=s, seed=11)
                out[(ell, sv)] = x.cpu().numpy()
            if case != "elasticity_3_fields":  # -K of the thermal form is symmetric positive definite
                x, st = mf.iterative_Solve(A, -K, -b, 1e-9 * float(mf.normalized_norm(b)), Sv_func=mf.cg_, maxiter=2000, max_pass=4)
                assert st.converged == 1
                out[(ell, "cg")] = x.cpu().numpy()
        finally:
            _lib.lib.mfem_debug_set_ell(1 | (6 << 4))
            _lib.lib.mfem_debug_set_sell(1)
    if case == "thermal_odd_n":
        mode = C.c_int32()
        _lib.check(_lib.lib.mfem_csr_solver_layout(brick.ctx._h, A._h, C.byref(mode), None, None, None))
        assert mode.value == 2  # (smaller bricks have too many boundary rows: > 10 % padding rules the uniform layouts out)
    _lib.lib.mfem_debug_set_layout_min_rows(262144, 1000000)
    for key in [k for k in out if k[0] != 0]:
        a, c = out[key], out[(0, key[1])]
        tol = 1e-7 if key[1] == "cg" else 1e-10
        assert np.abs(a - c).max() <= tol * np.abs(c).max(), key


def test_solver_layout_inspector(mf):
    """mfem_csr_solver_layout: lattice stencil -> diagonal-slotted (mode 2), 3-field blocks -> explicit columns (mode 1),
    hex-27 rows of 27..125 entries -> CSR tile kernel (mode 0); and a diagonal-slotted solve at a size with many regular
    blocks equals the CSR-kernel solve."""
    import ctypes as C

    import torch
    from metafem_jl_amd import _lib

    def layout(brick, A):
        mode, slots, npad, reg = C.c_int32(), C.c_int32(), C.c_int64(), C.c_int64()
        _lib.check(_lib.lib.mfem_csr_solver_layout(brick.ctx._h, A._h, C.byref(mode), C.byref(slots), C.byref(npad), C.byref(reg)))
        return mode.value, slots.value, npad.value, reg.value

    b0 = mf.make_Brick((1.0, 1.0, 1.0), (24, 24, 24))
    assert layout(b0, b0.pattern(1))[0] == 0        # default thresholds: 15 625 rows are launch-bound -> CSR tile kernel
    _lib.lib.mfem_debug_set_layout_min_rows(0, 0)
    b1 = mf.make_Brick((1.0, 1.0, 1.0), (24, 24, 24))
    A1 = b1.pattern(1)
    m, slots, npad, reg = layout(b1, A1)
    assert (m, slots) == (2, 27) and npad % 128 == 0 and npad >= A1.n and 0.5 * A1.n <= reg <= A1.n
    assert layout(b1, b1.pattern(3))[:2] == (5, 81)  # field-major 3-field 27-point matrix: symmetric lattice tiles (if the values of the solve are symmetric)
    _lib.lib.mfem_debug_set_lat8(0)
    try:
        b1s = mf.make_Brick((1.0, 1.0, 1.0), (24, 24, 24))
        assert layout(b1s, b1s.pattern(3))[:2] == (2, 81)  # ... else diagonal slots: one diagonal list per row field
    finally:
        _lib.lib.mfem_debug_set_lat8(1)
    b3 = mf.make_Brick((1.0, 1.0, 1.0), (6, 6, 6))
    assert layout(b3, b3.pattern(1))[0] == 3        # small brick: boundary rows would need > 10 % padding in a uniform layout
    b27 = mf.make_Brick((1.0, 1.0, 1.0), (4, 4, 4), 2, 5)
    assert layout(b27, b27.pattern(1))[0] == 4      # the hex-27 lattice stencil: symmetric lattice tiles (if the values of the solve are symmetric)
    _lib.lib.mfem_debug_set_lat27(0)
    try:
        b27s = mf.make_Brick((1.0, 1.0, 1.0), (4, 4, 4), 2, 5)
        assert layout(b27s, b27s.pattern(1))[0] == 3  # 27 / 45 / 75 / 125 entries per row: row-sorted sliced ELL
    finally:
        _lib.lib.mfem_debug_set_lat27(1)
    b2 = mf.make_Brick((1.0, 1.0, 1.0), (3, 3, 3))
    assert layout(b2, b2.pattern(1))[0] == 0        # 64 rows: less than one block, CSR tile kernel
    K = b1.assemble_thermal(A1, 0.6, 25.0, 293.15, 0x3F)
    rhs = mf.FEM_rand(A1.n, 5, 0) - 0.5
    res = {}
    for ell in (1, 0):
        _lib.lib.mfem_debug_set_ell(ell | (6 << 4))
        _lib.lib.mfem_debug_set_sell(ell)
        try:
            res[ell] = mf.iterative_Solve(A1, K, rhs, 1e-300, Sv_func=mf.bicgstabl_GS_, maxiter=8, max_pass=1, s=2, seed=3)[0].cpu().numpy()
        finally:
            _lib.lib.mfem_debug_set_ell(1 | (6 << 4))
            _lib.lib.mfem_debug_set_sell(1)
    _lib.lib.mfem_debug_set_layout_min_rows(262144, 1000000)
    assert np.abs(res[1] - res[0]).max() <= 1e-10 * np.abs(res[0]).max()


@pytest.mark.parametrize("rp_dtype,base", [("int32", 1), ("int64", 1), ("int32", 0)])
def test_slot_major_layouts_on_caller_supplied_csr(mf, rp_dtype, base):
    """The reference hands CUSPARSE a 1-based Int32 CSR (04_GPU_Utils.jl:131).  A lattice matrix supplied that way takes the
    diagonal-slotted layout, a randomly perturbed (but still near-uniform) pattern the explicit-column layout; both solve
    like scipy."""
    import ctypes as C

    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    import torch
    from metafem_jl_amd import _lib

    m = 28
    n = m ** 3
    idx = np.arange(n).reshape(m, m, m)
    rows, cols, vals = [], [], []
    rng = np.random.default_rng(0)
    for di in (-1, 0, 1):
        for dj in (-1, 0, 1):
            for dk in (-1, 0, 1):
                src = idx[max(0, -di):m - max(0, di), max(0, -dj):m - max(0, dj), max(0, -dk):m - max(0, dk)]
                dst = idx[max(0, di):m - max(0, -di), max(0, dj):m - max(0, -dj), max(0, dk):m - max(0, -dk)]
                rows.append(src.ravel()); cols.append(dst.ravel())
                vals.append(np.full(src.size, 30.0 if (di, dj, dk) == (0, 0, 0) else -1.0 + 0.2 * rng.random(src.size)))
    M = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n))
    M.sort_indices()
    b = rng.standard_normal(n)
    _lib.lib.mfem_debug_set_layout_min_rows(0, 0)
    for perturb in (False, True):
        A_ = M.copy()
        if perturb:  # swap one neighbour of every 7th row for a far column: rows stay 27 long but leave the diagonals
            A_ = A_.tolil()
            for r in range(0, n, 7):
                c_old = A_.rows[r][0]
                c_new = (r + n // 2) % n
                if c_new not in A_.rows[r]:
                    A_[r, c_new] = A_[r, c_old]
                    A_[r, c_old] = 0.0
            A_ = A_.tocsr()
            A_.eliminate_zeros()
            A_.sort_indices()
        ref = spl.spsolve(A_.tocsc(), b)
        rp = torch.tensor(A_.indptr + base, dtype=getattr(torch, rp_dtype), device="cuda")
        ci = torch.tensor(A_.indices + base, dtype=torch.int32, device="cuda")
        A = mf.FEM_SpMat_CSR(rp, ci, n, index_base=base)
        mode = C.c_int32()
        _lib.check(_lib.lib.mfem_csr_solver_layout(A.ctx._h, A._h, C.byref(mode), None, None, None))
        assert mode.value == (1 if perturb else 2)
        x, st = mf.iterative_Solve(A, torch.tensor(A_.data, device="cuda"), torch.tensor(b, device="cuda"),
                                   1e-12 * float(np.linalg.norm(b) / np.sqrt(n)), Sv_func=mf.bicgstabl_GS_, maxiter=500, max_pass=5, s=2)
        assert st.converged == 1
        assert np.abs(x.cpu().numpy() - ref).max() <= 1e-9 * np.abs(ref).max()
    _lib.lib.mfem_debug_set_layout_min_rows(262144, 1000000)


def _sym_brick(mf, slab=None):
    """21 x 64 x 64 control points: a lattice plane is 4096 rows = 8 chunks of 512, so the symmetric sweep kernel applies."""
    brick = mf.make_Brick((1.0, 2.0, 1.5), (20, 63, 63))
    if slab is not None:
        brick.set_slab(*slab)
    A = brick.pattern(1)
    K = brick.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
    return brick, A, K


@pytest.mark.parametrize("slab", [None, (5, 17)])
def test_symmetric_sweep_spmv_is_bitwise_the_plain_kernel(mf, slab):
    """27-point lattice stencil + bitwise symmetric values: the sweep kernel (lower diagonals mirrored through LDS) must give the
    same y as the plain diagonal-slotted kernel bit for bit, on the whole mesh and on a slab with ghost planes on both sides."""
    import ctypes as C

    import torch
    from metafem_jl_amd import _lib, parallel as par

    brick, A, K = _sym_brick(mf, slab)
    nloc = A.n if slab is None else par.local_vector_length(slab[0], slab[1], 64, 64, 1)
    x = mf.FEM_rand(nloc, 5, 0) - 0.5
    _lib.lib.mfem_debug_set_layout_min_rows(0, 0)
    try:
        ent, sym = C.c_int64(), C.c_int32()
        _lib.check(_lib.lib.mfem_csr_solver_layout_entries(brick.ctx._h, A._h, C.byref(ent), C.byref(sym)))
        assert sym.value == 2 and ent.value < 0.8 * 27 * A.n  # a third and more of the entries come from LDS (2 = wave-private patch sweep)
        ys = []
        for knob in (1 << 22, 1 << 23, 0, 1 << 27):  # plain kernel, workgroup-tile sweep, wave-private patch sweep (values bound in one pass / in two)
            _lib.lib.mfem_debug_set_ell(1 | knob)
            before = _lib.lib.mfem_debug_sym_spmv_count()
            y = torch.full((A.n,), 3.0, dtype=torch.float64, device="cuda")
            _lib.check(_lib.lib.mfem_spmv_solver_layout(brick.ctx._h, A._h, K.data_ptr(), x.data_ptr(), y.data_ptr(), 1.0, 0.0))
            assert (_lib.lib.mfem_debug_sym_spmv_count() > before) == (knob != 1 << 22)
            ys.append(y)
        assert all(torch.equal(ys[0], yy) for yy in ys[1:])
        yc = torch.zeros(A.n, dtype=torch.float64, device="cuda")
        _lib.lib.mfem_debug_set_ell(0)
        mf.mul_(yc, A, K, x)  # CSR kernel
        assert float((yc - ys[1]).abs().max()) <= 1e-13 * float(yc.abs().max())
        # alpha / beta form
        _lib.lib.mfem_debug_set_ell(1)
        y2 = ys[1].clone()
        _lib.check(_lib.lib.mfem_spmv_solver_layout(brick.ctx._h, A._h, K.data_ptr(), x.data_ptr(), y2.data_ptr(), -0.5, 2.0))
        assert float((y2 - 1.5 * ys[1]).abs().max()) <= 1e-12 * float(yc.abs().max())
    finally:
        _lib.lib.mfem_debug_set_ell(1)
        _lib.lib.mfem_debug_set_layout_min_rows(262144, 1000000)


def test_symmetric_sweep_is_refused_for_unsymmetric_values(mf):
    """One entry changed by one ulp: the per-solve check fails, the plain kernel runs, and the result is the unsymmetric product."""
    import torch
    from metafem_jl_amd import _lib

    brick, A, K = _sym_brick(mf)
    x = mf.FEM_rand(A.n, 7, 0) - 0.5
    rp = A.rowptr.cpu().numpy()
    row = 9 * 4096 + 31 * 64 + 17  # an interior control point
    K2 = K.clone()
    K2[int(rp[row]) + 20] = torch.nextafter(K2[int(rp[row]) + 20], torch.tensor(float("inf"), dtype=torch.float64, device="cuda"))
    _lib.lib.mfem_debug_set_layout_min_rows(0, 0)
    try:
        before = _lib.lib.mfem_debug_sym_spmv_count()
        y = torch.zeros(A.n, dtype=torch.float64, device="cuda")
        _lib.check(_lib.lib.mfem_spmv_solver_layout(brick.ctx._h, A._h, K2.data_ptr(), x.data_ptr(), y.data_ptr(), 1.0, 0.0))
        assert _lib.lib.mfem_debug_sym_spmv_count() == before
        yc = torch.zeros(A.n, dtype=torch.float64, device="cuda")
        _lib.lib.mfem_debug_set_ell(0)
        mf.mul_(yc, A, K2, x)
        assert float((yc - y).abs().max()) <= 1e-13 * float(yc.abs().max())
        # and the symmetric values of the same pattern go through the sweep kernel again
        _lib.lib.mfem_debug_set_ell(1)
        _lib.check(_lib.lib.mfem_spmv_solver_layout(brick.ctx._h, A._h, K.data_ptr(), x.data_ptr(), y.data_ptr(), 1.0, 0.0))
        assert _lib.lib.mfem_debug_sym_spmv_count() > before
    finally:
        _lib.lib.mfem_debug_set_ell(1)
        _lib.lib.mfem_debug_set_layout_min_rows(262144, 1000000)


def test_fill_fingerprint_gives_the_check_passes_verdict(mf):
    """Round 6: whether the swept rows of a solve's copy are bitwise symmetric is decided by a fingerprint the fill pass sums while it
    writes them -- sign(c - r) x hash(pair) x bits(value), 64-bit wrap-around arithmetic: symmetric pairs cancel exactly in any order -- instead of a
    separate pass over the copy (k_spmv_symp<1>, a quarter of the per-solve work; bit 30 of the "ell" knob restores it).  Same verdicts: symmetric values,
    one ulp off in an interior pair, one ulp off next to the lattice edge, a NaN."""
    import torch
    from metafem_jl_amd import _lib

    brick, A, K = _sym_brick(mf)
    x = mf.FEM_rand(A.n, 7, 0) - 0.5
    rp = A.rowptr.cpu().numpy()
    inf = torch.tensor(float("inf"), dtype=torch.float64, device="cuda")
    cases = {"symmetric": K}
    for name, row, slot in (("interior pair", 9 * 4096 + 31 * 64 + 17, 20), ("upper entry", 15 * 4096 + 5 * 64 + 40, 25), ("lower entry", 18 * 4096 + 40 * 64 + 3, 2),
                            ("next to the edge", 2 * 4096 + 1 * 64 + 1, 14)):
        K2 = K.clone()
        k = int(rp[row]) + slot
        K2[k] = torch.nextafter(K2[k], inf)
        cases[name] = K2
    K2 = K.clone()
    K2[int(rp[17 * 4096 + 17 * 64 + 17]) + 5] = float("nan")
    cases["nan"] = K2
    _lib.lib.mfem_debug_set_layout_min_rows(0, 0)
    try:
        verdicts = {}
        for bit30 in (0, 1):
            _lib.lib.mfem_debug_set_ell(1 | (bit30 << 30))
            f0 = int(_lib.lib.mfem_debug_symp_fingerprint_count())
            for name, Kc in cases.items():
                before = _lib.lib.mfem_debug_sym_spmv_count()
                y = torch.zeros(A.n, dtype=torch.float64, device="cuda")
                _lib.check(_lib.lib.mfem_spmv_solver_layout(brick.ctx._h, A._h, Kc.data_ptr(), x.data_ptr(), y.data_ptr(), 1.0, 0.0))
                verdicts[(bit30, name)] = _lib.lib.mfem_debug_sym_spmv_count() > before
            assert (int(_lib.lib.mfem_debug_symp_fingerprint_count()) > f0) == (bit30 == 0)
        for name in cases:
            assert verdicts[(0, name)] == (name == "symmetric"), (name, verdicts)  # the fingerprint looks at EVERY pair among the swept rows
            assert verdicts[(1, name)] or not verdicts[(0, name)], (name, verdicts)  # ... the check pass at the pairs the sweep mirrors: never stricter
        assert verdicts[(1, "symmetric")] and not verdicts[(1, "interior pair")]
    finally:
        _lib.lib.mfem_debug_set_ell(1)
        _lib.lib.mfem_debug_set_layout_min_rows(262144, 1000000)


@pytest.mark.parametrize("n", [(9, 7, 11), (24, 20, 18)])
def test_cg_carrying_the_preconditioned_residual_matches_the_classic_recurrence(mf, n):
    """cg_variant 3 (the one-rank default) carries z = M^-1 r instead of r -- one vector stream less per iteration; same iterates
    as the classic recurrence up to round-off: iteration counts within one, solutions to 1e-10, with and without the Jacobi
    preconditioner, on vectors of odd and even length."""
    import torch

    brick = mf.make_Brick((1.0, 2.0, 1.5), n)
    A = brick.pattern(1)
    K = brick.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
    b = mf.FEM_rand(A.n, 11, 0) - 0.5
    for pr in (mf.Pr_Jacobi_, mf.Identity):
        sol = {}
        for var in (1, 3, 0):
            x, st = mf.iterative_Solve(A, K, b, 1e-11, Sv_func=mf.cg_, Pr_func=pr, maxiter=3000, max_pass=2, cg_variant=var)
            assert st.converged == 1
            sol[var] = (x, st.iterations)
        assert abs(sol[1][1] - sol[3][1]) <= 1 and sol[0][1] == sol[3][1]
        assert float((sol[1][0] - sol[3][0]).abs().max()) <= 1e-10 * float(sol[1][0].abs().max())
        assert torch.equal(sol[0][0], sol[3][0])  # auto = 3 on one rank
        r = b.clone()
        mf.mul_(r, A, K, sol[3][0], -1.0, 1.0)
        assert float(r.norm()) / A.n ** 0.5 <= 2e-11


def test_cg_with_symmetric_sweep_matches_plain_kernel(mf):
    """Same CG, SpMVs on the sweep kernel vs the plain kernel: identical iteration count, solutions equal to round-off."""
    import torch
    from metafem_jl_amd import _lib

    brick, A, K = _sym_brick(mf)
    b = mf.FEM_rand(A.n, 11, 0)
    _lib.lib.mfem_debug_set_layout_min_rows(0, 0)
    try:
        out = []
        for knob in (1 << 22, 1 << 23, 0, 1 << 27):
            _lib.lib.mfem_debug_set_ell(1 | knob)
            before = _lib.lib.mfem_debug_sym_spmv_count()
            x, st = mf.iterative_Solve(A, K, b, 1e-11, Sv_func=mf.cg_, maxiter=2000, max_pass=2)
            assert st.converged == 1
            assert (_lib.lib.mfem_debug_sym_spmv_count() > before) == (knob != 1 << 22)
            out.append((x, st.iterations))
        for o in out[1:]:
            assert abs(out[0][1] - o[1]) <= 1
            assert float((out[0][0] - o[0]).abs().max()) <= 1e-9 * float(out[0][0].abs().max())
    finally:
        _lib.lib.mfem_debug_set_ell(1)
        _lib.lib.mfem_debug_set_layout_min_rows(262144, 1000000)


def test_scaled_cg_matches_the_jacobi_recurrence(mf):
    """cg_variant 4: plain CG on S^-1 A S^-1 (S = sqrt|diag|, folded into the layout copy as a * (t_r t_c), t = 1 / S, the product of the factors formed first, so the
    mirrored sweep still applies) -- the Jacobi-preconditioned iteration without the 1 / d stream.  On a matrix whose diagonal varies by three
    orders of magnitude: the sweep kernel runs, the solution equals the classic recurrence's to 1e-9 and the TRUE residual meets the tolerance;
    with unsymmetric values the variant falls back to the classic recurrence on the plain kernel."""
    import torch
    from metafem_jl_amd import _lib

    brick, A, K = _sym_brick(mf)
    n = A.n
    # symmetric rescaling of the thermal matrix: K' = T K T with T = diag(t), t in [1, 30] -- stays bitwise symmetric (t_r * t_c formed first)
    t = 1.0 + 29.0 * mf.FEM_rand(n, 3, 0)
    rows = torch.repeat_interleave(torch.arange(n, device="cuda"), (A.rowptr[1:] - A.rowptr[:-1]).long())
    cols = A.colidx.long() - int(getattr(A, "index_base", 0))
    Kt = K * (t[rows] * t[cols])
    b = mf.FEM_rand(n, 11, 0) - 0.5
    _lib.lib.mfem_debug_set_layout_min_rows(0, 0)
    try:
        c0 = _lib.lib.mfem_debug_sym_spmv_count()
        x3, st3 = mf.iterative_Solve(A, Kt, b, 1e-10, Sv_func=mf.cg_, maxiter=4000, max_pass=3, cg_variant=3)
        c1 = _lib.lib.mfem_debug_sym_spmv_count()
        x4, st4 = mf.iterative_Solve(A, Kt, b, 1e-10, Sv_func=mf.cg_, maxiter=4000, max_pass=3, cg_variant=4)
        c2 = _lib.lib.mfem_debug_sym_spmv_count()
        assert st3.converged == 1 and st4.converged == 1 and c1 > c0 and c2 > c1
        assert float((x3 - x4).abs().max()) <= 1e-9 * float(x3.abs().max())
        r = b.clone()
        mf.mul_(r, A, Kt, x4, -1.0, 1.0)
        assert float(r.norm()) / n ** 0.5 <= 1e-10
        assert abs(st4.final_res - float(r.norm()) / n ** 0.5) <= 1e-3 * st4.final_res + 1e-14  # the reported residual is the true one
        assert abs(st4.iterations - st3.iterations) <= 2  # (the same stopping rule: the kernels switch to the true residual norm near the tolerance)
        # unsymmetric values: the classic recurrence on the plain kernel
        rp = A.rowptr.cpu().numpy()
        row = 7 * 4096 + 21 * 64 + 33
        Ku = Kt.clone()
        Ku[int(rp[row]) + 20] *= 1.0 + 1e-3
        xa, sta = mf.iterative_Solve(A, Ku, b, 1e-30, Sv_func=mf.cg_, maxiter=30, max_pass=1, fixed_iterations=True, cg_variant=4)
        xb, stb = mf.iterative_Solve(A, Ku, b, 1e-30, Sv_func=mf.cg_, maxiter=30, max_pass=1, fixed_iterations=True, cg_variant=1)
        assert float((xa - xb).abs().max()) <= 1e-11 * float(xb.abs().max())
    finally:
        _lib.lib.mfem_debug_set_layout_min_rows(262144, 1000000)


def test_cg_graph_cache_follows_the_symmetry_of_the_values(mf):
    """Same buffers, first symmetric then unsymmetric values: the cached CG cycle graph of the sweep kernel must not be replayed
    on the unsymmetric matrix (the graph key carries the outcome of the per-solve symmetry check)."""
    import torch
    from metafem_jl_amd import _lib

    brick, A, K = _sym_brick(mf)
    b = mf.FEM_rand(A.n, 13, 0)
    rp = A.rowptr.cpu().numpy()
    row = 7 * 4096 + 21 * 64 + 33  # entry 20 of this row is mirrored by both sweep kernels (not at a patch / tile edge)
    Kw = K.clone()
    _lib.lib.mfem_debug_set_layout_min_rows(0, 0)
    try:
        c0 = _lib.lib.mfem_debug_sym_spmv_count()
        x1, st1 = mf.iterative_Solve(A, Kw, b, 1e-30, Sv_func=mf.cg_, maxiter=40, max_pass=1, fixed_iterations=True)
        c1 = _lib.lib.mfem_debug_sym_spmv_count()
        assert c1 > c0
        # break the symmetry visibly (BiCGStab-type perturbation is irrelevant here: CG just iterates 40 times)
        Kw[int(rp[row]) + 20] *= 1.0 + 1e-3
        x2, st2 = mf.iterative_Solve(A, Kw, b, 1e-30, Sv_func=mf.cg_, maxiter=40, max_pass=1, fixed_iterations=True)
        assert _lib.lib.mfem_debug_sym_spmv_count() == c1  # plain kernel
        _lib.lib.mfem_debug_set_ell(1 | (1 << 22))
        x3, st3 = mf.iterative_Solve(A, Kw, b, 1e-30, Sv_func=mf.cg_, maxiter=40, max_pass=1, fixed_iterations=True)
        assert float((x2 - x3).abs().max()) <= 1e-12 * float(x3.abs().max())
        assert float((x2 - x1).abs().max()) > 0.0
    finally:
        _lib.lib.mfem_debug_set_ell(1)
        _lib.lib.mfem_debug_set_layout_min_rows(262144, 1000000)


@pytest.mark.parametrize("n", [(20, 64, 64), (24, 70, 58), (12, 95, 47), (16, 33, 64), (14, 20, 129), (40, 12, 30)])
def test_symmetric_sweep_with_drifting_tiles(mf, n):
    """Lattice planes that are not a whole number of 512-row tiles (4225 = 8.25 tiles, ...): the tiles drift against the plane from
    step to step and the mirror lookups have to follow; also lattice lines of 48 .. 96 points.  y must equal the plain kernel's."""
    import ctypes as C

    import torch
    from metafem_jl_amd import _lib

    brick = mf.make_Brick((1.0, 2.0, 1.5), n)
    A = brick.pattern(1)
    K = brick.assemble_thermal(A, 0.6, 25.0, 293.15, 0x2D)
    x = mf.FEM_rand(A.n, 5, 0) - 0.5
    _lib.lib.mfem_debug_set_layout_min_rows(0, 0)
    try:
        ent, sym = C.c_int64(), C.c_int32()
        _lib.check(_lib.lib.mfem_csr_solver_layout_entries(brick.ctx._h, A._h, C.byref(ent), C.byref(sym)))
        assert sym.value == 2
        ys = []
        for knob in (1 << 22, 0, 1 << 27):  # (the workgroup-tile sweep needs >= 8 tiles per plane: not all of these lattices qualify)
            _lib.lib.mfem_debug_set_ell(1 | knob)
            before = _lib.lib.mfem_debug_sym_spmv_count()
            y = torch.full((A.n,), -2.0, dtype=torch.float64, device="cuda")
            _lib.check(_lib.lib.mfem_spmv_solver_layout(brick.ctx._h, A._h, K.data_ptr(), x.data_ptr(), y.data_ptr(), 1.0, 0.0))
            assert (_lib.lib.mfem_debug_sym_spmv_count() > before) == (knob != 1 << 22)
            ys.append(y)
        assert torch.equal(ys[0], ys[1]) and torch.equal(ys[0], ys[2])
        # alpha / beta form through the same kernels
        y2 = ys[1].clone()
        _lib.check(_lib.lib.mfem_spmv_solver_layout(brick.ctx._h, A._h, K.data_ptr(), x.data_ptr(), y2.data_ptr(), -0.5, 2.0))
        assert float((y2 - 1.5 * ys[1]).abs().max()) <= 1e-12 * float(ys[1].abs().max())
    finally:
        _lib.lib.mfem_debug_set_ell(1)
        _lib.lib.mfem_debug_set_layout_min_rows(262144, 1000000)


@pytest.mark.parametrize("rp_dtype,base", [("int32", 1), ("int64", 0)])
def test_patch_sweep_on_a_caller_supplied_symmetric_lattice_matrix(mf, rp_dtype, base):
    """A symmetric 27-point lattice operator handed over the way the reference hands matrices to CUSPARSE (1-based Int32 CSR,
    04_GPU_Utils.jl:131), on a lattice with three different extents: the inspector recognises the lattice from the CSR pattern alone, CG
    runs on the wave-private patch sweep (bitwise-symmetric values) and solves like scipy; with one entry changed by one ulp the solve
    silently takes the plain kernel and still solves the (then unsymmetric) system."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    import torch
    from metafem_jl_amd import _lib

    m0, m1, m2 = 20, 19, 37
    n = m0 * m1 * m2
    idx = np.arange(n).reshape(m0, m1, m2)
    rows, cols, vals = [], [], []
    for di in (-1, 0, 1):
        for dj in (-1, 0, 1):
            for dk in (-1, 0, 1):
                src = idx[max(0, -di):m0 - max(0, di), max(0, -dj):m1 - max(0, dj), max(0, -dk):m2 - max(0, dk)].ravel()
                dst = idx[max(0, di):m0 - max(0, -di), max(0, dj):m1 - max(0, -dj), max(0, dk):m2 - max(0, -dk)].ravel()
                lo, hi = np.minimum(src, dst), np.maximum(src, dst)
                w = -1.0 + 0.2 * (((lo * 2654435761 + hi * 40503) % 1000) / 1000.0)  # a function of the unordered pair: symmetric bit for bit
                rows.append(src); cols.append(dst)
                vals.append(np.where(src == dst, 30.0, w))
    M = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n))
    M.sort_indices()
    assert (M - M.T).nnz == 0
    rng = np.random.default_rng(1)
    b = rng.standard_normal(n)
    rp = torch.tensor(M.indptr + base, dtype=getattr(torch, rp_dtype), device="cuda")
    ci = torch.tensor(M.indices + base, dtype=torch.int32, device="cuda")
    tol = 1e-12 * float(np.linalg.norm(b) / np.sqrt(n))
    _lib.lib.mfem_debug_set_layout_min_rows(0, 0)
    try:
        A = mf.FEM_SpMat_CSR(rp, ci, n, index_base=base)
        K = torch.tensor(M.data, device="cuda")
        before = _lib.lib.mfem_debug_sym_spmv_count()
        x, st = mf.iterative_Solve(A, K, torch.tensor(b, device="cuda"), tol, Sv_func=mf.cg_, maxiter=500, max_pass=3)
        assert st.converged == 1 and _lib.lib.mfem_debug_sym_spmv_count() > before
        ref = spl.spsolve(M.tocsc(), b)
        assert np.abs(x.cpu().numpy() - ref).max() <= 1e-9 * np.abs(ref).max()
        # one ulp off the symmetry in a mirrored pair: no sweep, same quality of solution for the perturbed matrix
        r = (9 * m1 + 9) * m2 + 17
        k = int(M.indptr[r]) + 20
        K2 = K.clone()
        K2[k] = torch.nextafter(K2[k], torch.tensor(float("inf"), dtype=torch.float64, device="cuda"))
        before = _lib.lib.mfem_debug_sym_spmv_count()
        x2, st2 = mf.iterative_Solve(A, K2, torch.tensor(b, device="cuda"), tol, Sv_func=mf.cg_, maxiter=500, max_pass=3, cg_variant=3)
        assert st2.converged == 1 and _lib.lib.mfem_debug_sym_spmv_count() == before
        assert np.abs(x2.cpu().numpy() - ref).max() <= 1e-9 * np.abs(ref).max()
        # the default on this layout is the scaled CG (cg_variant 4): its check sees the SCALED copy a * (t_r t_c), where a difference of one ulp
        # can vanish in the rounding of the product (the iteration then runs on a matrix that IS symmetric); eight ulps cannot
        K3 = K.clone()
        for _ in range(8):
            K3[k] = torch.nextafter(K3[k], torch.tensor(float("inf"), dtype=torch.float64, device="cuda"))
        before = _lib.lib.mfem_debug_sym_spmv_count()
        x3, st3 = mf.iterative_Solve(A, K3, torch.tensor(b, device="cuda"), tol, Sv_func=mf.cg_, maxiter=500, max_pass=3)
        assert st3.converged == 1 and _lib.lib.mfem_debug_sym_spmv_count() == before
        assert np.abs(x3.cpu().numpy() - ref).max() <= 1e-9 * np.abs(ref).max()
    finally:
        _lib.lib.mfem_debug_set_layout_min_rows(262144, 1000000)


@pytest.mark.parametrize("s", [1, 4, 8, 11])
def test_idrs_merged_biorthogonalisation_equals_the_literal_loop(mf, s):
    """idrs! (04_IDRs.jl:62-73): the merged form of the bi-orthogonalisation (one multi-dot pass, alphas by forward substitution with M, one vector
    kernel; default) against the literal loop (mfem_debug_set_idrs(1)): the first cycles agree to round-off, the converged solutions to the solver
    tolerance; s = 11 needs two chunks of dot products."""
    import torch
    from metafem_jl_amd import _lib

    brick = mf.make_Brick((1.0, 2.0, 1.5), (9, 7, 11))
    A = brick.pattern(1)
    K = brick.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
    b = mf.FEM_rand(A.n, 11, 0) - 0.5
    out = {}
    try:
        for literal in (0, 1):
            _lib.lib.mfem_debug_set_idrs(literal)
            x1, _ = mf.iterative_Solve(A, K, b, 1e-300, Sv_func=mf.idrs_, s=s, maxiter=2 * (s + 1), max_pass=1, fixed_iterations=True)
            x2, st = mf.iterative_Solve(A, K, b, 1e-11, Sv_func=mf.idrs_, s=s, maxiter=3000, max_pass=4)
            assert st.converged == 1
            out[literal] = (x1.clone(), x2.clone(), st.iterations)
    finally:
        _lib.lib.mfem_debug_set_idrs(0)
    assert float((out[0][0] - out[1][0]).abs().max()) <= 1e-11 * float(out[1][0].abs().max())
    assert float((out[0][1] - out[1][1]).abs().max()) <= 1e-8 * float(out[1][1].abs().max())
    assert abs(out[0][2] - out[1][2]) <= max(4, out[1][2] // 10)


@pytest.mark.parametrize("s", [1, 4, 8, 11, 20])
def test_idrs_sign_shadow_vectors_and_fused_update(mf, s):
    """Round 6: idrs! (04_IDRs.jl:26-95) no longer streams its shadow vectors.  P (:35, an unseeded rand in the reference) is the +-1 vector family of the
    seed's sign words; P' g reads g only (kk_sign_dots), and the update of step k runs in one pass with the combination of step k + 1
    (ki_update_combine).  Checked: (a) the generated signs ARE the oracle's fem_sign vectors -- the same solve with those vectors handed in as explicit
    shadow vectors (the streamed multi-dot path) gives the same iterates; (b) the fused pass equals the two kernels (mfem_debug_set_idrs(4)) to
    round-off; (c) U(0,1) vectors (mfem_debug_set_idrs(2), the default until round 5) still converge to the same solution; s = 11 / 20: two / three chunks
    of dot products and the run-time form of the kernels; odd n."""
    import torch
    from metafem_jl_amd import _lib
    from oracle import solvers

    brick = mf.make_Brick((1.0, 2.0, 1.5), (9, 7, 11))  # 10 * 8 * 12 = 960 rows ... + an odd-length case below
    A = brick.pattern(1)
    K = brick.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
    b = mf.FEM_rand(A.n, 11, 0) - 0.5
    seed = 0xC0FFEE
    P = torch.tensor(np.concatenate([solvers.fem_sign(seed, k, A.n) for k in range(s)]), device="cuda")
    fixed = dict(Sv_func=mf.idrs_, s=s, maxiter=2 * (s + 1) + 3, max_pass=1, fixed_iterations=True, seed=seed)
    x_sign, _ = mf.iterative_Solve(A, K, b, 1e-300, **fixed)
    x_expl, _ = mf.iterative_Solve(A, K, b, 1e-300, shadow=P, **fixed)
    scale = float(x_expl.abs().max())
    assert float((x_sign - x_expl).abs().max()) <= 1e-11 * scale  # (a): other summation order of the dot products only
    conv = dict(Sv_func=mf.idrs_, s=s, maxiter=3000, max_pass=4, seed=seed)
    x_ref, st_ref = mf.iterative_Solve(A, K, b, 1e-11, **conv)
    assert st_ref.converged == 1
    try:
        _lib.lib.mfem_debug_set_idrs(4)
        x_unfused, _ = mf.iterative_Solve(A, K, b, 1e-300, **fixed)
        assert float((x_unfused - x_sign).abs().max()) <= 1e-12 * scale  # (b): the same products and sums (contracted differently by the compiler)
        _lib.lib.mfem_debug_set_idrs(2)
        x_uni, st_uni = mf.iterative_Solve(A, K, b, 1e-11, **conv)
        assert st_uni.converged == 1 and float((x_uni - x_ref).abs().max()) <= 1e-8 * float(x_ref.abs().max())  # (c)
    finally:
        _lib.lib.mfem_debug_set_idrs(0)
    # odd n (the last entry shares its 16 bytes with padding): 7 * 9 * 11 = 693 rows
    brick = mf.make_Brick((1.0, 1.0, 1.0), (6, 8, 10))
    A = brick.pattern(1)
    K = brick.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
    assert A.n % 2 == 1
    b = mf.FEM_rand(A.n, 12, 0) - 0.5
    P = torch.tensor(np.concatenate([solvers.fem_sign(seed, k, A.n) for k in range(s)]), device="cuda")
    x_sign, _ = mf.iterative_Solve(A, K, b, 1e-300, **fixed)
    x_expl, _ = mf.iterative_Solve(A, K, b, 1e-300, shadow=P, **fixed)
    assert float((x_sign - x_expl).abs().max()) <= 1e-11 * float(x_expl.abs().max())


@pytest.mark.parametrize("l", [1, 2])
def test_bicgstabl_fused_form_equals_the_literal_sequence(mf, l):
    """bicgstabl_GS! (03_BiCGstabl.jl:41-94): the fused form (dot products produced by the SpMVs, the minimal-residual part on the Gram matrix of
    R[0..l], the updates of a sweep in one kernel; default) against the literal operation sequence (mfem_debug_set_bicgstabl(1)): the first sweeps
    agree to round-off, the converged solutions to the solver tolerance.  (l > 2 keeps the literal sequence: the Gram-matrix form of the modified
    Gram-Schmidt loop squares the condition of R[1..l] -- measured 1e-8 instead of 1e-11 at l = 6.)"""
    from metafem_jl_amd import _lib

    brick = mf.make_Brick((1.0, 2.0, 1.5), (9, 7, 11))
    A = brick.pattern(1)
    K = brick.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
    b = mf.FEM_rand(A.n, 11, 0) - 0.5
    out = {}
    try:
        for literal in (0, 1):
            _lib.lib.mfem_debug_set_bicgstabl(literal)
            x1, _ = mf.iterative_Solve(A, K, b, 1e-300, Sv_func=mf.bicgstabl_GS_, s=l, maxiter=3 * l, max_pass=1, fixed_iterations=True)
            x2, st = mf.iterative_Solve(A, K, b, 1e-11, Sv_func=mf.bicgstabl_GS_, s=l, maxiter=3000, max_pass=4)
            assert st.converged == 1
            out[literal] = (x1.clone(), x2.clone(), st.iterations)
    finally:
        _lib.lib.mfem_debug_set_bicgstabl(0)
    assert float((out[0][0] - out[1][0]).abs().max()) <= 1e-11 * float(out[1][0].abs().max())
    assert float((out[0][1] - out[1][1]).abs().max()) <= 1e-8 * float(out[1][1].abs().max())
    assert abs(out[0][2] - out[1][2]) <= max(2 * l, out[1][2] // 10)


@pytest.mark.parametrize("elems", [(20, 12, 96), (20, 14, 63), (40, 30, 24), (40, 13, 40), (40, 17, 33)])
def test_patch_aligned_fill_is_bitwise_the_row_tile_copy(mf, elems):
    """The per-solve patch-major copy of the swept rows is made by k_symp_fill (a workgroup per patch step: aligned slot pieces, the edge block through LDS,
    short rows decoded from their columns); k_dia_vals' row tiles (bit 29 of mfem_debug_set_ell) and the two-pass bind (bit 27) must give the same copy:
    the sweep's y bit for bit, and the scaled CG (cg_variant 4: the S^-1 A S^-1 copy) the same iterates -- on lattices whose lines are odd, shorter than a
    patch, one point longer than whole patches, and whose last strip has 1, 2 or 3 lines (lattice = elements + 1 per direction)."""
    import ctypes as C

    import torch
    from metafem_jl_amd import _lib

    brick = mf.make_Brick((1.0, 0.7, 1.3), elems)
    A = brick.pattern(1)
    K = brick.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
    x = mf.FEM_rand(A.n, 5, 0) - 0.5
    b = torch.ones(A.n, dtype=torch.float64, device="cuda")
    _lib.lib.mfem_debug_set_layout_min_rows(0, 0)
    try:
        ent, sym = C.c_int64(), C.c_int32()
        _lib.check(_lib.lib.mfem_csr_solver_layout_entries(brick.ctx._h, A._h, C.byref(ent), C.byref(sym)))
        assert sym.value == 2  # (the patch sweep is what runs)
        ys, xs = [], []
        for knob in (0, 1 << 29, 3 << 28, 1 << 27):
            _lib.lib.mfem_debug_set_ell(1 | knob)
            y = torch.full((A.n,), 3.0, dtype=torch.float64, device="cuda")
            _lib.check(_lib.lib.mfem_spmv_solver_layout(brick.ctx._h, A._h, K.data_ptr(), x.data_ptr(), y.data_ptr(), 1.0, 0.0))
            ys.append(y)
            xx, st = mf.iterative_Solve(A, K, b, 1e-30, Sv_func=mf.cg_, maxiter=12, max_pass=1, fixed_iterations=True, cg_variant=4)
            xs.append(xx)
        assert all(torch.equal(ys[0], yy) for yy in ys[1:])
        assert all(torch.equal(xs[0], xx) for xx in xs[1:])
        yc = torch.zeros(A.n, dtype=torch.float64, device="cuda")
        _lib.lib.mfem_debug_set_ell(0)
        mf.mul_(yc, A, K, x)  # CSR kernel
        assert float((yc - ys[0]).abs().max()) <= 1e-13 * float(yc.abs().max())
    finally:
        _lib.lib.mfem_debug_set_ell(1)
        _lib.lib.mfem_debug_set_layout_min_rows(262144, 1000000)
