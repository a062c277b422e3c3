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


def test_rccl_world1_solve_equals_plain_solve(mf):
    import torch
    from metafem_jl_amd import parallel as par

    brick = mf.make_Brick((1.0, 1.0, 1.0), (10, 9, 8))
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
    finally:
        comm.close()
