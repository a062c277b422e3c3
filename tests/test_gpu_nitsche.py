"""GPU parity: the weakly imposed (Nitsche-type) Dirichlet faces of the thermal form,
    fix_boundary = h_penalty*Bilinear(T, Tw - T) + k*Bilinear(T, n{i}*T{;i})      examples/thermal_conduction/2D_Script.jl:58
fused into mfem_brick_assemble_thermal / mfem_brick_residual_thermal (csrc/assemble_nitsche.hip), hex-8 and hex-27, against the oracle's
term-by-term assembly of problems.thermal_fixed on stored facet tables -- and the reference's own solver configuration on the NONSYMMETRIC
K it produces (idrs!(s = 8) / bicgstabl_GS!(2) with Pr_Jacobi!, 02_Preconditioner.jl:32-76) against spsolve."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

K_COND, H, TENV, SRC = 0.6, 25.0, 293.15, 1600.0   # examples/thermal_conduction/3D_Script.jl:21-25
H_PEN, TW = 1000.0, 1173.15                          # examples/thermal_conduction/2D_Script.jl:46-47 (h_penalty, Tw)


def _distort(c):
    out = c.copy()
    out[:, 0] += 0.03 * np.sin(3 * c[:, 1]) * np.cos(2 * c[:, 2])
    out[:, 1] += 0.02 * np.sin(2 * c[:, 0] + c[:, 2])
    out[:, 2] += 0.025 * c[:, 0] * c[:, 1]
    return out


def _oracle(x, n, order, itg, distorted, robin, fixed):
    from oracle import fem, mesh as om, problems, reference_element as re_

    disc = re_.initialize_classical_element(3, "CUBE", order, 1, itg)
    msh = om.lattice_mesh(x, n, disc)
    if distorted:
        msh.coords[:] = _distort(msh.coords)
    fac = om.boundary_facets_structured(x, n, 3)
    bnd = []
    if robin:
        bnd.append((fac.select(np.isin(fac.element_eindex, robin)), problems.thermal_convection(H, TENV)))
    if fixed:
        bnd.append((fac.select(np.isin(fac.element_eindex, fixed)), problems.thermal_fixed(3, H_PEN, TW, K_COND)))
    od = fem.FEMDomain(msh, disc, 1, problems.thermal_domain(3, K_COND), bnd)
    od.controlpoints["s"] = np.full(msh.ncp, SRC)
    return od


def _bits(faces):
    return sum(1 << f for f in faces)


@pytest.mark.parametrize("order,n,itg,distorted,robin,fixed", [
    (1, (1, 1, 1), 3, False, [], [4]),
    (1, (4, 3, 5), 3, False, [0, 1, 2, 3, 5], [4]),          # the bench leg's shape: Dirichlet on x = 0, convection elsewhere
    (1, (5, 3, 6), 3, True, [1, 3], [4, 2, 0]),
    (1, (3, 4, 2), 5, True, [], [0, 1, 2, 3, 4, 5]),
    (1, (6, 5, 4), 1, True, [5], [3]),
    (1, (4, 4, 3), 7, True, [2], [4, 5]),
    (2, (1, 1, 1), 5, False, [], [4]),
    (2, (2, 2, 3), 5, False, [0, 1, 2, 3, 5], [4]),
    (2, (3, 2, 2), 5, True, [1], [4, 2, 0]),
    (2, (2, 3, 1), 4, True, [], [0, 1, 2, 3, 4, 5]),
    (2, (2, 2, 2), 7, True, [5], [3, 1]),
])
def test_nitsche_matrix_and_residual_match_oracle(mf, order, n, itg, distorted, robin, fixed):
    import torch

    x = (1.0, 1.5, 0.75)
    od = _oracle(x, n, order, itg, distorted, robin, fixed)
    od.update_time()
    od.K_linear_func()
    rng = np.random.default_rng(5)
    od.x_star[:] = 300.0 + 20.0 * rng.standard_normal(od.basicfield_size)
    od.K_nonlinear_func()

    brick = mf.make_Brick(x, n, order, itg)
    if distorted:
        for d in range(3):
            brick.coords_view(d).copy_(torch.tensor(od.mesh.coords[:, d], device="cuda"))
    A = brick.pattern(1)
    kw = dict(fixed_faces=_bits(fixed), h_penalty=H_PEN, Tw=TW)
    K = brick.assemble_thermal(A, K_COND, H, TENV, _bits(robin), **kw).cpu().numpy()
    assert np.max(np.abs(K - od.K_linear)) <= 1e-12 * np.max(np.abs(od.K_linear))
    s = torch.full((brick.ncp,), SRC, dtype=torch.float64, device="cuda")
    R = brick.residual_thermal(torch.tensor(od.x_star, device="cuda"), K_COND, H, TENV, _bits(robin), s=s, **kw).cpu().numpy()
    assert np.max(np.abs(R - od.residue)) <= 1e-11 * np.max(np.abs(od.residue))
    # the second term has no mirrored counterpart: K must NOT be symmetric (what sends cg! away and the solver layouts to their
    # nonsymmetric paths), and the asymmetry sits in the rows of the face nodes and of their element neighbours only
    import scipy.sparse as sp

    M = sp.csr_matrix((K, od.pattern.colidx, od.pattern.rowptr), shape=(od.pattern.n,) * 2)
    D = (M - M.T).tocsr()
    D.eliminate_zeros()
    assert abs(D).max() > 1e-6 * abs(M).max()


@pytest.mark.parametrize("order,n", [(1, (10, 8, 6)), (2, (4, 3, 3))])
@pytest.mark.parametrize("solver", ["idrs8", "bicgstabl2"])
def test_reference_solvers_on_the_nonsymmetric_matrix(mf, order, n, solver):
    """One Newton step of the (linear) problem with the reference's solver configuration: idrs!(s = 8) -- the default Sv_func!
    (src/MetaFEM.jl:36-37, 04_IDRs.jl:26-95) -- and bicgstabl_GS!(2), right Jacobi, on the nonsymmetric K; against spsolve to 1e-10."""
    from oracle import solvers

    x = (1.0, 1.0, 1.0)
    robin, fixed = [0, 1, 2, 3, 5], [4]
    od = _oracle(x, n, order, 3 if order == 1 else 5, False, robin, fixed)
    od.converge_tol = 1e-9
    od.linear_solver = lambda d: solvers.solver_lu_cpu(d.pattern.rowptr, d.pattern.colidx, d.K_total, d.residue)
    od.update_one_step()

    dom = mf.ThermalDomain(mf.make_Brick(x, n, order, 3 if order == 1 else 5), K_COND, H, TENV, _bits(robin),
                           fixed_faces=_bits(fixed), h_penalty=H_PEN, Tw=TW)
    dom.s.fill_(SRC)
    dom.converge_tol = 1e-9
    stats = []

    def linear_solver(gf):
        if solver == "idrs8":
            dx, st = mf.iterative_Solve(gf.A, gf.K_total, gf.residue, 1e-12, Sv_func=mf.idrs_, maxiter=2000, max_pass=10, s=8)
        else:
            dx, st = mf.iterative_Solve(gf.A, gf.K_total, gf.residue, 1e-12, Sv_func=mf.bicgstabl_GS_, maxiter=2000, max_pass=10, s=2)
        stats.append(st)
        return dx

    dom.linear_solver = linear_solver
    hist = dom.update_OneStep()
    assert len(hist) == 2 and hist[1] < 1e-9, hist
    assert stats[0].converged == 1
    got = dom.x.cpu().numpy()
    assert np.max(np.abs(got - od.x)) <= 1e-10 * np.max(np.abs(od.x))
    # the Dirichlet value is reproduced on the fixed face to the penalty's accuracy (2D_Ceramic_Strip.vtk:4132 shows the same for the 2-D script)
    m = dom.brick.m
    face = got.reshape(m)[0]
    assert np.max(np.abs(face - TW)) < 0.05 * TW
