"""GPU parity: device pattern build for unstructured connectivity (assemble_SparseID! replacement) and the fully
generic path built on it -- the reference's generated call sequence executed with mfem_op_* on device."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _quad8_problem(nx=10, ny=6):
    """examples/thermal_conduction/2D_Script.jl on a smaller strip (quad-8 serendipity, Nitsche + convection/radiation)."""
    from oracle import fem, mesh as om, problems, reference_element as re_

    L1, L2 = 0.02, 0.01
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
    return dom


@pytest.mark.parametrize("base", [0, 1])
@pytest.mark.parametrize("kind,F", [("quad8", 1), ("hex27", 1), ("hex8", 3)])
def test_pattern_build_matches_oracle(mf, kind, F, base):
    import torch
    from oracle import mesh as om, operators as oo, reference_element as re_

    if kind == "quad8":
        cp_ids, ncp = _quad8_problem().mesh.cp_ids, _quad8_problem().mesh.ncp
    elif kind == "hex27":
        disc = re_.initialize_classical_element(3, "CUBE", 2, 1, 5)
        vert, conn = om.make_brick((1.0, 1.0, 1.0), (3, 2, 2))
        m = om.mesh_classical(vert, conn, disc)  # unstructured-style numbering (vertices, edges, faces, centres)
        cp_ids, ncp = m.cp_ids, m.ncp
    else:
        disc = re_.initialize_classical_element(3, "CUBE", 1, 1, 3)
        m = om.lattice_mesh((1.0, 1.0, 1.0), (4, 3, 2), disc)
        cp_ids, ncp = m.cp_ids, m.ncp
    blocks = [(i, j) for i in range(F) for j in range(F)]
    pat = oo.assemble_sparse_id(cp_ids, ncp, blocks)
    conn_t = torch.tensor(np.ascontiguousarray(cp_ids.T) + base, dtype=torch.int32, device="cuda")  # (nel, itp) C == [itp, nel] F
    A, slots = mf.assemble_SparseID(conn_t, ncp, n_fields=F, index_base=base)
    assert A.n == pat.n and A.nnz == pat.nnz
    assert np.array_equal(A.rowptr.cpu().numpy(), pat.rowptr)
    assert np.array_equal(A.colidx.cpu().numpy(), pat.colidx)
    s = slots.cpu().numpy()  # (F*F, nel, itp_b, itp_a)
    for u, blk in enumerate(blocks):
        ref = pat.sparse_ids_by_el(blk)  # [a, b, e]
        assert np.array_equal(s[u].transpose(2, 1, 0), ref + base)


def test_generic_path_reproduces_the_oracle_newton_system(mf):
    """K_linear_func / K_nonlinear_func replayed term by term with the device operators on a device-built pattern
    (no fused fast path): quad-8, boundary launches with facet tables, normals, nonlinear (T^3) gradient."""
    import torch
    from oracle import solvers

    od = _quad8_problem()
    od.update_time()
    od.x_star[:] = 900.0 + 50.0 * np.random.default_rng(0).standard_normal(od.basicfield_size)
    od.K_linear_func()
    od.K_nonlinear_func()

    dev = "cuda"
    t64 = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
    i32 = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.int32, device=dev)
    conn_t = i32(od.mesh.cp_ids.T + 1)
    A, slots = mf.assemble_SparseID(conn_t, od.mesh.ncp, n_fields=1, index_base=1)
    sl = slots[0].reshape(-1)
    K_lin = torch.zeros(A.nnz, dtype=torch.float64, device=dev)
    x_star = t64(od.x_star)
    residue = torch.zeros(A.n, dtype=torch.float64, device=dev)
    parts = list(od._parts())
    K_tot = None
    for phase in ("linear", "nonlinear"):
        if phase == "nonlinear":
            K_tot = K_lin.clone()
        for wf, vals, w, host, el, fg in parts:
            itg, itp, nsd, nh = vals.shape
            N = t64(vals.ravel(order="F"))
            dims = (itg, itp, nsd, nh)
            hostd, eld = i32(host + 1), i32(el + 1)
            env = {}
            if phase == "nonlinear":
                for name, pos, s_, td in wf.inner_vars:
                    tgt = torch.zeros(itg * len(el), dtype=torch.float64, device=dev)
                    mf._Var_Basic(N, s_, 0, conn_t, x_star, tgt, hostd, eld, dims=dims, index_base=1)
                    env[name] = tgt.cpu().numpy().reshape(len(el), itg).T
            for name, sym, s_ in wf.cp_ext_vars:
                tgt = torch.zeros(itg * len(el), dtype=torch.float64, device=dev)
                mf._Var_Basic(N, s_, 0, conn_t, t64(od.controlpoints[sym]), tgt, hostd, eld, dims=dims, index_base=1)
                env[name] = tgt.cpu().numpy().reshape(len(el), itg).T
            for name, comp in wf.normals:
                env[name] = fg.normal_directions[:, comp, :][:, host]
            if phase == "linear":
                for g in wf.linear_gradients:
                    v = np.broadcast_to(g.fn(env) * w[:, host], w[:, host].shape)  # vals = @. coeff * K_params * w (host side)
                    mf._Kval_Basic(N, g.dual_s, g.base_s, t64(v.ravel(order="F")), sl, 0, K_lin, hostd, eld, dims=dims, index_base=1)
            else:
                for r in wf.residues:
                    v = np.broadcast_to(r.fn(env) * w[:, host], w[:, host].shape)
                    mf._Res_Basic(N, r.dual_s, t64(v.ravel(order="F")), 0, conn_t, residue, hostd, eld, dims=dims, index_base=1)
                for g in wf.nonlinear_gradients:
                    v = np.broadcast_to(g.fn(env) * w[:, host], w[:, host].shape)
                    mf._Kval_Basic(N, g.dual_s, g.base_s, t64(v.ravel(order="F")), sl, 0, K_tot, hostd, eld, dims=dims, index_base=1)
    assert np.max(np.abs(K_lin.cpu().numpy() - od.K_linear)) <= 1e-12 * np.abs(od.K_linear).max()
    assert np.max(np.abs(K_tot.cpu().numpy() - od.K_total)) <= 1e-12 * np.abs(od.K_total).max()
    assert np.max(np.abs(residue.cpu().numpy() - od.residue)) <= 1e-11 * np.abs(od.residue).max()
    # and the solve of that nonsymmetric system with the reference's default solver
    ref = solvers.solver_lu_cpu(od.pattern.rowptr, od.pattern.colidx, od.K_total, od.residue)
    dx, st = mf.iterative_Solve(A, K_tot, residue, 1e-10 * solvers.normalized_norm(od.residue), Sv_func=mf.idrs_, maxiter=2000,
                                max_pass=10, s=8)
    assert st.converged == 1 and np.abs(dx.cpu().numpy() - ref).max() <= 1e-8 * np.abs(ref).max()
