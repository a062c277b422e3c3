"""One rank of the multi-rank GPU tests (tests/test_gpu_multirank.py starts `world` of these on cuda:0).

Every rank builds its slab of a structured problem, attaches the host-callback communicator (gloo moves the data: several
ranks share ONE GPU here, which RCCL refuses) and runs the product's own multi-rank code paths -- mfem_halo_exchange,
mfem_halo_reduce, the Jacobi vectors, mfem_solve with CG (classic and single-reduction), BiCGStab(2), IDR(8) -- against the
single-rank solve of the global problem, which the rank computes itself on a second context.  Exit code 0 = all checks passed.

usage: multirank_worker.py <case> <world> <rank> <port>
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

K_COND, H, TENV, SRC = 0.6, 25.0, 293.15, 1600.0
LAM, MU, TAU = 0.5769, 0.3846, 1000.0

CASES = {
    # name: (elements, order, itg_order, fields)
    "thermal_hex8": ((30, 63, 63), 1, 3, 1),      # planes of 64 x 64 nodes: the symmetric sweep kernel is eligible on every slab
    "thermal_hex8_small": ((11, 5, 4), 1, 3, 1),  # CSR tile kernel (default thresholds), uneven slabs
    "elasticity_hex8": ((12, 6, 5), 1, 3, 3),
    "thermal_hex27": ((8, 4, 3), 2, 5, 1),
    # round 5: the reference's NONSYMMETRIC matrices on slabs -- Nitsche faces on x = 0 (asymmetric rows in rank 0's slab only: its tiles carry a remainder,
    # the other ranks' do not) ...
    "nitsche_hex8": ((40, 6, 5), 1, 3, 1),
    "nitsche_hex27": ((20, 4, 3), 2, 5, 1),
    # ... and an elasticity matrix whose rows in the first lattice planes are ALL perturbed: rank 0's tiles REFUSE the values (too many rows for a
    # remainder) and its solve starts over on another layout while the other ranks keep theirs -- the start-over must issue no collective
    "asym_one_slab_hex8": ((24, 6, 5), 1, 3, 3),
}
H_PEN, TW = 1000.0, 1173.15


def main():
    case, world, rank, port = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    import torch
    import torch.distributed as dist

    # MFEM_WORKER_TRANSPORT=rccl: one GPU per rank and the library's RCCL transport (mfem_comm_create) -- needs `world` GPUs; default: all
    # ranks on cuda:0 through the host-callback communicator
    rccl = os.environ.get("MFEM_WORKER_TRANSPORT") == "rccl"
    devid = rank if rccl else 0
    if rccl:
        torch.cuda.set_device(devid)
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, device_id=torch.device(f"cuda:{devid}"))
    else:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    import metafem_jl_amd as mf
    from metafem_jl_amd import _lib, parallel as par

    n, order, itg, F = CASES[case]
    x_len = (2.0, 1.0, 1.0)
    dev = f"cuda:{devid}"
    force_layouts = case != "thermal_hex8_small"
    if force_layouts:
        _lib.lib.mfem_debug_set_layout_min_rows(0, 0)  # solver layouts (diagonal slots / symmetric sweep / sliced) on these small systems
    report = {"case": case, "world": world, "rank": rank, "checks": {}}
    ok = True

    def check(name, cond, **info):
        nonlocal ok
        report["checks"][name] = {"ok": bool(cond), **{k: (float(v) if isinstance(v, (float, np.floating)) else v) for k, v in info.items()}}
        ok = ok and bool(cond)

    # ---- global problem on its own context (no communicator): the single-rank reference ------------------------------
    gctx = mf.Context(devid)
    gb = mf.Brick(x_len, n, order, itg, ctx=gctx)
    gA = gb.pattern(F)
    m0, m1, m2 = gb.m
    pl = m1 * m2
    ncp = m0 * pl
    nonsym = case.startswith("nitsche") or case.startswith("asym")
    fix = dict(fixed_faces=mf.FACE_BITS["x0"], h_penalty=H_PEN, Tw=TW) if case.startswith("nitsche") else {}
    robin = 0x3F & ~mf.FACE_BITS["x0"] if case.startswith("nitsche") else 0x3F

    def perturb(K_, rowptr_, first_global_plane, n_rows_per_field):
        """asym_one_slab: every entry of the rows in global lattice planes 0..3 scaled by 1 + 1e-3 u(global row, position in row) -- the same numbers
        on the global matrix and on the slab that owns those planes (rows of a slab keep the global order of their entries)."""
        if not case.startswith("asym"):
            return K_
        rp = rowptr_.to(torch.int64)
        rows = torch.repeat_interleave(torch.arange(rp.numel() - 1, device=dev), rp[1:] - rp[:-1])
        posn = torch.arange(K_.numel(), device=dev) - rp[rows]
        node = rows % n_rows_per_field
        fld = rows // n_rows_per_field
        gplane = node // pl + first_global_plane
        gnode = node + first_global_plane * pl
        u = (((fld * ncp + gnode) * 131 + posn * 7919) % 1000).to(torch.float64) / 1000.0
        return torch.where(gplane < 4, K_ * (1.0 + 1e-3 * u), K_)

    if F == 1:
        gK = gb.assemble_thermal(gA, K_COND, H, TENV, robin, **fix)
        gs = torch.full((gA.n,), SRC, dtype=torch.float64, device=dev)
        gR = gb.residual_thermal(torch.zeros(gA.n, dtype=torch.float64, device=dev), K_COND, H, TENV, robin, s=gs, **fix)
    else:
        gK = gb.assemble_elasticity(gA, LAM, MU, TAU, mf.FACE_BITS["x0"])
        gR = gb.residual_elasticity(torch.zeros(gA.n, dtype=torch.float64, device=dev), LAM, MU, TAU, mf.FACE_BITS["x0"],
                                    mf.FACE_BITS["y1"], (0.0, 1.0, 0.0, 0.0, 0.0, 0.3))
    gK = perturb(gK, gA.rowptr, 0, ncp)
    g_d = mf.jacobi_by_diagonal(gA, gK).cpu().numpy()
    g_dc = mf.jacobi2_by_column(gA, gK).cpu().numpy()
    g_shadow = mf.FEM_rand(8 * gA.n, 0x5EED, 7, ctx=gctx)  # shadow vectors of BiCGStab / IDR(8): the SAME global vectors on every layout

    def gsolve(sv, **kw):
        sh = g_shadow[:gA.n] if sv == mf.bicgstabl_GS_ else g_shadow if sv == mf.idrs_ else None
        return mf.iterative_Solve(gA, gK, gR, 1e-11, Sv_func=sv, maxiter=4000, max_pass=3, shadow=sh, **kw)

    # ---- this rank's slab -----------------------------------------------------------------------------------------------
    ctx = mf.Context(devid)
    sb = mf.Brick(x_len, n, order, itg, ctx=ctx)
    lo, hi = par.slab_planes(m0, world, rank, order)
    sb.set_slab(lo, hi)
    A = sb.pattern(F)
    n_owned = (hi - lo) * pl
    nloc = par.local_vector_length(lo, hi, m1, m2, F, order)
    assert A.n == F * n_owned and A.ncols == (nloc if world > 1 and (lo > 0 or hi < m0) else A.n), (A.n, A.ncols, nloc)
    comm = par.SlabComm(ctx, sb, rank, world, n_fields=F) if rccl else par.HostSlabComm(ctx, sb, rank, world, n_fields=F, poison=True)
    if F == 1:
        K = sb.assemble_thermal(A, K_COND, H, TENV, robin, **fix)
        s_loc = torch.full((nloc,), SRC, dtype=torch.float64, device=dev)
        R = sb.residual_thermal(torch.zeros(nloc, dtype=torch.float64, device=dev), K_COND, H, TENV, robin, s=s_loc, **fix)
    else:
        K = sb.assemble_elasticity(A, LAM, MU, TAU, mf.FACE_BITS["x0"])
        R = sb.residual_elasticity(torch.zeros(nloc, dtype=torch.float64, device=dev), LAM, MU, TAU, mf.FACE_BITS["x0"],
                                   mf.FACE_BITS["y1"], (0.0, 1.0, 0.0, 0.0, 0.0, 0.3))

    K = perturb(K, A.rowptr, lo, n_owned)

    def owned(gvec):  # the entries of a global field-major vector this rank owns, in local order
        gvec = np.asarray(gvec)
        return np.concatenate([gvec[f * ncp + lo * pl:f * ncp + hi * pl] for f in range(F)])

    def expected_local(gvec):  # [owned | ghosts] with the ghost blocks this rank's neighbours fill
        gvec = np.asarray(gvec)
        out = np.full(nloc, np.nan)
        out[:F * n_owned] = owned(gvec)
        jj, kk = np.meshgrid(np.arange(m1), np.arange(m2), indexing="ij")
        for f in range(F):
            for i in list(range(max(lo - order, 0), lo)) + list(range(hi, min(hi + order, m0))):
                li = par.slab_local_index(np.full(jj.size, i), jj.ravel(), kk.ravel(), f, lo, hi, m1, m2, F, order)
                out[li] = gvec[f * ncp + i * pl + jj.ravel() * m2 + kk.ravel()]
        return out

    assert np.array_equal(R.cpu().numpy(), owned(gR.cpu().numpy())), "slab residual differs from the global rows"

    # ---- halo exchange: ghost planes bitwise equal to the neighbours' planes ------------------------------------------
    gv = np.random.default_rng(5).standard_normal(F * ncp)
    exp = expected_local(gv)
    xl = torch.full((nloc,), -7.0, dtype=torch.float64, device=dev)
    xl[:F * n_owned] = torch.tensor(owned(gv), device=dev)
    comm.halo_(xl)
    got = xl.cpu().numpy()
    filled = ~np.isnan(exp)
    check("halo_ghost_planes_bitwise", np.array_equal(got[filled], exp[filled]), unreferenced=int((~filled).sum()))

    # ---- Jacobi vectors: |diag| with halo, column norms with reverse halo -------------------------------------------------
    d = mf.jacobi_by_diagonal(A, K)
    comm.halo_(d)
    dl = d.cpu().numpy()
    e = expected_local(g_d)
    check("jacobi_diag_with_halo_bitwise", np.array_equal(dl[filled], e[filled]))
    dc = mf.jacobi2_by_column(A, K).cpu().numpy()
    e = expected_local(g_dc)
    err = np.abs(dc[filled] - e[filled]).max() / np.abs(e[filled]).max()
    check("jacobi_colnorm_is_global", err < 1e-14, rel_err=err)

    # ---- all-reduce of device scalars -----------------------------------------------------------------------------------
    t = torch.tensor([1.0 + rank, 0.5], dtype=torch.float64, device=dev)
    comm.allreduce_(t)
    check("allreduce", t.cpu().tolist() == [world * (world + 1) / 2.0, 0.5 * world])

    # ---- the solvers -------------------------------------------------------------------------------------------------------
    shadow_loc = {mf.bicgstabl_GS_: torch.tensor(owned(g_shadow[:gA.n].cpu().numpy()), device=dev),
                  mf.idrs_: torch.cat([torch.tensor(owned(g_shadow[k * gA.n:(k + 1) * gA.n].cpu().numpy()), device=dev) for k in range(8)])}

    def lsolve(sv, **kw):
        return mf.iterative_Solve(A, K, R, 1e-11, Sv_func=sv, maxiter=4000, max_pass=3, shadow=shadow_loc.get(sv), **kw)

    def relerr(xl_, xg_):
        xg_o = owned(xg_.cpu().numpy())
        scale = np.abs(xg_.cpu().numpy()).max()
        return float(np.abs(xl_.cpu().numpy() - xg_o).max() / scale)

    if nonsym:
        # ---- the reference's solvers on a nonsymmetric matrix (cg! does not apply) -----------------------------------------
        lat8_0 = int(_lib.lib.mfem_debug_lat8_spmv_count())
        lat27_0 = int(_lib.lib.mfem_debug_lat27_spmv_count())
        rem_0 = int(_lib.lib.mfem_debug_rem_spmv_count())
        for overlap in (1, 0):
            _lib.lib.mfem_debug_set_halo_overlap(overlap)
            tag = "overlap" if overlap else "blocking"
            for sv, name, s_par in ((mf.bicgstabl_GS_, "bicgstabl2", 2), (mf.idrs_, "idrs8", 8)):
                xg, sg = gsolve(sv, s=s_par, Pr_func=mf.Pr_Jacobi_)
                x1, s1 = lsolve(sv, s=s_par, Pr_func=mf.Pr_Jacobi_)
                # (idrs! on a nonsymmetric matrix is sensitive to the summation order of its dot products: the iteration counts of `world` ranks and of one
                # rank differ by up to ~20 % here while the solutions agree to 1e-9)
                check(f"{name}_diag_{tag}", s1.converged == 1 and sg.converged == 1 and relerr(x1, xg) <= 1e-8 and
                      abs(s1.iterations - sg.iterations) <= max(6, sg.iterations // 3), iters=(s1.iterations, sg.iterations), rel_err=relerr(x1, xg),
                      converged=(int(s1.converged), int(sg.converged)), passes=(int(s1.passes), int(sg.passes)), final_res=(float(s1.final_res), float(sg.final_res)))
        _lib.lib.mfem_debug_set_halo_overlap(1)
        tiles = (int(_lib.lib.mfem_debug_lat8_spmv_count()) - lat8_0) + (int(_lib.lib.mfem_debug_lat27_spmv_count()) - lat27_0)
        rem = int(_lib.lib.mfem_debug_rem_spmv_count()) - rem_0
        if case.startswith("nitsche"):
            # every rank ran its slab on the tiles (the global single-rank solves count too); only processes that hold the x = 0 planes applied a remainder
            check("tiles_served_the_nonsymmetric_solves", tiles > 0, tiles=tiles)
            check("remainder_applied", rem > 0, remainder_products=rem)
        else:
            # rank 0's slab refused (its first planes are perturbed in every row); the start-over ran without a hang -- that this line is reached is the test
            check("solves_completed_with_a_rank_local_refusal", True, tiles=tiles, remainder_products=rem)
        check("callbacks_ran", rccl or (comm.calls["exchange"] > 10 and comm.calls["allreduce"] > 10))
        comm.close()
        dist.barrier()
        dist.destroy_process_group()
        print("MULTIRANK_REPORT " + json.dumps(report), flush=True)
        sys.exit(0 if ok else 1)

    # classic CG runs the same recurrence on every layout; what differs between 1 and `world` ranks is the order in which the
    # partial sums of the dot products are added.  1e-12 of max|x| for the thermal operators; the penalty-constrained
    # elasticity operator (tau = 1000) is worse conditioned and amplifies that round-off to a few 1e-12.
    tol_classic = 1e-11 if F == 3 else 1e-12
    sym0 = int(_lib.lib.mfem_debug_sym_spmv_count())
    lat8_0 = int(_lib.lib.mfem_debug_lat8_spmv_count())
    lat27_0 = int(_lib.lib.mfem_debug_lat27_spmv_count())
    for overlap in (1, 0):
        _lib.lib.mfem_debug_set_halo_overlap(overlap)
        tag = "overlap" if overlap else "blocking"
        xg, sg = gsolve(mf.cg_, cg_variant=1)
        x1, s1 = lsolve(mf.cg_, cg_variant=1)
        # (the lattice-tile layouts of the 3-field and hex-27 cases add in an order that is not fixed from run to run: their iteration counts may
        # differ by one when the stop test sits on the edge; the other cases run deterministic layouts: equal counts)
        it_ok = s1.iterations == sg.iterations if case.startswith("thermal_hex8") else abs(s1.iterations - sg.iterations) <= 1
        check(f"cg_classic_{tag}", s1.converged == 1 and it_ok and relerr(x1, xg) <= tol_classic,
              iters=(s1.iterations, sg.iterations), rel_err=relerr(x1, xg))
        x3, s3 = lsolve(mf.cg_, cg_variant=3)  # classic recurrence carrying z = M^-1 r (ghost entries masked)
        check(f"cg_classic_z_carried_{tag}", s3.converged == 1 and abs(s3.iterations - sg.iterations) <= 1 and relerr(x3, xg) <= 1e-10,
              iters=(s3.iterations, sg.iterations), rel_err=relerr(x3, xg))
        x2, s2 = lsolve(mf.cg_)  # auto: single-reduction form with world > 1
        check(f"cg_single_reduction_{tag}", s2.converged == 1 and abs(s2.iterations - sg.iterations) <= 2 and relerr(x2, xg) <= 1e-10,
              iters=(s2.iterations, sg.iterations), rel_err=relerr(x2, xg))
        x4, s4 = lsolve(mf.cg_, cg_variant=4)  # plain CG on S^-1 A S^-1 in the classic two-reduction form (on the mode-2 layout: the scaling of the
        check(f"cg_scaled_classic_form_{tag}",  # ghost columns by their owners' S; elsewhere variant 3 runs), world ranks
              s4.converged == 1 and abs(s4.iterations - sg.iterations) <= 2 and relerr(x4, xg) <= 1e-10, iters=(s4.iterations, sg.iterations), rel_err=relerr(x4, xg))
        xg2, sg2 = gsolve(mf.cg_, cg_variant=2)  # the single-reduction form on one rank
        check(f"cg_single_reduction_one_rank_{tag}", sg2.converged == 1 and relerr(owned_t(xg2, owned, dev), xg) <= 1e-10,
              iters=(sg2.iterations, sg.iterations))
    _lib.lib.mfem_debug_set_halo_overlap(1)
    # the auto choice of the CG form is size-dependent with a communicator (single reduction below 2e7 rows per rank, the classic recurrence -- one
    # vector stream less, one all-reduce more -- above): force the large-system branch on this small problem
    _lib.lib.mfem_debug_set_cg_single_max_rows(0)
    xg, sg = gsolve(mf.cg_, cg_variant=1)
    xa, sa = lsolve(mf.cg_)
    _lib.lib.mfem_debug_set_cg_single_max_rows(20000000)
    check("cg_auto_takes_the_classic_recurrence_on_large_systems", sa.converged == 1 and abs(sa.iterations - sg.iterations) <= 2 and relerr(xa, xg) <= 1e-10,
          iters=(sa.iterations, sg.iterations), rel_err=relerr(xa, xg))
    if not rccl:
        # ---- the communication SCHEDULE of one iteration (design budget), counted on the host transport's callbacks: fixed-count
        #      solves of 10 and 60 iterations differ by 50 iterations' worth of calls -- one halo exchange per SpMV and ONE all-reduce per CG iteration in
        #      the single-reduction form (two in the classic recurrence); bicgstabl_GS!(2): 4 SpMVs and at most 10 reduction groups per sweep (= 2 iterations)
        def calls_per_iteration(sv, iters_unit=1, **kw):
            c = {}
            for nit in (10, 60):
                e0, a0 = comm.calls["exchange"], comm.calls["allreduce"]
                mf.iterative_Solve(A, K, R, 1e-300, Sv_func=sv, maxiter=nit, max_pass=1, fixed_iterations=True, shadow=shadow_loc.get(sv), **kw)
                c[nit] = (comm.calls["exchange"] - e0, comm.calls["allreduce"] - a0)
            return (c[60][0] - c[10][0]) / (50.0 / iters_unit), (c[60][1] - c[10][1]) / (50.0 / iters_unit)
        ex, ar = calls_per_iteration(mf.cg_)
        check("schedule_cg_single_reduction_1_halo_1_allreduce_per_iteration", ex == 1.0 and ar == 1.0, exchanges=ex, allreduces=ar)
        ex, ar = calls_per_iteration(mf.cg_, cg_variant=1)
        check("schedule_cg_classic_1_halo_2_allreduces_per_iteration", ex == 1.0 and ar == 2.0, exchanges=ex, allreduces=ar)
        ex, ar = calls_per_iteration(mf.bicgstabl_GS_, iters_unit=2, s=2)
        check("schedule_bicgstabl2_4_halos_at_most_10_allreduces_per_sweep", ex == 4.0 and 1.0 <= ar <= 10.0, exchanges=ex, allreduces=ar)
    if case == "thermal_hex8":
        check("symmetric_sweep_kernel_ran", int(_lib.lib.mfem_debug_sym_spmv_count()) > sym0)
    if case == "elasticity_hex8":  # the slab solves ran on the symmetric lattice tiles (mode 5: ghost planes staged, lower-ghost terms from the CSR values)
        check("lattice_tiles_ran_on_the_slab", int(_lib.lib.mfem_debug_lat8_spmv_count()) > lat8_0)
    if case == "thermal_hex27":  # (mode 4: two ghost planes per side)
        check("lattice_tiles_ran_on_the_slab", int(_lib.lib.mfem_debug_lat27_spmv_count()) > lat27_0)
    for sv, name, s_par in ((mf.bicgstabl_GS_, "bicgstabl2", 2), (mf.idrs_, "idrs8", 8)):
        for pr, pname in ((mf.Pr_Jacobi_, "diag"), (mf.Pr_Jacobi_colnorm_, "colnorm")):
            xg, sg = gsolve(sv, s=s_par, Pr_func=pr)
            x1, s1 = lsolve(sv, s=s_par, Pr_func=pr)
            # same recurrences on the same shadow vectors: the iterates agree to round-off, so do the iteration counts (+- a few
            # when the stop test sits on the edge)
            check(f"{name}_{pname}", s1.converged == 1 and sg.converged == 1 and relerr(x1, xg) <= 1e-8 and
                  abs(s1.iterations - sg.iterations) <= max(4, sg.iterations // 10),
                  iters=(s1.iterations, sg.iterations), rel_err=relerr(x1, xg))
    # round 6: idrs! with its DEFAULT shadow vectors -- the +-1 family of the seed's sign words, generated per rank from the local row index, never stored;
    # P' g is a rank-local sign-dot + the all-reduce -- converges on the slabs to the single-rank solution (other shadow vectors than the global solve's:
    # the converged solutions agree, not the iterates)
    xg_, sg_ = gsolve(mf.idrs_, s=8, Pr_func=mf.Pr_Jacobi_)
    xd_, sd_ = mf.iterative_Solve(A, K, R, 1e-11, Sv_func=mf.idrs_, maxiter=4000, max_pass=3, s=8, Pr_func=mf.Pr_Jacobi_)
    check("idrs8_generated_sign_shadows_on_slabs", sd_.converged == 1 and relerr(xd_, xg_) <= 1e-7 and sd_.iterations <= 2 * sg_.iterations + 20,
          iters=(sd_.iterations, sg_.iterations), rel_err=relerr(xd_, xg_))
    if rccl:
        # hipGraph replay of the Krylov cycles WITH the communicator's calls recorded (round 6: mfem_debug_set_graphs bit 1 / MFEM_GRAPH_COMM=1): the un-captured
        # sequence above is the reference -- same iterates, same iteration counts
        _lib.lib.mfem_debug_set_graphs(3, 0)
        g0 = int(_lib.lib.mfem_debug_graph_comm_count())
        for sv, name, kw in ((mf.cg_, "cg", dict(cg_variant=1)), (mf.cg_, "cg_single", dict()), (mf.bicgstabl_GS_, "bicgstabl2", dict(s=2)), (mf.idrs_, "idrs8", dict(s=8))):
            _lib.lib.mfem_debug_set_graphs(1, 0)
            x0_, s0_ = lsolve(sv, **kw)
            _lib.lib.mfem_debug_set_graphs(3, 0)
            x1_, s1_ = lsolve(sv, **kw)
            check(f"graph_with_rccl_{name}", s1_.converged == 1 and s1_.iterations == s0_.iterations and relerr(x1_, x0_) <= 1e-12,
                  iters=(s1_.iterations, s0_.iterations), rel_err=relerr(x1_, x0_))
        check("graph_with_rccl_cycles_were_captured", int(_lib.lib.mfem_debug_graph_comm_count()) > g0, captured=int(_lib.lib.mfem_debug_graph_comm_count()) - g0)
        _lib.lib.mfem_debug_set_graphs(1, 0)
        hw, hn, aw, an = C.c_double(), C.c_int64(), C.c_double(), C.c_int64()
        _lib.check(_lib.lib.mfem_prof_comm_enable(ctx._h, 1))
        lsolve(mf.cg_)
        _lib.check(_lib.lib.mfem_prof_comm_read(ctx._h, C.byref(hw), C.byref(hn), C.byref(aw), C.byref(an), 1))
        _lib.check(_lib.lib.mfem_prof_comm_enable(ctx._h, 0))
        check("rccl_exposed_communication_is_timed", hn.value > 10 and an.value > 10 and hw.value >= 0.0 and aw.value > 0.0,
              halo_waits=int(hn.value), halo_wait_ms=hw.value, allreduces=int(an.value), allreduce_ms=aw.value)
    else:
        check("callbacks_ran", comm.calls["exchange"] > 10 and comm.calls["allreduce"] > 10, **comm.calls)
    comm.close()
    dist.barrier()
    dist.destroy_process_group()
    print("MULTIRANK_REPORT " + json.dumps(report), flush=True)
    sys.exit(0 if ok else 1)


def owned_t(xg, owned, dev):
    import torch
    return torch.tensor(owned(xg.cpu().numpy()), device=dev)


if __name__ == "__main__":
    main()
