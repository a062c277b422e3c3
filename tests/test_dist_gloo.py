"""N > 1 path on CPU: slab partition + halo protocol + all-reduce semantics with world_size 2/3 over gloo.

Every rank extracts its slab of the global oracle system with the HOST restatement of the device numbering
(metafem_jl_amd.parallel.slab_local_index), then runs the distributed Jacobi-PCG exactly as csrc/krylov.hip
sequences it (halo exchange of p before each SpMV, all-reduce of p.Ap and of (r.z, r.r)), and the gathered
result must equal the single-rank oracle solve.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _global_system(n, order=1):
    from oracle import fem, mesh as om, problems, reference_element as re_

    x = (2.0, 1.0, 1.0)
    disc = re_.initialize_classical_element(3, "CUBE", order, 1, 3 if order == 1 else 5)
    msh = om.lattice_mesh(x, n, disc)
    fac = om.boundary_facets_structured(x, n, 3)
    od = fem.FEMDomain(msh, disc, 1, problems.thermal_domain(3, 0.6), [(fac, problems.thermal_convection(25.0, 293.15))])
    od.controlpoints["s"] = np.full(msh.ncp, 1600.0)
    od.update_time(); od.K_linear_func(); od.update_x_star(); od.K_nonlinear_func()
    return od


def _worker(rank, world, port, n, out_q, order=1):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import metafem_jl_amd.parallel as par  # host logic only; no device call is made
    from oracle import solvers

    od = _global_system(n, order)
    m0, m1, m2 = order * n[0] + 1, order * n[1] + 1, order * n[2] + 1
    pl = m1 * m2
    hp = order * pl  # halo thickness: `order` planes per side
    lo, hi = par.slab_planes(m0, world, rank, order)
    n_owned = (hi - lo) * pl
    A = solvers.csr(od.pattern.rowptr, od.pattern.colidx, od.K_total, od.pattern.n)
    rows = np.arange(lo * pl, hi * pl)
    Aloc = A[rows].tocoo()
    gi, gj, gk = Aloc.col // pl, (Aloc.col % pl) // m2, Aloc.col % m2
    assert gi.min() >= lo - order and gi.max() <= hi + order - 1  # `order` ghost planes per side suffice
    lcol = par.slab_local_index(gi, gj, gk, 0, lo, hi, m1, m2, 1, order)
    nloc = par.local_vector_length(lo, hi, m1, m2, 1, order)
    import scipy.sparse as sp

    Al = sp.csr_matrix((Aloc.data, (Aloc.row, lcol)), shape=(n_owned, nloc))
    b = od.residue[rows]
    d = np.abs(A.diagonal()[rows])

    def allreduce(v):
        t = torch.tensor(np.atleast_1d(v), dtype=torch.float64)
        dist.all_reduce(t)
        return t.numpy()

    def halo(v):  # the ncclSend/ncclRecv group of csrc/comm.hip::mfem_comm_halo
        reqs = []
        lo_buf, hi_buf = torch.zeros(hp, dtype=torch.float64), torch.zeros(hp, dtype=torch.float64)
        if rank > 0:
            reqs.append(dist.isend(torch.tensor(v[:hp].copy()), rank - 1))
            reqs.append(dist.irecv(lo_buf, rank - 1))
        if rank < world - 1:
            reqs.append(dist.isend(torch.tensor(v[n_owned - hp:n_owned].copy()), rank + 1))
            reqs.append(dist.irecv(hi_buf, rank + 1))
        for r in reqs:
            r.wait()
        if rank > 0:
            v[n_owned:n_owned + hp] = lo_buf.numpy()
        if rank < world - 1:
            v[n_owned + hp:n_owned + 2 * hp] = hi_buf.numpy()

    n_glob = int(allreduce(float(n_owned))[0])
    x = np.zeros(nloc); r = np.zeros(nloc); p = np.zeros(nloc)
    r[:n_owned] = b
    p[:n_owned] = r[:n_owned] / d
    rz = allreduce(r[:n_owned] @ p[:n_owned])[0]
    its = 0
    for _ in range(500):
        halo(p)
        Ap = Al @ p
        alpha = rz / allreduce(p[:n_owned] @ Ap)[0]
        x[:n_owned] += alpha * p[:n_owned]
        r[:n_owned] -= alpha * Ap
        z = r[:n_owned] / d
        s = allreduce(np.array([r[:n_owned] @ z, r[:n_owned] @ r[:n_owned]]))
        its += 1
        if np.sqrt(s[1] / n_glob) <= 1e-10:
            break
        p[:n_owned] = z + (s[0] / rz) * p[:n_owned]
        rz = s[0]
    parts = [None] * world
    dist.all_gather_object(parts, (lo, hi, x[:n_owned].copy(), its))
    if rank == 0:
        full = np.concatenate([q[2] for q in sorted(parts, key=lambda t: t[0])])
        info = solvers.SolveInfo()
        ref = solvers.solve_cg_jacobi(od.pattern.rowptr, od.pattern.colidx, od.K_total, od.residue, 1e-10, 500, info=info)
        out_q.put((float(np.abs(full - ref).max() / np.abs(ref).max()), its, info.iters))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n,order", [(2, (6, 3, 4), 1), (3, (7, 2, 2), 1), (2, (3, 2, 2), 2), (3, (4, 1, 2), 2)])
def test_distributed_cg_matches_single_rank(world, n, order):
    """order 2 = hex-27: slabs on element boundaries, two ghost planes per side in the same send/recv group."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q, order)) for r in range(world)]
    for p in procs:
        p.start()
    err, its, its_ref = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert err < 1e-12 and its == its_ref


def test_slab_planes_cover_without_overlap():
    sys.path.insert(0, ROOT)
    import metafem_jl_amd.parallel as par

    for m0, world in [(257, 1), (513, 2), (1025, 4), (2049, 8), (10, 3), (8, 8)]:
        spans = [par.slab_planes(m0, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == m0
        assert all(spans[r][1] == spans[r + 1][0] for r in range(world - 1))
        assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
    with pytest.raises(ValueError):
        par.slab_planes(3, 4, 0)
    # order 2: slabs start and end on element boundaries (even planes), the last rank owns the closing plane
    for m0, world in [(257, 1), (257, 2), (257, 8), (9, 4), (11, 3)]:
        spans = [par.slab_planes(m0, world, r, 2) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == m0
        assert all(spans[r][1] == spans[r + 1][0] and spans[r][1] % 2 == 0 for r in range(world - 1))
        assert all(h - l >= 2 for l, h in spans)
    with pytest.raises(ValueError):
        par.slab_planes(5, 3, 0, 2)


def test_local_index_layout():
    sys.path.insert(0, ROOT)
    import metafem_jl_amd.parallel as par

    lo, hi, m1, m2, F = 4, 7, 3, 5, 2
    pl = m1 * m2
    seen = set()
    for f in range(F):
        for i in range(lo - 1, hi + 1):
            for j in range(m1):
                for k in range(m2):
                    seen.add(int(par.slab_local_index(i, j, k, f, lo, hi, m1, m2, F)))
    assert seen == set(range(par.local_vector_length(lo, hi, m1, m2, F)))
    assert int(par.slab_local_index(lo, 0, 0, 1, lo, hi, m1, m2, F)) == (hi - lo) * pl  # field-major owned block
    assert int(par.slab_local_index(lo - 1, 0, 0, 0, lo, hi, m1, m2, F)) == F * (hi - lo) * pl  # ghosts behind owned
    # two ghost planes per side (order-2 lattices): still a bijection, low block = planes lo-2, lo-1 in that order
    seen = set()
    for f in range(F):
        for i in range(lo - 2, hi + 2):
            for j in range(m1):
                for k in range(m2):
                    seen.add(int(par.slab_local_index(i, j, k, f, lo, hi, m1, m2, F, 2)))
    assert seen == set(range(par.local_vector_length(lo, hi, m1, m2, F, 2)))
    base = F * (hi - lo) * pl
    assert int(par.slab_local_index(lo - 2, 0, 0, 0, lo, hi, m1, m2, F, 2)) == base
    assert int(par.slab_local_index(lo - 1, 0, 0, 0, lo, hi, m1, m2, F, 2)) == base + pl
    assert int(par.slab_local_index(hi, 0, 0, 0, lo, hi, m1, m2, F, 2)) == base + 2 * pl
    assert int(par.slab_local_index(hi + 1, 0, 0, 1, lo, hi, m1, m2, F, 2)) == base + 7 * pl
