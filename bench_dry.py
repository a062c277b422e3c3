"""bench_dry.py -- `bench.py --gpus N --dry 1`: everything bench.py does for an N-rank run EXCEPT device work, so that the first real 8-GPU run cannot
fail on host logic.  Exercised: the rank spawn (bench.self_launch, before any GPU call), the rendezvous (gloo), every rank's slab
planes / owned rows / ghost planes / halo bytes per SpMV (metafem.jl_amd/parallel.py -- the same functions the timed path uses), nnz of the owned rows
(closed form of the lattice stencil = what mfem_brick_pattern builds), the device bytes a rank needs against 288 GB, the max-over-ranks reduction of the
timing, and the compact line.  tests/test_bench_report.py runs it at world 8 on the CPU."""
from __future__ import annotations

import os
import time

HBM_BYTES = 288e9


def _couplings(i, m, order):
    """Number of lattice points in direction d that the row of point i couples to (points sharing an element): order 1: 3 (2 on the boundary);
    order 2: element-corner points (even) 5 (3 on the boundary), mid points 3."""
    if order == 1:
        return 3 - (i == 0) - (i == m - 1)
    if i % 2:
        return 3
    return 5 - 2 * (i == 0) - 2 * (i == m - 1)


def slab_plan(cfg, N, world, rank, strong):
    from metafem_jl_amd import parallel  # (host logic only; the library is loaded, no device is touched)

    order, F = cfg["order"], cfg["fields"]
    nx = N if strong else N * world
    m0, m1, m2 = order * nx + 1, order * N + 1, order * N + 1
    lo, hi = parallel.slab_planes(m0, world, rank, order)
    n_owned = (hi - lo) * m1 * m2
    xlen = parallel.local_vector_length(lo, hi, m1, m2, F, order)
    ci = sum(_couplings(i, m0, order) for i in range(lo, hi))
    cj = sum(_couplings(j, m1, order) for j in range(m1))
    ck = sum(_couplings(k, m2, order) for k in range(m2))
    nnz = F * F * ci * cj * ck
    n_rows = F * n_owned
    halo_bytes = F * order * m1 * m2 * 8  # per neighbour and SpMV
    neighbours = (rank > 0) + (rank < world - 1)
    n_vec = {"cg": 5, "bicgstabl2": 2 * 3 + 4, "idrs8": 3 * 8 + 5}[cfg["solver"]]
    csr = nnz * 12 + (n_rows + 1) * 8
    layout = nnz * 8  # (upper bound: the symmetric layouts store 0.52-0.61 of it)
    vectors = (n_vec + 3) * xlen * 8
    dev_bytes = csr + layout + vectors + 3 * 8 * (hi - lo + 2 * order) * m1 * m2
    return {"rank": rank, "planes": [lo, hi], "n_rows": n_rows, "local_vector_length": xlen, "nnz": nnz, "halo_bytes_per_neighbour_per_spmv": halo_bytes,
            "neighbours": neighbours, "est_device_bytes": dev_bytes, "fits_288GB": dev_bytes < 0.9 * HBM_BYTES, "n_global": F * m0 * m1 * m2,
            "lattice": [m0, m1, m2]}


def main(args, emit) -> int:
    import bench_legs as LC  # (config tables; importing it touches no device)

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    cfg, base = LC.config_of(args.config)
    if args.n <= 0:
        args.n = cfg["n"]
    dist = None
    if world > 1:
        import torch.distributed as dist  # gloo: no device

        dist.init_process_group("gloo")
    strong = args.scaling == "strong" and world > 1
    t0 = time.perf_counter()
    plans = {"weak" if not strong else "strong": slab_plan(cfg, args.n, world, rank, strong)}
    if world > 1 and not strong and args.strong_leg:
        plans["strong"] = slab_plan(cfg, args.n, world, rank, True)
    elapsed = time.perf_counter() - t0
    gathered = [plans]
    if dist is not None:
        import torch

        gathered = [None] * world
        dist.all_gather_object(gathered, plans)
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.barrier()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)  # (the timed path's max-over-ranks)
        elapsed = float(t.item())
    rc = 0
    if rank == 0:
        key = "strong" if strong else "weak"
        per_rank = [g[key] for g in gathered]
        # the slabs must tile the lattice: contiguous, disjoint, complete
        assert per_rank[0]["planes"][0] == 0 and per_rank[-1]["planes"][1] == per_rank[0]["lattice"][0]
        assert all(a["planes"][1] == b["planes"][0] for a, b in zip(per_rank, per_rank[1:]))
        out = {"metric": cfg["metric"], "value": None, "unit": "DOF-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": None, "higher_is_better": True, "scaling": key, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "dry": True,
               "config": {"workload": f"DRY RUN (no device work): {cfg['title']}, {per_rank[0]['lattice']} lattice in {world} slabs along i",
                          "n_dof": per_rank[0]["n_global"], "nnz": sum(p["nnz"] for p in per_rank), "parallelism": f"slab decomposition x{world} (planned)"},
               "roofline": None, "cpu_baseline": None,
               "dry_plan": {k: [g[k] for g in gathered] for k in gathered[0]}}
        if not all(p["fits_288GB"] for plan in out["dry_plan"].values() for p in plan):
            out["errors"] = ["a rank's estimated device bytes exceed 0.9 x 288 GB"]
            rc = 1
        emit(out, args)
    if dist is not None:
        dist.destroy_process_group()
    return rc
