#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X MetaFEM backend (contract: see task prompt / DESIGN.md §Measurement).

Metric (BASELINE.json): DOF-updates/s = n_DOF x Krylov iterations / (t_assembly + t_solve) on 3-D thermal
conduction, linear hex-8.  One *step* = one pass of the hot path over one synthetic mesh:
    K_linear_func  (fused hex-8 assembly of K, CSR order)           mfem_brick_assemble_thermal
  + K_nonlinear_func (matrix-free residual R at x* = 0)              mfem_brick_residual_thermal
  + `--iters` Jacobi-PCG iterations on K delta = R (fixed count)     mfem_solve(fixed_iterations)
N = 1 workload: configs[1] of BASELINE.json, 256^3 elements (16 974 593 DOF, nnz 454 756 609).
N > 1: weak scaling -- every rank owns a 256-element-thick slab of a (256 N) x 256 x 256 mesh (slab
decomposition along i, one ghost node plane per neighbour, RCCL all-reduce of the CG scalars).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

K_COND, H, TENV, SRC = 0.6, 25.0, 293.15, 1600.0  # examples/thermal_conduction/3D_Script.jl:21-25,56
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def cpu_baseline(n_cpu: int, iters: int):
    """Reference algorithm restated in C/OpenMP (oracle/c), timed on this box's host cores."""
    # size and pin the OpenMP team before the runtime starts (must precede loading liboracle.so): the container may
    # see every host CPU but own only a cgroup quota of them -- oversubscribing the quota throttles all threads
    ncpu = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            ncpu = max(1, min(ncpu, int(int(quota) / int(period))))
    except Exception:
        pass
    try:
        ncpu = min(ncpu, int(os.environ.get("OMP_NUM_THREADS", ncpu)))
    except ValueError:
        pass
    os.environ["OMP_NUM_THREADS"] = str(ncpu)
    os.environ.setdefault("OMP_PROC_BIND", "spread")
    import numpy as np  # noqa: F401
    from oracle import cport

    cport.lib().orc_set_num_threads(ncpu)  # the OpenMP runtime may already be initialised (torch / numpy import it)
    prob = cport.CThermal((n_cpu, n_cpu, n_cpu), k=K_COND, h=H, Tenv=TENV, src=SRC).setup()
    prob.timed_step(2)  # warm caches / page in
    reps = 3
    ta = ts = 0.0
    for _ in range(reps):
        a, b = prob.timed_step(iters)
        ta, ts = ta + a / reps, ts + b / reps
    cores = cport.lib().orc_num_threads()
    return {
        "value": prob.mesh.ncp * iters / (ta + ts),
        "unit": "DOF-updates/s",
        "cores": cores,
        "kind": "port",
        "sample": f"hex-8 {n_cpu}^3 thermal ({prob.mesh.ncp} DOF), mean of {reps} steps: 1 step = term-by-term assembly "
                  f"({ta:.2f} s) + {iters} Jacobi-CG iterations ({ts:.2f} s); C/OpenMP restatement of the "
                  f"reference algorithm (oracle/c/oracle.c), {cores} threads (cgroup CPU quota of the box)",
    }


def self_launch(n_ranks: int) -> int:
    import socket
    import subprocess

    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    # stdout carries the JSON line only (RCCL / gloo banners that reached rank 0's stdout go to stderr)
    lines = out0.decode(errors="replace").splitlines()
    json_lines = [ln for ln in lines if ln.startswith("{") and ln.rstrip().endswith("}")]
    for ln in lines:
        if not json_lines or ln is not json_lines[-1]:
            print(ln, file=sys.stderr)
    if json_lines:
        print(json_lines[-1], flush=True)
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0] + ([] if json_lines else [(0, "no JSON line")])
    if bad:
        print(f"bench.py: ranks failed (rank, exit code): {bad}", file=sys.stderr)
        return 1
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=256, help="elements per side of the per-GPU mesh")
    ap.add_argument("--iters", type=int, default=200, help="CG iterations per step")
    ap.add_argument("--cpu-n", type=int, default=192, help="elements per side of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-iters", type=int, default=400)
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: start the N ranks ourselves, as child processes and BEFORE anything here touches
        # the GPU; rank 0's JSON line is the only thing that reaches stdout; a failing rank fails the run
        sys.exit(self_launch(args.gpus))

    import ctypes as C

    import torch

    import metafem_jl_amd as mf
    from metafem_jl_amd import _lib

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # functional check of the N > 1 path on a 1-GPU box: all ranks on cuda:0, host-callback communicator over gloo
    # (never a measurement: the ranks share the GPU)
    host_comm = os.environ.get("MFEM_BENCH_HOST_COMM") == "1"
    if host_comm:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist  # type: ignore

        if host_comm:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))

    ctx = mf.Context(local_rank)
    N = args.n
    nx_global = N * world
    brick = mf.Brick((float(world), 1.0, 1.0), (nx_global, N, N), 1, 3, ctx=ctx)
    use_comm = world > 1 or os.environ.get("MFEM_BENCH_FORCE_COMM") == "1"  # the env var exercises the RCCL path at N = 1
    if use_comm:
        from metafem_jl_amd import parallel

        plo, phi = parallel.slab_planes(nx_global + 1, world, rank)
        brick.set_slab(plo, phi)
        # RCCL prints a version banner / warnings through C stdio on stdout: send them to stderr so that stdout
        # carries only the JSON line
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            comm = (parallel.HostSlabComm if host_comm else parallel.SlabComm)(ctx, brick, rank, world, n_fields=1)  # noqa: F841 (kept alive for the run)
            C.CDLL(None).fflush(None)
        finally:
            os.dup2(saved, 1)
            os.close(saved)
    A = brick.pattern(1)
    n_local = A.n
    dev = f"cuda:{local_rank}"
    K = torch.empty(A.nnz, dtype=torch.float64, device=dev)
    xlen = n_local + (2 * brick.m[1] * brick.m[2] if use_comm else 0)
    x_star = torch.zeros(xlen, dtype=torch.float64, device=dev)
    s = torch.full((xlen,), SRC, dtype=torch.float64, device=dev)
    R = torch.empty(n_local, dtype=torch.float64, device=dev)

    n_global = (nx_global + 1) * (N + 1) * (N + 1)

    def step():
        brick.assemble_thermal(A, K_COND, H, TENV, mf.ALL_FACES, out=K)
        brick.residual_thermal(x_star, K_COND, H, TENV, mf.ALL_FACES, s=s, out=R)
        _, st = mf.iterative_Solve(A, K, R, 1e-300, Sv_func=mf.cg_, Pr_func=mf.Pr_Jacobi_, maxiter=args.iters,
                                   max_pass=1, fixed_iterations=True)
        return st

    for _ in range(args.warmup):
        step()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    _lib.check(_lib.lib.mfem_prof_spmv_enable(ctx._h, 1))
    tot, cnt = C.c_double(), C.c_int64()
    _lib.check(_lib.lib.mfem_prof_spmv_read(ctx._h, C.byref(tot), C.byref(cnt), 1))
    sym_count0 = int(_lib.lib.mfem_debug_sym_spmv_count())
    barrier()
    t0 = time.perf_counter()
    solve_ms = 0.0
    iters_done = 0
    for _ in range(args.steps):
        st = step()
        solve_ms += st.solve_ms
        iters_done += st.iterations
    barrier()
    elapsed = time.perf_counter() - t0
    _lib.check(_lib.lib.mfem_prof_spmv_read(ctx._h, C.byref(tot), C.byref(cnt), 1))
    _lib.check(_lib.lib.mfem_prof_spmv_enable(ctx._h, 0))
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if host_comm else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        assert iters_done == args.iters * args.steps, (iters_done, args.iters, args.steps)
        value = n_global * args.iters * args.steps / elapsed
        spmv_ms = tot.value / max(cnt.value, 1)
        csr_bytes = A.nnz * 12 + A.n * 16 + (A.n + 1) * 8  # val 8 + col 4 per nz; x, y 8 each per row; i64 rowptr
        mode, slots, npad, reg = C.c_int32(), C.c_int32(), C.c_int64(), C.c_int64()
        _lib.check(_lib.lib.mfem_csr_solver_layout(ctx._h, A._h, C.byref(mode), C.byref(slots), C.byref(npad), C.byref(reg)))
        sym_used = False
        if mode.value == 2:
            # diagonal-slotted blocks read no column stream: 8 B per nonzero + x, y; rows in generic blocks also read 4 B columns
            kernel = "k_spmv_dia<2,3> (SpMV on the slot-major copy of the CSR matrix made once per solve; diagonal-slotted blocks)"
            spmv_bytes = A.nnz * 8 + A.n * 16 + max(A.n - reg.value, 0) * slots.value * 4
            ent, symf = C.c_int64(), C.c_int32()
            _lib.check(_lib.lib.mfem_csr_solver_layout_entries(ctx._h, A._h, C.byref(ent), C.byref(symf)))
            sym_used = bool(symf.value) and int(_lib.lib.mfem_debug_sym_spmv_count()) > sym_count0
            if sym_used:
                # symmetric sweep: the lower-diagonal entries a workgroup still holds in LDS are not read again; `ent` = 8-byte
                # matrix values one SpMV reads from memory (27 slots per padded row minus the mirrored ones)
                kernel = ("k_spmv_sym27 (+ k_spmv_dia<2,3> on the two boundary planes): SpMV on the slot-major copy of the CSR matrix "
                          "made once per solve; the values passed the per-solve bitwise symmetry check, so lower-diagonal entries "
                          "are mirrored through LDS (bitwise the same y as the plain diagonal-slotted kernel)")
                plain_bytes = spmv_bytes
                spmv_bytes = ent.value * 8 + A.n * 16 + max(A.n - reg.value, 0) * slots.value * 4
        elif mode.value == 1:
            kernel = "k_spmv_ell<2,1> (SpMV on the slot-major copy of the CSR matrix made once per solve; f64 val / i32 col)"
            spmv_bytes = A.nnz * 12 + A.n * 16
        else:
            kernel = "k_spmv_lds (CSR SpMV, i64 rowptr / i32 col / f64 val)"
            spmv_bytes = csr_bytes
        achieved = spmv_bytes / (spmv_ms * 1e-3) / 1e9
        csr_equiv = csr_bytes / (spmv_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "spmv_traffic.json")
        if os.path.exists(tpath) and N == 256 and world == 1:  # PMC traffic was measured on this single-GPU workload
            try:
                tj = json.load(open(tpath))
                same = tj.get("solver_layout_mode") == mode.value and bool(tj.get("symmetric_sweep", False)) == sym_used
                traffic = tj.get("hbm_bytes_per_launch") if same else None
            except Exception:
                traffic = None
        out = {
            "metric": "DOF-updates/sec (assembly+CG iter) on 3D hex thermal conduction",
            "value": value,
            "unit": "DOF-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"3D thermal conduction, linear hex-8, {nx_global}x{N}x{N} structured mesh "
                            f"(make_Brick), Robin on 6 faces: fused assembly (K + R) + {args.iters} Jacobi-CG iterations per step",
                "n_dof": n_global, "nnz_per_gpu": A.nnz, "cg_iters_per_step": args.iters,
                "parallelism": "single GPU" if world == 1 else f"slab decomposition x{world} (RCCL halo + all-reduce)",
                "solve_ms_per_step": solve_ms / args.steps,
                # the two halves of the metric on their own (whole job, all ranks)
                "assembly_ms_per_step": elapsed / args.steps * 1e3 - solve_ms / args.steps,
                "assembly_dof_per_s": n_global / max(elapsed / args.steps - solve_ms / args.steps * 1e-3, 1e-12),
                "solve_dof_updates_per_s": n_global * args.iters / (solve_ms / args.steps * 1e-3),
            },
            "roofline": {
                "kernel": kernel,
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes_per_launch": spmv_bytes, "avg_launch_ms": spmv_ms, "launches": cnt.value,
                **({"plain_diagonal_kernel_bytes_per_launch": plain_bytes,
                    "note": "algorithmic bytes = what this kernel design reads: 8 B per matrix entry not mirrored from LDS "
                            "(about 18.6 of 27 per row) + x + y; the plain diagonal-slotted kernel reads "
                            "plain_diagonal_kernel_bytes_per_launch"} if sym_used else {}),
                "csr_equivalent": {"bytes_per_launch": csr_bytes, "achieved": csr_equiv, "frac": csr_equiv / HBM_PEAK_GBS,
                                   "note": "the same launch priced with SURVEY 8(d)'s CSR formula (12 B per nonzero): what a CSR "
                                           "kernel would have to sustain to match this time"},
            },
        }
        if world == 1 and args.cpu_n > 0:
            out["cpu_baseline"] = cpu_baseline(args.cpu_n, args.cpu_iters)
        else:
            out["cpu_baseline"] = None
        C.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
