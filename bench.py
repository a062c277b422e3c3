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


def cpu_baseline(n_cpu: int, iters: int, repeats: int = 3):
    """Reference algorithm restated in C/OpenMP (oracle/c), timed on this box's host cores: the SAME step as the GPU leg
    (term-by-term assembly + `iters` Jacobi-CG iterations), on a smaller mesh; median of `repeats` steps."""
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
    times = sorted((prob.timed_step(iters) for _ in range(max(repeats, 1))), key=lambda t: t[0] + t[1])
    ta, ts = times[len(times) // 2]
    cores = cport.lib().orc_num_threads()
    return {
        "value": prob.mesh.ncp * iters / (ta + ts),
        "unit": "DOF-updates/s",
        "cores": cores,
        "kind": "port",
        "sample": f"hex-8 {n_cpu}^3 thermal ({prob.mesh.ncp} DOF = {prob.mesh.ncp / 16974593:.3f} of the 256^3 workload), median of "
                  f"{len(times)} steps: 1 step = term-by-term assembly ({ta:.2f} s) + {iters} Jacobi-CG iterations ({ts:.2f} s) -- the "
                  f"GPU leg's iterations per assembly; C/OpenMP restatement of the reference algorithm (oracle/c/oracle.c), "
                  f"{cores} threads (cgroup CPU quota of the box)",
        "all_step_seconds": [round(a + b, 3) for a, b in times],
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
    ap.add_argument("--cpu-repeats", type=int, default=3, help="timed CPU steps (the median is reported)")
    ap.add_argument("--target-n", type=int, default=512, help="also run this mesh size for a few steps at N = 1 (0 = skip)")
    ap.add_argument("--target-steps", type=int, default=2)
    ap.add_argument("--hex27-n", type=int, default=128, help="CSR-kernel roofline on the hex-27 matrix of this size (0 = skip)")
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
    dev = f"cuda:{local_rank}"
    use_comm = world > 1 or os.environ.get("MFEM_BENCH_FORCE_COMM") == "1"  # the env var exercises the RCCL path at N = 1

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def csr_kernel_roofline(A, K, launches=20):
        """The CSR kernel behind mul! (mfem_spmv_csr: caller's CSR arrays, no copy) on this matrix: live hip-event timing of
        `launches` launches, priced with SURVEY 8(d)'s CSR bytes.  This is the north_star's 'CSR SpMV % of HBM roofline'."""
        x = mf.FEM_rand(A.ncols, 0x5EED, 0, ctx=ctx)
        y = torch.empty(A.n, dtype=torch.float64, device=dev)
        for _ in range(3):
            mf.mul_(y, A, K, x)
        tot, cnt = C.c_double(), C.c_int64()
        _lib.check(_lib.lib.mfem_prof_spmv_enable(ctx._h, 1))
        _lib.check(_lib.lib.mfem_prof_spmv_read(ctx._h, C.byref(tot), C.byref(cnt), 1))
        for _ in range(launches):
            mf.mul_(y, A, K, x)
        _lib.check(_lib.lib.mfem_prof_spmv_read(ctx._h, C.byref(tot), C.byref(cnt), 1))
        _lib.check(_lib.lib.mfem_prof_spmv_enable(ctx._h, 0))
        ms = tot.value / max(cnt.value, 1)
        nbytes = A.nnz * 12 + A.n * 16 + (A.n + 1) * 8  # val 8 + col 4 per nonzero; x, y 8 per row; i64 row pointers
        return {"kernel": "mfem_spmv_csr (mul!): k_spmv_csr_w (rows of up to 64 entries, uniform length) / k_spmv_csr_rb (wide or uneven rows) on the caller's CSR arrays; tiles whose rows repeat the column offsets of their first row(s) -- found by an inspection when the pattern is created -- do not re-read their columns, so the kernel moves fewer bytes than the CSR formula this object is priced with",
                "avg_launch_ms": ms, "launches": int(cnt.value), "algorithmic_bytes_per_launch": nbytes,
                "achieved": nbytes / (ms * 1e-3) / 1e9, "frac": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "unit": "GB/s",
                "n": A.n, "nnz": A.nnz}

    def run_workload(N, steps, warmup, want_csr):
        """One workload: the hex-8 thermal problem on an (N * world) x N x N mesh, `steps` timed steps."""
        nx_global = N * world
        brick = mf.Brick((float(world), 1.0, 1.0), (nx_global, N, N), 1, 3, ctx=ctx)
        comm = None
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
                comm = (parallel.HostSlabComm if host_comm else parallel.SlabComm)(ctx, brick, rank, world, n_fields=1)
                C.CDLL(None).fflush(None)
            finally:
                os.dup2(saved, 1)
                os.close(saved)
        A = brick.pattern(1)
        n_local = A.n
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

        for _ in range(warmup):
            step()
        _lib.check(_lib.lib.mfem_prof_spmv_enable(ctx._h, 1))
        tot, cnt = C.c_double(), C.c_int64()
        _lib.check(_lib.lib.mfem_prof_spmv_read(ctx._h, C.byref(tot), C.byref(cnt), 1))
        sym_count0 = int(_lib.lib.mfem_debug_sym_spmv_count())
        barrier()
        t0 = time.perf_counter()
        solve_ms = 0.0
        iters_done = 0
        for _ in range(steps):
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
        res = {"N": N, "nx_global": nx_global, "n_global": n_global, "nnz": A.nnz, "n_local": A.n, "elapsed": elapsed,
               "steps": steps, "solve_ms": solve_ms, "iters_done": iters_done, "spmv_ms": tot.value / max(cnt.value, 1),
               "spmv_launches": int(cnt.value)}
        if rank == 0:
            assert iters_done == args.iters * steps, (iters_done, args.iters, steps)
            csr_bytes = A.nnz * 12 + A.n * 16 + (A.n + 1) * 8
            mode, slots, npad, reg = C.c_int32(), C.c_int32(), C.c_int64(), C.c_int64()
            _lib.check(_lib.lib.mfem_csr_solver_layout(ctx._h, A._h, C.byref(mode), C.byref(slots), C.byref(npad), C.byref(reg)))
            sym_used = False
            plain_bytes = None
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
                    if symf.value == 2:
                        kernel = ("k_spmv_symp<0> (one launch: the sweep, then the two boundary planes row by row): symmetric sweep on wave-private "
                                  "(j, k) patches of a patch-major copy of the CSR matrix made once per solve; the values passed the "
                                  "per-solve bitwise symmetry check, so 10.5 of a row's 13 lower-diagonal entries are mirrored through "
                                  "LDS (bitwise the same y as the plain diagonal-slotted kernel); x staged per lattice plane in LDS")
                    else:
                        kernel = ("k_spmv_sym27 (+ k_spmv_dia<2,3> on the two boundary planes): SpMV on the slot-major copy of the CSR "
                                  "matrix made once per solve; the values passed the per-solve bitwise symmetry check, so lower-diagonal "
                                  "entries are mirrored through LDS (bitwise the same y as the plain diagonal-slotted kernel)")
                    plain_bytes = spmv_bytes
                    byts = C.c_int64()
                    _lib.check(_lib.lib.mfem_csr_solver_layout_bytes(ctx._h, A._h, C.byref(byts)))
                    spmv_bytes = byts.value  # matrix entries read + columns of generic blocks + x as staged / read by design + y
            elif mode.value == 1:
                kernel = "k_spmv_ell<2,1> (SpMV on the slot-major copy of the CSR matrix made once per solve; f64 val / i32 col)"
                spmv_bytes = A.nnz * 12 + A.n * 16
            else:
                kernel = "mfem_spmv_csr kernel (CSR SpMV, i64 rowptr / i32 col / f64 val)"
                spmv_bytes = csr_bytes
            res.update(kernel=kernel, spmv_bytes=spmv_bytes, csr_bytes=csr_bytes, mode=mode.value, sym_used=sym_used,
                       sym_kind=(int(symf.value) if sym_used else 0), plain_bytes=plain_bytes)
            if want_csr and world == 1:
                res["csr_kernel"] = csr_kernel_roofline(A, K)
        if comm is not None:
            comm.close()
        del brick, A, K, x_star, s, R
        torch.cuda.empty_cache()
        return res

    main_res = run_workload(args.n, args.steps, args.warmup, want_csr=True)

    if rank == 0:
        r = main_res
        value = r["n_global"] * args.iters * r["steps"] / r["elapsed"]
        achieved = r["spmv_bytes"] / (r["spmv_ms"] * 1e-3) / 1e9
        csr_equiv = r["csr_bytes"] / (r["spmv_ms"] * 1e-3) / 1e9
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "spmv_traffic.json")
        if os.path.exists(tpath) and args.n == 256 and world == 1:  # PMC traffic was measured on this single-GPU workload
            try:
                tj = json.load(open(tpath))
                same = tj.get("solver_layout_mode") == r["mode"] and int(tj.get("symmetric_sweep", 0)) == r["sym_kind"]
                traffic = tj.get("hbm_bytes_per_launch") if same else None
                if traffic is not None:
                    traffic_source = ("profiles/spmv_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel on "
                                      "this workload (gfx950 x2 FETCH_SIZE correction), not collected in this run")
            except Exception:
                traffic = None
        solve_ms_step = r["solve_ms"] / r["steps"]
        out = {
            "metric": "DOF-updates/sec (assembly+CG iter) on 3D hex thermal conduction",
            "value": value,
            "unit": "DOF-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": r["elapsed"] / r["steps"] * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"3D thermal conduction, linear hex-8, {r['nx_global']}x{args.n}x{args.n} structured mesh "
                            f"(make_Brick), Robin on 6 faces: fused assembly (K + R) + {args.iters} Jacobi-CG iterations per step",
                "n_dof": r["n_global"], "nnz_per_gpu": r["nnz"], "cg_iters_per_step": args.iters,
                "parallelism": "single GPU" if world == 1 else f"slab decomposition x{world} (RCCL halo overlapped with the interior "
                                                               f"rows + one all-reduce per CG iteration)",
                "solve_ms_per_step": solve_ms_step,
                # the two halves of the metric on their own (whole job, all ranks)
                "assembly_ms_per_step": r["elapsed"] / r["steps"] * 1e3 - solve_ms_step,
                "assembly_dof_per_s": r["n_global"] / max(r["elapsed"] / r["steps"] - solve_ms_step * 1e-3, 1e-12),
                "solve_dof_updates_per_s": r["n_global"] * args.iters / (solve_ms_step * 1e-3),
            },
            "roofline": {
                "kernel": r["kernel"],
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                "algorithmic_bytes_per_launch": r["spmv_bytes"], "avg_launch_ms": r["spmv_ms"], "launches": r["spmv_launches"],
                **({"plain_diagonal_kernel_bytes_per_launch": r["plain_bytes"],
                    "note": "algorithmic bytes = what this kernel design reads: 8 B per matrix entry not mirrored from LDS "
                            "(14 upper-diagonal entries per row + the patch-edge entries; workgroup-tile sweep: about 18.6 of 27) "
                            "+ x as the kernel stages it (patch sweep: overlapping patch neighbourhoods, 1.6 n entries) + y "
                            "(mfem_csr_solver_layout_bytes); the plain diagonal-slotted kernel reads "
                            "plain_diagonal_kernel_bytes_per_launch"}
                   if r["sym_used"] else {}),
                "csr_equivalent": {"bytes_per_launch": r["csr_bytes"], "achieved": csr_equiv, "frac": csr_equiv / HBM_PEAK_GBS,
                                   "note": "the same launch priced with SURVEY 8(d)'s CSR formula (12 B per nonzero): what a CSR "
                                           "kernel would have to sustain to match this time"},
            },
        }
        if "csr_kernel" in r:
            # the north_star's own number: the CSR kernel behind mul! on this matrix, SURVEY 8(d) bytes, measured in this run
            out["roofline"]["csr_kernel"] = r["csr_kernel"]
    if world == 1 and args.target_n > 0 and args.target_n != args.n:
        # the configuration the north_star's target is quoted on (512^3 hex-8, 135 M DOF, int64 row pointers): a short run
        t = run_workload(args.target_n, args.target_steps, 1, want_csr=True)
        tv = t["n_global"] * args.iters * t["steps"] / t["elapsed"]
        out["target_512" if args.target_n == 512 else f"target_{args.target_n}"] = {
            "workload": f"3D thermal conduction, linear hex-8, {args.target_n}^3, {t['steps']} timed steps after 1 warm-up, same step as above",
            "value": tv, "unit": "DOF-updates/s", "n_dof": t["n_global"], "nnz": t["nnz"], "ms_per_step": t["elapsed"] / t["steps"] * 1e3,
            "solver_spmv": {"kernel": t["kernel"], "avg_launch_ms": t["spmv_ms"], "algorithmic_bytes_per_launch": t["spmv_bytes"],
                            "frac": t["spmv_bytes"] / (t["spmv_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "csr_equivalent_frac": t["csr_bytes"] / (t["spmv_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "csr_kernel": t.get("csr_kernel"),
        }
    if rank == 0 and world == 1 and args.hex27_n > 0:
        # the CSR kernel on the other matrix shape of the configs: hex-27 (27..125 entries per row), configs[3]'s size
        try:
            b27 = mf.Brick((1.0, 1.0, 1.0), (args.hex27_n,) * 3, 2, 5, ctx=ctx)
            A27 = b27.pattern(1)
            K27 = b27.assemble_thermal(A27, K_COND, H, TENV, mf.ALL_FACES)
            out["roofline"]["csr_kernel_hex27"] = dict(csr_kernel_roofline(A27, K27), matrix=f"hex-27 thermal {args.hex27_n}^3")
            del b27, A27, K27
            torch.cuda.empty_cache()
        except Exception as e:  # never lose the line over a side measurement
            out["roofline"]["csr_kernel_hex27"] = {"error": repr(e)}
    if rank == 0:
        if world == 1 and args.cpu_n > 0:
            cb = cpu_baseline(args.cpu_n, args.iters, args.cpu_repeats)
            out["cpu_baseline"] = cb
            out["vs_cpu_baseline"] = {"main_workload": out["value"] / cb["value"],
                                      **({"target_512": out["target_512"]["value"] / cb["value"]} if "target_512" in out else {}),
                                      "note": "GPU whole-step throughput / CPU-port whole-step throughput (same step definition, same CG "
                                              "iterations per assembly; the CPU sample is a smaller mesh, see cpu_baseline.sample)"}
        else:
            out["cpu_baseline"] = None
        C.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
