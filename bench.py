#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X MetaFEM backend (contract: see task prompt / DESIGN.md section 5).

Metric (BASELINE.json): DOF-updates/s = n_DOF x Krylov iterations / (t_assembly + t_solve).  One *step* = one pass of the hot
path over one synthetic mesh, i.e. what update_OneStep! (solver/04_Time_Domain.jl:59-80) runs for one Newton step with a capped
solver:
    K_linear_func    (fused assembly of K, CSR order)              mfem_brick_assemble_*
  + K_nonlinear_func (matrix-free residual R at x* = 0)            mfem_brick_residual_*
  + `--iters` Krylov iterations on K delta = R (fixed count)       mfem_solve

--config c2 (default): 3-D thermal conduction, linear hex-8, Jacobi-PCG.  N = 1 workload: the configuration the north_star's
    target is quoted on, 512^3 elements (135 005 697 DOF, nnz 3 630 961 153, int64 row pointers, 45 GB); configs[1] of
    BASELINE.json (256^3) runs in the same invocation as the `secondary_256` object.
--config c3: configs[2], linear elasticity hex-8 (3 DOF per node), 128^3 per GPU, BiCGStab(2) (bicgstabl_GS!).
--config c4: configs[3], thermal conduction on quadratic hex-27 (FP64-MFMA Ke), 128^3 per GPU, Jacobi-PCG.
N > 1 (every config), `--scaling weak` (default): every rank owns an `--n`-element-thick slab of an (n N) x n x n mesh (slab decomposition
along i, `order` ghost node planes per neighbour exchanged over RCCL beside the interior rows, one all-reduce per reduction group).
`--scaling strong`: THE n^3 mesh of the config (512^3 / 128^3 / 128^3) is cut into N slabs along i -- the north_star's "the mesh is
domain-decomposed across the 8 GPUs"; the line then says "scaling": "strong".
At N = 1 the default invocation also runs configs[1] (256^3: `secondary_256`), configs[2] (`secondary_c3`) and configs[3] (`secondary_c4`, with the
FP64-MFMA roofline of the hex-27 Ke kernels as `roofline_hex27_ke`) for a few timed steps each, so one driver line carries all four configs.
Every timed solve's residual is checked: the run FAILS if the last timed solve did not reduce ||r|| (`initial_res` / `final_res` are printed).

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
E_MOD, NU = 1.0, 0.3                               # examples/linear_elasticity/cantilever/3D_Script.jl:52-63 (E = 1, nu = 0.3)
LAM, MU = E_MOD * NU / ((1 + NU) * (1 - 2 * NU)), E_MOD / (2 * (1 + NU))
TAU = 1000.0 * E_MOD
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)

H_PEN, TW = 1000.0, 1173.15                       # examples/thermal_conduction/2D_Script.jl:46-47: h_penalty, Tw of the weakly imposed Dirichlet face

# The reference's OWN solver / boundary-condition path at scale (round 5): legs of the default line, a few timed steps each.
# (in the line: `ref_idrs8_256`, `nitsche_c2_256`, `nitsche_c4_128`; alone: --config ref_idrs8 | nitsche_c2 | nitsche_c4)
#   ref_idrs8        configs[1]'s mesh and form solved the way every example script does: idrs!(s = 8) -- the default Sv_func!, src/MetaFEM.jl:36-37,
#                    linear_solver/04_IDRs.jl:26-95 -- with Pr_Jacobi! (02_Preconditioner.jl:32-76)
#   nitsche_c2       the same mesh with the temperature FIXED on x = 0 the reference's way: h_penalty*Bilinear(T, Tw - T) + k*Bilinear(T, n{i}*T{;i})
#                    (thermal_conduction/2D_Script.jl:58) -- K is NONSYMMETRIC; convection on the other five faces; bicgstabl_GS!(2)
#   nitsche_c4       configs[3]'s mesh (hex-27) with the same face, bicgstabl_GS!(2)
REF_LEGS = {
    "ref_idrs8": dict(base="c2", n=256, solver="idrs8", nitsche=False),
    "nitsche_c2": dict(base="c2", n=256, solver="bicgstabl2", nitsche=True),
    "nitsche_c4": dict(base="c4", n=128, solver="bicgstabl2", nitsche=True),
}
SOLVER_TEXT = {"cg": "Jacobi-CG iterations", "bicgstabl2": "SpMV-equivalent steps of bicgstabl_GS! (s = 2, right Jacobi)",
               "idrs8": "SpMV-equivalent steps of idrs! (s = 8, right Jacobi)"}

CONFIGS = {
    "c2": dict(title="3D thermal conduction, linear hex-8", order=1, itg=3, fields=1, n=512, solver="cg",
               metric="DOF-updates/sec (assembly+CG iter) on 3D hex thermal conduction"),
    "c3": dict(title="linear elasticity, hex-8, 3 DOF per node (penalty on x = 0, traction on y = L)", order=1, itg=3, fields=3, n=128,
               solver="bicgstabl2", metric="DOF-updates/sec (assembly+BiCGStab(2) SpMV-steps) on 3D hex-8 linear elasticity"),
    "c4": dict(title="3D thermal conduction, quadratic hex-27 (FP64-MFMA Ke)", order=2, itg=5, fields=1, n=128, solver="cg",
               metric="DOF-updates/sec (assembly+CG iter) on 3D hex-27 thermal conduction"),
}


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
        "sample": f"hex-8 {n_cpu}^3 thermal ({prob.mesh.ncp} DOF = {prob.mesh.ncp / 135005697:.4f} of the 512^3 workload, "
                  f"{prob.mesh.ncp / 16974593:.3f} of the 256^3 one), median of "
                  f"{len(times)} steps: 1 step = term-by-term assembly ({ta:.2f} s) + {iters} Jacobi-CG iterations ({ts:.2f} s) -- the "
                  f"GPU leg's iterations per assembly; C/OpenMP restatement of the reference algorithm and DATA LAYOUT (oracle/c/oracle.c: "
                  f"stored per-element basis tables, 2 KB + 0.5 KB of slot ids per hex-8 element -- 512^3 would need 340 GB of host "
                  f"memory and 256^3 55 GB, so the sample stays at {n_cpu}^3 = 25 GB and the ratio below is a per-DOF throughput ratio), "
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
    # rank 0's stdout is drained by a thread while all ranks are watched: a rank that dies early (no such device, import error) would leave
    # the others waiting in the rendezvous for ever -- they are ended (these exact processes) and the failure is reported
    import threading
    import time

    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = False
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            failed = True
            time.sleep(5.0)  # (let the others notice by themselves first)
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        time.sleep(0.2)
    reader.join(timeout=30)
    out0 = chunks[0] if chunks else b""
    rcs = [p.wait() for p in procs]
    if failed:
        print("bench.py: a rank exited with an error; the remaining ranks were ended", file=sys.stderr)
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


TRAFFIC_FILE = "profiles/r05_traffic.json"


def load_traffic():
    """profiles/r05_traffic.json: HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate passes, gfx950 x2
    FETCH correction, calibrated in the same run) of THIS tree's kernels, keyed by '<kernel key>@<workload key>'.  Collected by
    tools/run_pmc_r03.sh, not in this run (the line says so in `traffic_source`)."""
    try:
        return json.load(open(os.path.join(ROOT, TRAFFIC_FILE)))
    except Exception:
        return {}


def collect_traffic_live(leg: str, timeout_s: int = 240):
    """HBM-side bytes per launch of the priced kernels on `leg`, COLLECTED IN THIS RUN: two child runs of tools/pmc_leg.py under
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes as MI355X_MICROARCH.md prescribes; the program
    itself directly after `--`; children of this process, started with subprocess -- nothing here is exec-replaced), summarised by
    tools/make_r03_traffic.py (gfx950 x2 FETCH correction, calibration on k_axpby in the same runs).  Returns (dict, None) or (None, reason)."""
    import shutil
    import subprocess
    import tempfile

    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="mfem_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for grp in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, f"{leg}_{grp}")
            with open(d + ".log", "w") as log:
                # its own process group: a pass that overruns is ended as a group (the profiler's launcher AND the program under it)
                pr = subprocess.Popen([rp, "--kernel-trace", "--pmc", grp, "-d", d, "-o", "out", "--output-format", "csv", "--",
                                       sys.executable, os.path.join(ROOT, "tools", "pmc_leg.py"), leg, "2"],
                                      cwd="/tmp", env=env, stdout=log, stderr=subprocess.STDOUT, start_new_session=True)
                try:
                    rc = pr.wait(timeout=timeout_s)
                except subprocess.TimeoutExpired:
                    import signal

                    os.killpg(pr.pid, signal.SIGKILL)  # exactly the group started above
                    pr.wait()
                    return None, f"rocprofv3 --pmc {grp} pass did not finish in {timeout_s} s"
            if rc != 0:
                return None, f"rocprofv3 --pmc {grp} pass exited with {rc}"
        out = os.path.join(tmp, "traffic.json")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_r03_traffic.py"), tmp, out, "this bench.py run"],
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120)
        if r.returncode != 0 or not os.path.exists(out):
            return None, "tools/make_r03_traffic.py failed"
        return json.load(open(out)), None
    except Exception as e:  # a missing counter, a timeout: the committed file stays the source
        return None, repr(e)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS) + sorted(REF_LEGS), default="c2")
    ap.add_argument("--n", type=int, default=0, help="elements per side of the per-GPU mesh (0 = the config's size: 512 / 128 / 128)")
    ap.add_argument("--iters", type=int, default=200, help="Krylov iterations per step (BiCGStab(2): SpMV-equivalent steps)")
    ap.add_argument("--cpu-n", type=int, default=192, help="elements per side of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-repeats", type=int, default=3, help="timed CPU steps (the median is reported)")
    ap.add_argument("--secondary-n", type=int, default=256, help="c2 at N = 1: also run this mesh size (configs[1]) for a few steps (0 = skip)")
    ap.add_argument("--secondary-steps", type=int, default=5)
    ap.add_argument("--hex27-n", type=int, default=128, help="c2 at N = 1: CSR-kernel roofline on the hex-27 matrix of this size (0 = skip)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="N > 1: weak = an --n thick slab per rank of an (n N) x n x n mesh; strong = the config's n^3 mesh cut into N slabs")
    ap.add_argument("--secondary-configs", type=int, default=1,
                    help="c2 at N = 1, default size: also run configs[2] (c3) and configs[3] (c4) for --secondary-steps timed steps each and the hex-27 "
                         "Ke MFMA roofline (0 = skip)")
    ap.add_argument("--secondary-config-n", type=int, default=0,
                    help="elements per side of the c3 / c4 legs and of the hex-27 Ke roofline of the default line (0 = the configs' own 128; a smaller "
                         "value makes the legs run at any --n: functional checks of the line's schema)")
    ap.add_argument("--ref-legs", type=int, default=1,
                    help="c2 at N = 1, default size: also run the reference's own solver / boundary-condition path (REF_LEGS: idrs!(8) on configs[1], the "
                         "Nitsche-Dirichlet nonsymmetric K on configs[1] and configs[3] under bicgstabl_GS!(2)) for --secondary-steps timed steps each (0 = skip)")
    ap.add_argument("--time-to-tol", type=int, default=1,
                    help="c3 legs: after the fixed-count steps, solve the same system to ||r|| / sqrt(n) <= 1e-8 ||r0|| with bicgstabl_GS!(2), idrs!(8) and cg! "
                         "and report iterations / ms / converged (0 = skip)")
    ap.add_argument("--strong-leg", type=int, default=1,
                    help="N > 1 with --scaling weak (the driver's command): also time THE config's n^3 mesh cut into N slabs for --secondary-steps steps and "
                         "report it as `strong_scaling` (value, ms/step, exposed communication) -- one SCALE run then answers the north_star's '>= 6x at 8 GPUs' "
                         "for the weak AND the strong reading (0 = skip)")
    ap.add_argument("--strong-leg-timeout", type=float, default=240.0, help="seconds after which the strong-scaling leg is given up (the weak line is printed without it)")
    ap.add_argument("--remainder", type=int, default=1,
                    help="0: switch the skew remainder of the lattice tiles off (mfem_debug_set_remainder): a nonsymmetric K then takes the layouts that read "
                         "every entry, as until round 4 -- the A/B of profiles/r05_nitsche_ab.txt")
    ap.add_argument("--ws-trial", type=int, default=1,
                    help="1 (default here): opt in to the library's workspace placement trial (mfem_debug_set_ws_trial; OFF by default in the library "
                         "since round 4) -- the line says so in config.workspace_placement_trial and prints the first step's wall time; 0 = as the library ships")
    ap.add_argument("--live-traffic", type=int, default=1,
                    help="N = 1, default sizes: collect the roofline objects' `traffic` in this run (two rocprofv3 --pmc child passes on the "
                         "headline workload, ~40 s) instead of reading profiles/r05_traffic.json (0 = read the file)")
    args = ap.parse_args()
    if args.config in REF_LEGS:  # one of the reference-solver legs alone (profiles: tools/run_profiles_r05.sh)
        leg = REF_LEGS[args.config]
        cfg = dict(CONFIGS[leg["base"]], solver=leg["solver"], nitsche=leg["nitsche"], n=leg["n"],
                   metric=f"DOF-updates/sec (assembly + {SOLVER_TEXT[leg['solver']]}) on " + CONFIGS[leg["base"]]["title"]
                          + (" with a Nitsche-Dirichlet face (nonsymmetric K)" if leg["nitsche"] else ""))
    else:
        cfg = CONFIGS[args.config]
    if args.n <= 0:
        args.n = cfg["n"]

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: start the N ranks ourselves, as child processes and BEFORE anything here touches
        # the GPU; rank 0's JSON line is the only thing that reaches stdout; a failing rank fails the run
        sys.exit(self_launch(args.gpus))

    import ctypes as C

    import torch

    import metafem_jl_amd as mf
    from metafem_jl_amd import _lib, parallel

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
    traffic_db = load_traffic()
    state = {"gloo_group": None, "transport": None}  # set when the RCCL transport had to be replaced (run_workload)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def traffic_of(key):
        t = traffic_db.get(key)
        if not t:
            return None, None
        return t.get("hbm_bytes_per_launch"), (f"profiles/r05_traffic.json['{key}']: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                               f"kernel on this workload (tree {t.get('tree', '?')}; gfx950 x2 FETCH_SIZE correction, calibrated "
                                               f"in the same run), not collected in this run")

    def csr_kernel_roofline(A, K, wkey, launches=20):
        """The CSR kernel behind mul! (mfem_spmv_csr: caller's CSR arrays, no copy) on this matrix: live hip-event timing of `launches`
        launches.  `achieved` / `frac` price the launch with the bytes the kernel moves BY DESIGN (mfem_csr_spmv_bytes: values, the
        columns it reads -- tiles whose rows repeat one column-offset list read only their first rows' columns --, x, y, row pointers);
        `csr_equivalent` with SURVEY 8(d)'s formula (12 B per nonzero), the north_star's 'CSR SpMV % of HBM roofline' of a kernel
        without that inspection; `frac_actual` with the PMC-measured traffic of profiles/r05_traffic.json."""
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
        formula = A.nnz * 12 + A.n * 16 + (A.n + 1) * 8  # val 8 + col 4 per nonzero; x, y 8 per row; i64 row pointers
        design, cols_read = A.spmv_bytes()
        gbs = lambda b: b / (ms * 1e-3) / 1e9
        traffic, src = traffic_of(f"csr_kernel@{wkey}")
        return {"kernel": "mfem_spmv_csr (mul!): k_spmv_csr_w (rows of up to 64 entries, uniform length) / k_spmv_csr_rb (wide or uneven rows) "
                          "on the caller's CSR arrays, no copy",
                "avg_launch_ms": ms, "launches": int(cnt.value), "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": design, "column_entries_read": cols_read,
                "achieved": gbs(design), "frac": gbs(design) / HBM_PEAK_GBS,
                "csr_equivalent": {"bytes_per_launch": formula, "achieved": gbs(formula), "frac": gbs(formula) / HBM_PEAK_GBS,
                                   "note": "SURVEY 8(d): nnz*12 + n*16 + (n+1)*8 -- what a CSR kernel that reads every column index moves"},
                "traffic": traffic, "traffic_source": src,
                "traffic_over_algorithmic": (traffic / design) if traffic else None,
                "frac_actual": (gbs(traffic) / HBM_PEAK_GBS) if traffic else None,
                "n": A.n, "nnz": A.nnz}

    def hex27_ke_roofline(n27, repeats=5):
        """north_star: 'MFMA used only for the dense per-element Ke = B^T D B contraction on high-order hex elements ... evidenced by MFMA
        utilisation (hex-27 Ke) against gfx950 peak'.  The element kernels of the hex-27 matrix assembly (pass 1: geometry + 63 FP64 MFMAs per
        element into a scratch of element matrices; pass 2: row-owner gather into CSR; no boundary faces) timed live with events on the stream they
        are launched on (the context runs on torch's current stream), priced with SURVEY 8(d)'s USEFUL flops: 2 * 27 * 27 * 81 = 118 098 per element.
        The MFMA-pipe busy fraction of pass 1 comes from the committed counter file (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128 SIMDs per XCD))."""
        PEAK_TFLOPS = 78.6  # MI355X_MICROARCH.md: FP64 matrix (MFMA) peak, dense
        b27 = mf.Brick((1.0, 1.0, 1.0), (n27,) * 3, 2, 5, ctx=ctx)
        A27 = b27.pattern(1)
        K27 = torch.empty(A27.nnz, dtype=torch.float64, device=dev)

        def timed_assembly():
            b27.assemble_thermal(A27, K_COND, 0.0, TENV, 0, out=K27)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(repeats):
                b27.assemble_thermal(A27, K_COND, 0.0, TENV, 0, out=K27)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / repeats

        # The make_Brick mesh is affine element by element, and the library then assembles without ever storing Ke (k_hex27_direct: round 4).  The MFMA
        # figure below is the GENERAL path's -- what a mesh with one curved element takes -- forced here on the same mesh (bit 9 of mfem_debug_set_hex27),
        # the default path's time is reported beside it against the bytes it must write.
        ms_direct = timed_assembly()
        Kd = K27.clone()
        _lib.lib.mfem_debug_set_hex27(1 << 9)
        try:
            ms = timed_assembly()
        finally:
            _lib.lib.mfem_debug_set_hex27(0)
        agree = float((K27 - Kd).abs().max() / K27.abs().max())
        assert agree <= 1e-13, f"hex-27: the scratch-free assembly and the two-pass MFMA assembly differ by {agree:.2e} of max |K|"
        nnz27, n27_rows = int(A27.nnz), int(A27.n)
        del Kd
        # ... and a mesh of GENERAL elements (the centre node of every element moved): round 5's row-owner kernel computes the rows from per-element G_q
        # (k_hex27_gq_lane + k_hex27_rows_gq, Ke never stored); the two-pass MFMA path on the same coordinates beside it (bit 11 turns the row-owner kernel off)
        m27 = 2 * n27 + 1
        odd = torch.arange(1, m27, 2, device=dev)
        centre = ((odd[:, None, None] * m27 + odd[None, :, None]) * m27 + odd[None, None, :]).reshape(-1)
        b27.coords_view(0)[centre] += 0.015 / n27
        rows_before = int(_lib.lib.mfem_debug_hex27_rows_count())
        ms_rows = timed_assembly()
        rows_ran = int(_lib.lib.mfem_debug_hex27_rows_count()) > rows_before
        Kd = K27.clone()
        _lib.lib.mfem_debug_set_hex27(1 << 11)
        try:
            ms_two_pass_general = timed_assembly()
        finally:
            _lib.lib.mfem_debug_set_hex27(0)
        agree_general = float((K27 - Kd).abs().max() / K27.abs().max())
        assert rows_ran and agree_general <= 2e-13, f"hex-27 general elements: rows from G_q against the two-pass MFMA path: {agree_general:.2e} of max |K| (ran: {rows_ran})"
        del Kd
        nel = n27 ** 3
        flops = 118098.0 * nel
        busy = None
        src = None
        for key in ("hex27_counters_MFMA@c4_%d" % n27,):
            ent = traffic_db.get(key, {}).get("per_kernel", {})
            for kname, cs in ent.items():
                if "k_hex27<true" in kname and cs.get("SQ_VALU_MFMA_BUSY_CYCLES", {}).get("mean"):
                    busy = cs["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / (cs["GRBM_GUI_ACTIVE"]["mean"] * 128.0)
                    src = (f"{TRAFFIC_FILE}['{key}']['{kname}']: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128) of the pass-1 kernel, "
                           f"rocprofv3 --pmc pass of tools/run_pmc_r03.sh (tree {traffic_db[key].get('tree', '?')}), not collected in this run")
        del b27, A27, K27
        torch.cuda.empty_cache()
        return {"kernel": "k_hex27<true, true> (pass 1: sum-factorised geometry + Ke = B^T D B on __builtin_amdgcn_mfma_f64_16x16x4f64, element matrices to "
                          "scratch) + k_hex27_gather_lds (pass 2: row-owner gather into sorted CSR); one matrix assembly without boundary faces",
                "matrix": f"hex-27 thermal {n27}^3 ({nel} elements)", "bound": "mfma", "unit": "TFLOP/s", "peak": PEAK_TFLOPS,
                "useful_flop_per_assembly": flops, "avg_assembly_ms": ms, "assemblies_timed": repeats,
                "achieved": flops / (ms * 1e-3) / 1e12, "frac": flops / (ms * 1e-3) / 1e12 / PEAK_TFLOPS,
                "mfma_pipe_busy_in_pass1": busy, "mfma_pipe_busy_source": src,
                "note": "useful flops only (SURVEY 8(d)): the B build and the zero-padding of the 27 x 81 operands to MFMA tiles are not counted; "
                        "measured with the scratch-free path of all-affine meshes turned off (this mesh would take it: `affine_mesh_assembly`)",
                "affine_mesh_assembly": {
                    "kernel": "k_hex27_affine_g0 (per element: affine test on the 27 nodes + G0 = -k adj(J) adj(J)^T / det) + k_hex27_direct (row-owner gather, "
                              "each (row, element) run computed from G0 and the 1-D reference integrals in registers, rows accumulated in LDS): the default "
                              "when every element is affine; Ke is never stored",
                    "avg_assembly_ms": ms_direct, "bound": "hbm", "unit": "GB/s", "peak": 8000.0,
                    "algorithmic_bytes_per_assembly": 8 * nnz27 + 2 * 48 * nel + 3 * 8 * n27_rows,
                    "achieved": (8 * nnz27 + 2 * 48 * nel + 3 * 8 * n27_rows) / (ms_direct * 1e-3) / 1e9,
                    "frac": (8 * nnz27 + 2 * 48 * nel + 3 * 8 * n27_rows) / (ms_direct * 1e-3) / 1e9 / 8000.0,
                    "max_rel_difference_to_the_mfma_path": agree,
                    "note": "algorithmic bytes = the CSR values written once + G0 written and read + the coordinates read"},
                "general_mesh_assembly": {
                    "kernel": "k_hex27_affine_g0 (affine test) + k_hex27_gq_lane (G_q = -k w det J^-1 J^-T at the 27 Gauss points, one lane per element) + k_hex27_rows_gq "
                              "(row owners compute every (row, element) run from G_q by sum factorisation in registers, rows accumulated in LDS without atomics): the "
                              "default from 30 % non-affine elements on (three Gauss points per direction); Ke is never stored",
                    "mesh": "the same mesh with the centre node of EVERY element moved (all elements non-affine)",
                    "avg_assembly_ms": ms_rows, "two_pass_mfma_path_same_mesh_ms": ms_two_pass_general,
                    "useful_flop_per_assembly": flops, "achieved_on_useful_flops_tflops": flops / (ms_rows * 1e-3) / 1e12,
                    "frac_of_fp64_peak_on_useful_flops": flops / (ms_rows * 1e-3) / 1e12 / PEAK_TFLOPS,
                    "executed_flop_per_assembly": 2.0 * 1053 * 27 * nel,
                    "max_rel_difference_to_the_mfma_path": agree_general,
                    "note": "useful flops = the 118 098 per element of Ke = B^T D B (the yardstick of the MFMA path); the kernel itself executes 2 x 1053 FMAs per "
                            "(row, element) run; counters and the development steps: profiles/r05_hex27_rows.txt"}}

    strong = args.scaling == "strong" and world > 1

    def run_workload(cfg, ckey, N, steps, warmup, want_csr, strong=strong):
        """One workload of config `ckey`, `steps` timed steps: weak scaling = an (N * world) x N x N mesh (an N-thick slab per rank), strong scaling =
        the N^3 mesh cut into `world` slabs along i."""
        order, F = cfg["order"], cfg["fields"]
        nx_global = N if strong else N * world
        brick = mf.Brick((1.0 if strong else float(world), 1.0, 1.0), (nx_global, N, N), order, cfg["itg"], ctx=ctx)
        m0, m1, m2 = brick.m
        comm = None
        if use_comm:
            plo, phi = parallel.slab_planes(m0, world, rank, order)
            brick.set_slab(plo, phi)
            # RCCL prints a version banner / warnings through C stdio on stdout: send them to stderr so that stdout
            # carries only the JSON line
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)
            try:
                if host_comm:
                    comm = parallel.HostSlabComm(ctx, brick, rank, world, n_fields=F)
                else:
                    ok, why = 1, ""
                    try:
                        comm = parallel.SlabComm(ctx, brick, rank, world, n_fields=F)
                        if os.environ.get("MFEM_BENCH_SIMULATE_RCCL_FAILURE") == "1":  # test hook for the fallback below
                            raise RuntimeError("simulated failure of the RCCL self-test")
                        if os.environ.get("MFEM_BENCH_SKIP_SELFTEST") != "1":
                            # every RCCL call of the solver's schedule once, on a ring, before the timed region: a transport problem
                            # shows up here with a message instead of as a hang inside the Krylov loop
                            _lib.check(_lib.lib.mfem_debug_comm_selftest(ctx._h, order * m1 * m2, 2))
                    except Exception as e:  # the library's RCCL transport is unusable on this rank
                        ok, why = 0, repr(e)
                    if dist is not None:  # all ranks take the same transport
                        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
                        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                        ok = int(flag.item())
                    if not ok:
                        # fall back to the host-callback transport (device -> pinned host -> gloo -> device): the same solver code path,
                        # a slower exchange -- the run still produces its line, and says so
                        print(f"bench.py rank {rank}: RCCL transport of the library not usable ({why or 'another rank failed'}); "
                              f"falling back to the host-callback transport over gloo", file=sys.stderr)
                        if comm is not None:
                            comm.close()
                        if state.get("gloo_group") is None and dist is not None:
                            state["gloo_group"] = dist.new_group(backend="gloo")
                        comm = parallel.HostSlabComm(ctx, brick, rank, world, n_fields=F, group=state.get("gloo_group"))
                        state["transport"] = "host callbacks over gloo (fallback: the library's RCCL self-test failed)"
                C.CDLL(None).fflush(None)
            finally:
                os.dup2(saved, 1)
                os.close(saved)
        A = brick.pattern(F)
        n_local = A.n
        K = torch.empty(A.nnz, dtype=torch.float64, device=dev)
        xlen = parallel.local_vector_length(brick.slab[0], brick.slab[1], m1, m2, F, order) if use_comm else n_local
        x_star = torch.zeros(xlen, dtype=torch.float64, device=dev)
        s = torch.full((xlen,), SRC, dtype=torch.float64, device=dev) if F == 1 else None
        R = torch.empty(n_local, dtype=torch.float64, device=dev)
        n_global = F * m0 * m1 * m2
        x0, y1 = mf.FACE_BITS["x0"], mf.FACE_BITS["y1"]

        if F == 1:
            fixed = x0 if cfg.get("nitsche") else 0
            fix = dict(fixed_faces=fixed, h_penalty=H_PEN if fixed else 0.0, Tw=TW if fixed else 0.0)
            robin = mf.ALL_FACES & ~fixed

            def assemble():
                brick.assemble_thermal(A, K_COND, H, TENV, robin, out=K, **fix)
                brick.residual_thermal(x_star, K_COND, H, TENV, robin, s=s, out=R, **fix)
        else:
            def assemble():
                brick.assemble_elasticity(A, LAM, MU, TAU, x0, out=K)
                brick.residual_elasticity(x_star, LAM, MU, TAU, x0, y1, (0.0, 1.0, 0.0, 0.0, 0.0, 0.0), out=R)

        if cfg["solver"] == "cg":
            def solve():
                return mf.iterative_Solve(A, K, R, 1e-300, Sv_func=mf.cg_, Pr_func=mf.Pr_Jacobi_, maxiter=args.iters, max_pass=1,
                                          fixed_iterations=True)
        elif cfg["solver"] == "idrs8":
            # idrs! with s = 8 (04_IDRs.jl:26-95): every inner step is one SpMV and advances `iter` by one
            def solve():
                return mf.iterative_Solve(A, K, R, 1e-300, Sv_func=mf.idrs_, Pr_func=mf.Pr_Jacobi_, maxiter=args.iters, max_pass=1, s=8,
                                          fixed_iterations=True)
        else:
            # bicgstabl_GS! with s = 2 (03_BiCGstabl.jl:18-96): one sweep = 4 SpMVs and advances the solver's `iter` by s = 2
            # (:93), so maxiter = iters / 2 gives `--iters` SpMV-equivalent steps per solve
            def solve():
                return mf.iterative_Solve(A, K, R, 1e-300, Sv_func=mf.bicgstabl_GS_, Pr_func=mf.Pr_Jacobi_, maxiter=max(args.iters // 2, 2),
                                          max_pass=1, s=2, fixed_iterations=True)

        def step():
            assemble()
            return solve()

        t_first = time.perf_counter()
        first_split = None
        for w in range(warmup):
            if w == 0:
                # the first step, split: assembly (tables, first launches) | the first solve = one-off plans of the solver layouts +
                # workspace allocation (+ the placement trial if opted in) + the solve itself
                assemble()
                torch.cuda.synchronize()
                t_asm1 = time.perf_counter() - t_first
                solve()
                torch.cuda.synchronize()
                t_first = time.perf_counter() - t_first
                first_split = {"assembly_s": t_asm1, "solve_s": t_first - t_asm1}
            else:
                t_w = time.perf_counter()
                _, st_w = step()
                torch.cuda.synchronize()
                if first_split is not None and "steady_step_s" not in first_split:
                    first_split["steady_step_s"] = time.perf_counter() - t_w
                    first_split["steady_solve_s"] = st_w.solve_ms * 1e-3
                    first_split["one_off_s"] = max(first_split["solve_s"] - st_w.solve_ms * 1e-3, 0.0)
                    first_split["note"] = ("one_off_s = first solve - a steady solve: pattern inspection and layout plans (once per pattern), workspace "
                                           "allocation, and the placement trial's extra allocations and timed products when --ws-trial 1")
        _lib.check(_lib.lib.mfem_prof_spmv_enable(ctx._h, 1))
        if comm is not None:
            _lib.check(_lib.lib.mfem_prof_comm_enable(ctx._h, 1))
            hw, hn, aw, an = C.c_double(), C.c_int64(), C.c_double(), C.c_int64()
            _lib.check(_lib.lib.mfem_prof_comm_read(ctx._h, C.byref(hw), C.byref(hn), C.byref(aw), C.byref(an), 1))
        tot, cnt = C.c_double(), C.c_int64()
        _lib.check(_lib.lib.mfem_prof_spmv_read(ctx._h, C.byref(tot), C.byref(cnt), 1))
        sym_count0 = int(_lib.lib.mfem_debug_sym_spmv_count())
        lat_count0 = int(_lib.lib.mfem_debug_lat27_spmv_count())
        lat8_count0 = int(_lib.lib.mfem_debug_lat8_spmv_count())
        rem_count0 = int(_lib.lib.mfem_debug_rem_spmv_count())
        barrier()
        t0 = time.perf_counter()
        solve_ms = 0.0
        iters_done = spmvs_done = 0
        st = dx_last = None
        for _ in range(steps):
            dx_last, st = step()
            solve_ms += st.solve_ms
            iters_done += st.iterations
            spmvs_done += st.spmv_count
        barrier()
        elapsed = time.perf_counter() - t0
        _lib.check(_lib.lib.mfem_prof_spmv_read(ctx._h, C.byref(tot), C.byref(cnt), 1))
        _lib.check(_lib.lib.mfem_prof_spmv_enable(ctx._h, 0))
        comm_exposed = None
        if comm is not None:
            # communication this rank's solver stream was exposed to during the timed steps (mfem_prof_comm_*), every rank's numbers on rank 0
            _lib.check(_lib.lib.mfem_prof_comm_read(ctx._h, C.byref(hw), C.byref(hn), C.byref(aw), C.byref(an), 1))
            _lib.check(_lib.lib.mfem_prof_comm_enable(ctx._h, 0))
            mine = torch.tensor([hw.value, float(hn.value), aw.value, float(an.value), solve_ms], dtype=torch.float64,
                                device="cpu" if (host_comm or state.get("gloo_group") is not None) else dev)
            if dist is not None:
                allr = [torch.empty_like(mine) for _ in range(world)]
                dist.all_gather(allr, mine, group=state.get("gloo_group") if mine.device.type == "cpu" and not host_comm else None)
            else:
                allr = [mine]
            comm_exposed = [{"rank": i, "halo_wait_ms_per_step": float(v[0]) / steps, "halo_waits_per_step": float(v[1]) / steps,
                             "allreduce_ms_per_step": float(v[2]) / steps, "allreduces_per_step": float(v[3]) / steps,
                             "solve_ms_per_step": float(v[4]) / steps,
                             "exposed_fraction_of_solve": (float(v[0]) + float(v[2])) / max(float(v[4]), 1e-12)} for i, v in enumerate(allr)]
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if host_comm else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        # one "DOF update" = one Krylov step on one unknown: a CG iteration (1 SpMV), or one SpMV-equivalent step of BiCGStab(2)
        updates = iters_done if cfg["solver"] == "cg" else spmvs_done
        res = {"N": N, "nx_global": nx_global, "n_global": n_global, "nnz": A.nnz, "n_local": A.n, "elapsed": elapsed,
               "steps": steps, "solve_ms": solve_ms, "iters_done": iters_done, "spmvs_done": spmvs_done, "updates": updates,
               "spmv_ms": tot.value / max(cnt.value, 1), "spmv_launches": int(cnt.value),
               # the LAST timed solve, as the library reports it (true residual ||b - A x|| / sqrt(n) recomputed after the pass, 02_Preconditioner.jl:53-55)
               "initial_res": st.initial_res if st else None, "final_res": st.final_res if st else None,
               "first_step_s": t_first if warmup > 0 else None, "first_step_split": first_split, "comm_exposed": comm_exposed, "cfg": cfg, "ckey": ckey}
        # A = S + N (csrc/spmv_rem.hip): did the timed solves run on symmetric lattice tiles + the sparse skew remainder of a nonsymmetric K, and how big was it
        rr_, re_, ra_ = C.c_int64(), C.c_int64(), C.c_double()
        _lib.check(_lib.lib.mfem_debug_remainder_info(A._h, C.byref(rr_), C.byref(re_), C.byref(ra_)))
        res["remainder"] = ({"rows": rr_.value, "entries": re_.value, "fraction_of_rows": rr_.value / max(A.n, 1),
                             "asymmetry_measured_on_the_tiles_alone": ra_.value,
                             "note": "the values are nonsymmetric in these rows (Nitsche face); the symmetric lattice tiles serve the solve and the mirrored "
                                     "entries' differences N[r][c] = A[r][c] - A[c][r] of these rows are applied behind them (k_rem_apply)"}
                            if int(_lib.lib.mfem_debug_rem_spmv_count()) > rem_count0 else None)
        if cfg.get("time_to_tol") and world == 1 and not use_comm and args.time_to_tol:
            # what a user of the script sees: the SAME system solved to a tolerance (||r|| / sqrt(n) <= 1e-8 ||r0||, 4 passes of 5000) by the reference's two
            # solvers and by cg! (K is symmetric here)
            ttt = {}
            for name, kw in (("bicgstabl_GS!(2)", dict(Sv_func=mf.bicgstabl_GS_, s=2)), ("idrs!(8)", dict(Sv_func=mf.idrs_, s=8)), ("cg!", dict(Sv_func=mf.cg_))):
                _, t_st = mf.iterative_Solve(A, K, R, 1e-8 * st.initial_res, Pr_func=mf.Pr_Jacobi_, maxiter=5000, max_pass=4, **kw)
                ttt[name] = {"converged": bool(t_st.converged), "passes": t_st.passes, "iterations": t_st.iterations, "spmvs": t_st.spmv_count,
                             "ms": t_st.solve_ms, "final_res_over_initial": t_st.final_res / st.initial_res}
            res["time_to_tol"] = {"target": "||r|| / sqrt(n) <= 1e-8 x ||r0|| / sqrt(n), right Jacobi, maxiter = 5000 per pass, max_pass = 4 "
                                            "(02_Preconditioner.jl:32-76 semantics: true residual between passes)", **ttt}
        if world == 1 and not use_comm and dx_last is not None:
            # ... and recomputed OUTSIDE the solver, after the timed region: ||R - K dx|| / sqrt(n) with mul! = the CSR kernel on the caller's arrays
            # (another kernel, another copy of the matrix than the solver layout the Krylov loop ran on)
            rr = torch.empty_like(R)
            mf.mul_(rr, A, K, dx_last)
            rr.sub_(R)
            res["final_res_recomputed"] = mf.normalized_norm(rr, ctx=ctx)
            del rr
        del dx_last
        if rank == 0:
            if cfg["solver"] == "cg":
                assert iters_done == args.iters * steps, (iters_done, args.iters, steps)
            csr_bytes = A.nnz * 12 + A.n * 16 + (A.n + 1) * 8
            mode, slots, npad, reg = C.c_int32(), C.c_int32(), C.c_int64(), C.c_int64()
            _lib.check(_lib.lib.mfem_csr_solver_layout(ctx._h, A._h, C.byref(mode), C.byref(slots), C.byref(npad), C.byref(reg)))
            lat_used = int(_lib.lib.mfem_debug_lat27_spmv_count()) > lat_count0
            if mode.value == 4 and not lat_used:
                # the structure allows the symmetric lattice tiles, but the values of these solves did not pass the symmetry measure: the sliced
                # layout ran -- ask for its accounting
                _lib.lib.mfem_debug_set_lat27(0)
                _lib.check(_lib.lib.mfem_csr_solver_layout(ctx._h, A._h, C.byref(mode), C.byref(slots), C.byref(npad), C.byref(reg)))
            lat8_used = int(_lib.lib.mfem_debug_lat8_spmv_count()) > lat8_count0
            lat8_one_field = False
            if mode.value == 5 and not lat8_used:  # (the same for the 3-field lattice tiles: the diagonal-slotted layout ran)
                _lib.lib.mfem_debug_set_lat8(0)
                _lib.check(_lib.lib.mfem_csr_solver_layout(ctx._h, A._h, C.byref(mode), C.byref(slots), C.byref(npad), C.byref(reg)))
            elif mode.value != 5 and lat8_used:
                # one field: the layout query answers for cg! (which keeps the bitwise patch sweep); idrs! / bicgstabl_GS! work on A D^-1 and ran on the tiles
                lat8_one_field = True
                _lib.lib.mfem_debug_set_lat8(3)
                _lib.check(_lib.lib.mfem_csr_solver_layout(ctx._h, A._h, C.byref(mode), C.byref(slots), C.byref(npad), C.byref(reg)))
            ent, symf, byts = C.c_int64(), C.c_int32(), C.c_int64()
            _lib.check(_lib.lib.mfem_csr_solver_layout_entries(ctx._h, A._h, C.byref(ent), C.byref(symf)))
            _lib.check(_lib.lib.mfem_csr_solver_layout_bytes(ctx._h, A._h, C.byref(byts)))
            if not lat_used:
                _lib.lib.mfem_debug_set_lat27(1)
            if not lat8_used or lat8_one_field:
                _lib.lib.mfem_debug_set_lat8(1)
            sym_used = bool(symf.value) and int(_lib.lib.mfem_debug_sym_spmv_count()) > sym_count0
            plain_bytes = None
            spmv_bytes = byts.value
            if mode.value == 2:
                # diagonal-slotted blocks read no column stream: 8 B per nonzero + x, y; rows in generic blocks also read 4 B columns
                kernel, kkey = "k_spmv_dia (SpMV on the slot-major copy of the CSR matrix made once per solve; diagonal-slotted blocks)", "k_spmv_dia"
                plain = A.nnz * 8 + A.n * 16 + max(A.n - reg.value, 0) * slots.value * 4
                if sym_used and symf.value == 2:
                    kernel = ("k_spmv_symp<0> (one launch: the sweep, then the two boundary planes row by row): symmetric sweep on wave-private "
                              "(j, k) patches of a patch-major copy of the CSR matrix made once per solve; the values passed the "
                              "per-solve bitwise symmetry check, so 10.5 of a row's 13 lower-diagonal entries are mirrored through "
                              "LDS (bitwise the same y as the plain diagonal-slotted kernel); x staged per lattice plane in LDS")
                    kkey, plain_bytes = "k_spmv_symp", plain
                elif sym_used:
                    kernel = ("k_spmv_sym27 (+ k_spmv_dia on the two boundary planes): SpMV on the slot-major copy of the CSR "
                              "matrix made once per solve; the values passed the per-solve bitwise symmetry check, so lower-diagonal "
                              "entries are mirrored through LDS (bitwise the same y as the plain diagonal-slotted kernel)")
                    kkey, plain_bytes = "k_spmv_sym27", plain
                else:
                    spmv_bytes = plain
            elif mode.value == 1:
                kernel, kkey = "k_spmv_ell (SpMV on the slot-major copy of the CSR matrix made once per solve; f64 val / i32 col)", "k_spmv_ell"
            elif mode.value == 4:
                kernel, kkey = ("k_spmv_lat27 + k_lat27_gather (two launches per SpMV): symmetric lattice tiles of the hex-27 matrix, copy made once per "
                                "solve; the values passed the per-solve symmetry measure (probe product against the CSR kernel within 4e-13 max |A[r][c]|), so only the "
                                "diagonal and the entries with column > row are stored and read (14..63 of a row's 27..125); x and y of a tile of "
                                "8 x 8 x 32 lattice points in LDS, mirrored products added there (ds_add_f64); the second launch sums the tiles' y blocks "
                                "in a fixed order; y equals the CSR kernel's to round-off"), "k_spmv_lat27"
                if cfg["solver"] == "cg" and world == 1 and lat_used and _lib.lib.mfem_debug_lat27_cg_fused():
                    # one rank: the CG iteration runs pass 2 inside its residual update (k_lat27_gather_cg); the SpMV launch the library times is pass 1
                    kernel, kkey = ("k_spmv_lat27 (pass 1 of the SpMV; pass 2 -- the sums over the tiles' y blocks -- runs inside the CG residual update, "
                                    "k_lat27_gather_cg, and p . A p comes from pass 1: A p is never stored): symmetric lattice tiles of the hex-27 matrix, copy made "
                                    "once per solve; the values passed the per-solve symmetry measure (probe product against the CSR kernel within 4e-13 max |A[r][c]|), "
                                    "so only the diagonal and the entries with column > row are stored and read (14..63 of a row's 27..125); x and y of a tile of "
                                    "8 x 8 x 32 lattice points in LDS, mirrored products added there (ds_add_f64)"), "k_spmv_lat27_pass1"
                    spmv_bytes = int(_lib.lib.mfem_debug_lat27_pass1_bytes(A._h))
            elif mode.value == 5:
                kernel, kkey = (f"k_spmv_lat8<{F}> + k_lat8_gather (two launches per SpMV): symmetric lattice tiles of the {F}-field 27-point matrix, copy made "
                                "once per solve; the values passed the per-solve symmetry measure (probe product against the CSR kernel within 4e-13 of each row's diagonal), so per node only "
                                + ("the 6 upper entries of its own 3 x 3 block and the blocks towards its 13 upper neighbours are stored and read (123 of "
                                   "243 values)" if F == 3 else "the diagonal and the entries towards its 13 upper neighbours are stored and read (14 of 27 values)") +
                                "; lane = node, x and y of a tile of 8 x 8 x 16 nodes in LDS, mirrored products added there (ds_add_f64), "
                                "the right Jacobi scaling applied to x while it is staged; the second launch sums the tiles' y blocks in a fixed "
                                "order; y equals the CSR kernel's to round-off"), "k_spmv_lat8"
            elif mode.value == 3:
                kernel, kkey = ("k_spmv_sell (rows sorted by length and diagonal-list signature, SELL-128 copy made once per solve; blocks "
                                "whose rows share one diagonal list read no columns)"), "k_spmv_sell"
            else:
                kernel, kkey = "mfem_spmv_csr kernel (CSR SpMV, i64 rowptr / i32 col / f64 val)", "csr_kernel"
                spmv_bytes = A.spmv_bytes()[0]
            if res["remainder"] and mode.value in (4, 5):
                kernel += (f" + k_rem_apply (A = S + N: the values are NONSYMMETRIC in {res['remainder']['rows']} rows -- the Nitsche face --; the tiles apply the "
                           "mirrored upper triangle S, a third launch adds the skew remainder N of those rows; the SAME symmetry measure passed on S + N)")
            res.update(kernel=kernel, kernel_key=kkey, spmv_bytes=spmv_bytes, csr_bytes=csr_bytes, mode=mode.value, sym_used=sym_used,
                       plain_bytes=plain_bytes)
            if want_csr and world == 1:
                res["csr_kernel"] = csr_kernel_roofline(A, K, f"{ckey}_{N}")
        if comm is not None:
            comm.close()
        del brick, A, K, x_star, s, R
        torch.cuda.empty_cache()
        return res

    def solver_roofline(r, wkey):
        achieved = r["spmv_bytes"] / (r["spmv_ms"] * 1e-3) / 1e9
        csr_equiv = r["csr_bytes"] / (r["spmv_ms"] * 1e-3) / 1e9
        traffic, src = traffic_of(f"{r['kernel_key']}@{wkey}") if world == 1 else (None, None)
        return {
            "kernel": r["kernel"],
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src,
            "traffic_over_algorithmic": (traffic / r["spmv_bytes"]) if traffic else None,
            "frac_actual": (traffic / (r["spmv_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
            "algorithmic_bytes_per_launch": r["spmv_bytes"], "avg_launch_ms": r["spmv_ms"], "launches": r["spmv_launches"],
            **({"plain_diagonal_kernel_bytes_per_launch": r["plain_bytes"],
                "note": "algorithmic bytes = what this kernel design reads: 8 B per matrix entry not mirrored from LDS "
                        "(14 upper-diagonal entries per row + the patch-edge entries; workgroup-tile sweep: about 18.6 of 27) "
                        "+ x as the kernel stages it (patch sweep: overlapping patch neighbourhoods, 1.6 n entries) + y "
                        "(mfem_csr_solver_layout_bytes); the plain diagonal-slotted kernel reads "
                        "plain_diagonal_kernel_bytes_per_launch"}
               if r["sym_used"] and r["kernel_key"] not in ("k_spmv_lat27", "k_spmv_lat27_pass1", "k_spmv_lat8") else {}),
            **({"note": "algorithmic bytes = what this kernel pair moves by design (mfem_csr_solver_layout_bytes): 8 B per stored entry (diagonal + upper "
                        "entries, padded to 68 wave steps per 128 rows, units cut by the lattice edge included) + x as the tiles stage it (4 320 cells per "
                        "2 048 rows) + the tiles' y blocks written and read again + y"}
               if r["kernel_key"] == "k_spmv_lat27" else {}),
            **({"note": "algorithmic bytes = what pass 1 moves by design (mfem_debug_lat27_pass1_bytes): 8 B per stored entry (diagonal + upper entries, padded "
                        "to 68 wave steps per 128 rows, units cut by the lattice edge included) + x as the tiles stage it (4 320 cells per 2 048 rows) + the "
                        "tiles' y blocks written; reading them back is part of the residual update kernel's bytes"}
               if r["kernel_key"] == "k_spmv_lat27_pass1" else {}),
            **({"note": "algorithmic bytes = what this kernel pair moves by design (mfem_csr_solver_layout_bytes): 8 B per stored entry (123 per node, "
                        "padded to 124 wave steps per 64 nodes, units cut by the lattice edge included) + x and the Jacobi scaling as the tiles stage them "
                        "(1 620 cells per field per 1 024 nodes) + the tiles' y blocks written and read again + y"}
               if r["kernel_key"] == "k_spmv_lat8" else {}),
            "csr_equivalent": {"bytes_per_launch": r["csr_bytes"], "achieved": csr_equiv, "frac": csr_equiv / HBM_PEAK_GBS,
                               "note": "the same launch priced with SURVEY 8(d)'s CSR formula (12 B per nonzero): what a CSR "
                                       "kernel would have to sustain to match this time"},
        }

    if args.ws_trial:
        _lib.check(_lib.lib.mfem_debug_set_ws_trial(1))
    if not args.remainder:
        _lib.check(_lib.lib.mfem_debug_set_remainder(0))
    main_res = run_workload(cfg, REF_LEGS[args.config]["base"] if args.config in REF_LEGS else args.config, args.n, args.steps, args.warmup, want_csr=True)

    def check_residual(r, what):
        """bench.py checks what it times: the last timed solve must have reduced the residual (and produced finite numbers)."""
        ir, fr = r["initial_res"], r["final_res"]
        # CG reduces the energy norm monotonically and, over 200 iterations, the residual too: final < initial is required.  bicgstabl_GS! is not
        # monotone -- on the penalty-constrained elasticity operator of c3 ||r|| hovers around ||r0|| for the first hundreds of steps (3.1e-6 -> 2.7e-6
        # or 3.3e-6 after 200 steps at 128^3, run to run: mode 5 is not bitwise reproducible) -- so its legs have to stay finite and within 2 x ||r0||.
        # Independent of convergence, the residual the solver reports must be the one recomputed outside it with the CSR kernel (one rank).
        strict = r["cfg"]["solver"] == "cg"
        # (round 5: 2 x ||r0|| instead of 100 x -- a solve that DIVERGES is not a measurement; whether the same system converges in a sane number of steps is
        # the `time_to_tol` object of the c3 legs)
        # (shorter fixed counts than the default 200 -- the functional runs of the test-suite -- sit inside BiCGStab's initial hump: 40 steps leave 2.3 x ||r0|| on c3)
        loose = 2.0 if args.iters >= 200 else 10.0
        ok = ir is not None and fr is not None and fr == fr and ir == ir and fr < float("inf") and (fr < ir if strict else fr < loose * ir)
        if not ok:
            raise SystemExit(f"bench.py: {what}: the last timed solve did not reduce the residual (initial {ir}, final {fr}) -- the run is invalid")
        rc = r.get("final_res_recomputed")
        if rc is not None and not (abs(rc - fr) <= 1e-5 * max(fr, rc) + 1e-9 * ir):  # (two kernels, two summation orders: round-off of K dx against a small residual)
            raise SystemExit(f"bench.py: {what}: the solver reports a final residual of {fr}, the CSR kernel on the caller's matrix gives {rc} for the "
                             f"solution it returned -- the run is invalid")

    def secondary_object(t, title, wkey, baseline_config):
        tv = t["n_global"] * t["updates"] / t["elapsed"]
        sms = t["solve_ms"] / t["steps"]
        return {"workload": title, "baseline_config": baseline_config, "value": tv, "unit": "DOF-updates/s", "n_dof": t["n_global"], "nnz": t["nnz"],
                "steps": t["steps"], "ms_per_step": t["elapsed"] / t["steps"] * 1e3, "solve_ms_per_step": sms,
                "assembly_ms_per_step": t["elapsed"] / t["steps"] * 1e3 - sms, "krylov_steps_per_step": t["updates"] / t["steps"],
                "initial_res": t["initial_res"], "final_res": t["final_res"], "final_res_recomputed": t.get("final_res_recomputed"),
                "roofline": solver_roofline(t, wkey), "csr_kernel": t.get("csr_kernel"),
                **({"remainder": t["remainder"]} if t.get("remainder") else {}),
                **({"time_to_tol": t["time_to_tol"]} if t.get("time_to_tol") else {})}

    if rank == 0:
        r = main_res
        per_step_updates = r["updates"] / r["steps"]
        value = r["n_global"] * r["updates"] / r["elapsed"]
        solve_ms_step = r["solve_ms"] / r["steps"]
        out = {
            "metric": cfg["metric"],
            "value": value,
            "unit": "DOF-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": r["elapsed"] / r["steps"] * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{cfg['title']}, {r['nx_global']}x{args.n}x{args.n} structured mesh (make_Brick)"
                            + (f" cut into {world} slabs along i (strong scaling)" if strong else "") + ": fused assembly (K + R) + "
                            + (f"{args.iters} Jacobi-CG iterations per step" if cfg["solver"] == "cg" else
                               f"{per_step_updates:.0f} {SOLVER_TEXT[cfg['solver']]} per step")
                            + (", temperature fixed on x = 0 by the reference's Nitsche form (2D_Script.jl:58: nonsymmetric K)" if cfg.get("nitsche") else ""),
                "baseline_config": {"c2": "the north_star target size (512^3 hex-8, 1 GPU); configs[1] (256^3) is `secondary_256`",
                                    "c3": "configs[2]", "c4": "configs[3]", "ref_idrs8": "configs[1] solved with the reference's default solver",
                                    "nitsche_c2": "configs[1] mesh and form, Dirichlet face added", "nitsche_c4": "configs[3] mesh and form, Dirichlet face added"
                                    }[args.config] if args.n == cfg["n"] else f"{args.config} at a non-default size",
                "n_dof": r["n_global"], "nnz_per_gpu": r["nnz"], "krylov_steps_per_step": per_step_updates,
                "parallelism": "single GPU" if world == 1 else f"slab decomposition x{world} ("
                               + (state["transport"] or ("host callbacks over gloo, ranks sharing one GPU: functional run" if host_comm else "RCCL"))
                               + " halo overlapped with the interior rows + one all-reduce per reduction group)",
                "solve_ms_per_step": solve_ms_step,
                # the two halves of the metric on their own (whole job, all ranks)
                "assembly_ms_per_step": r["elapsed"] / r["steps"] * 1e3 - solve_ms_step,
                "assembly_dof_per_s": r["n_global"] / max(r["elapsed"] / r["steps"] - solve_ms_step * 1e-3, 1e-12),
                "solve_dof_updates_per_s": r["n_global"] * per_step_updates / (solve_ms_step * 1e-3),
                # the last timed solve (true residuals, ||.||_2 / sqrt(n)): the run fails unless final < initial
                "initial_res": r["initial_res"], "final_res": r["final_res"],
                # the same residual recomputed after the timed region with mul! (the CSR kernel on the caller's arrays): must agree to 1e-5 relative
                "final_res_recomputed": r.get("final_res_recomputed"),
                "first_step_s": r["first_step_s"], "first_step_split": r.get("first_step_split"),
                "workspace_placement_trial": ("on (--ws-trial 1: the first solve times the SpMV on up to three allocations of the workspace and keeps the "
                                              "fastest; its cost is inside first_step_s, outside the timed region)" if args.ws_trial else
                                              "off (library default since round 4)"),
            },
            "roofline": solver_roofline(r, f"{args.config}_{args.n}"),
        }
        if r["comm_exposed"] is not None:
            # what the solver stream of every rank waited for (hip-event pairs around the wait for the halo stream and around each all-reduce)
            out["comm_exposed"] = r["comm_exposed"]
        check_residual(r, f"{args.config} {args.n}^3")
        if r.get("remainder"):
            out["remainder"] = r["remainder"]
        if "csr_kernel" in r:
            # the north_star's own number: the CSR kernel behind mul! on this matrix, measured in this run
            out["roofline"]["csr_kernel"] = r["csr_kernel"]
    strong_res = None
    if world > 1 and not strong and args.strong_leg:
        # the same config as ONE mesh of its own size cut into `world` slabs (what --scaling strong times), a few steps: every rank takes part.
        # The weak measurement above is what the driver asked for and is COMPLETE at this point: a watchdog on every rank makes sure a problem in this extra
        # leg (it builds a second communicator) can only cost the extra object, never the line -- past the deadline rank 0 prints the line it has and
        # all ranks leave.
        import threading

        def bail():
            if rank == 0:
                out["strong_scaling"] = {"error": f"the strong-scaling leg did not finish within {args.strong_leg_timeout} s; the line above is the complete weak measurement"}
                out["cpu_baseline"] = None
                sys.stdout.write(json.dumps(out) + "\n")
                sys.stdout.flush()
            os._exit(0)

        watchdog = threading.Timer(args.strong_leg_timeout, bail)
        watchdog.daemon = True
        watchdog.start()
        try:
            strong_res = run_workload(cfg, REF_LEGS[args.config]["base"] if args.config in REF_LEGS else args.config, args.n, args.secondary_steps, 1,
                                      want_csr=False, strong=True)
        except Exception as e:  # (an error on one rank usually shows on all of them: every rank reports and the line still goes out)
            strong_res = None
            if rank == 0:
                out["strong_scaling"] = {"error": repr(e)}
        finally:
            watchdog.cancel()
        if rank == 0 and strong_res is not None:
            try:
                check_residual(strong_res, f"{args.config} {args.n}^3 strong")
            except SystemExit as e:  # (the extra leg must not take the line down)
                out["strong_scaling"] = {"error": str(e)}
                strong_res = None
        if rank == 0 and strong_res is not None:
            t = strong_res
            out["strong_scaling"] = {
                "workload": f"{cfg['title']}, THE {args.n}^3 mesh cut into {world} slabs along i, {t['steps']} timed steps after 1 warm-up, same step as above",
                "scaling": "strong", "value": t["n_global"] * t["updates"] / t["elapsed"], "unit": "DOF-updates/s", "n_dof": t["n_global"],
                "ms_per_step": t["elapsed"] / t["steps"] * 1e3, "solve_ms_per_step": t["solve_ms"] / t["steps"],
                "initial_res": t["initial_res"], "final_res": t["final_res"], "comm_exposed": t["comm_exposed"],
                "note": "speed-up over one GPU = this value / the N = 1 line's value on the same mesh (the driver computes it from its own runs)"}
    if world == 1 and args.config == "c2" and args.secondary_n > 0 and args.secondary_n != args.n:
        # configs[1] of BASELINE.json (256^3 hex-8; also the matrix the round-1 / round-2 lines were quoted on)
        t = run_workload(cfg, "c2", args.secondary_n, args.secondary_steps, 1, want_csr=True)
        check_residual(t, f"c2 {args.secondary_n}^3")
        out[f"secondary_{args.secondary_n}"] = secondary_object(
            t, f"{cfg['title']}, {args.secondary_n}^3, {t['steps']} timed steps after 1 warm-up, same step as above", f"c2_{args.secondary_n}",
            "configs[1]")
    if world == 1 and args.config == "c2" and args.secondary_configs and (args.n == cfg["n"] or args.secondary_config_n > 0):
        # configs[2] and configs[3] of BASELINE.json in the same invocation (what `--config c3` / `--config c4` run, fewer steps): the driver's one
        # line then carries all four configs
        for ck in ("c3", "c4"):
            c = dict(CONFIGS[ck])
            c["time_to_tol"] = ck == "c3"
            if args.secondary_config_n > 0:
                c["n"] = args.secondary_config_n
            t = run_workload(c, ck, c["n"], args.secondary_steps, 1, want_csr=True)
            check_residual(t, f"{ck} {c['n']}^3")
            per = t["updates"] / t["steps"]
            out[f"secondary_{ck}"] = secondary_object(
                t, f"{c['title']}, {c['n']}^3, {t['steps']} timed steps after 1 warm-up: fused assembly (K + R) + "
                   + (f"{args.iters} Jacobi-CG iterations" if c["solver"] == "cg" else f"{per:.0f} SpMV-equivalent steps of bicgstabl_GS! (s = 2, right Jacobi)")
                   + " per step", f"{ck}_{c['n']}", {"c3": "configs[2]", "c4": "configs[3]"}[ck])
            out[f"secondary_{ck}"]["metric"] = c["metric"]
        out["roofline_hex27_ke"] = hex27_ke_roofline(args.secondary_config_n or CONFIGS["c4"]["n"])
    if world == 1 and args.config == "c2" and args.ref_legs and (args.n == cfg["n"] or args.secondary_config_n > 0):
        # the reference's own solver / boundary-condition path at scale (REF_LEGS above)
        for lk, leg in REF_LEGS.items():
            c = dict(CONFIGS[leg["base"]], solver=leg["solver"], nitsche=leg["nitsche"], n=args.secondary_config_n or leg["n"])
            t = run_workload(c, leg["base"], c["n"], args.secondary_steps, 1, want_csr=leg["nitsche"])
            check_residual(t, f"{lk} {c['n']}^3")
            lk = f"{lk}_{leg['n']}"  # (the key names the leg at its own size: ref_idrs8_256, nitsche_c2_256, nitsche_c4_128)
            per = t["updates"] / t["steps"]
            obj = secondary_object(
                t, f"{c['title']}, {c['n']}^3" + (", temperature fixed on x = 0 by the reference's Nitsche form (h_penalty*Bilinear(T, Tw - T) + k*Bilinear(T, n{i}*T{;i}), "
                                                   "2D_Script.jl:58: NONSYMMETRIC K), convection on the other faces" if leg["nitsche"] else "")
                   + f", {t['steps']} timed steps after 1 warm-up: fused assembly (K + R) + {per:.0f} {SOLVER_TEXT[leg['solver']]} per step",
                f"{lk[:lk.rfind('_')]}_{c['n']}", {"c2": "configs[1]", "c4": "configs[3]"}[leg["base"]] + (" mesh and form, Dirichlet face added" if leg["nitsche"] else " solved with the reference's default solver"))
            obj["metric"] = f"DOF-updates/sec (assembly + {SOLVER_TEXT[leg['solver']]})"
            out[lk] = obj
    if rank == 0 and world == 1 and args.config == "c2" and args.hex27_n > 0:
        # the CSR kernel on the other matrix shape of the configs: hex-27 (27..125 entries per row), configs[3]'s size
        try:
            b27 = mf.Brick((1.0, 1.0, 1.0), (args.hex27_n,) * 3, 2, 5, ctx=ctx)
            A27 = b27.pattern(1)
            K27 = b27.assemble_thermal(A27, K_COND, H, TENV, mf.ALL_FACES)
            out["roofline"]["csr_kernel_hex27"] = dict(csr_kernel_roofline(A27, K27, f"c4_{args.hex27_n}"), matrix=f"hex-27 thermal {args.hex27_n}^3")
            del b27, A27, K27
            torch.cuda.empty_cache()
        except Exception as e:  # never lose the line over a side measurement
            out["roofline"]["csr_kernel_hex27"] = {"error": repr(e)}
    if rank == 0 and world == 1 and args.live_traffic and args.n == cfg["n"]:
        # the counters of the headline workload's two priced kernels, observed in THIS run (the parent's GPU work is over and its
        # memory released; the children set up the same matrix themselves)
        leg = f"{args.config}_{args.n}"
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        live, why = collect_traffic_live(leg)
        rf = out["roofline"]
        for obj, key in ((rf, f"{main_res['kernel_key']}@{leg}"), (rf.get("csr_kernel"), f"csr_kernel@{leg}")):
            if obj is None:
                continue
            ent = live.get(key) if live else None
            if ent and ent.get("hbm_bytes_per_launch"):
                tb = ent["hbm_bytes_per_launch"]
                obj["traffic_committed_file"] = obj.get("traffic")
                obj["traffic"] = tb
                obj["traffic_over_algorithmic"] = tb / obj["algorithmic_bytes_per_launch"]
                obj["frac_actual"] = tb / (obj["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
                cal = ent.get("calibration") or {}
                obj["traffic_source"] = (f"collected in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (two child passes of "
                                         f"tools/pmc_leg.py {leg}; {ent.get('launches_measured')} launches; FETCH_SIZE x 2 per MI355X_MICROARCH.md; in-run "
                                         f"calibration on k_axpby: read {cal.get('FETCH_SIZE_x2_bytes', 0) / max(cal.get('expected_read_bytes', 1), 1):.4f} x, "
                                         f"written {cal.get('WRITE_SIZE_bytes', 0) / max(cal.get('expected_write_bytes', 1), 1):.4f} x of the known bytes)")
            elif obj.get("traffic_source"):
                obj["traffic_source"] += f"; live collection not available ({why or 'kernel missing in the counter output'})"
    if rank == 0:
        if world == 1 and args.config == "c2" and args.cpu_n > 0:
            cb = cpu_baseline(args.cpu_n, args.iters, args.cpu_repeats)
            out["cpu_baseline"] = cb
            sk = f"secondary_{args.secondary_n}"
            out["vs_cpu_baseline"] = {"main_workload": out["value"] / cb["value"],
                                      **({sk: out[sk]["value"] / cb["value"]} if sk in out else {}),
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
