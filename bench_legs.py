"""bench_legs.py -- the workloads bench.py times.  One *step* of a leg = what update_OneStep! (solver/04_Time_Domain.jl:59-80)
runs for one Newton step with a capped solver: K_linear_func (fused K) + K_nonlinear_func (matrix-free R at x* = 0) + a fixed number of Krylov steps.
Imported by bench.py only after the ranks exist (it imports torch and the product library)."""
from __future__ import annotations

import ctypes as C
import json
import os
import sys
import time

import torch

import metafem_jl_amd as mf
from bench_report import FP64_MFMA_PEAK_TFLOPS, HBM_PEAK_GBS, KERNEL_NOTES
from metafem_jl_amd import _lib, parallel

ROOT = os.path.dirname(os.path.abspath(__file__))

K_COND, H, TENV, SRC = 0.6, 25.0, 293.15, 1600.0  # examples/thermal_conduction/3D_Script.jl:21-25,56
E_MOD, NU = 1.0, 0.3                               # examples/linear_elasticity/cantilever/3D_Script.jl:52-63 (E = 1, nu = 0.3)
LAM, MU = E_MOD * NU / ((1 + NU) * (1 - 2 * NU)), E_MOD / (2 * (1 + NU))
TAU = 1000.0 * E_MOD
H_PEN, TW = 1000.0, 1173.15                       # examples/thermal_conduction/2D_Script.jl:46-47: h_penalty, Tw of the weakly imposed Dirichlet face

# The reference's OWN solver / boundary-condition path at scale (round 5): legs of the default line, a few timed steps each
#   ref_idrs8   configs[1]'s mesh and form solved the way every example script does: idrs!(s = 8) (src/MetaFEM.jl:36-37, 04_IDRs.jl:26-95) + Pr_Jacobi!
#   nitsche_c2  the same mesh with T FIXED on x = 0 the reference's way (thermal_conduction/2D_Script.jl:58): K is NONSYMMETRIC; bicgstabl_GS!(2)
#   nitsche_c4  configs[3]'s mesh (hex-27) with the same face, bicgstabl_GS!(2)
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
TRAFFIC_FILES = ("profiles/r06_traffic.json", "profiles/r05_traffic.json")


def config_of(name):
    """CONFIGS entry of a --config value (a BASELINE config or one of the reference-path legs alone) and the key of its base config."""
    if name in REF_LEGS:
        leg = REF_LEGS[name]
        return dict(CONFIGS[leg["base"]], solver=leg["solver"], nitsche=leg["nitsche"], n=leg["n"],
                    metric=f"DOF-updates/sec (assembly + {SOLVER_TEXT[leg['solver']]}) on " + CONFIGS[leg["base"]]["title"]
                           + (" with a Nitsche-Dirichlet face (nonsymmetric K)" if leg["nitsche"] else "")), leg["base"]
    return CONFIGS[name], name


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
    from oracle import cport  # (test infrastructure: the only place bench.py touches oracle/)

    cport.lib().orc_set_num_threads(ncpu)  # the OpenMP runtime may already be initialised (torch / numpy import it)
    prob = cport.CThermal((n_cpu, n_cpu, n_cpu), k=K_COND, h=H, Tenv=TENV, src=SRC).setup()
    prob.timed_step(2)  # warm caches / page in
    times = sorted((prob.timed_step(iters) for _ in range(max(repeats, 1))), key=lambda t: t[0] + t[1])
    ta, ts = times[len(times) // 2]
    cores = cport.lib().orc_num_threads()
    return {
        "value": prob.mesh.ncp * iters / (ta + ts), "unit": "DOF-updates/s", "cores": cores, "kind": "port",
        "sample_short": f"oracle/c (C/OpenMP port) hex-8 {n_cpu}^3 thermal, {prob.mesh.ncp} DOF: assembly {ta:.2f}s + {iters} CG {ts:.2f}s, median of {len(times)}",
        "sample": f"hex-8 {n_cpu}^3 thermal ({prob.mesh.ncp} DOF = {prob.mesh.ncp / 135005697:.4f} of the 512^3 workload, {prob.mesh.ncp / 16974593:.3f} of the 256^3 "
                  f"one), median of {len(times)} steps: 1 step = term-by-term assembly ({ta:.2f} s) + {iters} Jacobi-CG iterations ({ts:.2f} s) -- the GPU leg's "
                  f"iterations per assembly; C/OpenMP restatement of the reference algorithm and DATA LAYOUT (oracle/c/oracle.c: stored per-element basis tables, "
                  f"2 KB + 0.5 KB of slot ids per hex-8 element -- 512^3 would need 340 GB of host memory and 256^3 55 GB, so the sample stays at {n_cpu}^3 and "
                  f"the ratio is a per-DOF throughput ratio), {cores} threads (cgroup CPU quota of the box)",
        "all_step_seconds": [round(a + b, 3) for a, b in times],
    }


def load_traffic():
    """HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate passes, gfx950 x2 FETCH correction, calibrated in the same run) of
    this tree's kernels, keyed '<kernel key>@<workload key>' (tools/run_pmc_r03.sh); not collected in this run (the full object says so in `traffic_source`)."""
    for f in TRAFFIC_FILES:
        try:
            return json.load(open(os.path.join(ROOT, f))), f
        except Exception:
            continue
    return {}, None


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


class ResidualCheckFailed(RuntimeError):
    pass


class Bench:
    """Process-wide state of one bench.py run on one rank (context, communicator choice, counters file) + the legs."""

    def __init__(self, args):
        self.args = args
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={self.world}")
        # functional check of the N > 1 path on a 1-GPU box: all ranks on cuda:0, host-callback communicator over gloo (never a measurement)
        self.host_comm = os.environ.get("MFEM_BENCH_HOST_COMM") == "1"
        if self.host_comm:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist

            self.dist = dist
            if self.host_comm:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        self.ctx = mf.Context(local_rank)
        self.dev = f"cuda:{local_rank}"
        self.use_comm = self.world > 1 or os.environ.get("MFEM_BENCH_FORCE_COMM") == "1"  # the env var exercises the RCCL path at N = 1
        self.traffic_db, self.traffic_file = load_traffic()
        self.state = {"gloo_group": None, "transport": None}  # set when the RCCL transport had to be replaced (run_workload)
        self.strong = args.scaling == "strong" and self.world > 1

    # ---- helpers ---------------------------------------------------------------------------------------------------------------------------------
    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        torch.cuda.synchronize()

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()

    def traffic_of(self, key):
        t = self.traffic_db.get(key)
        if not t:
            return None, None
        return t.get("hbm_bytes_per_launch"), (f"{self.traffic_file}['{key}']: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel on this workload "
                                               f"(tree {t.get('tree', '?')}; gfx950 x2 FETCH_SIZE correction, calibrated in the same run), not collected in this run")

    def spmv_timer(self):
        """hip-event pairs around every SpMV on the context's stream (mfem_prof_spmv_*): returns a function -> (average ms, launches) since the call."""
        ctx = self.ctx
        tot, cnt = C.c_double(), C.c_int64()
        _lib.check(_lib.lib.mfem_prof_spmv_enable(ctx._h, 1))
        _lib.check(_lib.lib.mfem_prof_spmv_read(ctx._h, C.byref(tot), C.byref(cnt), 1))

        def read(disable=True):
            _lib.check(_lib.lib.mfem_prof_spmv_read(ctx._h, C.byref(tot), C.byref(cnt), 1))
            if disable:
                _lib.check(_lib.lib.mfem_prof_spmv_enable(ctx._h, 0))
            return tot.value / max(cnt.value, 1), int(cnt.value)
        return read

    def csr_kernel_roofline(self, A, K, wkey, launches=20):
        """The CSR kernel behind mul! (mfem_spmv_csr: caller's CSR arrays, no copy; 04_GPU_Utils.jl:131) on this matrix: live hip-event timing of `launches`
        launches.  `achieved` / `frac` price the launch with the bytes the kernel moves BY DESIGN (mfem_csr_spmv_bytes: values, the columns it reads -- tiles
        whose rows repeat one column-offset list read only their first rows' columns --, x, y, row pointers); `csr_equivalent` with SURVEY 8(d)'s formula
        (12 B per nonzero); `frac_actual` with the PMC-measured traffic."""
        x = mf.FEM_rand(A.ncols, 0x5EED, 0, ctx=self.ctx)
        y = torch.empty(A.n, dtype=torch.float64, device=self.dev)
        for _ in range(3):
            mf.mul_(y, A, K, x)
        read = self.spmv_timer()
        for _ in range(launches):
            mf.mul_(y, A, K, x)
        ms, n_l = read()
        formula = A.nnz * 12 + A.n * 16 + (A.n + 1) * 8  # val 8 + col 4 per nonzero; x, y 8 per row; i64 row pointers
        design, cols_read = A.spmv_bytes()
        gbs = lambda b: b / (ms * 1e-3) / 1e9
        traffic, src = self.traffic_of(f"csr_kernel@{wkey}")
        return {"kernel_key": "csr_kernel", "kernel_note": KERNEL_NOTES["csr_kernel"],
                "avg_launch_ms": ms, "launches": n_l, "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": design, "column_entries_read": cols_read,
                "achieved": gbs(design), "frac": gbs(design) / HBM_PEAK_GBS,
                "csr_equivalent": {"bytes_per_launch": formula, "achieved": gbs(formula), "frac": gbs(formula) / HBM_PEAK_GBS,
                                   "note": "SURVEY 8(d): nnz*12 + n*16 + (n+1)*8 -- what a CSR kernel that reads every column index moves"},
                "traffic": traffic, "traffic_source": src,
                "traffic_over_algorithmic": (traffic / design) if traffic else None,
                "frac_actual": (gbs(traffic) / HBM_PEAK_GBS) if traffic else None,
                "n": A.n, "nnz": A.nnz}

    # ---- hex-27 Ke on FP64 MFMA ------------------------------------------------------------------------------------------------------------------
    def hex27_ke_roofline(self, n27, repeats=5):
        """north_star: 'MFMA used only for the dense per-element Ke = B^T D B contraction on high-order hex elements ... evidenced by MFMA utilisation
        (hex-27 Ke) against gfx950 peak'.  The element kernels of the hex-27 matrix assembly (pass 1: geometry + FP64 MFMAs per element into a scratch of
        element matrices; pass 2: row-owner gather into CSR; no boundary faces) timed live with events on the stream they are launched on (the context runs
        on torch's current stream), priced with SURVEY 8(d)'s USEFUL flops: 2 * 27 * 27 * 81 = 118 098 per element.  The MFMA-pipe busy fraction of pass 1
        comes from the committed counter file (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128 SIMDs per XCD))."""
        ctx, dev = self.ctx, self.dev
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

        # The make_Brick mesh is affine element by element, and the library then assembles without ever storing Ke (k_hex27_direct).  The MFMA figure is
        # the GENERAL two-pass path's, forced here on the same mesh (bit 9 of mfem_debug_set_hex27); the default path's time is reported beside it.
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
        # ... and a mesh of GENERAL elements (the centre node of every element moved): the row-owner kernel computes the rows from per-element G_q
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
        busy = src = None
        key = "hex27_counters_MFMA@c4_%d" % n27
        for kname, cs in self.traffic_db.get(key, {}).get("per_kernel", {}).items():
            if "k_hex27<true" in kname and cs.get("SQ_VALU_MFMA_BUSY_CYCLES", {}).get("mean"):
                busy = cs["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / (cs["GRBM_GUI_ACTIVE"]["mean"] * 128.0)
                src = (f"{self.traffic_file}['{key}']['{kname}']: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128) of the pass-1 kernel, rocprofv3 --pmc pass "
                       f"(tree {self.traffic_db[key].get('tree', '?')}), not collected in this run")
        del b27, A27, K27
        torch.cuda.empty_cache()
        direct_bytes = 8 * nnz27 + 2 * 48 * nel + 3 * 8 * n27_rows
        return {"kernel": "k_hex27<true, true> (pass 1: sum-factorised geometry + Ke = B^T D B on __builtin_amdgcn_mfma_f64_16x16x4f64, element matrices to scratch) "
                          "+ k_hex27_gather_lds (pass 2: row-owner gather into sorted CSR); one matrix assembly without boundary faces",
                "matrix": f"hex-27 thermal {n27}^3 ({nel} elements)", "bound": "mfma", "unit": "TFLOP/s", "peak": FP64_MFMA_PEAK_TFLOPS,
                "useful_flop_per_assembly": flops, "avg_assembly_ms": ms, "assemblies_timed": repeats,
                "achieved": flops / (ms * 1e-3) / 1e12, "frac": flops / (ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                "mfma_pipe_busy_in_pass1": busy, "mfma_pipe_busy_source": src,
                "note": "useful flops only (SURVEY 8(d)): the B build and the zero-padding of the 27 x 81 operands to MFMA tiles are not counted; measured with the "
                        "scratch-free path of all-affine meshes turned off (this mesh would take it: `affine_mesh_assembly`)",
                "affine_mesh_assembly": {
                    "kernel": "k_hex27_affine_g0 + k_hex27_direct (row-owner gather, each (row, element) run computed from G0 and the 1-D reference integrals in "
                              "registers): the default when every element is affine; Ke is never stored",
                    "avg_assembly_ms": ms_direct, "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "algorithmic_bytes_per_assembly": direct_bytes,
                    "achieved": direct_bytes / (ms_direct * 1e-3) / 1e9, "frac": direct_bytes / (ms_direct * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "max_rel_difference_to_the_mfma_path": agree,
                    "note": "algorithmic bytes = the CSR values written once + G0 written and read + the coordinates read"},
                "general_mesh_assembly": {
                    "kernel": "k_hex27_affine_g0 + k_hex27_gq_lane (G_q at the 27 Gauss points, one lane per element) + k_hex27_rows_gq (row owners compute every "
                              "(row, element) run from G_q by sum factorisation in registers): the default from 30 % non-affine elements on; Ke is never stored",
                    "mesh": "the same mesh with the centre node of EVERY element moved (all elements non-affine)",
                    "avg_assembly_ms": ms_rows, "two_pass_mfma_path_same_mesh_ms": ms_two_pass_general,
                    "useful_flop_per_assembly": flops, "achieved_on_useful_flops_tflops": flops / (ms_rows * 1e-3) / 1e12,
                    "frac_of_fp64_peak_on_useful_flops": flops / (ms_rows * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                    "executed_flop_per_assembly": 2.0 * 1053 * 27 * nel, "max_rel_difference_to_the_mfma_path": agree_general}}

    # ---- communicator ----------------------------------------------------------------------------------------------------------------------------
    def make_comm(self, brick, F, order, m1, m2):
        ctx, rank, world, dist, dev, state = self.ctx, self.rank, self.world, self.dist, self.dev, self.state
        # RCCL prints a version banner / warnings through C stdio on stdout: send them to stderr so that stdout carries only the JSON line
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        comm = None
        try:
            if self.host_comm:
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
                    # fall back to the host-callback transport (device -> pinned host -> gloo -> device): the same solver code path, a slower exchange
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
        return comm

    # ---- one workload ----------------------------------------------------------------------------------------------------------------------------
    def run_workload(self, cfg, ckey, N, steps, warmup, want_csr, strong=None, iters=None):
        """One workload of config `ckey`, `steps` timed steps: weak scaling = an (N * world) x N x N mesh (an N-thick slab per rank), strong scaling =
        the N^3 mesh cut into `world` slabs along i."""
        args, ctx, dev, world, rank, dist, use_comm, host_comm, state = (self.args, self.ctx, self.dev, self.world, self.rank, self.dist, self.use_comm,
                                                                         self.host_comm, self.state)
        strong = self.strong if strong is None else strong
        iters = args.iters if iters is None else iters
        order, F = cfg["order"], cfg["fields"]
        nx_global = N if strong else N * world
        brick = mf.Brick((1.0 if strong else float(world), 1.0, 1.0), (nx_global, N, N), order, cfg["itg"], ctx=ctx)
        m0, m1, m2 = brick.m
        comm = None
        if use_comm:
            plo, phi = parallel.slab_planes(m0, world, rank, order)
            brick.set_slab(plo, phi)
            comm = self.make_comm(brick, F, order, m1, m2)
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
                return mf.iterative_Solve(A, K, R, 1e-300, Sv_func=mf.cg_, Pr_func=mf.Pr_Jacobi_, maxiter=iters, max_pass=1, fixed_iterations=True)
        elif cfg["solver"] == "idrs8":
            # idrs! with s = 8 (04_IDRs.jl:26-95): every inner step is one SpMV and advances `iter` by one
            def solve():
                return mf.iterative_Solve(A, K, R, 1e-300, Sv_func=mf.idrs_, Pr_func=mf.Pr_Jacobi_, maxiter=iters, max_pass=1, s=8, fixed_iterations=True)
        else:
            # bicgstabl_GS! with s = 2 (03_BiCGstabl.jl:18-96): one sweep = 4 SpMVs and advances the solver's `iter` by s = 2 (:93), so maxiter = iters / 2
            # gives `iters` SpMV-equivalent steps per solve
            def solve():
                return mf.iterative_Solve(A, K, R, 1e-300, Sv_func=mf.bicgstabl_GS_, Pr_func=mf.Pr_Jacobi_, maxiter=max(iters // 2, 2), max_pass=1, s=2,
                                          fixed_iterations=True)

        def step():
            assemble()
            return solve()

        t_first = time.perf_counter()
        first_split = None
        for w in range(warmup):
            if w == 0:
                # the first step, split: assembly (tables, first launches) | the first solve = one-off plans of the solver layouts + workspace allocation
                # (+ the placement trial if opted in) + the solve itself
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
        read_spmv = self.spmv_timer()
        if comm is not None:
            _lib.check(_lib.lib.mfem_prof_comm_enable(ctx._h, 1))
            hw, hn, aw, an = C.c_double(), C.c_int64(), C.c_double(), C.c_int64()
            _lib.check(_lib.lib.mfem_prof_comm_read(ctx._h, C.byref(hw), C.byref(hn), C.byref(aw), C.byref(an), 1))
        sym_count0 = int(_lib.lib.mfem_debug_sym_spmv_count())
        lat_count0 = int(_lib.lib.mfem_debug_lat27_spmv_count())
        lat8_count0 = int(_lib.lib.mfem_debug_lat8_spmv_count())
        rem_count0 = int(_lib.lib.mfem_debug_rem_spmv_count())
        self.barrier()
        t0 = time.perf_counter()
        solve_ms = 0.0
        iters_done = spmvs_done = 0
        st = dx_last = None
        for _ in range(steps):
            dx_last, st = step()
            solve_ms += st.solve_ms
            iters_done += st.iterations
            spmvs_done += st.spmv_count
        self.barrier()
        elapsed = time.perf_counter() - t0
        spmv_ms, spmv_launches = read_spmv()
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
        # one "DOF update" = one Krylov step on one unknown: a CG iteration (1 SpMV), or one SpMV-equivalent step of BiCGStab(2) / IDR(s)
        updates = iters_done if cfg["solver"] == "cg" else spmvs_done
        res = {"N": N, "nx_global": nx_global, "n_global": n_global, "nnz": A.nnz, "n_local": A.n, "elapsed": elapsed, "iters": iters,
               "steps": steps, "solve_ms": solve_ms, "iters_done": iters_done, "spmvs_done": spmvs_done, "updates": updates,
               "spmv_ms": spmv_ms, "spmv_launches": spmv_launches,
               # the LAST timed solve, as the library reports it (true residual ||b - A x|| / sqrt(n) recomputed after the pass, 02_Preconditioner.jl:53-55)
               "initial_res": st.initial_res if st else None, "final_res": st.final_res if st else None,
               "first_step_s": t_first if warmup > 0 else None, "first_step_split": first_split, "comm_exposed": comm_exposed, "cfg": cfg, "ckey": ckey}
        # A = S + N (csrc/spmv_rem.hip): did the timed solves run on symmetric lattice tiles + the sparse skew remainder of a nonsymmetric K, and how big was it
        rr_, re_, ra_ = C.c_int64(), C.c_int64(), C.c_double()
        _lib.check(_lib.lib.mfem_debug_remainder_info(A._h, C.byref(rr_), C.byref(re_), C.byref(ra_)))
        res["remainder"] = ({"rows": rr_.value, "entries": re_.value, "fraction_of_rows": rr_.value / max(A.n, 1),
                             "asymmetry_measured_on_the_tiles_alone": ra_.value}
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
        if cfg.get("newton_like") and world == 1 and not use_comm:
            # per-solve cost for Newton-sized solves: assembly + a SHORT solve (20 iterations): the per-solve layout copy, the Jacobi
            # diagonal and the symmetry probe -- what the reference pays as its K_total[K_val_ids] gather per solve (02_Preconditioner.jl:35) -- are a large
            # share of such a step (04_Time_Domain.jl:66-78: one solve per Newton iteration)
            short = int(cfg["newton_like"])
            saved_iters = iters
            iters = short  # (the closures read `iters`)
            step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            sm = 0.0
            for _ in range(5):
                _, st_n = step()
                sm += st_n.solve_ms
            torch.cuda.synchronize()
            el = (time.perf_counter() - t1) / 5
            iters = saved_iters
            res["newton_like"] = {"iterations": short, "ms_per_step": el * 1e3, "solve_ms": sm / 5, "assembly_ms": el * 1e3 - sm / 5,
                                  "value": n_global * short / el, "unit": "DOF-updates/s", "n_dof": n_global,
                                  # per-solve work = a short solve minus its share of a long one
                                  "per_solve_ms": sm / 5 - short * (solve_ms / max(updates, 1))}
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
                assert iters_done == iters * steps, (iters_done, iters, steps)
            self.describe_layout(res, A, F, lat_count0, lat8_count0, sym_count0)
            if want_csr and world == 1:
                res["csr_kernel"] = self.csr_kernel_roofline(A, K, f"{ckey}_{N}")
        if comm is not None:
            comm.close()
        del brick, A, K, x_star, s, R
        torch.cuda.empty_cache()
        return res

    def describe_layout(self, res, A, F, lat_count0, lat8_count0, sym_count0):
        """Which solver layout the timed Krylov loop ran on (modes 0-5) and the bytes one SpMV on it moves by design."""
        ctx, world, cfg = self.ctx, self.world, res["cfg"]
        csr_bytes = A.nnz * 12 + A.n * 16 + (A.n + 1) * 8
        mode, slots, npad, reg = C.c_int32(), C.c_int32(), C.c_int64(), C.c_int64()
        query = lambda: _lib.check(_lib.lib.mfem_csr_solver_layout(ctx._h, A._h, C.byref(mode), C.byref(slots), C.byref(npad), C.byref(reg)))
        query()
        lat_used = int(_lib.lib.mfem_debug_lat27_spmv_count()) > lat_count0
        if mode.value == 4 and not lat_used:
            # the structure allows the symmetric lattice tiles, but the values of these solves did not pass the symmetry measure: the sliced layout ran
            _lib.lib.mfem_debug_set_lat27(0)
            query()
        lat8_used = int(_lib.lib.mfem_debug_lat8_spmv_count()) > lat8_count0
        lat8_one_field = False
        if mode.value == 5 and not lat8_used:  # (the same for the 3-field lattice tiles: the diagonal-slotted layout ran)
            _lib.lib.mfem_debug_set_lat8(0)
            query()
        elif mode.value != 5 and lat8_used:
            # one field: the layout query answers for cg! (which keeps the bitwise patch sweep); idrs! / bicgstabl_GS! work on A D^-1 and ran on the tiles
            lat8_one_field = True
            _lib.lib.mfem_debug_set_lat8(3)
            query()
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
            kkey = "k_spmv_dia"
            plain = A.nnz * 8 + A.n * 16 + max(A.n - reg.value, 0) * slots.value * 4
            if sym_used:
                kkey, plain_bytes = ("k_spmv_symp" if symf.value == 2 else "k_spmv_sym27"), plain
            else:
                spmv_bytes = plain
        elif mode.value == 1:
            kkey = "k_spmv_ell"
        elif mode.value == 4:
            kkey = "k_spmv_lat27"
            if cfg["solver"] == "cg" and world == 1 and lat_used and _lib.lib.mfem_debug_lat27_cg_fused():
                # one rank: the CG iteration runs pass 2 inside its residual update (k_lat27_gather_cg); the SpMV launch the library times is pass 1
                kkey = "k_spmv_lat27_pass1"
                spmv_bytes = int(_lib.lib.mfem_debug_lat27_pass1_bytes(A._h))
        elif mode.value == 5:
            kkey = "k_spmv_lat8"
        elif mode.value == 3:
            kkey = "k_spmv_bsell" if int(_lib.lib.mfem_debug_bsell_fields(A._h)) > 0 else "k_spmv_sell"
        else:
            kkey = "csr_kernel"
            spmv_bytes = A.spmv_bytes()[0]
        note = KERNEL_NOTES.get(kkey, kkey).replace("<F>", f"<{F}>")
        name = kkey
        if res["remainder"] and mode.value in (4, 5):
            note += " " + KERNEL_NOTES["remainder"]
            name += "+k_rem_apply"
        res.update(kernel=note, kernel_key=kkey, kernel_name=name, spmv_bytes=spmv_bytes, csr_bytes=csr_bytes, mode=mode.value, sym_used=sym_used,
                   plain_bytes=plain_bytes)

    # ---- the unstructured path: what mesh_Classical(...; itp_type = :Serendipity, itp_order = 2) users get -----------------------------------------------
    def unstructured_mesh(self, n, block=512, seed=0x5EED, shape="CUBE"):
        """An n^3 brick of serendipity hex-20 elements as an UNSTRUCTURED mesh: vertices + 8-vertex connectivity (make_Brick) -> control points the way
        mesh_Classical numbers them (vertices first, then one node per unique edge: 3_InitializeMesh.jl:70-163) -> element order shuffled in blocks of
        `block` elements -- no lattice structure is left for the library to find.  Host side (numpy), cached for the two field counts."""
        import numpy as np

        from metafem_jl_amd import element, mesh as pm

        key = (n, block, seed, shape)
        if getattr(self, "_umesh_key", None) != key:
            # examples/thermal_conduction/3D_Script.jl:39-40: itp_order = 2, itg_order = 5; shape = "SIMPLEX": the brick cut into tetrahedra -- the 10-node
            # tetrahedron of the reference's simplex meshes (read_Mesh of .inp / .mphtxt files, e.g. examples/*/3D_Script.jl with the pikachu mesh)
            space = element.classical_space(3, "Serendipity", 2, 5, shape=shape) if shape != "CUBE" else element.classical_space(3, "Serendipity", 2, 5)
            vert, conn = pm.make_Brick((1.0, 1.0, 1.0), (n, n, n), shape=shape) if shape != "CUBE" else pm.make_Brick((1.0, 1.0, 1.0), (n, n, n))
            nel = conn.shape[1]
            nb = (nel + block - 1) // block
            perm = (np.random.default_rng(seed).permutation(nb)[:, None] * block + np.arange(block)[None, :]).ravel()
            perm = perm[perm < nel]
            msh = pm.mesh_Classical(vert, conn[:, perm], space)
            fac = pm.get_BoundaryMesh(msh)
            self._umesh_key, self._umesh = key, (space, msh, fac)
        return self._umesh

    def unstructured_leg(self, n, fields, steps, iters=None, shape="CUBE"):
        """One step on the unstructured hex-20 mesh = K_linear_func (constant-coefficient terms: mfem_mesh_assemble_elements_rows + _facets on the pattern of
        mfem_pattern_build) + K_nonlinear_func (generic S3 operators: residual at x* = 0) + `iters` steps of idrs!(s = 8) with Pr_Jacobi! -- the solver and
        preconditioner every example script selects (3D_Script.jl:49).  fields = 1: the thermal form of thermal_conduction/3D_Script.jl:29-32;
        fields = 3: linear elasticity with a penalty wall on x = 0 and a traction on y = 1 (cantilever/3D_Script.jl:52-63)."""
        import numpy as np

        from metafem_jl_amd import generic as G, physics

        args, ctx, dev = self.args, self.ctx, self.dev
        iters = args.iters if iters is None else iters
        t_host = time.perf_counter()
        space, msh, fac = self.unstructured_mesh(n, shape=shape)
        t_host = time.perf_counter() - t_host
        tag = "u20" if shape == "CUBE" else "tet10"
        t_setup = time.perf_counter()
        if fields == 1:
            wf = physics.thermal_domain(3, K_COND)
            bnd = [(fac.element_ID, fac.element_eindex, physics.thermal_convection(H, TENV))]
        else:
            wf = physics.elasticity_domain(3, LAM, MU)
            c = fac.centroid
            wall, top = fac.select(np.abs(c[:, 0]) < 1e-9), fac.select(np.abs(c[:, 1] - 1.0) < 1e-9)
            bnd = [(wall.element_ID, wall.element_eindex, physics.penalty([0, 1, 2], TAU)),
                   (top.element_ID, top.element_eindex, physics.traction(3, "sl", rows=[1]))]
        gd = G.GenericDomain(ctx, space, msh.coords, msh.cp_ids, fields, wf, bnd)
        if fields == 1:
            gd.controlpoints["s"] = torch.full((msh.ncp,), SRC, dtype=torch.float64, device=dev)
        else:
            for v in (2, 4, 6):  # Voigt ids of row 1 of the nodal traction tensor: sigma_22 = 1
                gd.controlpoints[f"sl{v}"] = torch.full((msh.ncp,), 1.0 if v == 2 else 0.0, dtype=torch.float64, device=dev)
        A = gd.A
        torch.cuda.synchronize()
        t_setup = time.perf_counter() - t_setup

        def solve():
            return mf.iterative_Solve(A, gd.K_total, gd.residue, 1e-300, Sv_func=mf.idrs_, Pr_func=mf.Pr_Jacobi_, maxiter=iters, max_pass=1, s=8,
                                      fixed_iterations=True)

        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]

        def step(timed=False):
            if timed:
                ev[0].record()
            gd.K_linear_func()
            if timed:
                ev[1].record()
            gd.K_nonlinear_func()
            if timed:
                ev[2].record()
            return solve()

        rows_before = int(_lib.lib.mfem_debug_mesh_rows_count()) if hasattr(_lib.lib, "mfem_debug_mesh_rows_count") else None
        step()
        torch.cuda.synchronize()
        lat_count0 = int(_lib.lib.mfem_debug_lat27_spmv_count())
        lat8_count0 = int(_lib.lib.mfem_debug_lat8_spmv_count())
        sym_count0 = int(_lib.lib.mfem_debug_sym_spmv_count())
        read_spmv = self.spmv_timer()
        k_ms = r_ms = solve_ms = 0.0
        spmvs = 0
        st = dx = None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            dx, st = step(timed=True)
            solve_ms += st.solve_ms
            spmvs += st.spmv_count
            torch.cuda.synchronize()
            k_ms += ev[0].elapsed_time(ev[1])
            r_ms += ev[1].elapsed_time(ev[2])
        elapsed = time.perf_counter() - t0
        spmv_ms, spmv_launches = read_spmv()
        n_dof = fields * msh.ncp
        res = {"cfg": {"solver": "idrs8"}, "remainder": None, "iters": iters, "initial_res": st.initial_res, "final_res": st.final_res,
               "spmv_ms": spmv_ms, "spmv_launches": spmv_launches}
        rr = torch.empty_like(gd.residue)
        mf.mul_(rr, A, gd.K_total, dx)
        rr.sub_(gd.residue)
        res["final_res_recomputed"] = mf.normalized_norm(rr, ctx=ctx)
        del rr, dx
        # (idrs! on the penalty-constrained hex-20 elasticity operator starts with a hump of one to two decades: ||r|| after a fixed count says little here --
        # finite, and within 100 x ||r0||; whether the same system CONVERGES is tests/test_gpu_u20.py)
        self.check_residual(res, f"{tag} {n}^3 x {fields}", loose=100.0)
        self.describe_layout(res, A, fields, lat_count0, lat8_count0, sym_count0)
        csr = self.csr_kernel_roofline(A, gd.K_total, f"{tag}_{fields}_{n}")
        rf = self.solver_roofline(res, f"{tag}_{fields}_{n}")
        # SURVEY 8(d), assembly (matrix): each CSR value written once + connectivity read + coordinates read
        asm_bytes = A.nnz * 8 + msh.cp_ids.size * 4 + msh.ncp * 3 * 8
        k_ms, r_ms = k_ms / steps, r_ms / steps
        rows_ran = None if rows_before is None else int(_lib.lib.mfem_debug_mesh_rows_count()) > rows_before
        out = {"workload": ("UNSTRUCTURED serendipity hex-20 mesh, " if shape == "CUBE" else "UNSTRUCTURED 10-node tetrahedron mesh (an n^3 brick cut into tetrahedra), ") +
                           f"{n}^3 {'elements' if shape == 'CUBE' else 'cells'} ({msh.nel} elements, {msh.ncp} control points numbered by mesh_Classical, element "
                           f"order shuffled in blocks of 512), {fields} field(s): " + ("thermal conduction + convection faces" if fields == 1 else
                           "linear elasticity, penalty wall on x = 0, traction on y = 1") + f"; step = K_linear_func (fused) + K_nonlinear_func (S3 operators) + "
                           f"{iters} SpMV-equivalent steps of idrs!(8) with Pr_Jacobi!; {steps} timed steps after 1 warm-up",
               "baseline_config": "the path every shipped example takes (itp_type = :Serendipity, itp_order = 2; idrs!(s = 8))",
               "value": n_dof * spmvs / elapsed, "unit": "DOF-updates/s", "n_dof": n_dof, "nnz": A.nnz, "steps": steps, "ms_per_step": elapsed / steps * 1e3,
               "solve_ms_per_step": solve_ms / steps, "assembly_ms": k_ms, "residual_ms": r_ms,
               "assembly_bytes": asm_bytes, "assembly_frac": asm_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
               "assembly_ns_per_element": k_ms * 1e6 / msh.nel, "row_owner_assembly_ran": rows_ran,
               "max_row_nnz": int(A.nnz / max(A.n, 1) + 0.5), "initial_res": st.initial_res, "final_res": st.final_res,
               "final_res_recomputed": res["final_res_recomputed"], "roofline": rf, "csr_kernel": csr,
               "host_mesh_s": t_host, "device_setup_s": t_setup}
        del gd, A
        torch.cuda.empty_cache()
        return out

    # ---- objects of the full result --------------------------------------------------------------------------------------------------------------
    def solver_roofline(self, r, wkey):
        achieved = r["spmv_bytes"] / (r["spmv_ms"] * 1e-3) / 1e9
        csr_equiv = r["csr_bytes"] / (r["spmv_ms"] * 1e-3) / 1e9
        traffic, src = self.traffic_of(f"{r['kernel_key']}@{wkey}") if self.world == 1 else (None, None)
        return {
            "kernel_key": r["kernel_name"], "kernel_note": r["kernel"], "solver_layout_mode": r["mode"],
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src,
            "traffic_over_algorithmic": (traffic / r["spmv_bytes"]) if traffic else None,
            "frac_actual": (traffic / (r["spmv_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
            "algorithmic_bytes_per_launch": r["spmv_bytes"], "avg_launch_ms": r["spmv_ms"], "launches": r["spmv_launches"],
            "algorithmic_bytes_note": "what this kernel design moves per SpMV (mfem_csr_solver_layout_bytes / mfem_debug_lat27_pass1_bytes / mfem_csr_spmv_bytes): "
                                      "8 B per stored matrix entry (symmetric layouts store the diagonal + upper entries only) + x as the kernel stages it + y "
                                      "(+ the tiles' y blocks written and read again for the two-launch lattice tiles)",
            **({"plain_diagonal_kernel_bytes_per_launch": r["plain_bytes"]} if r.get("plain_bytes") else {}),
            "csr_equivalent": {"bytes_per_launch": r["csr_bytes"], "achieved": csr_equiv, "frac": csr_equiv / HBM_PEAK_GBS,
                               "note": "the same launch priced with SURVEY 8(d)'s CSR formula (12 B per nonzero): what a CSR kernel would have to sustain to "
                                       "match this time (reported, never `frac`)"},
        }

    def check_residual(self, r, what, loose=None):
        """bench.py checks what it times: the last timed solve must have reduced the residual (and produced finite numbers).  CG reduces the energy norm
        monotonically and, over 200 iterations, the residual too: final < initial is required.  bicgstabl_GS! / idrs! are not monotone -- on the
        penalty-constrained elasticity operator of c3 ||r|| hovers around ||r0|| for the first hundreds of steps -- so their legs have to stay finite and
        within 2 x ||r0|| (10 x for the short fixed counts of the functional runs, which sit inside BiCGStab's initial hump).  Independent of convergence,
        the residual the solver reports must be the one recomputed outside it with the CSR kernel (one rank)."""
        ir, fr = r["initial_res"], r["final_res"]
        strict = r["cfg"]["solver"] == "cg"
        if loose is None:
            loose = 2.0 if r.get("iters", self.args.iters) >= 200 else 10.0
        ok = ir is not None and fr is not None and fr == fr and ir == ir and fr < float("inf") and (fr < ir if strict else fr < loose * ir)
        if not ok:
            raise ResidualCheckFailed(f"{what}: the last timed solve did not reduce the residual (initial {ir}, final {fr}) -- the measurement is invalid")
        rc = r.get("final_res_recomputed")
        if rc is not None and not (abs(rc - fr) <= 1e-5 * max(fr, rc) + 1e-9 * ir):  # (two kernels, two summation orders)
            raise ResidualCheckFailed(f"{what}: the solver reports a final residual of {fr}, the CSR kernel on the caller's matrix gives {rc} for the "
                                      f"solution it returned -- the measurement is invalid")

    def secondary_object(self, t, title, wkey, baseline_config):
        tv = t["n_global"] * t["updates"] / t["elapsed"]
        sms = t["solve_ms"] / t["steps"]
        return {"workload": title, "baseline_config": baseline_config, "value": tv, "unit": "DOF-updates/s", "n_dof": t["n_global"], "nnz": t["nnz"],
                "steps": t["steps"], "ms_per_step": t["elapsed"] / t["steps"] * 1e3, "solve_ms_per_step": sms,
                "assembly_ms_per_step": t["elapsed"] / t["steps"] * 1e3 - sms, "krylov_steps_per_step": t["updates"] / t["steps"],
                "initial_res": t["initial_res"], "final_res": t["final_res"], "final_res_recomputed": t.get("final_res_recomputed"),
                "roofline": self.solver_roofline(t, wkey), "csr_kernel": t.get("csr_kernel"),
                **({"remainder": t["remainder"]} if t.get("remainder") else {}),
                **({"time_to_tol": t["time_to_tol"]} if t.get("time_to_tol") else {})}

    def headline_object(self, r, cfg, config_name):
        args, world = self.args, self.world
        per_step_updates = r["updates"] / r["steps"]
        solve_ms_step = r["solve_ms"] / r["steps"]
        n = r["N"]
        return {
            "metric": cfg["metric"], "value": r["n_global"] * r["updates"] / r["elapsed"], "unit": "DOF-updates/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["elapsed"] / r["steps"] * 1e3, "higher_is_better": True,
            "scaling": "strong" if self.strong else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": f"{cfg['title']}, {r['nx_global']}x{n}x{n} structured mesh (make_Brick)"
                            + (f" cut into {world} slabs along i (strong scaling)" if self.strong else "") + ": fused assembly (K + R) + "
                            + (f"{r['iters']} Jacobi-CG iterations per step" if cfg["solver"] == "cg" else f"{per_step_updates:.0f} {SOLVER_TEXT[cfg['solver']]} per step")
                            + (", temperature fixed on x = 0 by the reference's Nitsche form (2D_Script.jl:58: nonsymmetric K)" if cfg.get("nitsche") else ""),
                "baseline_config": {"c2": "the north_star target size (512^3 hex-8, 1 GPU); configs[1] (256^3) is `secondary_256`",
                                    "c3": "configs[2]", "c4": "configs[3]", "ref_idrs8": "configs[1] solved with the reference's default solver",
                                    "nitsche_c2": "configs[1] mesh and form, Dirichlet face added", "nitsche_c4": "configs[3] mesh and form, Dirichlet face added"
                                    }[config_name] if n == cfg["n"] else f"{config_name} at a non-default size",
                "n_dof": r["n_global"], "nnz": r["nnz"], "nnz_per_gpu": r["nnz"], "krylov_steps_per_step": per_step_updates,
                "parallelism": "single GPU" if world == 1 else f"slab decomposition x{world} ("
                               + (self.state["transport"] or ("host callbacks over gloo, ranks sharing one GPU: functional run" if self.host_comm else "RCCL"))
                               + " halo overlapped with the interior rows + one all-reduce per reduction group)",
                "solve_ms_per_step": solve_ms_step,
                # the two halves of the metric on their own (whole job, all ranks)
                "assembly_ms_per_step": r["elapsed"] / r["steps"] * 1e3 - solve_ms_step,
                "assembly_dof_per_s": r["n_global"] / max(r["elapsed"] / r["steps"] - solve_ms_step * 1e-3, 1e-12),
                "solve_dof_updates_per_s": r["n_global"] * per_step_updates / (solve_ms_step * 1e-3),
                # the last timed solve (true residuals, ||.||_2 / sqrt(n)): the run fails unless final < initial; recomputed after the timed region with mul!
                "initial_res": r["initial_res"], "final_res": r["final_res"], "final_res_recomputed": r.get("final_res_recomputed"),
                "first_step_s": r["first_step_s"], "first_step_split": r.get("first_step_split"),
                "workspace_placement_trial": ("on (--ws-trial 1: the first solve times the SpMV on up to three allocations of the workspace and keeps the "
                                              "fastest; its cost is inside first_step_s, outside the timed region)" if args.ws_trial else
                                              "off (library default since round 4)"),
            },
            "roofline": self.solver_roofline(r, f"{config_name}_{n}"),
        }

    def strong_object(self, t, cfg):
        return {"workload": f"{cfg['title']}, THE {t['N']}^3 mesh cut into {self.world} slabs along i, {t['steps']} timed steps after 1 warm-up, same step as above",
                "scaling": "strong", "value": t["n_global"] * t["updates"] / t["elapsed"], "unit": "DOF-updates/s", "n_dof": t["n_global"],
                "ms_per_step": t["elapsed"] / t["steps"] * 1e3, "solve_ms_per_step": t["solve_ms"] / t["steps"],
                "initial_res": t["initial_res"], "final_res": t["final_res"], "comm_exposed": t["comm_exposed"],
                "note": "speed-up over one GPU = this value / the N = 1 line's value on the same mesh (the driver computes it from its own runs)"}

    def apply_live_traffic(self, out, main_res, leg):
        """The counters of the headline workload's two priced kernels, observed in THIS run (the parent's GPU work is over and its memory released; the
        children set up the same matrix themselves)."""
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
