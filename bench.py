#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X MetaFEM backend (contract: task prompt / DESIGN.md section 5).

Metric (BASELINE.json): DOF-updates/s = n_DOF x Krylov iterations / (t_assembly + t_solve).  One *step* = what update_OneStep!
(solver/04_Time_Domain.jl:59-80) runs for one Newton step with a capped solver: fused K + matrix-free R + `--iters` Krylov steps (bench_legs.py).

--config c2 (default): 3-D thermal conduction, linear hex-8, Jacobi-PCG; N = 1 workload = the size the north_star's target is quoted on, 512^3
    (135 005 697 DOF, nnz 3.63e9, int64 row pointers).  The default N = 1 invocation also runs configs[1] (256^3), configs[2] (c3: hex-8 elasticity,
    bicgstabl_GS!(2)), configs[3] (c4: hex-27, CG) + the FP64-MFMA roofline of the hex-27 Ke kernels, the reference's own solver / boundary-condition path
    (idrs!(8); Nitsche-Dirichlet nonsymmetric K), a Newton-sized step and the unstructured hex-20 path for a few timed steps each.
--config c3 | c4 | ref_idrs8 | nitsche_c2 | nitsche_c4: that leg alone, at any N.
N > 1: `--scaling weak` (default) = an --n thick slab per rank; `--scaling strong` = the config's n^3 mesh cut into N slabs.  `--dry 1`: everything except
device work (rank spawn, slab planes, halo sizes, workspace bytes per rank, the compact line) -- runs without a GPU (tests/test_bench_report.py at world 8).

Output: rank 0 prints ONE compact JSON line (<= 8 KB: bench_report.compact_line -- contract keys, `roofline`, `cpu_baseline`, one small object per leg) and
writes the full result object (notes, sources, per-leg rooflines, time_to_tol, ...) to the side file the line names in `full`.  A failing SECONDARY leg
becomes {"error": ...} in both; only an invalid headline measurement fails the run.
"""
from __future__ import annotations

import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from bench_report import compact_line  # noqa: E402  (no torch: safe before the ranks exist)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=["c2", "c3", "c4", "nitsche_c2", "nitsche_c4", "ref_idrs8"], default="c2")
    ap.add_argument("--n", type=int, default=0, help="elements per side of the per-GPU mesh (0 = the config's size: 512 / 128 / 128)")
    ap.add_argument("--iters", type=int, default=200, help="Krylov iterations per step (BiCGStab(2) / IDR(s): SpMV-equivalent steps)")
    ap.add_argument("--cpu-n", type=int, default=192, help="elements per side of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-repeats", type=int, default=3, help="timed CPU steps (the median is reported)")
    ap.add_argument("--secondary-n", type=int, default=256, help="c2 at N = 1: also run this mesh size (configs[1]) for a few steps (0 = skip)")
    ap.add_argument("--secondary-steps", type=int, default=5)
    ap.add_argument("--hex27-n", type=int, default=128, help="c2 at N = 1: CSR-kernel roofline on the hex-27 matrix of this size (0 = skip)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--secondary-configs", type=int, default=1, help="c2 at N = 1, default size: also run configs[2], configs[3] and the hex-27 Ke MFMA roofline (0 = skip)")
    ap.add_argument("--secondary-config-n", type=int, default=0,
                    help="elements per side of the secondary legs of the default line (0 = the configs' own sizes; a smaller value makes the legs run at any --n: "
                         "functional checks of the line's schema)")
    ap.add_argument("--ref-legs", type=int, default=1, help="c2 at N = 1, default size: also run the reference's own solver / boundary-condition path (0 = skip)")
    ap.add_argument("--newton-like", type=int, default=20, help="secondary 256^3 leg: also time assembly + a solve of this many iterations (0 = skip)")
    ap.add_argument("--tet10-n", type=int, default=64, help="c2 at N = 1, default size: the unstructured legs on a 10-node tetrahedron mesh (an n^3 brick cut into tetrahedra), thermal and "
                    "3-field elasticity with idrs!(8); 0 = skip")
    ap.add_argument("--u20-n", type=int, default=96, help="c2 at N = 1, default size: unstructured serendipity hex-20 legs (thermal, elasticity) of this many elements "
                                                          "per side through mfem_pattern_build + mfem_mesh_assemble_elements_rows + idrs!(8) (0 = skip)")
    ap.add_argument("--time-to-tol", type=int, default=1, help="c3 legs: also solve the same system to 1e-8 ||r0|| with bicgstabl_GS!(2), idrs!(8), cg! (0 = skip)")
    ap.add_argument("--strong-leg", type=int, default=1, help="N > 1, weak: also time THE config's n^3 mesh cut into N slabs -> `strong_scaling` (0 = skip)")
    ap.add_argument("--strong-leg-timeout", type=float, default=240.0)
    ap.add_argument("--remainder", type=int, default=1, help="0: switch the skew remainder of the lattice tiles off (A/B of profiles/r05_nitsche_ab.txt)")
    ap.add_argument("--ws-trial", type=int, default=1, help="1: opt in to the library's workspace placement trial (mfem_debug_set_ws_trial); 0 = as the library ships")
    ap.add_argument("--live-traffic", type=int, default=1, help="N = 1, config sizes: collect the headline rooflines' `traffic` in this run (two rocprofv3 --pmc child passes)")
    ap.add_argument("--dry", type=int, default=0, help="1: no device work -- rank spawn, slab planes, halo sizes, workspace bytes per rank, the compact line (CPU only)")
    ap.add_argument("--full-out", default="", help="path of the full result object (default gpurun_out/bench_full.json; '-' = do not write)")
    return ap.parse_args(argv)


def self_launch(n_ranks: int) -> int:
    """plain `python bench.py --gpus N`: start the N ranks as child processes BEFORE anything here touches the GPU; rank 0's JSON line is the only thing
    that reaches stdout; a failing rank fails the run (and the others are ended: they would wait in the rendezvous for ever)."""
    import socket
    import subprocess
    import threading
    import time

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


def emit(out, args):
    """Write the full object beside the line, print the line (rank 0)."""
    path = args.full_out or os.path.join("gpurun_out", "bench_full.json")
    named = None
    if path != "-":
        try:
            full_path = path if os.path.isabs(path) else os.path.join(ROOT, path)
            os.makedirs(os.path.dirname(full_path) or ".", exist_ok=True)
            with open(full_path, "w") as f:
                json.dump(out, f, indent=1, default=str)
            named = path
        except OSError as e:  # (a read-only tree: the line still goes out)
            print(f"bench.py: could not write the full result object to {path}: {e}", file=sys.stderr)
    sys.stdout.write(compact_line(out, named) + "\n")
    sys.stdout.flush()


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus))
    if args.dry:
        import bench_dry

        sys.exit(bench_dry.main(args, emit))

    import ctypes as C

    import torch

    import bench_legs as L
    from metafem_jl_amd import _lib

    cfg, base = L.config_of(args.config)
    if args.n <= 0:
        args.n = cfg["n"]
    B = L.Bench(args)
    rank, world = B.rank, B.world
    if args.ws_trial:
        _lib.check(_lib.lib.mfem_debug_set_ws_trial(1))
    if not args.remainder:
        _lib.check(_lib.lib.mfem_debug_set_remainder(0))

    main_res = B.run_workload(cfg, base, args.n, args.steps, args.warmup, want_csr=True)
    out = {"errors": []}
    if rank == 0:
        out.update(B.headline_object(main_res, cfg, args.config))
        if main_res["comm_exposed"] is not None:
            out["comm_exposed"] = main_res["comm_exposed"]
        try:
            B.check_residual(main_res, f"{args.config} {args.n}^3")
        except L.ResidualCheckFailed as e:  # the HEADLINE measurement is invalid: no line
            raise SystemExit(f"bench.py: {e}")
        if main_res.get("remainder"):
            out["remainder"] = main_res["remainder"]
        if "csr_kernel" in main_res:
            out["roofline"]["csr_kernel"] = main_res["csr_kernel"]  # the north_star's own number: the CSR kernel behind mul! on this matrix, this run

    def guarded(key, fn):
        """A secondary leg never takes the line down: its failure is the leg's object."""
        try:
            obj = fn()
            if obj is not None:
                out[key] = obj
        except (Exception, SystemExit) as e:
            out[key] = {"error": f"{type(e).__name__}: {e}"}
            out["errors"].append(f"{key}: {type(e).__name__}: {e}")
            torch.cuda.empty_cache()

    if world > 1 and not B.strong and args.strong_leg:
        # the same config as ONE mesh of its own size cut into `world` slabs, a few steps: every rank takes part.  The weak measurement above is COMPLETE at
        # this point: a watchdog on every rank makes sure a problem in this extra leg (it builds a second communicator) can only cost the extra object,
        # never the line -- past the deadline rank 0 prints the line it has and all ranks leave.
        import threading

        def bail():
            if rank == 0:
                out["strong_scaling"] = {"error": f"the strong-scaling leg did not finish within {args.strong_leg_timeout} s; the line is the complete weak measurement"}
                out["cpu_baseline"] = None
                emit(out, args)
            os._exit(0)

        watchdog = threading.Timer(args.strong_leg_timeout, bail)
        watchdog.daemon = True
        watchdog.start()
        strong_res = None
        try:
            strong_res = B.run_workload(cfg, base, args.n, args.secondary_steps, 1, want_csr=False, strong=True)
            if rank == 0:
                B.check_residual(strong_res, f"{args.config} {args.n}^3 strong")
                out["strong_scaling"] = B.strong_object(strong_res, cfg)
        except Exception as e:  # (an error on one rank usually shows on all of them: every rank reports and the line still goes out)
            if rank == 0:
                out["strong_scaling"] = {"error": f"{type(e).__name__}: {e}"}
        finally:
            watchdog.cancel()

    default_line = world == 1 and args.config == "c2" and (args.n == cfg["n"] or args.secondary_config_n > 0)
    if world == 1 and args.config == "c2" and args.secondary_n > 0 and args.secondary_n != args.n:
        def leg_256():  # configs[1] of BASELINE.json (256^3 hex-8)
            c = dict(cfg, newton_like=args.newton_like)
            t = B.run_workload(c, "c2", args.secondary_n, args.secondary_steps, 1, want_csr=True)
            B.check_residual(t, f"c2 {args.secondary_n}^3")
            if t.get("newton_like"):
                out["newton_like"] = dict(t["newton_like"], workload=f"{cfg['title']}, {args.secondary_n}^3: fused assembly (K + R) + a {args.newton_like}-iteration "
                                                                       "Jacobi-CG solve per step (a Newton-sized solve: the per-solve layout copy, diagonal and symmetry "
                                                                       "probe are inside solve_ms; per_solve_ms = solve_ms - its iterations' share of a 200-iteration solve)")
            return B.secondary_object(t, f"{cfg['title']}, {args.secondary_n}^3, {t['steps']} timed steps after 1 warm-up, same step as the headline",
                                      f"c2_{args.secondary_n}", "configs[1]")
        guarded(f"secondary_{args.secondary_n}", leg_256)
    if default_line and args.secondary_configs:
        for ck in ("c3", "c4"):  # configs[2] and configs[3] of BASELINE.json in the same invocation
            def leg(ck=ck):
                c = dict(L.CONFIGS[ck], time_to_tol=(ck == "c3"))
                if args.secondary_config_n > 0:
                    c["n"] = args.secondary_config_n
                t = B.run_workload(c, ck, c["n"], args.secondary_steps, 1, want_csr=True)
                B.check_residual(t, f"{ck} {c['n']}^3")
                per = t["updates"] / t["steps"]
                obj = B.secondary_object(t, f"{c['title']}, {c['n']}^3, {t['steps']} timed steps after 1 warm-up: fused assembly (K + R) + "
                                         + (f"{args.iters} Jacobi-CG iterations" if c["solver"] == "cg" else f"{per:.0f} {L.SOLVER_TEXT[c['solver']]}") + " per step",
                                         f"{ck}_{c['n']}", {"c3": "configs[2]", "c4": "configs[3]"}[ck])
                obj["metric"] = c["metric"]
                return obj
            guarded(f"secondary_{ck}", leg)
        guarded("roofline_hex27_ke", lambda: B.hex27_ke_roofline(args.secondary_config_n or L.CONFIGS["c4"]["n"]))
    if default_line and args.ref_legs:
        for lk, legd in L.REF_LEGS.items():  # the reference's own solver / boundary-condition path at scale
            def leg(lk=lk, legd=legd):
                c = dict(L.CONFIGS[legd["base"]], solver=legd["solver"], nitsche=legd["nitsche"], n=args.secondary_config_n or legd["n"])
                t = B.run_workload(c, legd["base"], c["n"], args.secondary_steps, 1, want_csr=legd["nitsche"])
                B.check_residual(t, f"{lk} {c['n']}^3")
                per = t["updates"] / t["steps"]
                obj = B.secondary_object(
                    t, f"{c['title']}, {c['n']}^3" + (", temperature fixed on x = 0 by the reference's Nitsche form (2D_Script.jl:58: NONSYMMETRIC K), convection "
                                                       "on the other faces" if legd["nitsche"] else "")
                    + f", {t['steps']} timed steps after 1 warm-up: fused assembly (K + R) + {per:.0f} {L.SOLVER_TEXT[legd['solver']]} per step",
                    f"{lk}_{c['n']}", {"c2": "configs[1]", "c4": "configs[3]"}[legd["base"]]
                    + (" mesh and form, Dirichlet face added" if legd["nitsche"] else " solved with the reference's default solver"))
                obj["metric"] = f"DOF-updates/sec (assembly + {L.SOLVER_TEXT[legd['solver']]})"
                return obj
            guarded(f"{lk}_{legd['n']}", leg)  # (the key names the leg at its own size: ref_idrs8_256, nitsche_c2_256, nitsche_c4_128)
    if default_line and args.u20_n > 0 and hasattr(B, "unstructured_leg"):
        n20 = min(args.u20_n, args.secondary_config_n) if args.secondary_config_n > 0 else args.u20_n
        for fields, key in ((1, "u20_thermal"), (3, "u20_elasticity")):
            guarded(f"{key}_{args.u20_n}", lambda fields=fields: B.unstructured_leg(n20, fields, max(args.secondary_steps // 2, 2)))
    if default_line and args.tet10_n > 0 and hasattr(B, "unstructured_leg"):
        nt = min(args.tet10_n, args.secondary_config_n) if args.secondary_config_n > 0 else args.tet10_n
        for fields, key in ((1, "tet10_thermal"), (3, "tet10_elasticity")):
            guarded(f"{key}_{args.tet10_n}", lambda fields=fields: B.unstructured_leg(nt, fields, max(args.secondary_steps // 2, 2), shape="SIMPLEX"))
    if rank == 0 and world == 1 and args.config == "c2" and args.hex27_n > 0:
        def csr27():  # the CSR kernel on the other matrix shape of the configs: hex-27 (27..125 entries per row), configs[3]'s size
            import metafem_jl_amd as mf

            b27 = mf.Brick((1.0, 1.0, 1.0), (args.hex27_n,) * 3, 2, 5, ctx=B.ctx)
            A27 = b27.pattern(1)
            K27 = b27.assemble_thermal(A27, L.K_COND, L.H, L.TENV, mf.ALL_FACES)
            out["roofline"]["csr_kernel_hex27"] = dict(B.csr_kernel_roofline(A27, K27, f"c4_{args.hex27_n}"), matrix=f"hex-27 thermal {args.hex27_n}^3")
            del b27, A27, K27
            torch.cuda.empty_cache()
        try:
            csr27()
        except Exception as e:  # never lose the line over a side measurement
            out["roofline"]["csr_kernel_hex27"] = {"error": repr(e)}
    if rank == 0 and world == 1 and args.live_traffic and args.n == cfg["n"]:
        try:
            B.apply_live_traffic(out, main_res, f"{args.config}_{args.n}")
        except Exception as e:
            out["errors"].append(f"live traffic: {e!r}")
    if rank == 0:
        out["cpu_baseline"] = None
        if world == 1 and args.config == "c2" and args.cpu_n > 0:
            def cpu():
                cb = L.cpu_baseline(args.cpu_n, args.iters, args.cpu_repeats)
                sk = f"secondary_{args.secondary_n}"
                out["vs_cpu_baseline"] = {"main_workload": out["value"] / cb["value"],
                                          **({sk: out[sk]["value"] / cb["value"]} if "value" in out.get(sk, {}) else {}),
                                          "note": "GPU whole-step throughput / CPU-port whole-step throughput (same step definition, same CG iterations per "
                                                  "assembly; the CPU sample is a smaller mesh, see cpu_baseline.sample)"}
                return cb
            guarded("cpu_baseline", cpu)
        if not out["errors"]:
            del out["errors"]
        C.CDLL(None).fflush(None)
        emit(out, args)
    B.close()


if __name__ == "__main__":
    main()
