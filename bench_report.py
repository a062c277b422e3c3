"""bench_report.py -- the ONE stdout line of bench.py, built from the full result object (no torch, no GPU: importable in the CPU suite).

The driver parses the last stdout line; round 5's line had grown to 32 KB of which 20 KB were prose and was not parsed.  Contract
of this module: `compact_line(full)` is at most MAX_LINE_BYTES (8 KB) long for ANY full object, always carries the contract keys, `roofline` and
`cpu_baseline` as numbers, and per secondary leg only {value, ms_per_step, spmv_ms, frac, frac_actual, csr_frac} (or {"error": ...}).  Everything else
(notes, sources, time_to_tol, first_step_split, hex-27 sub-objects) stays in the full object, which bench.py writes to a side file named in the line.
tests/test_bench_report.py runs it on canned objects (among them round 5's committed 32 KB line)."""
from __future__ import annotations

import json
import math

MAX_LINE_BYTES = 8192
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s
FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X_MICROARCH.md: FP64 matrix peak, dense

# What each kernel key of a roofline object IS: prose for the full file only (the line carries the key).
KERNEL_NOTES = {
    "k_spmv_symp": "k_spmv_symp<0> (one launch: the sweep, then the two boundary planes row by row): symmetric sweep on wave-private (j, k) patches of a "
                   "patch-major copy of the CSR matrix made once per solve; the values passed the per-solve bitwise symmetry check, so 10.5 of a row's 13 "
                   "lower-diagonal entries are mirrored through LDS (bitwise the same y as the plain diagonal-slotted kernel); x staged per lattice plane in LDS",
    "k_spmv_sym27": "k_spmv_sym27 (+ k_spmv_dia on the two boundary planes): SpMV on the slot-major copy of the CSR matrix made once per solve; lower-diagonal "
                    "entries mirrored through LDS (bitwise the same y as the plain diagonal-slotted kernel)",
    "k_spmv_dia": "k_spmv_dia: SpMV on the slot-major copy of the CSR matrix made once per solve; diagonal-slotted blocks read no column stream",
    "k_spmv_ell": "k_spmv_ell: SpMV on the slot-major copy of the CSR matrix made once per solve; f64 val / i32 col",
    "k_spmv_lat27": "k_spmv_lat27 + k_lat27_gather (two launches per SpMV): symmetric lattice tiles of the hex-27 matrix, copy made once per solve; only the "
                    "diagonal and the entries with column > row are stored and read (14..63 of a row's 27..125); x and y of a tile of 8 x 8 x 32 lattice points "
                    "in LDS; the second launch sums the tiles' y blocks in a fixed order",
    "k_spmv_lat27_pass1": "k_spmv_lat27 (pass 1 of the SpMV; pass 2 -- the sums over the tiles' y blocks -- runs inside the CG residual update, k_lat27_gather_cg, "
                          "and p . A p comes from pass 1: A p is never stored)",
    "k_spmv_lat8": "k_spmv_lat8<F> + k_lat8_gather (two launches per SpMV): symmetric lattice tiles of the F-field 27-point matrix, copy made once per solve; per "
                   "node only the upper entries are stored and read (123 of 243 values for F = 3, 14 of 27 for F = 1); lane = node, x and y of a tile of "
                   "8 x 8 x 16 nodes in LDS, the right Jacobi scaling applied to x while it is staged; the second launch sums the tiles' y blocks in a fixed order",
    "k_spmv_sell": "k_spmv_sell: rows sorted by length and diagonal-list signature, SELL-128 copy made once per solve; blocks whose rows share one diagonal "
                   "list read no columns",
    "k_spmv_bsell": "k_spmv_bsell<F>: node-blocked sliced layout of a field-major F-field matrix (nodes sorted by their number of coupled nodes, 64 per block, "
                    "copy made once per solve): a lane owns a node -- one column index and F gathers of x per F x F values",
    "csr_kernel": "mfem_spmv_csr (mul!): k_spmv_csr_w (rows of up to 64 entries, uniform length) / k_spmv_csr_rb (wide or uneven rows) on the caller's CSR "
                  "arrays (i64 rowptr / i32 col / f64 val), no copy",
    "remainder": "+ k_rem_apply (A = S + N: the values are NONSYMMETRIC in a few rows -- the Nitsche face --; the tiles apply the mirrored upper triangle S, a "
                 "third launch adds the skew remainder N of those rows; the SAME symmetry measure passed on S + N)",
}

CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
CONFIG_KEYS = ("workload", "baseline_config", "n_dof", "nnz", "krylov_steps_per_step", "parallelism", "solve_ms_per_step", "assembly_ms_per_step",
               "initial_res", "final_res", "final_res_recomputed", "first_step_s")
LEG_PREFIXES = ("secondary_", "ref_", "nitsche_", "u20_", "tet10_", "newton_like")


def sig(x, digits=5):
    """Numbers to `digits` significant digits (a line of 15-digit floats is twice as long and no more informative); everything else unchanged."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        if not math.isfinite(x):
            return None  # (NaN / Infinity are not JSON)
        return float(f"{x:.{digits - 1}e}")
    if isinstance(x, dict):
        return {k: sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [sig(v, digits) for v in x]
    return x


def clip(s, n):
    if not isinstance(s, str):
        return s
    return s if len(s) <= n else s[: n - 3] + "..."


def compact_csr_kernel(ck):
    """`csr_kernel` as numbers: the CSR kernel behind mul! (04_GPU_Utils.jl:131) on the leg's matrix."""
    if not isinstance(ck, dict):
        return None
    if "error" in ck:
        return {"error": clip(str(ck["error"]), 160)}
    return {"ms": ck.get("avg_launch_ms"), "bytes": ck.get("algorithmic_bytes_per_launch"), "frac": ck.get("frac"),
            "frac_actual": ck.get("frac_actual"), "csr_frac": (ck.get("csr_equivalent") or {}).get("frac"), "traffic": ck.get("traffic")}


def compact_roofline(rf):
    if not isinstance(rf, dict):
        return None
    out = {"kernel": clip(rf.get("kernel_key") or rf.get("kernel"), 100), "bound": rf.get("bound"), "achieved": rf.get("achieved"), "peak": rf.get("peak"),
           "unit": rf.get("unit"), "frac": rf.get("frac"), "traffic": rf.get("traffic"), "frac_actual": rf.get("frac_actual"),
           "avg_launch_ms": rf.get("avg_launch_ms"), "launches": rf.get("launches"), "bytes_per_launch": rf.get("algorithmic_bytes_per_launch"),
           "traffic_live": bool(str(rf.get("traffic_source") or "").startswith("collected in this run"))}
    if rf.get("csr_equivalent"):
        out["csr_frac"] = rf["csr_equivalent"].get("frac")
    for k in ("csr_kernel", "csr_kernel_hex27"):
        if k in rf:
            out[k] = compact_csr_kernel(rf[k])
    return out


def compact_leg(obj):
    """Per secondary leg: {value, ms_per_step, spmv_ms, frac, frac_actual, csr_frac} (+ a short kernel key), or {"error": ...}."""
    if not isinstance(obj, dict):
        return None
    if "error" in obj:
        return {"error": clip(str(obj["error"]), 200)}
    rf = obj.get("roofline") or {}
    ck = obj.get("csr_kernel") or {}
    out = {"value": obj.get("value"), "ms_per_step": obj.get("ms_per_step"), "spmv_ms": rf.get("avg_launch_ms"), "kernel": clip(rf.get("kernel_key"), 40),
           "frac": rf.get("frac"), "frac_actual": rf.get("frac_actual"),
           "csr_frac": ck.get("frac_actual") if ck.get("frac_actual") is not None else ck.get("frac")}
    for k in ("assembly_ms", "assembly_frac", "n_dof", "iterations", "solve_ms", "per_solve_ms"):  # (legs that are not "assembly + N Krylov steps")
        if obj.get(k) is not None:
            out[k] = obj[k]
    return out


def compact_mfma(ke):
    if not isinstance(ke, dict):
        return None
    if "error" in ke:
        return {"error": clip(str(ke["error"]), 200)}
    out = {"kernel": "k_hex27<true,true>+k_hex27_gather_lds", "bound": "mfma", "achieved": ke.get("achieved"), "peak": ke.get("peak"), "unit": ke.get("unit"),
           "frac": ke.get("frac"), "avg_assembly_ms": ke.get("avg_assembly_ms"), "mfma_pipe_busy_in_pass1": ke.get("mfma_pipe_busy_in_pass1")}
    for sub, key in (("affine_mesh_assembly", "affine_ms"), ("general_mesh_assembly", "general_ms")):
        if isinstance(ke.get(sub), dict):
            out[key] = ke[sub].get("avg_assembly_ms")
    return out


def compact_comm(ce):
    """Per-rank exposed communication -> worst rank's numbers (the full per-rank list stays in the side file)."""
    if not isinstance(ce, list) or not ce:
        return None
    worst = max(ce, key=lambda c: c.get("exposed_fraction_of_solve") or 0.0)
    return {k: worst.get(k) for k in ("rank", "halo_wait_ms_per_step", "allreduce_ms_per_step", "solve_ms_per_step", "exposed_fraction_of_solve")}


def compact_dict(full, full_path=None):
    out = {k: full.get(k) for k in CONTRACT_KEYS}
    cfg = full.get("config") or {}
    out["config"] = {k: (clip(cfg.get(k), 300) if isinstance(cfg.get(k), str) else cfg.get(k)) for k in CONFIG_KEYS if k in cfg}
    if "nnz" not in out["config"] and "nnz_per_gpu" in cfg:
        out["config"]["nnz"] = cfg["nnz_per_gpu"]
    out["roofline"] = compact_roofline(full.get("roofline"))
    cb = full.get("cpu_baseline")
    out["cpu_baseline"] = ({"error": clip(str(cb["error"]), 200)} if isinstance(cb, dict) and "error" in cb else
                           {"value": cb.get("value"), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                            "sample": clip(cb.get("sample_short") or cb.get("sample"), 120)} if isinstance(cb, dict) else None)
    if isinstance(full.get("vs_cpu_baseline"), dict):
        out["vs_cpu_baseline"] = full["vs_cpu_baseline"].get("main_workload")
    legs = {}
    for k, v in full.items():
        if k.startswith(LEG_PREFIXES) and isinstance(v, dict):
            legs[k] = compact_leg(v)
    out.update(legs)
    if "roofline_hex27_ke" in full:
        out["roofline_hex27_ke"] = compact_mfma(full["roofline_hex27_ke"])
    if full.get("comm_exposed"):
        out["comm_exposed"] = compact_comm(full["comm_exposed"])
    ss = full.get("strong_scaling")
    if isinstance(ss, dict):
        out["strong_scaling"] = ({"error": clip(str(ss["error"]), 200)} if "error" in ss else
                                 {"scaling": "strong", "value": ss.get("value"), "n_dof": ss.get("n_dof"), "ms_per_step": ss.get("ms_per_step"),
                                  "solve_ms_per_step": ss.get("solve_ms_per_step"), "comm_exposed": compact_comm(ss.get("comm_exposed"))})
    if full.get("dry"):
        # --dry 1: per scaling mode the planned slabs' extremes (the per-rank plan is in the side file)
        out["dry"] = True
        out["dry_plan"] = {k: {"ranks": len(v), "max_rows": max(p["n_rows"] for p in v), "min_rows": min(p["n_rows"] for p in v),
                               "max_nnz": max(p["nnz"] for p in v), "halo_bytes": max(p["halo_bytes_per_neighbour_per_spmv"] for p in v),
                               "max_est_device_bytes": max(p["est_device_bytes"] for p in v), "fits_288GB": all(p["fits_288GB"] for p in v)}
                           for k, v in (full.get("dry_plan") or {}).items()}
    if full.get("errors"):
        out["errors"] = [clip(str(e), 160) for e in full["errors"]][:8]
    if full_path:
        out["full"] = full_path
    return sig(out)


def compact_line(full, full_path=None) -> str:
    """The line.  Never longer than MAX_LINE_BYTES: optional objects are dropped (named in `dropped`) until it fits -- with the fixed schema above that does
    not happen for any object bench.py produces (about 3-5 KB), but a future leg must not be able to un-parse the driver's record again."""
    d = compact_dict(full, full_path)
    line = json.dumps(d, separators=(",", ":"), allow_nan=False)
    droppable = ["vs_cpu_baseline", "roofline_hex27_ke", "comm_exposed", "strong_scaling", "errors"] + [k for k in d if k.startswith(LEG_PREFIXES)]  # (popped from the end)
    dropped = []
    while len(line.encode()) > MAX_LINE_BYTES and droppable:
        k = droppable.pop()
        if k in d:
            del d[k]
            dropped.append(k)
            d["dropped"] = dropped
            line = json.dumps(d, separators=(",", ":"), allow_nan=False)
    if len(line.encode()) > MAX_LINE_BYTES:  # (only the contract keys are left: clip the strings hard)
        d["config"] = {k: clip(v, 80) if isinstance(v, str) else v for k, v in d["config"].items()}
        d["metric"] = clip(d.get("metric"), 120)
        line = json.dumps(d, separators=(",", ":"), allow_nan=False)
    assert len(line.encode()) <= MAX_LINE_BYTES, len(line)
    return line
