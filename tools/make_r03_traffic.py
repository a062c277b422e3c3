"""gpurun_out/pmc_r03/<leg>_<group>/**/out_counter_collection.csv (tools/run_pmc_r03.sh) -> r03_traffic.json: HBM-side bytes per launch
of every kernel bench.py prices, keyed '<kernel key>@<leg>', with the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE counts a
coalesced streaming read at 1/2: doubled) and its calibration in the same run on mfem_axpby (2 n doubles read, n written).
usage: make_r03_traffic.py <pmc dir> <out json> [tree id]"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

src, dst = sys.argv[1], sys.argv[2]
tree = sys.argv[3] if len(sys.argv) > 3 else os.environ.get("MFEM_TREE", "")
if not tree:
    try:
        tree = subprocess.check_output(["git", "-C", os.path.dirname(os.path.abspath(__file__)), "rev-parse", "--short", "HEAD"],
                                       stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        tree = "working tree of the gpurun snapshot"


def means(d):
    """{kernel name: {counter: (launches, mean)}} of one pass directory"""
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    out = collections.defaultdict(dict)
    for (k, c), v in agg.items():
        out[k][c] = (len(v), sum(v) / len(v))
    return out


KEYS = [  # (kernel key of bench.py, substrings of the kernels one "launch" consists of)
    ("csr_kernel", ["k_spmv_csr_w", "k_spmv_csr_rb"]),
    ("k_spmv_symp", ["k_spmv_symp<0>", "k_spmv_dia_outside"]),
    ("k_spmv_sym27", ["k_spmv_sym27"]),
    ("k_spmv_dia", ["k_spmv_dia<"]),
    ("k_spmv_ell", ["k_spmv_ell<"]),
    ("k_spmv_sell", ["k_spmv_sell"]),
    ("k_spmv_bsell", ["k_spmv_bsell"]),
    ("k_spmv_lat27", ["k_spmv_lat27(", "k_spmv_lat27d(", "k_lat27_gather<", "k_lat27_gather_st<", "k_rem_apply"]),  # (+ the skew remainder of a nonsymmetric K, where one is carried)
    ("k_spmv_lat27_pass1", ["k_spmv_lat27(", "k_spmv_lat27d("]),  # the fused CG iteration times pass 1 alone (pass 2 runs inside the residual update: k_lat27_gather_cg)
    ("k_spmv_lat8", ["k_spmv_lat8<", "k_lat8_gather<", "k_lat8_gather_st<", "k_rem_apply"]),  # one SpMV = the tile pass + the pass that sums the tiles' y blocks (+ the remainder)
]
ASSEMBLY = ["k_brick_nitsche", "k_rem_", "k_probe", "k_thermal_matrix", "k_thermal_residual", "k_elasticity_matrix", "k_elasticity_residual", "k_elast", "k_hex27", "k_dia_vals", "k_ell_vals",
            "k_sell_vals", "k_sell_fill", "k_bsell_fill", "k_l27_fill", "k_l27d_fill", "k_l8_fill", "k_symp_bind", "k_jacobi", "k_ell_diag", "k_sell_diag", "k_mat_div"]
out = {}
legs = sorted({os.path.basename(p)[:-len("_FETCH_SIZE")] for p in glob.glob(os.path.join(src, "*_FETCH_SIZE"))})
for leg in legs:
    F = means(os.path.join(src, leg + "_FETCH_SIZE"))
    W = means(os.path.join(src, leg + "_WRITE_SIZE"))
    kb = lambda M, k, c: M.get(k, {}).get(c, (0, 0.0))
    cal = [k for k in F if "k_axpby" in k or "axpby" in k.lower()]
    log = open(os.path.join(src, leg + "_FETCH_SIZE.log")).read() if os.path.exists(os.path.join(src, leg + "_FETCH_SIZE.log")) else ""
    info = {}
    for ln in log.splitlines():
        if ln.startswith("LEG "):
            tok = ln.split()
            info = {tok[i]: int(tok[i + 1]) for i in range(2, len(tok) - 1, 2)}
    n = info.get("n", 0)
    calib = None
    if cal and n:
        c = cal[0]
        calib = {"kernel": c.split("(")[0], "expected_read_bytes": 2 * n * 8, "FETCH_SIZE_x2_bytes": kb(F, c, "FETCH_SIZE")[1] * 1024 * 2,
                 "expected_write_bytes": n * 8, "WRITE_SIZE_bytes": kb(W, c, "WRITE_SIZE")[1] * 1024}
    per_kernel = {}
    for k in sorted(set(F) | set(W)):
        nl, f = kb(F, k, "FETCH_SIZE")
        _, w = kb(W, k, "WRITE_SIZE")
        per_kernel[k.split("(")[0][:90]] = {"launches": nl, "fetch_bytes_x2": f * 1024 * 2, "write_bytes": w * 1024, "hbm_bytes": f * 2048 + w * 1024}
    for key, subs in KEYS:
        ks = [k for k in set(F) | set(W) if any(s in k for s in subs if s != "k_rem_apply")]
        if not ks:
            continue
        if "k_rem_apply" in subs:  # the skew remainder rides with whichever tile kernel of the leg it follows
            ks += [k for k in set(F) | set(W) if "k_rem_apply" in k]
        main_launches = max(kb(F, k, "FETCH_SIZE")[0] for k in ks)
        # a "launch" = the main kernel + its companion launch (rows outside the swept planes), each at its own mean
        fetch = sum(kb(F, k, "FETCH_SIZE")[1] for k in ks) * 1024 * 2
        write = sum(kb(W, k, "WRITE_SIZE")[1] for k in ks) * 1024
        out[f"{key}@{leg}"] = {"kernels": [k.split("(")[0][:90] for k in ks], "launches_measured": main_launches,
                               "fetch_bytes_x2": fetch, "write_bytes": write, "hbm_bytes_per_launch": fetch + write,
                               "design_bytes_csr_kernel": info.get("csr_design_bytes") if key == "csr_kernel" else None,
                               "tree": tree, "workload": {**info, "leg": leg}, "calibration": calib,
                               "correction": "gfx950: FETCH_SIZE x 2 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as read; separate --pmc passes"}
    out[f"all_kernels@{leg}"] = {"tree": tree, "calibration": calib, "per_kernel": per_kernel,
                                 "assembly_and_bind": {k: v for k, v in per_kernel.items() if any(s in k for s in ASSEMBLY)}}
# matrix-core counters of the hex-27 Ke kernel
for tag in ("MFMA", "SQ"):
    d = os.path.join(src, "c4_128_" + tag)
    if os.path.isdir(d):
        M = means(d)
        out[f"hex27_counters_{tag}@c4_128"] = {"tree": tree, "per_kernel": {k.split("(")[0][:90]: {c: {"launches": v[0], "mean": v[1]} for c, v in cs.items()}
                                                                              for k, cs in M.items() if "hex27" in k}}
json.dump(out, open(dst, "w"), indent=1)
for k, v in out.items():
    if "hbm_bytes_per_launch" in v:
        print(f"{k:28s} {v['hbm_bytes_per_launch'] / 1e9:8.3f} GB per launch  ({', '.join(v['kernels'])})")
