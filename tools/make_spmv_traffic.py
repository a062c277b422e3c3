"""gpurun_out/pmc_ell_summary.json (tools/run_pmc_ell.sh) -> profiles/spmv_traffic.json: HBM bytes per SpMV of the solver-layout kernel(s)
inside the CG loop at 256^3, FETCH_SIZE doubled (gfx950 counts coalesced streaming reads at 1/2), the calibration on k_cg_update beside it."""
import json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(os.path.join(R, "gpurun_out", "pmc_ell_summary.json")))
F, W = d["FETCH_SIZE"], d["WRITE_SIZE"]
def pick(tag):
    return [k for k in F if tag in k]
sweep = pick("k_spmv_symp<0>") or pick("k_spmv_sym27")
outside = pick("k_spmv_dia_outside")
kind = 2 if pick("k_spmv_symp<0>") else 1
ks = sweep + outside
fetch = sum(F[k]["mean_KB"] for k in ks) * 1024 * 2
write = sum(W[k]["mean_KB"] for k in ks) * 1024
cal = pick("k_cg_update")[0]
n, nnz = 16974593, 454756609
out = {
    "kernel": " + ".join(k.split("(")[0] for k in ks) + " (per SpMV: the sweep launch incl. the rows outside the swept planes)",
    "solver_layout_mode": 2, "symmetric_sweep": kind,
    "workload": f"hex-8 256^3 thermal K (n={n}, nnz={nnz}), inside the CG loop",
    "FETCH_SIZE_KB_mean": {k: F[k]["mean_KB"] for k in ks}, "WRITE_SIZE_KB_mean": {k: W[k]["mean_KB"] for k in ks},
    "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of coalesced streaming reads (MI355X_MICROARCH.md HBM section); calibration in "
                  f"the same run on k_cg_update: 3 vectors read = {3 * n * 8 / 1e6:.1f} MB, FETCH_SIZE x 2 = {F[cal]['mean_KB'] * 2048 / 1e6:.1f} MB; "
                  f"1 vector written = {n * 8 / 1e6:.1f} MB, WRITE_SIZE = {W[cal]['mean_KB'] * 1024 / 1e6:.1f} MB",
    "hbm_bytes_per_launch": fetch + write,
    "note": "tools/run_pmc_ell.sh + tools/make_spmv_traffic.py; separate --pmc passes for FETCH_SIZE and WRITE_SIZE",
    "all_kernels": d,
}
json.dump(out, open(os.path.join(R, "profiles", "spmv_traffic.json"), "w"), indent=1)
print(out["kernel"], out["hbm_bytes_per_launch"])
