#!/bin/bash
# Round 5: the SQ / TA / TCP / TCC counters of the CSR kernels behind mul! as they are today, the same groups as
# profiles/r02_csr_hex27_rowblock_counters.json: hex-27 128^3 (k_spmv_csr_rb), hex-8 elasticity 128^3 (k_spmv_csr_rb), hex-8 thermal 256^3 (k_spmv_csr_w).
# One rocprofv3 --pmc pass per group (tools/run_pmc.sh), per-kernel means -> gpurun_out/pmc_csr_r05_<leg>.json -> profiles/r05_csr_counters.json
R=$GRAFT_REPO_ROOT
G1="SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"
G2="SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD"
G3="TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
G4="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
G7="GRBM_GUI_ACTIVE TA_BUSY_avr TA_TA_BUSY_sum"
G8="SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM"
bash $R/tools/run_pmc.sh csr_r05_c4 "tools/csr_hex27_once.py 128" "$G1" "$G2" "$G3" "$G4" "FETCH_SIZE" "WRITE_SIZE" "$G7" "$G8" > $R/gpurun_out/pmc_csr_r05_c4.txt 2>&1 || exit 1
echo "done c4"
bash $R/tools/run_pmc.sh csr_r05_c3 "tools/csr_c3_once.py 128" "$G1" "$G2" "$G3" "$G4" "FETCH_SIZE" "WRITE_SIZE" "$G7" "$G8" > $R/gpurun_out/pmc_csr_r05_c3.txt 2>&1 || exit 1
echo "done c3"
bash $R/tools/run_pmc.sh csr_r05_c2 "tools/csr_once.py 0 256 5" "$G1" "$G2" "$G3" "$G4" "FETCH_SIZE" "WRITE_SIZE" "$G7" "$G8" > $R/gpurun_out/pmc_csr_r05_c2.txt 2>&1 || exit 1
echo "done c2"
python3 - <<'PY'
import json, os
R = os.environ["GRAFT_REPO_ROOT"]
out = {}
for leg in ("c4", "c3", "c2"):
    d = json.load(open(f"{R}/gpurun_out/pmc_csr_r05_{leg}.json"))
    out[leg] = {k: {c: v["mean"] for c, v in cs.items()} | {"launches": max(v["launches"] for v in cs.values())} for k, cs in d.items() if "k_spmv_csr" in k}
json.dump(out, open(f"{R}/gpurun_out/r05_csr_counters.json", "w"), indent=1)
for leg, ks in out.items():
    for k, v in ks.items():
        print(leg, k[:60]); [print(f"    {c:34s} {x:.5g}") for c, x in sorted(v.items())]
PY
