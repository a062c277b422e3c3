#!/bin/bash
# SQ counters + kernel stats of the row-owner path of general hex-27 elements (k_hex27_gq_lane + k_hex27_rows_gq) on a fully distorted 128^3 mesh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/rows_pmc
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/rows_pmc/stats -o r --output-format csv -- python3 $R/tools/hex27_rows_once.py 128 5 > $R/gpurun_out/rows_pmc/stats.log 2>&1 || exit 1
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_MOPS_F64" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/rows_pmc/p$i -o out --output-format csv -- python3 $R/tools/hex27_rows_once.py 128 2 > $R/gpurun_out/rows_pmc/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<'PY'
import csv, collections, glob, os
R = os.environ["GRAFT_REPO_ROOT"]
for f in glob.glob(f"{R}/gpurun_out/rows_pmc/stats/**/r_kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:6]: print(r["Name"][:60], r["Calls"], r["AverageNs"])
agg = collections.defaultdict(list)
for f in glob.glob(f"{R}/gpurun_out/rows_pmc/p*/**/out_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "hex27_rows" in r["Kernel_Name"] or "gq_lane" in r["Kernel_Name"]: agg[(r["Kernel_Name"][:20], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()): print(f"{k:22s} {c:34s} {sum(v)/len(v):.5g}")
PY
