#!/bin/bash
# Placement experiment (round 4): durations + translation / fabric counters of the solver SpMV on the workspace candidates of one process.
# Counter passes only with --kernel-trace (no other trace domain), the program itself after `--`.  usage: tools/run_placement_counters.sh [N]
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
N=${1:-512}
O=$R/gpurun_out/placement
mkdir -p $O
cd /tmp && export TMPDIR=/tmp MFEM_WS_TRIAL_VERBOSE=1
python3 $R/tools/placement_counters.py $N > $O/plain.log 2>&1 || { echo "plain run failed"; tail -5 $O/plain.log; exit 1; }
grep -E "TRIAL|ws trial" $O/plain.log
i=0
for c in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE" \
         "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum" \
         "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum TCC_BUSY_sum" \
         "TCC_EA0_RDREQ"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c -d $O/pass$i -o out --output-format csv -- python3 $R/tools/placement_counters.py $N > $O/pass$i.log 2>&1 \
    || { echo "pass $i failed"; tail -5 $O/pass$i.log; }
done
python3 $R/tools/placement_counters_summary.py $O/pass1 $O/pass2 $O/pass3 > $O/summary.txt 2>&1
# per-channel spread of the fabric read requests (pass 4: TCC_EA0_RDREQ with its instance dimension)
python3 - $O/pass4 >> $O/summary.txt 2>&1 <<'PY'
import collections, csv, glob, os, sys
d = sys.argv[1]
print("== pass4: TCC_EA0_RDREQ per dispatch, min / max / mean over the counter's instances (channels)")
rows = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    rd = csv.DictReader(open(f))
    print("columns:", rd.fieldnames)
    for r in rd:
        if "k_spmv_symp<0>" in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])].append(float(r["Counter_Value"]))
for k, v in sorted(rows.items()):
    print(k, len(v), min(v), max(v), sum(v) / len(v))
for ln in open(d + ".log"):
    if ln.startswith("TRIAL"):
        print(ln.rstrip())
PY
cat $O/summary.txt
