#!/bin/bash
# wave-time counters of chosen kernels of one pmc_leg.py workload: where do their waves spend their cycles (issue, memory wait, LDS)?
#   usage: bash tools/run_pmc_wave_time.sh <leg: c2_256 | c2_512 | c3_128 | c4_128> <kernel name substring> [more substrings]     -> gpurun_out/pmc_wave_<leg>.txt
# Counter passes only with --kernel-trace; the program itself directly after `--`.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
leg=$1; shift
O=$R/gpurun_out/pmc_wave_$leg
mkdir -p $O
i=0
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" \
         "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c -d $O/pass$i -o out --output-format csv -- python3 $R/tools/pmc_leg.py $leg 2 > $O/pass$i.log 2>&1 || { echo "pass $i failed"; tail -3 $O/pass$i.log; }
done
python3 - $O "$@" > $R/gpurun_out/pmc_wave_$leg.txt <<'PY'
import csv, collections, glob, sys
pats = sys.argv[2:]
for d in sorted(glob.glob(sys.argv[1] + "/pass*/out_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(d)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        if not any(p in k for p in pats): continue
        print(k)
        for c, v in cs.items():
            print(f"    {c:36s} launches {len(v):3d} mean {sum(v) / len(v):16.0f}")
PY
cat $R/gpurun_out/pmc_wave_$leg.txt
