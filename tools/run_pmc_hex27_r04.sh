#!/bin/bash
# Round 4: where the waves of the hex-27 pass-1 kernel spend their cycles (SQ wave-time counters; MI355X_MICROARCH.md: WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY
# ~ WAVE_CYCLES, quad-cycle units) and what they issue.  Counter passes with --kernel-trace only, the program itself after `--`.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_hex27_r04
mkdir -p $O
i=0
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR" \
         "SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c -d $O/pass$i -o out --output-format csv -- python3 $R/tools/hex27_only.py 128 > $O/pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 $O/pass$i.log; }
done
python3 - $O <<'PY'
import csv, collections, glob, sys
for d in sorted(glob.glob(sys.argv[1] + "/pass*/out_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(d)):
        agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        if "hex27" not in k: continue
        print(k)
        for c, v in cs.items():
            print(f"    {c:36s} launches {len(v):3d} mean {sum(v) / len(v):16.0f}")
PY
