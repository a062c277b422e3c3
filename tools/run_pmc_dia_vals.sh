#!/bin/bash
# wave-time counters of the per-solve layout copy k_dia_vals (and the kernels around it) at 256^3: where do its waves wait?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_dia_vals
mkdir -p $O
i=0
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" \
         "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE" \
         "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c -d $O/pass$i -o out --output-format csv -- python3 $R/tools/pmc_leg.py c2_256 2 > $O/pass$i.log 2>&1 || { echo "pass $i failed"; tail -3 $O/pass$i.log; }
done
python3 - $O <<'PY'
import csv, collections, glob, sys
for d in sorted(glob.glob(sys.argv[1] + "/pass*/out_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(d)):
        agg[r["Kernel_Name"][:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        if "k_dia_vals" not in k and "k_spmv_symp<1>" not in k: continue
        print(k)
        for c, v in cs.items():
            print(f"    {c:36s} launches {len(v):3d} mean {sum(v) / len(v):16.0f}")
PY
