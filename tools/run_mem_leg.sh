# usage: run_mem_leg.sh <leg> <kernel substring>    memory-path counters (four passes, at most four counters of a block each -- more is refused by the hardware: L1 = TCP, address / data units = TA / TD, L2 = TCC)
# of the kernels of a pmc_leg.py leg whose name contains the substring
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
leg=$1; pat=$2
O=$R/gpurun_out/mem_$leg
mkdir -p $O
timeout -k 10 240 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE -d $O/a -o out --output-format csv -- python3 $R/tools/pmc_leg.py $leg 1 > $O/a.log 2>&1 || echo "pass a failed: $(grep -m1 -i "error code" $O/a.log)"
timeout -k 10 240 rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum -d $O/b -o out --output-format csv -- python3 $R/tools/pmc_leg.py $leg 1 > $O/b.log 2>&1 || echo "pass b failed: $(grep -m1 -i "error code" $O/b.log)"
timeout -k 10 240 rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum -d $O/c -o out --output-format csv -- python3 $R/tools/pmc_leg.py $leg 1 > $O/c.log 2>&1 || echo "pass c failed: $(grep -m1 -i "error code" $O/c.log)"
timeout -k 10 240 rocprofv3 --kernel-trace --pmc TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum TA_DATA_STALLED_BY_TC_CYCLES_sum -d $O/d -o out --output-format csv -- python3 $R/tools/pmc_leg.py $leg 1 > $O/d.log 2>&1 || echo "pass d failed: $(grep -m1 -i "error code" $O/d.log)"
python3 - "$leg" "$pat" <<'PY'
import csv,glob,collections,os,sys
R=os.environ["GRAFT_REPO_ROOT"]; leg,pat=sys.argv[1],sys.argv[2]
for p in ("a","b","c","d"):
    agg=collections.defaultdict(list)
    for f in glob.glob(f"{R}/gpurun_out/mem_{leg}/{p}/**/*counter_collection.csv",recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                agg[(r["Kernel_Name"].split("(")[0][:40],r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k,v in sorted(agg.items()): print(f"{leg:8s} {k[0]:42s} {k[1]:36s} x{len(v)} {sum(v)/len(v):.5g}")
PY
