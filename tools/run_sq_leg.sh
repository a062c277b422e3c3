# usage: run_sq_leg.sh <leg> <kernel substring>    SQ counters (two passes) of the kernels of a pmc_leg.py leg whose name contains the substring
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
leg=$1; pat=$2
mkdir -p $R/gpurun_out/sq_$leg
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES -d $R/gpurun_out/sq_$leg/a -o out --output-format csv -- python3 $R/tools/pmc_leg.py $leg 1 > $R/gpurun_out/sq_$leg/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE -d $R/gpurun_out/sq_$leg/b -o out --output-format csv -- python3 $R/tools/pmc_leg.py $leg 1 > $R/gpurun_out/sq_$leg/b.log 2>&1
python3 - "$leg" "$pat" <<'PY'
import csv,glob,collections,os,sys
R=os.environ["GRAFT_REPO_ROOT"]; leg,pat=sys.argv[1],sys.argv[2]
for p in ("a","b"):
    agg=collections.defaultdict(list)
    for f in glob.glob(f"{R}/gpurun_out/sq_{leg}/{p}/**/*counter_collection.csv",recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                agg[(r["Kernel_Name"].split("(")[0][:40],r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k,v in sorted(agg.items()): print(f"{k[0]:42s} {k[1]:26s} x{len(v)} {sum(v)/len(v):.4g}")
PY
