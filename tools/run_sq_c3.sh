cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/sq_c3
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES -d $R/gpurun_out/sq_c3/a -o out --output-format csv -- python3 $R/tools/pmc_leg.py c3_128 1 > $R/gpurun_out/sq_c3/a.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE -d $R/gpurun_out/sq_c3/b -o out --output-format csv -- python3 $R/tools/pmc_leg.py c3_128 1 > $R/gpurun_out/sq_c3/b.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/sq_c3/s -o out --output-format csv -- python3 $R/tools/pmc_leg.py c3_128 1 > $R/gpurun_out/sq_c3/s.log 2>&1
python3 - <<'PY'
import csv,glob,collections,os
R=os.environ["GRAFT_REPO_ROOT"]
for p in ("a","b"):
    agg=collections.defaultdict(list)
    for f in glob.glob(f"{R}/gpurun_out/sq_c3/{p}/**/*counter_collection.csv",recursive=True):
        for r in csv.DictReader(open(f)):
            if "elasticity" in r["Kernel_Name"] or "k_dia_vals" in r["Kernel_Name"] or "k_mat_div" in r["Kernel_Name"] or "jacobi" in r["Kernel_Name"]:
                agg[(r["Kernel_Name"][:40],r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k,v in sorted(agg.items()): print(k, len(v), sum(v)/len(v))
for f in glob.glob(f"{R}/gpurun_out/sq_c3/s/**/*kernel_stats.csv",recursive=True):
    for i,r in enumerate(csv.DictReader(open(f))):
        if i<14: print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
