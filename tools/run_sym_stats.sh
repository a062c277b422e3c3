cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/sym_stats -o out --output-format csv -- python3 $R/tools/probe_sym.py 256 > $R/gpurun_out/sym_stats.log 2>&1
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/sym_stats/out_kernel_stats.csv")))
for r in rows[:8]:
    print(r["Name"][:60], r["Calls"], r["TotalDurationNs"], r["AverageNs"])
PY
