cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/hex27_stats -o out --output-format csv -- python3 $R/tools/hex27_only.py 128 ${HEX27_VARIANT:-0} > $R/gpurun_out/hex27_stats.log 2>&1
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/hex27_stats/out_kernel_stats.csv")))
for r in rows[:14]:
    print(r["Name"][:70], r["Calls"], r["TotalDurationNs"], r["AverageNs"])
PY
