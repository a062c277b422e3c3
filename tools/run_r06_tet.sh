#!/bin/bash
export TMPDIR=/tmp; R=$(pwd); mkdir -p gpurun_out
cd /tmp
for f in 1 3; do
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_tet -o tet --output-format csv -- python3 $R/tools/tet10_leg.py 64 $f > $R/gpurun_out/tet.log 2>&1
grep "^tet-10 64" $R/gpurun_out/tet.log | cut -c1-220
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$R/gpurun_out/prof_tet/tet_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:14]:
    print('  ', r['Name'][:80].ljust(80), r['Calls'].rjust(6), f"{float(r['TotalDurationNs'])/1e6:9.1f} ms", f"{float(r['AverageNs'])/1e3:9.1f} us", f"{100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
done
