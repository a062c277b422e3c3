#!/bin/bash
export TMPDIR=/tmp; R=$(pwd); mkdir -p gpurun_out
timeout -k 10 850 python -m pytest tests/test_gpu_u20.py -x -q > gpurun_out/u20_tests.log 2>&1 || { echo "u20 tests failed"; tail -40 gpurun_out/u20_tests.log | cut -c1-250; exit 1; }
tail -3 gpurun_out/u20_tests.log
cd /tmp
for n in 256 512; do
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_ps$n -o ps --output-format csv -- python3 $R/tools/per_solve.py $n 6 20 > $R/gpurun_out/ps$n.log 2>&1 || { echo "per_solve $n failed"; tail -5 $R/gpurun_out/ps$n.log; exit 1; }
rm -f $R/gpurun_out/prof_ps$n/*kernel_trace.csv
grep "solve_ms" $R/gpurun_out/ps$n.log
python3 - $R/gpurun_out/prof_ps$n/ps_kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print(f"{r['Name'][:70]:70s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e6:9.4f} ms total {float(r['TotalDurationNs'])/1e6:9.3f} ms")
PY
done
