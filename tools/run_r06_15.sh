#!/bin/bash
export TMPDIR=/tmp; R=$(pwd); mkdir -p gpurun_out
cd /tmp
for f in 1 3; do
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_u20e -o u20 --output-format csv -- python3 $R/tools/u20_leg.py 96 $f 2 > $R/gpurun_out/u20e.log 2>&1
grep -E "k_op_res|k_op_var" $R/gpurun_out/prof_u20e/u20_kernel_stats.csv | cut -c1-60,150-260
grep "^{" $R/gpurun_out/u20e.log | python3 -c "import sys,json; o=json.loads(sys.stdin.readline()); print({k: round(o[k],2) for k in ('value','ms_per_step','assembly_ms','residual_ms')})"
done
