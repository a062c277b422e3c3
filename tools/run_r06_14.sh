#!/bin/bash
export TMPDIR=/tmp; R=$(pwd); mkdir -p gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_u20d -o u20 --output-format csv -- python3 $R/tools/u20_leg.py 96 3 2 > $R/gpurun_out/u20d.log 2>&1
grep -E "k_bsell_fill|k_spmv_bsell" $R/gpurun_out/prof_u20d/u20_kernel_stats.csv | cut -c1-50,150-230
grep "^{" $R/gpurun_out/u20d.log | python3 -c "import sys,json; o=json.loads(sys.stdin.readline()); print({k: round(o[k],2) for k in ('value','ms_per_step','assembly_ms','residual_ms')})"
