#!/bin/bash
export TMPDIR=/tmp; R=$(pwd); mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_krylov.py tests/test_gpu_nitsche.py tests/test_gpu_remainder.py tests/test_gpu_slab.py -x -q > gpurun_out/t16.log 2>&1 || { echo "tests failed"; tail -30 gpurun_out/t16.log | cut -c1-250; exit 1; }
tail -2 gpurun_out/t16.log
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_u20f -o u20 --output-format csv -- python3 $R/tools/u20_leg.py 96 1 2 > $R/gpurun_out/u20f.log 2>&1
grep -E "ki_ortho|ki_step_end|kk_sign_dots|ki_omega|ki_store" $R/gpurun_out/prof_u20f/u20_kernel_stats.csv | cut -c1-40,90-200
grep "^{" $R/gpurun_out/u20f.log | python3 -c "import sys,json; o=json.loads(sys.stdin.readline()); print({k: round(o[k],2) for k in ('value','ms_per_step','assembly_ms','residual_ms','solve_ms_per_step')})"
