#!/bin/bash
export TMPDIR=/tmp; R=$(pwd); mkdir -p gpurun_out
timeout -k 10 850 python -m pytest tests/test_gpu_u20.py tests/test_gpu_unstructured.py tests/test_gpu_generic.py tests/test_gpu_physics.py tests/test_gpu_cylinder.py -x -q > gpurun_out/t9.log 2>&1 || { echo "tests failed"; tail -40 gpurun_out/t9.log | cut -c1-250; exit 1; }
tail -2 gpurun_out/t9.log
cd /tmp
for f in 1 3; do
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_u20c -o u20 --output-format csv -- python3 $R/tools/u20_leg.py 96 $f 2 > $R/gpurun_out/u20c.log 2>&1
grep -E "k_mesh_assemble|k_mesh_gather" $R/gpurun_out/prof_u20c/u20_kernel_stats.csv | cut -c1-40,120-200
grep "^{" $R/gpurun_out/u20c.log | python3 -c "import sys,json; o=json.loads(sys.stdin.readline()); print({k: round(o[k],2) for k in ('value','ms_per_step','assembly_ms','residual_ms')})"
done
