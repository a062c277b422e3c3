#!/bin/bash
# round 6, second GPU call: idrs! with generated sign shadow vectors + fused update/combine; SELL sort keeps mesh order on unstructured patterns
R=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_krylov.py tests/test_gpu_lat8.py tests/test_gpu_lat27.py tests/test_gpu_remainder.py tests/test_gpu_nitsche.py tests/test_gpu_unstructured.py tests/test_gpu_generic.py -x -q > gpurun_out/t2.log 2>&1 || { echo "tests failed"; tail -40 gpurun_out/t2.log; exit 1; }
tail -3 gpurun_out/t2.log
timeout -k 10 600 python tools/idrs_ab.py > gpurun_out/r06_idrs_streams.txt 2>&1 || { echo "idrs ab failed"; tail -20 gpurun_out/r06_idrs_streams.txt; exit 1; }
cat gpurun_out/r06_idrs_streams.txt
timeout -k 10 400 python tools/u20_leg.py 96 1,3 2 > gpurun_out/u20_96b.log 2>&1 || { echo "u20 failed"; tail -20 gpurun_out/u20_96b.log; exit 1; }
python - <<'PY'
import json
for ln in open('gpurun_out/u20_96b.log'):
    if ln.startswith('{'):
        o=json.loads(ln); print({k:o[k] for k in ('n_dof','value','ms_per_step','solve_ms_per_step','assembly_ms','residual_ms','final_res','initial_res')}, o['roofline']['kernel_key'], round(o['roofline']['avg_launch_ms'],3), round(o['roofline']['frac'],3), 'csr', round(o['csr_kernel']['avg_launch_ms'],3), round(o['csr_kernel']['frac'],3))
PY
