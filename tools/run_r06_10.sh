#!/bin/bash
export TMPDIR=/tmp; R=$(pwd); mkdir -p gpurun_out
timeout -k 10 850 python -m pytest tests/test_gpu_u20.py tests/test_gpu_unstructured.py tests/test_gpu_generic.py tests/test_gpu_physics.py tests/test_gpu_cylinder.py -x -q > gpurun_out/t10.log 2>&1 || { echo "tests failed"; tail -40 gpurun_out/t10.log | cut -c1-250; exit 1; }
tail -2 gpurun_out/t10.log
timeout -k 10 600 python tools/u20_assembly_ab.py 96 1,3 2>&1 | grep "^fields"
