#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_lat8.py tests/test_gpu_remainder.py tests/test_gpu_slab.py -x -q > gpurun_out/t4.log 2>&1 || { echo "tests failed"; tail -40 gpurun_out/t4.log; exit 1; }
tail -2 gpurun_out/t4.log
bash tools/ab_libs.sh "tools/lat8_chunk_ab.py" ab_tmp "chunk  8" 2>&1 | grep -v amdgpu.ids
