#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout -k 10 850 python -m pytest tests/test_gpu_lat27.py tests/test_gpu_hex27.py -x -q -k "not c4_hex27_128" > gpurun_out/t8.log 2>&1 || { echo "tests failed"; tail -40 gpurun_out/t8.log | cut -c1-250; exit 1; }
tail -3 gpurun_out/t8.log
python tools/lat27_ab.py 2>&1 | grep -v amdgpu.ids
