#!/bin/bash
export TMPDIR=/tmp; R=$(pwd); mkdir -p gpurun_out
timeout -k 10 850 python -m pytest tests/test_gpu_krylov.py tests/test_gpu_fullsize.py -x -q -k "sym or sweep or fingerprint or t512 or c2_256 or patch" > gpurun_out/t6.log 2>&1 || { echo "tests failed"; tail -40 gpurun_out/t6.log | cut -c1-250; exit 1; }
tail -3 gpurun_out/t6.log
for n in 256 512; do python tools/per_solve.py $n 6 20 2>&1 | grep solve_ms; done
