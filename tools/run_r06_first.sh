#!/bin/bash
# round 6, first GPU call: the bench tests (line schema), the default line, the u20 legs under rocprofv3 --stats
R=$(pwd); mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
cd $R
true
true
timeout -k 10 600 python -m pytest tests/test_gpu_bench.py -x -q -k "single_gpu_line or default_line or two_ranks_on_one" > gpurun_out/bench_tests.log 2>&1 || { echo "bench tests failed"; tail -40 gpurun_out/bench_tests.log; exit 1; }
tail -3 gpurun_out/bench_tests.log
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_u20 -o u20 --output-format csv -- python3 $R/tools/u20_leg.py 96 1,3 2 > $R/gpurun_out/u20_96.log 2>&1 || { echo "u20 96 failed"; tail -20 $R/gpurun_out/u20_96.log; exit 1; }
cd $R
rm -f gpurun_out/prof_u20/*kernel_trace.csv
tail -2 gpurun_out/u20_96.log | cut -c1-900
head -25 gpurun_out/prof_u20/*kernel_stats.csv | cut -c1-200
timeout -k 10 600 python bench.py --steps 10 --warmup 3 > gpurun_out/r06_bench_first.json 2> gpurun_out/r06_bench_first.err || { echo "bench failed"; tail -20 gpurun_out/r06_bench_first.err; exit 1; }
wc -c gpurun_out/r06_bench_first.json; cat gpurun_out/r06_bench_first.json
