#!/bin/bash
# Regenerates everything under profiles/ that comes from a GPU run in round 3 (use through gpurun: outputs land in gpurun_out/, copy
# the summaries you want judged into profiles/ afterwards).  Each rocprofv3 invocation runs `python3 <script>` directly (no env /
# bash -c hop), counters in their own passes with --kernel-trace only.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
cd $R
bash tools/run_pmc_r03.sh > gpurun_out/r03_pmc.log 2>&1                                       # FETCH_SIZE / WRITE_SIZE of every priced kernel -> gpurun_out/r03_traffic.json
cp gpurun_out/r03_traffic.json profiles/r03_traffic.json                                        # (so that the bench lines below carry `traffic`)
python bench.py 2> gpurun_out/bench_err.log | tail -1 > gpurun_out/r03_bench_n1.json            # default bench line (512^3 headline + 256^3 + hex-27 + CPU)
python bench.py --config c3 2>> gpurun_out/bench_err.log | tail -1 > gpurun_out/r03_bench_c3.json
python bench.py --config c4 2>> gpurun_out/bench_err.log | tail -1 > gpurun_out/r03_bench_c4.json
bash tools/run_bench_profile.sh > gpurun_out/bench_profile.log 2>&1                             # rocprofv3 --stats of the bench
python tools/configs_roofline.py > gpurun_out/r03_c3_c4_roofline.json 2>> gpurun_out/bench_err.log
bash tools/run_solve_trace.sh > gpurun_out/r03_solve_trace.txt 2>&1                             # kernels of one solve per config
python tools/el_time.py > gpurun_out/r03_el_time.txt 2>&1
ls -la gpurun_out | tail -12
