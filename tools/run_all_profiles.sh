#!/bin/bash
# Regenerates everything under profiles/ that comes from a GPU run (use through gpurun: outputs land in gpurun_out/, copy the
# summaries you want judged into profiles/ afterwards).  Each rocprofv3 invocation runs `python3 <script>` directly (no env /
# bash -c hop), counters in their own passes with --kernel-trace only.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
cd $R
python bench.py 2> gpurun_out/bench_err.log | tail -1 > gpurun_out/bench_n1.json               # default bench line
python bench.py --n 512 --steps 2 --warmup 1 --cpu-n 0 2>> gpurun_out/bench_err.log | tail -1 > gpurun_out/bench_512.json
bash tools/run_bench_profile.sh > gpurun_out/bench_profile.log 2>&1                           # rocprofv3 --stats of the bench
bash tools/run_pmc_ell.sh > gpurun_out/pmc.log 2>&1                                           # FETCH_SIZE / WRITE_SIZE of the SpMV
(python tools/probe_c3.py 128; python tools/probe_hex27.py 128) > gpurun_out/other_configs.log 2>&1
python tools/probe_ell.py 256 > gpurun_out/ell.log 2>&1                                       # CSR vs slot-major vs diagonal slots
python tools/probe_small_solve.py > gpurun_out/small_solve.log 2>&1                           # launch-bound regime, graphs on/off
python tools/probe_generic.py 96 > gpurun_out/generic.log 2>&1                                # generic S3 path at scale
ls -la gpurun_out | tail -20
