# rocprofv3 kernel stats of the default bench command (round-tagged copy goes to profiles/)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/bench_stats -o out --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 > $R/gpurun_out/bench_stats.json 2> $R/gpurun_out/bench_stats.err
tail -1 $R/gpurun_out/bench_stats.json | cut -c1-600
head -8 $R/gpurun_out/bench_stats/out_kernel_stats.csv | cut -c1-160
