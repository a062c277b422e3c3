# usage: run_bench_stats_cfg.sh c3|c4     rocprofv3 kernel stats of `bench.py --config <cfg>` (3 steps, no CPU leg, traffic from the committed file):
# the per-kernel averages of the timed workload, to be compared with the line's live avg_launch_ms (two launches per SpMV on the lattice tiles)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
c=$1
rm -rf $R/gpurun_out/bench_stats_$c
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/bench_stats_$c -o out --output-format csv -- python3 $R/bench.py --config $c --steps 3 --warmup 1 --cpu-n 0 --live-traffic 0 > $R/gpurun_out/bench_stats_$c.json 2> $R/gpurun_out/bench_stats_$c.err
tail -1 $R/gpurun_out/bench_stats_$c.json | cut -c1-300
head -6 $R/gpurun_out/bench_stats_$c/out_kernel_stats.csv | cut -c1-140
