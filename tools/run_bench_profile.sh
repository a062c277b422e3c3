# rocprofv3 kernel stats of the bench command restricted to the 256^3 workload (no 512^3 / hex-27 / CPU legs, so that the per-kernel
# averages are those of the timed workload and agree with the line's live avg_launch_ms); round-tagged copies go to profiles/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/bench_stats -o out --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --target-n 0 --cpu-n 0 --hex27-n 0 > $R/gpurun_out/bench_stats.json 2> $R/gpurun_out/bench_stats.err
tail -1 $R/gpurun_out/bench_stats.json | cut -c1-600
head -8 $R/gpurun_out/bench_stats/out_kernel_stats.csv | cut -c1-160
