# rocprofv3 kernel stats of the bench command restricted to the headline workload (512^3; no 256^3 / hex-27 / CPU legs, so that the
# per-kernel averages are those of the timed workload and agree with the line's live avg_launch_ms); round-tagged copies go to profiles/.
# The csr_kernel object inside the line still runs (20 + 3 launches of k_spmv_csr_w appear in the stats).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/bench_stats -o out --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --secondary-n 0 --cpu-n 0 --hex27-n 0 > $R/gpurun_out/bench_stats.json 2> $R/gpurun_out/bench_stats.err
tail -1 $R/gpurun_out/bench_stats.json | cut -c1-600
head -8 $R/gpurun_out/bench_stats/out_kernel_stats.csv | cut -c1-160
