cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_xcd_$c -o out --output-format csv -- python3 $R/tools/pmc_xcd.py 512 4096 > $R/gpurun_out/pmc_xcd_$c.log 2>&1
done
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pmc_xcd_stats -o out --output-format csv -- python3 $R/tools/pmc_xcd.py 512 4096 > $R/gpurun_out/pmc_xcd_stats.log 2>&1
ls -R $R/gpurun_out/pmc_xcd_FETCH_SIZE | head
