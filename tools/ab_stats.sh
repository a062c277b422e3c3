#!/bin/bash
# per-kernel times of library variants: tools/ab_stats.sh "<script.py args>" <dir with lib*.so>   (rocprofv3 --kernel-trace --stats, top kernels)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cp $R/metafem.jl_amd/libmetafem_mi355x.so /tmp/lib_keep.so
trap 'cp /tmp/lib_keep.so $R/metafem.jl_amd/libmetafem_mi355x.so' EXIT  # restored on every way out
for f in $R/$2/lib*.so; do
  cp "$f" $R/metafem.jl_amd/libmetafem_mi355x.so
  n=$(basename $f .so)
  rm -rf $R/gpurun_out/abs_$n
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/abs_$n -o out --output-format csv -- python3 $R/$1 > /dev/null 2>&1
  echo "== $n"
  head -7 $R/gpurun_out/abs_$n/out_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
done
cp /tmp/lib_keep.so $R/metafem.jl_amd/libmetafem_mi355x.so
