#!/bin/bash
# A/B of library variants on ONE box: tools/ab_libs.sh "<script.py args>" <dir with lib*.so> [grep pattern]; the shipped library is restored afterwards
set -e
cp metafem.jl_amd/libmetafem_mi355x.so /tmp/lib_keep.so
trap 'cp /tmp/lib_keep.so metafem.jl_amd/libmetafem_mi355x.so' EXIT  # restored on EVERY way out: a failing variant or a grep miss must not leave an experiment in the tree
for rep in 1 2; do
  for f in "$2"/lib*.so; do
    cp "$f" metafem.jl_amd/libmetafem_mi355x.so
    echo "== $(basename $f)"
    python $1 | grep -E "${3:-matrix ms [0-9]}"
  done
done
