#!/bin/bash
# Builds variants of the library that differ in the row-block CSR kernel's tile capacity / gathers in flight / resident workgroups / waves per SIMD into ab_tmp/
# (HERE, on the CPU: hipcc cross-compiles), to be timed on one box with tools/ab_libs.sh "tools/rb_time.py" ab_tmp "matrix ms".
# usage: bash tools/rb_sweep.sh
set -e
cd "$(dirname "$0")/../metafem.jl_amd/csrc"
mkdir -p ../../ab_tmp
build() {  # name, flags
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-value $2 -c spmv.hip -o /tmp/spmv_$1.o
  objs=$(ls *.o | grep -v '^spmv\.o$' | tr '\n' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../ab_tmp/lib_$1.so $objs /tmp/spmv_$1.o -L/opt/rocm/lib -lrccl
  echo "built $1"
}
build cap1536_ng16_wg8_eu2 ""
build cap1024_ng8_wg12_eu3 "-DRB_CAP=1024 -DRB_NG=8 -DRB_WG_PER_CU=12 -DRB_WAVES_PER_EU=3"
build cap1024_ng12_wg12_eu3 "-DRB_CAP=1024 -DRB_NG=12 -DRB_WG_PER_CU=12 -DRB_WAVES_PER_EU=3"
build cap1280_ng8_wg12_eu3 "-DRB_CAP=1280 -DRB_NG=8 -DRB_WG_PER_CU=12 -DRB_WAVES_PER_EU=3"
build cap768_ng8_wg16_eu4 "-DRB_CAP=768 -DRB_NG=8 -DRB_WG_PER_CU=16 -DRB_WAVES_PER_EU=4"
build cap1536_ng12_wg8_eu2 "-DRB_NG=12"
build cap1792_ng16_wg8_eu2 "-DRB_CAP=1792"
