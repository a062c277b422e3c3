#!/bin/bash
# Round 6, on the GPU box: everything profiles/r06_* is made from (the BASELINE configs + the reference's own solver / boundary-condition legs of bench.py: REF_LEGS).
#   1. HBM traffic of the priced kernels (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, + the hex-27 MFMA / SQ passes) -> gpurun_out/r06_traffic.json
#   2. rocprofv3 --kernel-trace --stats of one bench leg per config (512^3 alone, 256^3 alone, c3, c4) -> gpurun_out/prof_r06/<leg>_kernel_stats.csv
#   3. the bench lines: default invocation (driver's command), --config c3, --config c4
# Counter passes only with --kernel-trace; the program itself directly after `--`.   usage: bash tools/run_profiles_r06.sh [pmc|stats|lines ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
what=${@:-"pmc stats lines"}
mkdir -p $R/gpurun_out/pmc_r06 $R/gpurun_out/prof_r06
if echo "$what" | grep -q pmc; then
  for leg in c2_256 c2_512 c3_128 c4_128 ref_idrs8_256 nitsche_c2_256 nitsche_c4_128 u20_1_96 u20_3_96 tet10_1_64 tet10_3_64; do
    for grp in FETCH_SIZE WRITE_SIZE; do
      timeout -k 10 400 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/pmc_r06/${leg}_$grp -o out --output-format csv -- python3 $R/tools/pmc_leg.py $leg > $R/gpurun_out/pmc_r06/${leg}_$grp.log 2>&1 || { echo "pass $leg $grp failed"; tail -5 $R/gpurun_out/pmc_r06/${leg}_$grp.log; exit 1; }
      echo "done $leg $grp"
    done
  done
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_r06/c4_128_MFMA -o out --output-format csv -- python3 $R/tools/pmc_leg.py c4_128 1 > $R/gpurun_out/pmc_r06/c4_128_MFMA.log 2>&1 || { echo "MFMA pass failed"; exit 1; }
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU -d $R/gpurun_out/pmc_r06/c4_128_SQ -o out --output-format csv -- python3 $R/tools/pmc_leg.py c4_128 1 > $R/gpurun_out/pmc_r06/c4_128_SQ.log 2>&1 || echo "SQ pass failed (non-fatal)"
  python3 $R/tools/make_r03_traffic.py $R/gpurun_out/pmc_r06 $R/gpurun_out/r06_traffic.json || exit 1
  echo "traffic summary written"
fi
if echo "$what" | grep -q stats; then
  i=0
  for leg in "c2_512:--config c2 --secondary-n 0 --secondary-configs 0 --ref-legs 0 --hex27-n 0 --u20-n 0 --tet10-n 0" "c2_256:--config c2 --n 256 --secondary-n 0 --secondary-configs 0 --hex27-n 0" "c3_128:--config c3" "c4_128:--config c4" "ref_idrs8_256:--config ref_idrs8" "nitsche_c2_256:--config nitsche_c2" "nitsche_c4_128:--config nitsche_c4"; do
    name=${leg%%:*}; args=${leg#*:}
    timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r06/$name -o bench --output-format csv -- python3 $R/bench.py $args --steps 3 --warmup 1 --live-traffic 0 --cpu-n 0 --full-out gpurun_out/prof_r06/${name}_full.json > $R/gpurun_out/prof_r06/${name}_under_rocprof.json 2> $R/gpurun_out/prof_r06/${name}.err || { echo "stats $name failed"; tail -3 $R/gpurun_out/prof_r06/${name}.err; exit 1; }
    cp $R/gpurun_out/prof_r06/$name/bench_kernel_stats.csv $R/gpurun_out/prof_r06/${name}_kernel_stats.csv
    rm -f $R/gpurun_out/prof_r06/$name/bench_kernel_trace.csv
    echo "done stats $name"
  done
  # the unstructured hex-20 legs (tools/u20_leg.py = bench_legs.Bench.unstructured_leg alone)
  for f in 1 3; do
    timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r06/u20_$f -o bench --output-format csv -- python3 $R/tools/u20_leg.py 96 $f 2 > $R/gpurun_out/prof_r06/u20_${f}_96_under_rocprof.json 2> $R/gpurun_out/prof_r06/u20_$f.err || { echo "stats u20 $f failed"; tail -3 $R/gpurun_out/prof_r06/u20_$f.err; exit 1; }
    cp $R/gpurun_out/prof_r06/u20_$f/bench_kernel_stats.csv $R/gpurun_out/prof_r06/u20_${f}_96_kernel_stats.csv
    rm -f $R/gpurun_out/prof_r06/u20_$f/bench_kernel_trace.csv
    echo "done stats u20_$f"
  done
  for f in 1 3; do  # ... and on the tet-10 mesh
    timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r06/tet10_$f -o bench --output-format csv -- python3 $R/tools/u20_leg.py 64 $f 2 200 SIMPLEX > $R/gpurun_out/prof_r06/tet10_${f}_64_under_rocprof.json 2> $R/gpurun_out/prof_r06/tet10_$f.err || { echo "stats tet10 $f failed"; tail -3 $R/gpurun_out/prof_r06/tet10_$f.err; exit 1; }
    cp $R/gpurun_out/prof_r06/tet10_$f/bench_kernel_stats.csv $R/gpurun_out/prof_r06/tet10_${f}_64_kernel_stats.csv
    rm -f $R/gpurun_out/prof_r06/tet10_$f/bench_kernel_trace.csv
    echo "done stats tet10_$f"
  done
fi
if echo "$what" | grep -q lines; then
  timeout -k 10 400 python3 $R/bench.py --steps 20 --warmup 5 --full-out gpurun_out/r06_bench_n1_full.json > $R/gpurun_out/r06_bench_n1.json 2> $R/gpurun_out/r06_bench_n1.err || { echo "default line failed"; tail -3 $R/gpurun_out/r06_bench_n1.err; exit 1; }
  timeout -k 10 200 python3 $R/bench.py --config c3 --steps 10 --warmup 2 --full-out gpurun_out/r06_bench_c3_full.json > $R/gpurun_out/r06_bench_c3.json 2> $R/gpurun_out/r06_bench_c3.err || { echo "c3 line failed"; exit 1; }
  timeout -k 10 200 python3 $R/bench.py --config c4 --steps 10 --warmup 2 --full-out gpurun_out/r06_bench_c4_full.json > $R/gpurun_out/r06_bench_c4.json 2> $R/gpurun_out/r06_bench_c4.err || { echo "c4 line failed"; exit 1; }
  echo "done lines"
fi
