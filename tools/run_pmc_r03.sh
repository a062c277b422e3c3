# usage (on the GPU box): bash tools/run_pmc_r03.sh [legs...]     default legs: c2_256 c2_512 c3_128 c4_128
# One rocprofv3 --pmc pass per counter group and leg (FETCH_SIZE and WRITE_SIZE cannot share a pass; never combined with trace
# domains other than --kernel-trace; the program itself directly after `--`).  Raw CSVs under gpurun_out/pmc_r03/, the summary
# profiles/r03_traffic.json is written by tools/make_r03_traffic.py (run at the end; copy it back from gpurun_out/).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
legs=${@:-"c2_256 c2_512 c3_128 c4_128"}
mkdir -p $R/gpurun_out/pmc_r03
for leg in $legs; do
  for grp in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/pmc_r03/${leg}_$grp -o out --output-format csv -- python3 $R/tools/pmc_leg.py $leg > $R/gpurun_out/pmc_r03/${leg}_$grp.log 2>&1 || { echo "pass $leg $grp failed"; tail -5 $R/gpurun_out/pmc_r03/${leg}_$grp.log; exit 1; }
    echo "done $leg $grp"
  done
done
# hex-27 Ke kernel: matrix-core counters (own pass)
if echo "$legs" | grep -q c4_128; then
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_r03/c4_128_MFMA -o out --output-format csv -- python3 $R/tools/pmc_leg.py c4_128 1 > $R/gpurun_out/pmc_r03/c4_128_MFMA.log 2>&1 || { echo "MFMA pass failed"; tail -5 $R/gpurun_out/pmc_r03/c4_128_MFMA.log; exit 1; }
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU -d $R/gpurun_out/pmc_r03/c4_128_SQ -o out --output-format csv -- python3 $R/tools/pmc_leg.py c4_128 1 > $R/gpurun_out/pmc_r03/c4_128_SQ.log 2>&1 || echo "SQ pass failed (non-fatal)"
  echo "done c4_128 MFMA/SQ"
fi
python3 $R/tools/make_r03_traffic.py $R/gpurun_out/pmc_r03 $R/gpurun_out/r03_traffic.json
