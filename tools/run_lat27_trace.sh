# kernel times of the C4 leg (hex-27 128^3, 20 CG iterations on the lattice-tile layout)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/lat27_trace
rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/t -o out --output-format csv -- python3 $R/tools/pmc_leg.py ${LEG:-c4_128} 1 > $O/t.log 2>&1 || echo "trace pass failed"
python3 - <<'PY'
import csv,glob,os
R=os.environ["GRAFT_REPO_ROOT"]; O=f"{R}/gpurun_out/lat27_trace"
for f in glob.glob(f"{O}/t/**/*kernel_stats.csv",recursive=True):
    for r in list(csv.DictReader(open(f))):
        if any(k in r['Name'] for k in ('lat27','l27','lat8','l8','k_cg','sell','bicg','k_dia','k_spmv','k_ax','k_dot','probe')): print(f"{r['Name'][:60]:60s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:10.1f} pct {r['Percentage']}")
PY
