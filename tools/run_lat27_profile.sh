# kernel times and HBM traffic of the C4 leg (hex-27 128^3, 20 CG iterations on the lattice-tile layout): three rocprofv3 passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/lat27_prof
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/t -o out --output-format csv -- python3 $R/tools/pmc_leg.py c4_128 1 > $O/t.log 2>&1 || echo "trace pass failed"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f -o out --output-format csv -- python3 $R/tools/pmc_leg.py c4_128 1 > $O/f.log 2>&1 || echo "fetch pass failed"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/w -o out --output-format csv -- python3 $R/tools/pmc_leg.py c4_128 1 > $O/w.log 2>&1 || echo "write pass failed"
python3 - <<'PY'
import csv,glob,collections,os
R=os.environ["GRAFT_REPO_ROOT"]; O=f"{R}/gpurun_out/lat27_prof"
for f in glob.glob(f"{O}/t/**/*kernel_stats.csv",recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]:
        print(f"{r['Name'][:60]:60s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:10.1f} pct {r['Percentage']}")
for p,scale,name in (("f",2*1024.0,"FETCH x2 KB->B"),("w",1024.0,"WRITE KB->B")):
    agg=collections.defaultdict(list)
    for f in glob.glob(f"{O}/{p}/**/*counter_collection.csv",recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0][:50]].append(float(r["Counter_Value"]))
    for k,v in sorted(agg.items()):
        if any(s in k for s in ("lat27","l27","axpby","k_cg")): print(f"{name:16s} {k:52s} x{len(v)} {sum(v)/len(v)*scale/1e9:.4f} GB")
PY
