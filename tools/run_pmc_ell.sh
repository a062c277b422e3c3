cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_ell_$c -o out --output-format csv -- python3 $R/tools/pmc_sym.py > $R/gpurun_out/pmc_ell_$c.log 2>&1
done
python3 - <<PY
import csv, collections, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = list(csv.DictReader(open("$R/gpurun_out/pmc_ell_%s/out_counter_collection.csv" % c)))
    agg = collections.defaultdict(list)
    for r in rows:
        agg[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    out[c] = {k: {"launches": len(v), "mean_KB": sum(v) / len(v)} for k, v in agg.items()}
    for k, v in out[c].items():
        print(c, k, v)
json.dump(out, open("$R/gpurun_out/pmc_ell_summary.json", "w"), indent=1)
PY
