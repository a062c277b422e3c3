cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in "FETCH_SIZE" "WRITE_SIZE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_sell_$tag -o out --output-format csv -- python3 $R/tools/pmc_sell.py > $R/gpurun_out/pmc_sell_$tag.log 2>&1
done
python3 - <<PY
import csv, collections, glob
for d in sorted(glob.glob("$R/gpurun_out/pmc_sell_*/out_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(d)):
        agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        if "spmv" not in k: continue
        print(k, {c: (len(v), sum(v) / len(v)) for c, v in cs.items()})
PY
