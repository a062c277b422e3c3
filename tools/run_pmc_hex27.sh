# PMC passes over the hex-27 assembly at 128^3 (default two-pass variant): MFMA busy cycles, LDS activity, HBM bytes per kernel.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $c | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_hex27_$tag -o out --output-format csv -- python3 $R/tools/hex27_only.py 128 > $R/gpurun_out/pmc_hex27_$tag.log 2>&1
done
python3 - <<PY
import csv, collections, glob
for d in sorted(glob.glob("$R/gpurun_out/pmc_hex27_*/out_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(d)):
        agg[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        if "hex27" not in k: continue
        print(k, {c: (len(v), sum(v) / len(v)) for c, v in cs.items()})
PY
