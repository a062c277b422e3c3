# usage: run_pmc.sh <tag> "<program and args after python3>" "<counters of pass 1>" ["<counters of pass 2>" ...]
# One rocprofv3 --pmc pass per counter group (never combined with trace domains other than --kernel-trace); per-kernel means
# go to gpurun_out/pmc_<tag>.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; prog=$2; shift 2
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/pmc_${tag}_$i -o out --output-format csv -- python3 $R/$prog > $R/gpurun_out/pmc_${tag}_$i.log 2>&1
done
python3 - "$tag" "$i" <<'PY'
import csv, collections, json, os, sys, glob
tag, npass = sys.argv[1], int(sys.argv[2])
R = os.environ["GRAFT_REPO_ROOT"]
out = collections.defaultdict(dict)
for i in range(1, npass + 1):
    for f in glob.glob(f"{R}/gpurun_out/pmc_{tag}_{i}/**/out_counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"][:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in agg.items():
            out[k][c] = {"launches": len(v), "mean": sum(v) / len(v)}
json.dump(out, open(f"{R}/gpurun_out/pmc_{tag}.json", "w"), indent=1)
for k, d in out.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:40s} {v['mean']:.4g}  (x{v['launches']})")
PY
