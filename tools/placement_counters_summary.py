"""Per-dispatch table of the placement experiment: for every k_spmv_symp<0> dispatch of tools/placement_counters.py (in dispatch order: warm + 2
timed per workspace candidate) its duration (kernel trace) and the counters of the pass.  usage: placement_counters_summary.py <pass dir> ..."""
import collections
import csv
import glob
import os
import sys

for d in sys.argv[1:]:
    print(f"== {os.path.basename(d.rstrip('/'))}")
    dur = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Dispatch_Id"]] = (r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
    rows = collections.OrderedDict()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_spmv_symp<0>" not in r["Kernel_Name"]:
                continue
            rows.setdefault(int(r["Dispatch_Id"]), collections.OrderedDict())
            key = r["Counter_Name"]
            rows[int(r["Dispatch_Id"])][key] = rows[int(r["Dispatch_Id"])].get(key, 0.0) + float(r["Counter_Value"])
    names = []
    for v in rows.values():
        for k in v:
            if k not in names:
                names.append(k)
    print("dispatch  ms      " + "  ".join(f"{n:>28s}" for n in names))
    for i, (did, v) in enumerate(sorted(rows.items())):
        ms = dur.get(str(did), ("", float("nan")))[1]
        print(f"{did:8d}  {ms:6.3f}  " + "  ".join(f"{v.get(n, float('nan')):28.0f}" for n in names))
    log = d.rstrip("/") + ".log"
    if os.path.exists(log):
        for ln in open(log):
            if ln.startswith("TRIAL") or ln.startswith("ws trial"):
                print(ln.rstrip())
