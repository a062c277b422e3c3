"""gpurun_out/pmc_<tag>.json (tools/run_pmc.sh) -> a compact per-kernel table: python tools/make_pmc_table.py <tag> <out.json> [kernel substring ...]"""
import json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, out = sys.argv[1], sys.argv[2]
want = sys.argv[3:]
d = json.load(open(os.path.join(R, "gpurun_out", f"pmc_{tag}.json")))
res = {k: {c: v["mean"] for c, v in cs.items()} | {"launches": max(v["launches"] for v in cs.values())}
       for k, cs in d.items() if not want or any(w in k for w in want)}
json.dump(res, open(os.path.join(R, out), "w"), indent=1)
for k, v in res.items():
    print(k[:70], {c: f"{x:.4g}" for c, x in v.items()})
