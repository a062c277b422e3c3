"""The unstructured hex-20 legs of bench.py alone (for rocprofv3 --kernel-trace --stats and A/B runs).  usage: u20_leg.py [n = 96] [fields = 1,3] [steps = 3] [iters = 200] [shape = CUBE | SIMPLEX]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import bench_legs as L  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
fields = [int(f) for f in (sys.argv[2] if len(sys.argv) > 2 else "1,3").split(",")]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 200
shape = sys.argv[5] if len(sys.argv) > 5 else "CUBE"  # "SIMPLEX": the tet-10 legs
args = bench.parse_args(["--iters", str(iters)])
if os.environ.get("MFEM_SELL_KNOB"):  # A/B: bit 3 = field-periodic blocks read their whole column stream
    from metafem_jl_amd import _lib
    _lib.lib.mfem_debug_set_sell(1 | int(os.environ["MFEM_SELL_KNOB"]))
if os.environ.get("MFEM_BSELL"):  # A/B: 0 = the row-sorted sliced layout instead of the node-blocked one
    from metafem_jl_amd import _lib
    _lib.lib.mfem_debug_set_bsell(int(os.environ["MFEM_BSELL"]))
B = L.Bench(args)
for f in fields:
    o = B.unstructured_leg(n, f, steps, shape=shape)
    o["roofline"].pop("kernel_note", None)
    o["csr_kernel"].pop("kernel_note", None)
    print(json.dumps({k: v for k, v in o.items() if k != "workload"}, default=str), flush=True)
