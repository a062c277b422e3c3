"""On-box probe: every variant of the CSR kernel behind mul! (mfem_spmv_csr) on the hex-8 256^3 and hex-27 128^3 matrices,
timed with events around 30 launches; prints ms, GB/s on SURVEY 8(d)'s CSR bytes and the fraction of 8 TB/s.
usage: probe_csr.py [hex8|hex27|both] [n] [variants,...] [grid_mults,...]"""
import json, sys
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import metafem_jl_amd as mf
from metafem_jl_amd import _lib

which = sys.argv[1] if len(sys.argv) > 1 else "both"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 0
variants = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 4, 5, 6, 7]
mults = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0]
out = []
for kind in (["hex8", "hex27"] if which == "both" else [which]):
    n = N or (256 if kind == "hex8" else 128)
    b = mf.make_Brick((1.0, 1.0, 1.0), (n, n, n), 1 if kind == "hex8" else 2, 3 if kind == "hex8" else 5)
    A = b.pattern(1)
    K = b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
    x = mf.FEM_rand(A.n, 3, 0)
    y = torch.zeros(A.n, dtype=torch.float64, device="cuda")
    bytes_ = A.nnz * 12 + A.n * 16 + (A.n + 1) * 8
    ref = None
    for var in variants:
        for mult in mults:
            _lib.lib.mfem_debug_set_spmv(var << 16, mult)
            for _ in range(3):
                mf.mul_(y, A, K, x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                mf.mul_(y, A, K, x)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 30
            if ref is None:
                ref = y.clone()
            err = float((y - ref).abs().max() / ref.abs().max())
            rec = {"matrix": f"{kind} {n}^3", "variant": var, "wg_per_cu": mult, "ms": round(ms, 4), "GBps": round(bytes_ / ms / 1e6, 1),
                   "frac_of_8TBps": round(bytes_ / ms / 1e6 / 8000, 4), "err_vs_first": err}
            print(json.dumps(rec), flush=True)
            out.append(rec)
    _lib.lib.mfem_debug_set_spmv(0, 8)
    del b, A, K, x, y
    torch.cuda.empty_cache()
