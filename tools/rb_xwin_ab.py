"""A/B of the x-window form of the row-block CSR kernel (k_spmv_csr_rb, round 5) against its gather form: bitwise equality of y and time per launch
on the BASELINE matrices that take that kernel (configs[2]: 81-entry rows; configs[3]: 27..125-entry rows) and smaller / odd-sized ones.
usage: rb_xwin_ab.py [n]    -> profiles/r05_csr_rb_xwin.txt"""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import metafem_jl_amd as mf
from metafem_jl_amd import _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
lam, mu = 0.5769230769230769, 0.38461538461538464


def run(name, A, K, launches=20):
    x = mf.FEM_rand(A.ncols, 3, 0) - 0.5
    ys, ts = {}, {}
    for on in (1, 0, 1, 0):
        _lib.lib.mfem_debug_set_rb_xwin(on)
        y = torch.zeros(A.n, dtype=torch.float64, device="cuda")
        for _ in range(3):
            mf.mul_(y, A, K, x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(launches):
            mf.mul_(y, A, K, x, 1.0, 0.0)
        e1.record()
        torch.cuda.synchronize()
        ts.setdefault(on, []).append(e0.elapsed_time(e1) / launches)
        ys[on] = y
    _lib.lib.mfem_debug_set_rb_xwin(1)
    same = bool(torch.equal(ys[0], ys[1]))
    byts, cols = A.spmv_bytes()
    t1, t0 = min(ts[1]), min(ts[0])
    print(f"{name:34s} n {A.n:10d} nnz {A.nnz:11d}  windows {t1:7.4f} ms ({byts / t1 / 1e6 / 8000:.3f} of 8 TB/s on {byts / 1e9:.2f} GB)  gathers {t0:7.4f} ms "
          f"({byts / t0 / 1e6 / 8000:.3f})  bitwise equal: {same}", flush=True)
    assert same


for dims in ((N, N, N), (N // 2 + 3, N // 2, N // 2 - 5)):
    b = mf.make_Brick((1.0, 1.0, 1.0), dims, 1, 3)
    A = b.pattern(3)
    K = b.assemble_elasticity(A, lam, mu, 1000.0, mf.FACE_BITS["x0"])
    run(f"hex-8 elasticity {dims}", A, K)
    del b, A, K
    b = mf.make_Brick((1.0, 1.0, 1.0), dims, 2, 5)
    A = b.pattern(1)
    K = b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
    run(f"hex-27 thermal {dims}", A, K)
    del b, A, K
    torch.cuda.empty_cache()
