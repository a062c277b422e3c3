"""A/B of the XCD-strip walk of the CSR kernel behind mul! (k_spmv_csr_w, round 5) against the round-robin walk: bitwise equality of y and time per
launch on the hex-8 thermal matrices (512^3: strips are the default there; 256^3: forced for the comparison).
usage: csr_strips_ab.py [n ...]    -> profiles/r05_csr_strips.txt"""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import metafem_jl_amd as mf
from metafem_jl_amd import _lib

for N in [int(a) for a in sys.argv[1:]] or [256, 512]:
    b = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N), 1, 3)
    A = b.pattern(1)
    K = b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
    x = mf.FEM_rand(A.n, 3, 0) - 0.5
    ys, ts = {}, {}
    for on in (1, 0, 1, 0):
        _lib.lib.mfem_debug_set_csr_strips(on, 0 if on else -1)  # (threshold 0: strips whatever the plane size)
        y = torch.zeros(A.n, dtype=torch.float64, device="cuda")
        for _ in range(3):
            mf.mul_(y, A, K, x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            mf.mul_(y, A, K, x)
        e1.record()
        torch.cuda.synchronize()
        ts.setdefault(on, []).append(e0.elapsed_time(e1) / 20)
        ys[on] = y
    _lib.lib.mfem_debug_set_csr_strips(1, 3 << 20)
    byts, cols = A.spmv_bytes()
    t1, t0 = min(ts[1]), min(ts[0])
    print(f"hex-8 thermal {N}^3  n {A.n} nnz {A.nnz} design bytes {byts / 1e9:.3f} GB:  XCD strips {t1:.4f} ms ({byts / t1 / 1e6 / 8000:.3f} of 8 TB/s)   round-robin {t0:.4f} ms "
          f"({byts / t0 / 1e6 / 8000:.3f})   bitwise equal: {bool(torch.equal(ys[0], ys[1]))}", flush=True)
    assert torch.equal(ys[0], ys[1])
    del b, A, K, x, ys
    torch.cuda.empty_cache()
