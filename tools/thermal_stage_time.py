"""hex-8 thermal matrix sweep at N^3: rows staged through LDS and written as contiguous streams (default) against the per-thread write-out of round 3
(bit 1 of mfem_debug_set_hex8_thermal); ms per assembly (no Robin faces), best of 3 x 5, and the two matrices compared bit for bit.  usage: thermal_stage_time.py N [N ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metafem_jl_amd as mf
from metafem_jl_amd import _lib
for N in [int(a) for a in sys.argv[1:]] or [256]:
    brick = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N))
    A = brick.pattern(1)
    K = torch.empty(A.nnz, dtype=torch.float64, device="cuda")
    def timed():
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(5):
                brick.assemble_thermal(A, 0.6, 0.0, 293.15, 0, out=K)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 5)
        return best
    brick.assemble_thermal(A, 0.6, 0.0, 293.15, 0, out=K)
    Ks = {}
    for knob, tag in ((0, "staged rows"), (2, "per-thread rows"), (0, "staged rows"), (2, "per-thread rows")):
        _lib.lib.mfem_debug_set_hex8_thermal(knob)
        ms = timed()
        Ks[tag] = K.clone() if N <= 256 else None
        print(f"N {N} {tag:16s}: {ms:.3f} ms per assembly = {A.nnz * 8 / (ms * 1e-3) / 8e12:.3f} of HBM on the nnz * 8 it writes", flush=True)
    xs = mf.FEM_rand(A.n, 1, 0); R = torch.empty_like(xs)
    def timed_res():
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(5):
                brick.residual_thermal(xs, 0.6, 0.0, 293.15, 0, s=xs, out=R)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 5)
        return best
    for knob, tag in ((0, "affine shortcut"), (4, "general path"), (0, "affine shortcut"), (4, "general path")):
        _lib.lib.mfem_debug_set_hex8_thermal(knob)
        msK = timed()
        print(f"N {N} {tag:16s}: matrix {msK:.3f} ms, residual {timed_res():.3f} ms", flush=True)
    _lib.lib.mfem_debug_set_hex8_thermal(0)
    if N <= 256:
        print(f"N {N}: bitwise equal: {bool(torch.equal(Ks['staged rows'], Ks['per-thread rows']))}", flush=True)
    del brick, A, K, Ks
    torch.cuda.empty_cache()
