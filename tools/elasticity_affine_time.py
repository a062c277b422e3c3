"""hex-8 elasticity matrix assembly at N^3 with and without the affine-element shortcut (bit 5 of mfem_debug_set_elasticity): ms per assembly, best of 3 x 5.
usage: elasticity_affine_time.py [N]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metafem_jl_amd as mf
from metafem_jl_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
lam, mu = 0.5769230769230769, 0.38461538461538464
brick = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N))
A = brick.pattern(3)
K = torch.empty(A.nnz, dtype=torch.float64, device="cuda")
def timed():
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(5):
            brick.assemble_elasticity(A, lam, mu, 1000.0, mf.FACE_BITS["x0"], out=K)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 5)
    return best
brick.assemble_elasticity(A, lam, mu, 1000.0, mf.FACE_BITS["x0"], out=K)
Ks = {}
for knob, tag in ((0, "affine shortcut"), (1 << 5, "general path"), (0, "affine shortcut"), (1 << 5, "general path")):
    _lib.lib.mfem_debug_set_elasticity(knob)
    ms = timed()
    Ks[tag] = K.clone()
    print(f"N {N} {tag:16s}: {ms:.3f} ms per assembly ({A.nnz * 8 / (ms * 1e-3) / 1e12:.2f} TB/s of the nnz * 8 it writes = {A.nnz * 8 / (ms * 1e-3) / 8e12:.3f} of HBM)", flush=True)
_lib.lib.mfem_debug_set_elasticity(0)
d = float((Ks["affine shortcut"] - Ks["general path"]).abs().max() / Ks["general path"].abs().max())
print(f"max |K_affine - K_general| / max |K| = {d:.2e}")
