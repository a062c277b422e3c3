import sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import metafem_jl_amd as mf
from metafem_jl_amd import _lib
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for w in ("c3", "c4"):
    for var in (0, 7, 3):
        _lib.lib.mfem_debug_set_spmv(var << 16, 0)
        if w == "c3":
            b = mf.make_Brick((1.0, 1.0, 1.0), (128, 128, 128), 1, 3); A = b.pattern(3); K = b.assemble_elasticity(A, 0.5769230769230769, 0.38461538461538464, 1000.0, mf.FACE_BITS["x0"])
        else:
            b = mf.make_Brick((1.0, 1.0, 1.0), (128, 128, 128), 2, 5); A = b.pattern(1); K = b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
        x = mf.FEM_rand(A.n, 3, 0); y = torch.zeros(A.n, dtype=torch.float64, device="cuda")
        ms = t(lambda: mf.mul_(y, A, K, x))
        byts, cols = A.spmv_bytes()
        print(f"{w} variant {var}: {ms:.3f} ms  {byts / ms / 1e6 / 8000:.3f}", flush=True)
        del b, A, K, x, y
        torch.cuda.empty_cache()
_lib.lib.mfem_debug_set_spmv(0, 0)
