"""Time of the per-solve layout copy alone (k_dia_vals through mfem_spmv_solver_layout's bind) at N^3: hip events around 5 binds, minus the SpMV each bind is followed by
(measured separately on the bound layout through the CG loop is not possible here, so: total of bind + one SpMV, and the SpMV alone from a 40-iteration solve).
usage: dia_vals_only.py N"""
import os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metafem_jl_amd as mf
from metafem_jl_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
brick = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N))
A = brick.pattern(1)
K = brick.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
x = mf.FEM_rand(A.n, 5, 0) - 0.5
y = torch.empty_like(x)
def bind_ms(reps=5):
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps):
            _lib.check(_lib.lib.mfem_spmv_solver_layout(brick.ctx._h, A._h, K.data_ptr(), x.data_ptr(), y.data_ptr(), 1.0, 0.0))
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best
_lib.check(_lib.lib.mfem_spmv_solver_layout(brick.ctx._h, A._h, K.data_ptr(), x.data_ptr(), y.data_ptr(), 1.0, 0.0))
print(f"N {N}: bind (layout copy + symmetry check) + one SpMV: {bind_ms():.3f} ms", flush=True)
