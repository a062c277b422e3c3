"""Pass 1 of the lattice tiles (mode 5), steps per register buffer 8 / 12 / 16 (bits 4-5 of mfem_debug_set_lat8): SpMV pair ms (hip events around every
product of a 200-step solve) on C3 (three fields, 128^3, bicgstabl_GS!(2)) and on the one-field 256^3 matrix (idrs!(8))."""
import ctypes as C
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import metafem_jl_amd as mf
from metafem_jl_amd import _lib

lam, mu = 0.5769230769230769, 0.38461538461538464
ctx = mf.default_context()


def timed(A, K, R, **kw):
    tot, cnt = C.c_double(), C.c_int64()
    _lib.check(_lib.lib.mfem_prof_spmv_enable(ctx._h, 1))
    _lib.check(_lib.lib.mfem_prof_spmv_read(ctx._h, C.byref(tot), C.byref(cnt), 1))
    ms = []
    for _ in range(4):
        _, st = mf.iterative_Solve(A, K, R, 1e-300, Pr_func=mf.Pr_Jacobi_, max_pass=1, fixed_iterations=True, **kw)
        ms.append(st.solve_ms)
    _lib.check(_lib.lib.mfem_prof_spmv_read(ctx._h, C.byref(tot), C.byref(cnt), 1))
    _lib.check(_lib.lib.mfem_prof_spmv_enable(ctx._h, 0))
    return tot.value / max(cnt.value, 1), sorted(ms)[1]


b = mf.make_Brick((1.0, 1.0, 1.0), (128, 128, 128), 1, 3)
A = b.pattern(3)
K = b.assemble_elasticity(A, lam, mu, 1000.0, mf.FACE_BITS["x0"])
R = b.residual_elasticity(torch.zeros(A.n, dtype=torch.float64, device="cuda"), lam, mu, 1000.0, mf.FACE_BITS["x0"], mf.FACE_BITS["y1"], (0.0, 1.0, 0.0, 0.0, 0.0, 0.0))
for ch in (0, 0):
    _lib.lib.mfem_debug_set_lat8(1 | (ch << 4))
    sp, sv = timed(A, K, R, Sv_func=mf.bicgstabl_GS_, maxiter=100, s=2)
    print(f"c3_128 chunk {(8, 12, 16)[ch]:2d}: SpMV pair {sp:.4f} ms, 200-step solve {sv:.2f} ms", flush=True)
del b, A, K, R
torch.cuda.empty_cache()
b = mf.make_Brick((1.0, 1.0, 1.0), (256, 256, 256), 1, 3)
A = b.pattern(1)
K = b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
R = b.residual_thermal(torch.zeros(A.n, dtype=torch.float64, device="cuda"), 0.6, 25.0, 293.15, 0x3F, s=torch.full((A.n,), 1600.0, dtype=torch.float64, device="cuda"))
for ch in (0, 0):
    _lib.lib.mfem_debug_set_lat8(1 | (ch << 4))
    sp, sv = timed(A, K, R, Sv_func=mf.idrs_, maxiter=200, s=8)
    print(f"c2_256 (one field, idrs!(8)) chunk {(8, 12, 16)[ch]:2d}: SpMV pair {sp:.4f} ms, 200-step solve {sv:.2f} ms", flush=True)
_lib.lib.mfem_debug_set_lat8(1)
