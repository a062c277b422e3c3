"""Pass 1 of the hex-27 lattice tiles (mode 4): the deterministic form (lane = row, phase-major; default) against the four-lanes-per-row kernel (bit 3 of
the "lat27" knob), C4 128^3, CG 200 iterations: pass-1 ms (hip events around every product) and solve ms; and bicgstabl_GS!(2) (two-launch SpMV)."""
import ctypes as C
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import metafem_jl_amd as mf
from metafem_jl_amd import _lib

ctx = mf.default_context()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
b = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N), 2, 5)
A = b.pattern(1)
K = b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
R = b.residual_thermal(torch.zeros(A.n, dtype=torch.float64, device="cuda"), 0.6, 25.0, 293.15, 0x3F, s=torch.full((A.n,), 1600.0, dtype=torch.float64, device="cuda"))


def timed(**kw):
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


for knob, name in ((1, "lane = row, phase-major (deterministic)"), (1 | 8, "four lanes per row (round 5)"), (1, "lane = row"), (1 | 8, "four lanes per row")):
    _lib.lib.mfem_debug_set_lat27(knob)
    sp, sv = timed(Sv_func=mf.cg_, maxiter=200)
    sp2, sv2 = timed(Sv_func=mf.bicgstabl_GS_, maxiter=100, s=2)
    print(f"c4_{N} {name:42s}: CG pass 1 {sp:.4f} ms, 200 iterations {sv:.2f} ms | bicgstabl_GS!(2) SpMV pair {sp2:.4f} ms, 200 steps {sv2:.2f} ms", flush=True)
_lib.lib.mfem_debug_set_lat27(1)
