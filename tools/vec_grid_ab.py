import sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import metafem_jl_amd as mf
from metafem_jl_amd import _lib
N = 256
b = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N), 1, 3)
A = b.pattern(1)
K = b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
s = torch.full((A.n,), 1600.0, dtype=torch.float64, device="cuda")
R = b.residual_thermal(torch.zeros(A.n, dtype=torch.float64, device="cuda"), 0.6, 25.0, 293.15, 0x3F, s=s)
for mult in (3, 4, 5, 6, 8, 3):
    _lib.lib.mfem_debug_set_vec_grid(mult)
    best = 1e9
    for _ in range(3):
        x, st = mf.iterative_Solve(A, K, R, 1e-300, Sv_func=mf.idrs_, maxiter=200, max_pass=1, s=8, fixed_iterations=True)
        best = min(best, st.solve_ms)
    bb = 1e9
    for _ in range(3):
        x, st = mf.iterative_Solve(A, K, R, 1e-300, Sv_func=mf.bicgstabl_GS_, maxiter=100, max_pass=1, s=2, fixed_iterations=True)
        bb = min(bb, st.solve_ms)
    print(f"vec grid {mult} WG/CU: idrs8 200 steps {best:.1f} ms   bicgstabl2 100 steps {bb:.1f} ms", flush=True)
