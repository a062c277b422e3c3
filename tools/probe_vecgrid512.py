"""CG iteration time at 512^3 (hex-8 thermal) against the persistent-grid size of the streaming vector kernels (mfem_debug_set_vec_grid).
usage: probe_vecgrid512.py [N]"""
import sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import metafem_jl_amd as mf
from metafem_jl_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
b = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N), 1, 3)
A = b.pattern(1)
K = b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
rhs = mf.FEM_rand(A.n, 5, 0) - 0.5
def solve(it):
    best = 1e9
    for _ in range(2):
        _, st = mf.iterative_Solve(A, K, rhs, 1e-300, Sv_func=mf.cg_, maxiter=it, max_pass=1, fixed_iterations=True)
        best = min(best, st.solve_ms)
    return best
for g in (3, 2, 4, 5, 6, 8, 12, 16, 3):
    _lib.lib.mfem_debug_set_vec_grid(g)
    a, c = solve(20), solve(80)
    print(f"vec grid {g:2d} workgroups per CU: CG iteration {(c - a) / 60:.4f} ms", flush=True)
_lib.lib.mfem_debug_set_vec_grid(3)
