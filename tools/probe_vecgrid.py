"""CG iteration time at 256^3 vs the persistent grid of the vector kernels."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metafem_jl_amd as mf
from metafem_jl_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
brick = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N))
A = brick.pattern(1)
K = brick.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
b = torch.ones(A.n, dtype=torch.float64, device="cuda")
for mult in (3, 2, 4, 6, 8, 3):
    _lib.lib.mfem_debug_set_vec_grid(mult)
    mf.iterative_Solve(A, K, b, 1e-30, Sv_func=mf.cg_, maxiter=200, max_pass=1, fixed_iterations=True)
    _, st = mf.iterative_Solve(A, K, b, 1e-30, Sv_func=mf.cg_, maxiter=200, max_pass=1, fixed_iterations=True)
    print(f"vector kernels with {mult} workgroups per CU: {st.solve_ms/200:.4f} ms per CG iteration", flush=True)
