"""PMC probe: CG solve at 256^3 with the symmetric sweep SpMV (default) -- FETCH_SIZE / WRITE_SIZE per kernel."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metafem_jl_amd as mf
N = 256
brick = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N))
A = brick.pattern(1)
K = brick.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
b = torch.ones(A.n, dtype=torch.float64, device="cuda")
_, st = mf.iterative_Solve(A, K, b, 1e-30, Sv_func=mf.cg_, maxiter=20, max_pass=1, fixed_iterations=True)
print("nnz", A.nnz, "n", A.n, "ms/it", st.solve_ms / 20)
