import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metafem_jl_amd as mf
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
brick = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N), 2, 5)
A = brick.pattern(1)
K = torch.empty(A.nnz, dtype=torch.float64, device="cuda")
if len(sys.argv) > 2:  # assembly variant (mfem_debug_set_hex27)
    from metafem_jl_amd import _lib
    _lib.lib.mfem_debug_set_hex27(int(sys.argv[2]))
for _ in range(2):
    brick.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F, out=K)
x = mf.FEM_rand(A.n, 1, 0); y = torch.empty_like(x)
brick.residual_thermal(x, 0.6, 25.0, 293.15, 0x3F, s=x, out=y)
torch.cuda.synchronize()
