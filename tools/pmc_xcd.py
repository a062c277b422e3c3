"""PMC probe: SpMV with the dispatcher round-robin tile map vs XCD-contiguous runs (run length from argv), 12 launches each."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metafem_jl_amd as mf
from metafem_jl_amd import _lib
N = 256
brick = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N))
A = brick.pattern(1)
K = brick.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
x = mf.FEM_rand(A.n, 1, 0); y = torch.empty_like(x)
for run in [0] + [int(v) for v in sys.argv[1:]]:
    _lib.lib.mfem_debug_set_spmv((1 << 16) | run, 8)
    for _ in range(12): mf.mul_(y, A, K, x)
    torch.cuda.synchronize()
    brick.ctx.sync() if hasattr(brick.ctx, "sync") else None
    z = torch.zeros(1024, device="cuda"); z += 1  # marker kernel between the groups
torch.cuda.synchronize()
