"""The fused hex-8 kernels at their BASELINE sizes (thermal matrix + residual 256^3, elasticity matrix + residual 128^3), three
launches each: for rocprofv3 --kernel-trace --stats / --pmc."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metafem_jl_amd as mf
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
b = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N))
A = b.pattern(1)
K = torch.empty(A.nnz, dtype=torch.float64, device="cuda")
x = mf.FEM_rand(A.n, 1, 0) + 300.0
s = torch.full((A.n,), 1600.0, dtype=torch.float64, device="cuda")
R = torch.empty(A.n, dtype=torch.float64, device="cuda")
for _ in range(3):
    b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F, out=K)
    b.residual_thermal(x, 0.6, 25.0, 293.15, 0x3F, s=s, out=R)
torch.cuda.synchronize()
del A, K, b
torch.cuda.empty_cache()
M = N // 2
b = mf.make_Brick((1.0, 1.0, 1.0), (M, M, M))
A = b.pattern(3)
K = torch.empty(A.nnz, dtype=torch.float64, device="cuda")
x = 0.01 * mf.FEM_rand(A.n, 2, 0)
R = torch.empty(A.n, dtype=torch.float64, device="cuda")
for _ in range(3):
    b.assemble_elasticity(A, 0.5769, 0.3846, 1000.0, mf.FACE_BITS["x0"], out=K)
    b.residual_elasticity(x, 0.5769, 0.3846, 1000.0, mf.FACE_BITS["x0"], mf.FACE_BITS["y1"], (0.0, 1.0, 0.0, 0.0, 0.0, 0.3), out=R)
torch.cuda.synchronize()
print("done")
