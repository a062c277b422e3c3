"""A few launches of the CSR kernel behind mul! on the hex-8 N^3 matrix (for rocprofv3 --pmc passes).
usage: csr_once.py [variant] [N] [launches]"""
import sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import metafem_jl_amd as mf
from metafem_jl_amd import _lib
var = int(sys.argv[1]) if len(sys.argv) > 1 else 0
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
L = int(sys.argv[3]) if len(sys.argv) > 3 else 5
b = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N))
A = b.pattern(1)
K = b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
x = mf.FEM_rand(A.n, 3, 0)
y = torch.zeros(A.n, dtype=torch.float64, device="cuda")
_lib.lib.mfem_debug_set_spmv(var << 16, 8)
for _ in range(L):
    mf.mul_(y, A, K, x)
torch.cuda.synchronize()
print("nnz", A.nnz, "n", A.n)
