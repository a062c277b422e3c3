"""CG iterations on the hex-27 N^3 matrix with a chosen mfem_debug_set_sell knob (for rocprofv3 passes).  usage: sell_once.py [knob] [N] [iters]"""
import sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import metafem_jl_amd as mf
from metafem_jl_amd import _lib
knob = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0
N = int(sys.argv[2]) if len(sys.argv) > 2 else 128
its = int(sys.argv[3]) if len(sys.argv) > 3 else 10
_lib.lib.mfem_debug_set_sell(1 | knob)
b = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N), 2, 5)
A = b.pattern(1)
K = b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
rhs = torch.ones(A.n, dtype=torch.float64, device="cuda")
xs, st = mf.iterative_Solve(A, K, rhs, 1e-30, Sv_func=mf.cg_, maxiter=its, max_pass=1, fixed_iterations=True)
torch.cuda.synchronize()
print("n", A.n, "nnz", A.nnz, "ms/it", st.solve_ms / its)
