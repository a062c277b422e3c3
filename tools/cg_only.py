import sys
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metafem_jl_amd as mf
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
its = int(sys.argv[2]) if len(sys.argv) > 2 else 50
variants = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
brick = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N))
A = brick.pattern(1)
K = brick.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
b = torch.ones(A.n, dtype=torch.float64, device="cuda")
for var in variants:
    for _ in range(2):
        _, st = mf.iterative_Solve(A, K, b, 1e-30, Sv_func=mf.cg_, maxiter=its, max_pass=1, fixed_iterations=True, cg_variant=var)
        print(st.solve_ms / its, "ms/it", "cg_variant", var, "solve_ms", st.solve_ms)
import ctypes as C
from metafem_jl_amd import _lib
log = (C.c_double * 4)()
_lib.lib.mfem_debug_ws_trial_log(mf.default_context()._h, log)
print("workspace candidates tried (ms for two SpMVs):", [round(v, 3) for v in log[:3]])
