import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metafem_jl_amd as mf
from metafem_jl_amd import _lib
on = int(sys.argv[1])
_lib.lib.mfem_debug_set_ws_trial(on)
brick = mf.make_Brick((1.0, 1.0, 1.0), (512, 512, 512))
A = brick.pattern(1)
K = brick.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
b = torch.ones(A.n, dtype=torch.float64, device="cuda")
for k in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    _, st = mf.iterative_Solve(A, K, b, 1e-30, Sv_func=mf.cg_, maxiter=64, max_pass=1, fixed_iterations=True)
    torch.cuda.synchronize()
    print(f"trial {on} solve {k}: wall {1e3 * (time.time() - t0):.1f} ms, solve_ms {st.solve_ms:.1f}", flush=True)
