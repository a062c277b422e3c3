"""Per-solve work of a Newton-sized solve: `reps` x (assembly + a 20-iteration Jacobi-CG solve) on the hex-8 thermal brick, for
rocprofv3 --kernel-trace --stats.  usage: per_solve.py [n = 256] [reps = 6] [iters = 20]"""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import metafem_jl_amd as mf

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
b = mf.make_Brick((1.0, 1.0, 1.0), (n, n, n), 1, 3)
A = b.pattern(1)
K = torch.empty(A.nnz, dtype=torch.float64, device="cuda")
R = torch.empty(A.n, dtype=torch.float64, device="cuda")
s = torch.full((A.n,), 1600.0, dtype=torch.float64, device="cuda")
x0 = torch.zeros(A.n, dtype=torch.float64, device="cuda")
from metafem_jl_amd import _lib
for bit30 in (1, 0, 1, 0):  # A/B on one box: the separate symmetry check pass (bit 30 of the "ell" knob) against the fill's fingerprint (default)
  _lib.lib.mfem_debug_set_ell(1 | (bit30 << 30))
  ms = []
  for r in range(reps):
    b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F, out=K)
    b.residual_thermal(x0, 0.6, 25.0, 293.15, 0x3F, s=s, out=R)
    _, st = mf.iterative_Solve(A, K, R, 1e-300, Sv_func=mf.cg_, Pr_func=mf.Pr_Jacobi_, maxiter=iters, max_pass=1, fixed_iterations=True)
    ms.append(st.solve_ms)
  print(f"n = {n}, {'check pass   ' if bit30 else 'fingerprint  '}: solve_ms per rep {[round(m, 3) for m in ms]}")
_lib.lib.mfem_debug_set_ell(1)
