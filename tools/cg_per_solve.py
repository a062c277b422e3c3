"""Per-solve work and per-iteration time of CG at N^3 for the given cg_variants: two fixed-iteration solves (64 and 264 iterations), best of 3.
usage: cg_per_solve.py N variants   e.g. 256 3,4"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metafem_jl_amd as mf
from metafem_jl_amd import _lib
_lib.lib.mfem_debug_set_ws_trial(1)  # (512^3: compare on the same kind of backing memory)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [3, 4]
brick = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N))
A = brick.pattern(1)
K = brick.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
b = torch.ones(A.n, dtype=torch.float64, device="cuda")
def best(its, var):
    t = 1e9
    for _ in range(3):
        _, st = mf.iterative_Solve(A, K, b, 1e-30, Sv_func=mf.cg_, maxiter=its, max_pass=1, fixed_iterations=True, cg_variant=var)
        t = min(t, st.solve_ms)
    return t
for var in variants * 2:
    t0, t1 = best(64, var), best(264, var)
    per = (t1 - t0) / 200
    print(f"cg_variant {var}: per iteration {per:.4f} ms, per-solve work {t0 - 64 * per:.3f} ms")
