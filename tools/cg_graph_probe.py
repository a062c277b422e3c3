"""CG iteration time at N^3 with and without hipGraph replay of the iteration pairs (off by default above 4 M unknowns)."""
import sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import metafem_jl_amd as mf
from metafem_jl_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
b = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N))
A = b.pattern(1)
K = b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
rhs = torch.ones(A.n, dtype=torch.float64, device="cuda")
for graphs in (0, 1, 0, 1):
    _lib.lib.mfem_debug_set_graphs(1, (1 << 40) if graphs else 4000000)
    mf.iterative_Solve(A, K, rhs, 1e-30, Sv_func=mf.cg_, maxiter=50, max_pass=1, fixed_iterations=True)
    xs, st = mf.iterative_Solve(A, K, rhs, 1e-30, Sv_func=mf.cg_, maxiter=200, max_pass=1, fixed_iterations=True)
    print(f"N={N} graphs {'on ' if graphs else 'off'}: {st.solve_ms / 200:.4f} ms per iteration, |x| {float(xs.norm()):.12e}", flush=True)
