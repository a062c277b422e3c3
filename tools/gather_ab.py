"""A / B of pass 2 of the lattice-tile SpMVs in one process: the staged gather (default) against the kernel that walks the covering blocks (bit 1 of
mfem_debug_set_lat27, bit 2 of mfem_debug_set_lat8): solve time of C3 (hex-8 elasticity 128^3, 50 BiCGStab(2) sweeps) and C4 (hex-27 thermal 128^3,
200 CG iterations), alternating, best of 3.   usage: gather_ab.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metafem_jl_amd as mf
from metafem_jl_amd import _lib
lam, mu = 0.5769230769230769, 0.38461538461538464
N = 128
def best(fn, reps=3):
    t = 1e9
    for _ in range(reps):
        t = min(t, fn())
    return t
b = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N), 1, 3)
A = b.pattern(3)
K = b.assemble_elasticity(A, lam, mu, 1000.0, mf.FACE_BITS["x0"])
R = b.residual_elasticity(torch.zeros(A.n, dtype=torch.float64, device="cuda"), lam, mu, 1000.0, mf.FACE_BITS["x0"], mf.FACE_BITS["y1"], (0.0, 1.0, 0.0, 0.0, 0.0, 0.0))
c3 = lambda: mf.iterative_Solve(A, K, R, 1e-300, Sv_func=mf.bicgstabl_GS_, maxiter=50, max_pass=1, s=2, fixed_iterations=True)[1].solve_ms
for rnd in range(2):
    for knob, tag in ((1, "staged"), (1 | 4, "walking")):
        _lib.lib.mfem_debug_set_lat8(knob)
        print(f"C3 pass 2 {tag:8s}: solve {best(c3):.2f} ms", flush=True)
_lib.lib.mfem_debug_set_lat8(1)
del b, A, K, R
torch.cuda.empty_cache()
b = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N), 2, 5)
A = b.pattern(1)
K = b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
s = torch.full((A.n,), 1600.0, dtype=torch.float64, device="cuda")
R = b.residual_thermal(torch.zeros(A.n, dtype=torch.float64, device="cuda"), 0.6, 25.0, 293.15, 0x3F, s=s)
c4 = lambda: mf.iterative_Solve(A, K, R, 1e-300, Sv_func=mf.cg_, maxiter=200, max_pass=1, fixed_iterations=True)[1].solve_ms
xs = {}
for rnd in range(2):
    for knob, tag in ((1, "fused CG iteration (pass 2 inside the residual update)"), (1 | 4, "staged pass 2"), (1 | 4 | 2, "walking pass 2")):
        _lib.lib.mfem_debug_set_lat27(knob)
        print(f"C4 {tag:56s}: solve {best(c4):.2f} ms", flush=True)
        xs[tag] = mf.iterative_Solve(A, K, R, 1e-300, Sv_func=mf.cg_, maxiter=200, max_pass=1, fixed_iterations=True)[0]
_lib.lib.mfem_debug_set_lat27(1)
ks = list(xs)
for k in ks[1:]:
    print(f"C4 x after 200 iterations, {ks[0][:5]} against {k[:7]}: max |dx| / max |x| = {float((xs[ks[0]] - xs[k]).abs().max() / xs[k].abs().max()):.2e}", flush=True)
