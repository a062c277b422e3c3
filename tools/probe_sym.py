"""On-box probe: symmetric sweep variant of the diagonal-slotted SpMV (bit 22 of mfem_debug_set_ell) vs the plain kernel."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metafem_jl_amd as mf
from metafem_jl_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
if len(sys.argv) > 2:  # lift the size thresholds of the solver layouts (small meshes)
    _lib.lib.mfem_debug_set_layout_min_rows(0, 0)
brick = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N))
A = brick.pattern(1)
K = brick.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
x = mf.FEM_rand(A.n, 1, 0) - 0.5
rhs = torch.ones(A.n, dtype=torch.float64, device="cuda")
ys = []
KN = [int(a, 0) for a in sys.argv[3:]] or [0]
for knob in [1 << 22] + KN:  # bit 22 = symmetric sweep off
    _lib.lib.mfem_debug_set_ell(1 | knob)
    y = torch.full((A.n,), 3.0, dtype=torch.float64, device="cuda")
    _lib.check(_lib.lib.mfem_spmv_solver_layout(brick.ctx._h, A._h, K.data_ptr(), x.data_ptr(), y.data_ptr(), 1.0, 0.0))
    ys.append(y)
    mf.iterative_Solve(A, K, rhs, 1e-30, Sv_func=mf.cg_, maxiter=50, max_pass=1, fixed_iterations=True)
    xs, st = mf.iterative_Solve(A, K, rhs, 1e-30, Sv_func=mf.cg_, maxiter=100, max_pass=1, fixed_iterations=True)
    print(f"knob {knob:#x}: CG {st.solve_ms / 100:.4f} ms/it, |x| {float(xs.norm()):.15e}", flush=True)
print("SpMV bitwise equal:", [bool(torch.equal(ys[0], y)) for y in ys[1:]])
_lib.lib.mfem_debug_set_ell(1)
