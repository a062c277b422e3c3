"""One BiCGStab(2) (or IDR(8)) solve of config C3 (hex-8 elasticity 128^3) with a fixed number of iterations: for rocprofv3 --kernel-trace --stats.
usage: c3_solve_once.py [bicgstabl|idrs] [N] [iterations] [mfem_debug_set_ell knob]"""
import sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import metafem_jl_amd as mf
which = sys.argv[1] if len(sys.argv) > 1 else "bicgstabl"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 128
its = int(sys.argv[3]) if len(sys.argv) > 3 else 40
if len(sys.argv) > 4:
    from metafem_jl_amd import _lib
    _lib.lib.mfem_debug_set_ell(1 | int(sys.argv[4], 0))
brick = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N))
A = brick.pattern(3)
E, nu = 1.0, 0.3
lam, mu = E * nu / ((1 + nu) * (1 - 2 * nu)), E / (2 * (1 + nu))
K = brick.assemble_elasticity(A, lam, mu, 1000.0 * E, mf.FACE_BITS['x0'])
b = mf.FEM_rand(A.n, 1, 0) - 0.5
sv, s = (mf.bicgstabl_GS_, 2) if which == "bicgstabl" else (mf.idrs_, 8)
for _ in range(2):
    _, st = mf.iterative_Solve(A, K, b, 1e-300, Sv_func=sv, maxiter=its, max_pass=1, s=s)
torch.cuda.synchronize()
print(which, "solve_ms", st.solve_ms, "spmv", st.spmv_count, "ms per SpMV-equivalent", st.solve_ms / st.spmv_count)
