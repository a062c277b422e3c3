"""Config C3 (hex-8 elasticity, 3 DOF per node, 128^3): assembly, residual and Krylov timings; slot-major copy on/off."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metafem_jl_amd as mf
from metafem_jl_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
if len(sys.argv) > 2:
    _lib.lib.mfem_debug_set_vec_grid(int(sys.argv[2]))
def timeit(fn, reps=3, warm=1):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
brick = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N))
A = brick.pattern(3)
E, nu = 1.0, 0.3
lam, mu = E * nu / ((1 + nu) * (1 - 2 * nu)), E / (2 * (1 + nu))
K = torch.empty(A.nnz, dtype=torch.float64, device="cuda")
print(f"C3 N={N} n={A.n} nnz={A.nnz}")
for var in (1, 0):
    _lib.lib.mfem_debug_set_elasticity(var)
    ms = timeit(lambda: brick.assemble_elasticity(A, lam, mu, 1000.0 * E, mf.FACE_BITS['x0'], out=K))
    print(f"assemble elasticity ({'row-owner, global accumulation' if var else 'thread per (node, element), LDS rows'}) {ms:.2f} ms = {A.nnz*8/ms/1e6:.0f} GB/s of nnz*8")
    if var: K1 = K.clone()
print("variants bitwise equal:", bool(torch.equal(K1, K)))
xs = mf.FEM_rand(A.n, 2, 0) - 0.5
R = torch.empty(A.n, dtype=torch.float64, device="cuda")
ms = timeit(lambda: brick.residual_elasticity(xs, lam, mu, 1000.0 * E, mf.FACE_BITS['x0'], mf.FACE_BITS['y1'], [0, 1.0, 0, 0, 0, 0], out=R))
print(f"residual elasticity {ms:.2f} ms")
b = mf.FEM_rand(A.n, 1, 0) - 0.5
bytes_csr = A.nnz * 12 + A.n * 16 + (A.n + 1) * 8
x = torch.empty_like(b)
ms = timeit(lambda: mf.mul_(x, A, K, b), reps=10)
print(f"mul_ (CSR tile kernel) {ms:.3f} ms = {bytes_csr/ms/1e6:.0f} GB/s")
for ell in (0, 1):
    _lib.lib.mfem_debug_set_ell(ell | (6 << 4))
    for name, sv, s, it in (("bicgstabl(2)", mf.bicgstabl_GS_, 2, 40), ("idrs(8)", mf.idrs_, 8, 45)):
        mf.iterative_Solve(A, K, b, 1e-300, Sv_func=sv, maxiter=it, max_pass=1, s=s)
        _, st = mf.iterative_Solve(A, K, b, 1e-300, Sv_func=sv, maxiter=it, max_pass=1, s=s)
        print(f"slot-major copy {'on ' if ell else 'off'} {name}: {st.solve_ms:.1f} ms for {st.spmv_count} SpMV-equivalents = {st.solve_ms/st.spmv_count:.3f} ms each")
_lib.lib.mfem_debug_set_ell(1 | (6 << 4))
