"""C3 (hex-8 elasticity, 128^3) plain CSR mul!: the 2688-entry wave tile (32 rows of 81 entries) against the 1792-entry one (16 rows)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metafem_jl_amd as mf
from metafem_jl_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
def timeit(fn, reps=10, warm=2):
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
brick.assemble_elasticity(A, lam, mu, 1000.0 * E, mf.FACE_BITS['x0'], out=K)
b = mf.FEM_rand(A.n, 1, 0) - 0.5
bytes_csr = A.nnz * 12 + A.n * 16 + (A.n + 1) * 8
ys = []
for off in (1, 0, 1, 0):
    _lib.lib.mfem_debug_set_spmv(off << 27, 0)
    y = torch.empty_like(b)
    ms = timeit(lambda: mf.mul_(y, A, K, b))
    ys.append(y)
    print(f"tile {'1792' if off else '2688'}: {ms:.3f} ms = {bytes_csr/ms/1e6:.0f} GB/s = {bytes_csr/ms/1e6/8000:.3f} of 8 TB/s")
print("max rel diff", float((ys[0] - ys[1]).abs().max() / ys[0].abs().max()))
