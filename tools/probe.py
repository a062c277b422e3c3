"""Quick on-box perf probe (not part of the product): SpMV / CG / assembly timings at N^3."""
import ctypes as C
import sys
import time

import torch

import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metafem_jl_amd as mf
from metafem_jl_amd import _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lib = _lib.lib
lib.mfem_debug_set_spmv.argtypes = [C.c_int, C.c_int]


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


t0 = time.time()
brick = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N))
A = brick.pattern(1)
torch.cuda.synchronize()
print(f"N={N} n={A.n} nnz={A.nnz} pattern {time.time()-t0:.2f}s", flush=True)
K = torch.empty(A.nnz, dtype=torch.float64, device="cuda")
ms = timeit(lambda: brick.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F, out=K), reps=3, warm=1)
print(f"assemble_thermal {ms:.2f} ms  ({(A.nnz*8 + brick.nel*32 + brick.ncp*24)/ms/1e6:.1f} GB/s algorithmic)", flush=True)
x = mf.FEM_rand(A.n, 1, 0)
y = torch.empty_like(x)
ms = timeit(lambda: brick.residual_thermal(x, 0.6, 25.0, 293.15, 0x3F, s=x, out=y), reps=3, warm=1)
print(f"residual_thermal {ms:.2f} ms", flush=True)
bytes_spmv = A.nnz * 12 + A.n * 16 + (A.n + 1) * 8
# copy bandwidth reference
a = torch.empty(A.nnz, dtype=torch.float64, device="cuda")
ms = timeit(lambda: a.copy_(K))
print(f"torch copy {A.nnz*16/ms/1e6:.0f} GB/s")
for var in (0, 1, 2, 3):
  for xcd in (1, 0):
    for mult in (4, 8, 16):
        lib.mfem_debug_set_spmv(xcd | (var << 4), mult)
        ms = timeit(lambda: mf.mul_(y, A, K, x))
        print(f"spmv var={var} xcd={xcd} grid_mult={mult}: {ms:.3f} ms  {bytes_spmv/ms/1e6:.0f} GB/s ({bytes_spmv/ms/1e6/8000*100:.1f}% of 8 TB/s)", flush=True)
lib.mfem_debug_set_spmv(1, 8)
b = torch.ones(A.n, dtype=torch.float64, device="cuda")
for it in (50, 200):
    _, st = mf.iterative_Solve(A, K, b, 1e-30, Sv_func=mf.cg_, maxiter=it, max_pass=1, fixed_iterations=True)
    print(f"CG {it} its: {st.solve_ms:.1f} ms -> {st.solve_ms/it:.3f} ms/it", flush=True)
ms = timeit(lambda: mf.dot(x, y))
print(f"dot {ms:.3f} ms {A.n*16/ms/1e6:.0f} GB/s")
ms = timeit(lambda: mf.axpby_(0.5, x, 0.5, y))
print(f"axpby {ms:.3f} ms {A.n*24/ms/1e6:.0f} GB/s")
