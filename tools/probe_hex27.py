"""On-box probe for config C4 (hex-27, 128^3): assembly / residual / SpMV / CG timings."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metafem_jl_amd as mf
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
def timeit(fn, reps=5, warm=1):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
brick = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N), 2, 5)
A = brick.pattern(1)
print(f"hex27 N={N} n={A.n} nnz={A.nnz}", flush=True)
K = torch.empty(A.nnz, dtype=torch.float64, device="cuda")
from metafem_jl_amd import _lib
flops = brick.nel * (2 * 27 * 27 * 81)
_lib.lib.mfem_debug_set_hex27(1)
ms = timeit(lambda: brick.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F, out=K), reps=3)
print(f"assemble two-pass (scratch + gather) {ms:.2f} ms -> {flops/ms/1e9:.2f} TFLOP/s useful", flush=True)
if len(sys.argv) > 2 and sys.argv[2] == "chunks":  # element planes per scratch chunk (ring of planes)
    for P in (1, 2, 3, 4, 6, 8, 16, 32, 64, 128):
        _lib.lib.mfem_debug_set_hex27(1 | (P << 16))
        ms = timeit(lambda: brick.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F, out=K), reps=3)
        print(f"  two-pass, {P} element planes per chunk: {ms:.2f} ms", flush=True)
    sys.exit(0)
_lib.lib.mfem_debug_set_hex27(2)
ms = timeit(lambda: brick.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F, out=K), reps=3)
Ka = K.clone()
print(f"assemble FP64-atomics {ms:.2f} ms -> {flops/ms/1e9:.2f} TFLOP/s useful", flush=True)
_lib.lib.mfem_debug_set_hex27(3)
ms = timeit(lambda: brick.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F, out=K), reps=3)
print("atomics vs colours max rel diff", float((Ka - K).abs().max() / K.abs().max()))
print(f"assemble colour scatter {ms:.2f} ms -> {flops/ms/1e9:.2f} TFLOP/s useful Ke (2*27*27*81 per element), {A.nnz*8/ms/1e6:.0f} GB/s of nnz*8", flush=True)
_lib.lib.mfem_debug_set_hex27(0)
# (the phase ablation of round 1 -- kernel phases left out, wrong values -- was a timing probe inside the product kernel; it was removed
# from the library in round 2, its numbers are in profiles/r01_hex27_mfma_counters.txt)
x = mf.FEM_rand(A.n, 1, 0); y = torch.empty_like(x)
ms = timeit(lambda: brick.residual_thermal(x, 0.6, 25.0, 293.15, 0x3F, s=x, out=y), reps=3)
print(f"residual {ms:.2f} ms", flush=True)
b = A.nnz * 12 + A.n * 16 + (A.n + 1) * 8
ms = timeit(lambda: mf.mul_(y, A, K, x), reps=10, warm=2)
print(f"spmv {ms:.3f} ms {b/ms/1e6:.0f} GB/s ({b/ms/1e6/80:.1f}% of 8 TB/s)", flush=True)
rhs = torch.ones(A.n, dtype=torch.float64, device="cuda")
for sell in (0, 1):
    _lib.lib.mfem_debug_set_sell(sell)
    mf.iterative_Solve(A, K, rhs, 1e-30, Sv_func=mf.cg_, maxiter=50, max_pass=1, fixed_iterations=True)
    _, st = mf.iterative_Solve(A, K, rhs, 1e-30, Sv_func=mf.cg_, maxiter=50, max_pass=1, fixed_iterations=True)
    print(f"CG ({'row-sorted sliced ELL' if sell else 'CSR tile kernel'} in the loop) {st.solve_ms/50:.3f} ms/it", flush=True)
