"""On-box probe: the solver-layout SpMV (row-sorted sliced ELL) on hex-27 N^3, standalone time per sort window."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metafem_jl_amd as mf
from metafem_jl_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
windows = [int(a, 0) for a in sys.argv[2:]] or [0]  # upper bits of mfem_debug_set_sell (4 no block order, 8 XCD chunks, window << 8, unroll << 16, wg/cu << 24)
brick = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N), 2, 5)
A = brick.pattern(1)
K = brick.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
x = mf.FEM_rand(A.n, 1, 0); y = torch.empty_like(x)
y0 = torch.empty_like(x)
mf.mul_(y0, A, K, x)
def timeit(fn, reps=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for w in windows:
    _lib.lib.mfem_debug_set_sell(1 | w)
    A = brick.pattern(1)  # the layout plan is cached on the matrix handle
    mode, slots, padded = C.c_int32(), C.c_int32(), C.c_int64()
    _lib.check(_lib.lib.mfem_csr_solver_layout(brick.ctx._h, A._h, C.byref(mode), C.byref(slots), C.byref(padded), None))
    rhs = torch.ones(A.n, dtype=torch.float64, device="cuda")
    mf.iterative_Solve(A, K, rhs, 1e-30, Sv_func=mf.cg_, maxiter=50, max_pass=1, fixed_iterations=True)
    xs, st = mf.iterative_Solve(A, K, rhs, 1e-30, Sv_func=mf.cg_, maxiter=50, max_pass=1, fixed_iterations=True)
    print(f"knobs {w:#x}: mode {mode.value} padded rows {padded.value} CG {st.solve_ms / 50:.3f} ms/it, |x| {float(xs.norm()):.12e}", flush=True)
