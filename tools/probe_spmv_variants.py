"""Per-launch event timing (mfem_prof_spmv_*) of the tile-kernel variants: threads per workgroup x loads in flight."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metafem_jl_amd as mf
from metafem_jl_amd import _lib
lib = _lib.lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
brick = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N))
A = brick.pattern(1)
K = brick.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
ctx = brick.ctx
x = mf.FEM_rand(A.n, 1, 0); y = torch.empty_like(x)
bytes_spmv = A.nnz * 12 + A.n * 16 + (A.n + 1) * 8
def read():
    tot, cnt = C.c_double(), C.c_int64()
    _lib.check(lib.mfem_prof_spmv_read(ctx._h, C.byref(tot), C.byref(cnt), 1))
    return tot.value / max(cnt.value, 1)
_lib.check(lib.mfem_prof_spmv_enable(ctx._h, 1))
b = torch.ones(A.n, dtype=torch.float64, device="cuda")
ref = None
names = {1: "256 thr x8", 0: "256 thr x4", 4: "512 thr x4", 5: "512 thr x2", 6: "1024 thr x2", 2: "128 thr CAP 2016", 3: "64 thr CAP 2016",
         7: "64 thr CAP 1008"}
for var in (1, 2, 3, 7):
    for mult in (8, 16):
        lib.mfem_debug_set_spmv(var << 16, mult)
        for _ in range(10): mf.mul_(y, A, K, x)
        if ref is None: ref = y.clone()
        assert torch.equal(ref, y) or float((ref - y).abs().max()) <= 1e-12 * float(ref.abs().max()), "variant changed the result"
        read()
        for _ in range(30): mf.mul_(y, A, K, x)
        ms = read()
        mf.iterative_Solve(A, K, b, 1e-30, Sv_func=mf.cg_, maxiter=60, max_pass=1, fixed_iterations=True)
        ms2 = read()
        print(f"variant {var} ({names[var]}) grid_mult {mult}: standalone {ms:.4f} ms {bytes_spmv/ms/1e6:.0f} GB/s | in CG {ms2:.4f} ms {bytes_spmv/ms2/1e6:.0f} GB/s", flush=True)
lib.mfem_debug_set_spmv(1 << 16, 8)
