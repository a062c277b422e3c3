"""mul! (mfem_spmv_csr) with and without the per-tile x-window kernel on hex-27 N^3, hex-8 (2N)^3 and hex-8 elasticity N^3: time and max difference."""
import sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import metafem_jl_amd as mf
from metafem_jl_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
def case(name, b, nf, asm):
    A = b.pattern(nf)
    K = asm(b, A)
    x = mf.FEM_rand(A.n, 3, 0)
    nb = A.nnz * 12 + A.n * 16 + (A.n + 1) * 8
    ys = []
    for knob, tag in ((1 << 28, "other kernels"), (0, "default"), (2 << 28, "window kernel")):
        _lib.lib.mfem_debug_set_spmv(knob, 0)
        y = torch.zeros(A.n, dtype=torch.float64, device="cuda")
        c0 = _lib.lib.mfem_debug_xw_spmv_count()
        ms = timeit(lambda: mf.mul_(y, A, K, x))
        ys.append(y)
        print(f"{name}: {tag:14s} {ms:.3f} ms  frac {nb / ms / 1e6 / 8000:.3f}  (window launches {_lib.lib.mfem_debug_xw_spmv_count() - c0})", flush=True)
    print("   max rel diff vs other kernels:", [float((y - ys[0]).abs().max() / ys[0].abs().max()) for y in ys[1:]], flush=True)
    _lib.lib.mfem_debug_set_spmv(0, 0)
case(f"hex-27 {N}^3", mf.make_Brick((1.0, 1.0, 1.0), (N, N, N), 2, 5), 1, lambda b, A: b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F))
torch.cuda.empty_cache()
case(f"hex-8 {2*N}^3", mf.make_Brick((1.0, 1.0, 1.0), (2 * N, 2 * N, 2 * N)), 1, lambda b, A: b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F))
torch.cuda.empty_cache()
case(f"hex-8 elasticity {N}^3", mf.make_Brick((1.0, 1.0, 1.0), (N, N, N)), 3, lambda b, A: b.assemble_elasticity(A, 0.5769, 0.3846, 1000.0, mf.FACE_BITS["x0"]))
