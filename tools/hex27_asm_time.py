"""hex-27 thermal K + R at 128^3 (affine mesh: k_hex27_direct; R: 8 colour launches of k_hex27<false>) and the two-pass MFMA path forced.  usage: hex27_asm_time.py [N]"""
import sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import metafem_jl_amd as mf
from metafem_jl_amd import _lib
def t(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
b = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N), 2, 5)
A = b.pattern(1)
K = torch.empty(A.nnz, dtype=torch.float64, device="cuda")
x = torch.zeros(A.n, dtype=torch.float64, device="cuda")
s = torch.full((A.n,), 1600.0, dtype=torch.float64, device="cuda")
print(f"hex-27 {N}^3 affine mesh: K {t(lambda: b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F, out=K)):.3f} ms  R {t(lambda: b.residual_thermal(x, 0.6, 25.0, 293.15, 0x3F, s=s)):.3f} ms", flush=True)
_lib.lib.mfem_debug_set_hex27(1 << 9)
print(f"   two-pass MFMA path forced: K {t(lambda: b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F, out=K)):.3f} ms", flush=True)
_lib.lib.mfem_debug_set_hex27((1 << 9) | (1 << 8))
print(f"   ... with the affine shortcut off (general elements): K {t(lambda: b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F, out=K)):.3f} ms  R {t(lambda: b.residual_thermal(x, 0.6, 25.0, 293.15, 0x3F, s=s)):.3f} ms", flush=True)
_lib.lib.mfem_debug_set_hex27(0)
