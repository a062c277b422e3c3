"""hex-8 thermal kernels at 256^3 (and 512^3 with an argument): plane-sweep kernels against the tile kernels, hip-event times."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metafem_jl_amd as mf
from metafem_jl_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
b = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N))
A = b.pattern(1)
K = torch.empty(A.nnz, dtype=torch.float64, device="cuda")
x = mf.FEM_rand(A.n, 1, 0) + 300.0
s = torch.full((A.n,), 1600.0, dtype=torch.float64, device="cuda")
R = torch.empty(A.n, dtype=torch.float64, device="cuda")
def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
res = {}
for variant in (1, 0):
    _lib.lib.mfem_debug_set_hex8_thermal(variant)
    tm = timed(lambda: b.assemble_thermal(A, 0.6, 0.0, 293.15, 0, out=K))
    tmr = timed(lambda: b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F, out=K))
    tr = timed(lambda: b.residual_thermal(x, 0.6, 25.0, 293.15, 0x3F, s=s, out=R))
    res[variant] = (K.clone() if N <= 256 else None, R.clone())
    print(f"N={N} variant={variant} ({'tile' if variant else 'sweep'}): matrix {tm:.3f} ms  (+robin {tmr:.3f})  residual {tr:.3f} ms", flush=True)
_lib.lib.mfem_debug_set_hex8_thermal(0)
if res[0][0] is not None:
    print("max rel diff K", float((res[0][0] - res[1][0]).abs().max() / res[1][0].abs().max()))
print("max rel diff R", float((res[0][1] - res[1][1]).abs().max() / res[1][1].abs().max()))
