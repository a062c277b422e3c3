import sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import metafem_jl_amd as mf
from metafem_jl_amd import _lib
N = 128
b = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N))
A = b.pattern(3)
lam, mu = 0.5769230769230769, 0.38461538461538464
xs = 0.01 * (mf.FEM_rand(A.n, 2, 0) - 0.5)
R = torch.empty(A.n, dtype=torch.float64, device="cuda")
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
f = lambda: b.residual_elasticity(xs, lam, mu, 1000.0, mf.FACE_BITS["x0"], mf.FACE_BITS["y1"], (0.0, 1.0, 0.0, 0.0, 0.0, 0.0), out=R)
print("residual sweep ms", t(f))
_lib.lib.mfem_debug_set_elasticity(2)
print("residual per-point ms", t(f))
_lib.lib.mfem_debug_set_elasticity(0)
K = torch.empty(A.nnz, dtype=torch.float64, device="cuda")
print("matrix ms", t(lambda: b.assemble_elasticity(A, lam, mu, 1000.0, mf.FACE_BITS["x0"], out=K)))
for bits, what in ((4, "no accumulation phases"), (8, "no write-out"), (16, "no integration"), (12, "no phases, no write-out"), (28, "nothing but loads")):
    _lib.lib.mfem_debug_set_elasticity(bits)
    print("matrix ms,", what, t(lambda: b.assemble_elasticity(A, lam, mu, 1000.0, mf.FACE_BITS["x0"], out=K)))
_lib.lib.mfem_debug_set_elasticity(0)
