"""A/B of the layout-bind kernel inside one process: mfem_spmv_solver_layout (bind + one SpMV) timed with hip events, alternating a debug
knob.  usage: bind_ab.py <knob-bit-of-mfem_debug_set_ell> [c2|c3]"""
import sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import metafem_jl_amd as mf
from metafem_jl_amd import _lib
bit = int(sys.argv[1]); cfg = sys.argv[2] if len(sys.argv) > 2 else "c2"
if cfg == "c2":
    b = mf.make_Brick((1.0, 1.0, 1.0), (256,) * 3, 1, 3); A = b.pattern(1); K = b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
else:
    b = mf.make_Brick((1.0, 1.0, 1.0), (128,) * 3, 1, 3); A = b.pattern(3); K = b.assemble_elasticity(A, 0.5769, 0.3846, 1000.0, mf.FACE_BITS["x0"])
x = mf.FEM_rand(A.n, 3, 0); y = torch.zeros(A.n, dtype=torch.float64, device="cuda")
def t(reps=8):
    for _ in range(2): _lib.check(_lib.lib.mfem_spmv_solver_layout(b.ctx._h, A._h, K.data_ptr(), x.data_ptr(), y.data_ptr(), 1.0, 0.0))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): _lib.check(_lib.lib.mfem_spmv_solver_layout(b.ctx._h, A._h, K.data_ptr(), x.data_ptr(), y.data_ptr(), 1.0, 0.0))
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for rnd in range(3):
    _lib.lib.mfem_debug_set_ell(1); a = t()
    _lib.lib.mfem_debug_set_ell(1 | (1 << bit)); c = t()
    print(f"{cfg} round {rnd}: bind + SpMV  knob off {a:.3f} ms   knob on {c:.3f} ms", flush=True)
_lib.lib.mfem_debug_set_ell(1)
