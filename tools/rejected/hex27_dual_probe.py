"""hex-27 assembly at N^3: the two-pass sequence against the dual-role launches (both passes in one launch per chunk, on different chunks):
equality of K (bitwise) and time.  usage: hex27_dual_probe.py [N]"""
import sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import metafem_jl_amd as mf
from metafem_jl_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
b = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N), 2, 5)
A = b.pattern(1)
K0 = torch.empty(A.nnz, dtype=torch.float64, device="cuda"); K1 = torch.empty_like(K0)
def t(K, reps=5):
    for _ in range(2): b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F, out=K)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F, out=K)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
_lib.lib.mfem_debug_set_hex27(1)
print(f"two-pass sequence: {t(K0):.3f} ms", flush=True)
for share in (32, 48, 64, 96, 128):
    for P in (4, 8, 16):
        _lib.lib.mfem_debug_set_hex27(1 | 4 | (share << 8) | (P << 16))
        ms = t(K1)
        print(f"dual-role launches, gather share {share}/256, {P:2d} planes per chunk: {ms:.3f} ms   bitwise equal: {bool(torch.equal(K0, K1))}", flush=True)
_lib.lib.mfem_debug_set_hex27(0)
