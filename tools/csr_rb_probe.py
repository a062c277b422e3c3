"""mul! (mfem_spmv_csr) with a forced kernel variant on the BASELINE matrices: hex-8 thermal N^3 (27 per row), hex-8 elasticity (81 per row),
hex-27 (27 .. 125 per row).  usage: csr_rb_probe.py <kind: c2|c3|c4> <N> <variant: 0 auto, 1 product tile, 3 row blocks, 7 wave tiles>"""
import sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import metafem_jl_amd as mf
from metafem_jl_amd import _lib
kind, N, var = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
extra = int(sys.argv[4], 0) if len(sys.argv) > 4 else 0  # further knob bits (bit 25: no column inspection, bit 26: round-robin row-block tiles)
_lib.lib.mfem_debug_set_spmv((var << 16) | extra, 0)  # before the pattern exists: variant 3 also plans the row blocks
if kind == "c4":
    b = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N), 2, 5)
else:
    b = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N))
A = b.pattern(3 if kind == "c3" else 1)
K = mf.FEM_rand(A.nnz, 9, 0) - 0.5
x = mf.FEM_rand(A.n, 3, 0)
y = torch.zeros(A.n, dtype=torch.float64, device="cuda")
for _ in range(3): mf.mul_(y, A, K, x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): mf.mul_(y, A, K, x)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
nb = A.nnz * 12 + A.n * 16 + (A.n + 1) * 8
print(f"{kind} {N}^3 variant {var} knob {extra:#x}: {ms:.3f} ms  {nb / ms / 1e6:.0f} GB/s  frac {nb / ms / 1e6 / 8000:.3f}  checksum {float(y.sum()):.12e}")
