"""Where the time of k_thermal_matrix_sweep goes (hex-8 thermal matrix, 256^3 / 512^3): timing-only ablations behind bits 3-5 of the "hex8_thermal"
knob.  usage: thermal_sweep_ablate.py [N = 256]"""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import metafem_jl_amd as mf
from metafem_jl_amd import _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
b = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N))
A = b.pattern(1)
K = torch.empty(A.nnz, dtype=torch.float64, device="cuda")


def t(reps=10):
    for _ in range(3):
        b.assemble_thermal(A, 0.6, 0.0, 293.15, 0, out=K)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.assemble_thermal(A, 0.6, 0.0, 293.15, 0, out=K)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for bits, what in ((0, "everything"), (8, "no integration"), (16, "no gather from LDS"), (32, "no write-out"), (24, "no integration, no gather"), (56, "loads, LDS stores, barriers alone"),
                   (2, "rows written per thread (not staged)"), (4, "no affine shortcut")):
    _lib.lib.mfem_debug_set_hex8_thermal(bits)
    print(f"N {N}  {what:40s} {t():8.3f} ms", flush=True)
_lib.lib.mfem_debug_set_hex8_thermal(0)
