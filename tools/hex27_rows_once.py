"""A few hex-27 matrix assemblies of a fully distorted N^3 mesh through the row-owner kernel of general elements (k_hex27_gq_lane + k_hex27_rows_gq), for
rocprofv3 --kernel-trace --stats / --pmc.  usage: hex27_rows_once.py [N] [reps] [knob]"""
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import metafem_jl_amd as mf
from metafem_jl_amd import _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
knob = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0
n = (N, N, N)
brick = mf.make_Brick((1.0, 1.0, 1.0), n, 2, 5)
A = brick.pattern(1)
K = torch.empty(A.nnz, dtype=torch.float64, device="cuda")
m = [2 * v + 1 for v in n]
I, J, Kk = np.meshgrid(np.arange(N), np.arange(N), np.arange(N), indexing="ij")
cn = torch.tensor((((2 * I + 1) * m[1] + (2 * J + 1)) * m[2] + (2 * Kk + 1)).ravel(), device="cuda")
brick.coords_view(0)[cn] += 0.3 / N * 0.05
_lib.lib.mfem_debug_set_hex27(knob)
for _ in range(reps):
    brick.assemble_thermal(A, 0.6, 0.0, 293.15, 0, out=K)
torch.cuda.synchronize()
print("rows assemblies:", _lib.lib.mfem_debug_hex27_rows_count())
