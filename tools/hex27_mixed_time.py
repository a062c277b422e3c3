"""hex-27 matrix assembly time against the fraction of non-affine elements (round 5: per-element choice between the in-place computation of affine elements
and pass 1 + streamed runs for the others; rows of general elements computed in place from G_q), 128^3 by default: default policy, the row-owner kernel
of general elements forced, the per-element choice forced for every fraction, the two-pass path forced.
usage: hex27_mixed_time.py [n]   -> profiles/r05_hex27_mixed.txt"""
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import metafem_jl_amd as mf
from metafem_jl_amd import _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = (N, N, N)
brick = mf.make_Brick((1.0, 1.0, 1.0), n, 2, 5)
A = brick.pattern(1)
K = torch.empty(A.nnz, dtype=torch.float64, device="cuda")
base = [brick.coords_view(d).clone() for d in range(3)]
m = [2 * v + 1 for v in n]
I, J, Kk = np.meshgrid(np.arange(N), np.arange(N), np.arange(N), indexing="ij")
cn = torch.tensor((((2 * I + 1) * m[1] + (2 * J + 1)) * m[2] + (2 * Kk + 1)).ravel(), device="cuda")
perm = cn[torch.randperm(cn.numel(), device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))]


def timed(reps=5):
    brick.assemble_thermal(A, 0.6, 0.0, 293.15, 0, out=K)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        brick.assemble_thermal(A, 0.6, 0.0, 293.15, 0, out=K)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print(f"hex-27 {N}^3 thermal matrix assembly (no faces), ms per assembly; distorted = elements whose centre node is moved")
print(f"{'distorted %':>12s} {'default policy':>16s} {'rows from G_q':>16s} {'choice forced':>16s} {'two-pass forced':>16s}")
for pct in (0, 1, 5, 10, 25, 50, 75, 100):
    for d in range(3):
        brick.coords_view(d).copy_(base[d])
    k = (cn.numel() * pct) // 100
    if k:
        brick.coords_view(0)[perm[:k]] += 0.3 / N * 0.05
    row = []
    # default | the row-owner kernel of general elements from 1 % on (bits 2-7) | it off, the per-element choice for every fraction | both off: two-pass
    for knob in (0, 1 << 2, (100 << 24) | (1 << 11), (1 << 10) | (1 << 11)):
        _lib.lib.mfem_debug_set_hex27(knob)
        row.append(timed())
    _lib.lib.mfem_debug_set_hex27(0)
    print(f"{pct:12d} {row[0]:16.3f} {row[1]:16.3f} {row[2]:16.3f} {row[3]:16.3f}", flush=True)
