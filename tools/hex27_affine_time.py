"""hex-27 matrix assembly at N^3 with and without the affine-element shortcut (bit 8 of mfem_debug_set_hex27), and on a distorted mesh (general path
whatever the knob): ms per assembly without boundary faces, torch events on the stream the context runs on, best of 3 x 5.  usage: hex27_affine_time.py [N]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metafem_jl_amd as mf
from metafem_jl_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
brick = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N), 2, 5)
A = brick.pattern(1)
K = torch.empty(A.nnz, dtype=torch.float64, device="cuda")
def timed():
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(5):
            brick.assemble_thermal(A, 0.6, 0.0, 293.15, 0, out=K)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 5)
    return best
brick.assemble_thermal(A, 0.6, 0.0, 293.15, 0, out=K)
Kd = K.clone()
_lib.lib.mfem_debug_set_hex27(1 << 9)
brick.assemble_thermal(A, 0.6, 0.0, 293.15, 0, out=K)
print(f"N {N}: scratch-free assembly against the two-pass MFMA path: max |dK| / max |K| = {float((K - Kd).abs().max() / K.abs().max()):.2e}", flush=True)
for knob, tag in ((0, "scratch-free (all elements affine)"), (1 << 9, "two-pass, affine shortcut"), (1 << 8, "two-pass, general path"), (0, "scratch-free (all elements affine)"), (1 << 9, "two-pass, affine shortcut")):
    _lib.lib.mfem_debug_set_hex27(knob)
    ms = timed()
    print(f"N {N} uniform brick, {tag:34s}: {ms:.3f} ms per assembly = {118098.0 * N ** 3 / (ms * 1e-3) / 1e12:.2f} TFLOP/s useful = {118098.0 * N ** 3 / (ms * 1e-3) / 78.6e12:.3f} of the FP64 matrix peak", flush=True)
xs = mf.FEM_rand(A.n, 1, 0); R = torch.empty_like(xs)
def timed_res():
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(5):
            brick.residual_thermal(xs, 0.6, 0.0, 293.15, 0, s=xs, out=R)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 5)
    return best
for knob, tag in ((0, "affine shortcut"), (1 << 8, "general path"), (0, "affine shortcut"), (1 << 8, "general path")):
    _lib.lib.mfem_debug_set_hex27(knob)
    print(f"N {N} uniform brick, residual, {tag:16s}: {timed_res():.3f} ms", flush=True)
_lib.lib.mfem_debug_set_hex27(0)
x0 = brick.coords_view(0); x1 = brick.coords_view(1); x2 = brick.coords_view(2)
x0.add_(0.001 * torch.sin(3 * x1) * torch.cos(x2))
print(f"N {N} distorted brick (no element affine): {timed():.3f} ms per assembly", flush=True)
