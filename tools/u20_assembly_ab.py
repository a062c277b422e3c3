"""Where the time of the unstructured hex-20 element kernel goes: K_linear_func of the u20 legs with phases of k_mesh_assemble left out
(mfem_debug_set("mesh_abl", bits): timing only).  usage: u20_assembly_ab.py [n = 96] [fields = 1,3]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import bench_legs as L  # noqa: E402
from metafem_jl_amd import _lib, generic as G, physics  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
fields = [int(f) for f in (sys.argv[2] if len(sys.argv) > 2 else "1,3").split(",")]
B = L.Bench(bench.parse_args([]))
space, msh, fac = B.unstructured_mesh(n)
for f in fields:
    wf = physics.thermal_domain(3, L.K_COND) if f == 1 else physics.elasticity_domain(3, L.LAM, L.MU)
    gd = G.GenericDomain(B.ctx, space, msh.coords, msh.cp_ids, f, wf, [])
    gd.K_linear_func()
    torch.cuda.synchronize()
    for abl, what in ((0, "everything"), (1, "no pair products"), (2, "no stores"), (3, "no pair products, no stores"), (4, "no geometry"), (8, "no table"),
                      (16, "no coordinate gather"), (31, "nothing but the loop"), (29, "stores alone")):
        _lib.lib.mfem_debug_set_mesh_abl(abl)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        gd.K_linear_func()
        e0.record()
        for _ in range(3):
            gd.K_linear_func()
        e1.record()
        torch.cuda.synchronize()
        print(f"fields {f} abl {abl:2d} {what:32s} K_linear_func {e0.elapsed_time(e1) / 3:8.3f} ms", flush=True)
    _lib.lib.mfem_debug_set_mesh_abl(0)
    del gd
