"""K_linear_func on large unstructured tet-10 / hex-8 meshes with the staged persistent form of the row-owner element kernel on (from 8 nodes) and off
(from 16: the default).  usage: tet10_assembly_ab.py [n = 48]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import metafem_jl_amd as mf  # noqa: E402
from metafem_jl_amd import _lib, element, generic as G, mesh as pm, physics  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
for shape, itp_type, order, itg in (("SIMPLEX", "Serendipity", 2, 5), ("CUBE", "Lagrange", 1, 3)):
    space = element.classical_space(3, itp_type, order, itg, shape=shape)
    vert, conn = pm.make_Brick((1.0, 1.0, 1.0), (n, n, n), shape=shape)
    msh = pm.mesh_Classical(vert, conn, space)
    for fields in (1, 3):
        wf = physics.thermal_domain(3, 0.6) if fields == 1 else physics.elasticity_domain(3, 0.5769230769230769, 0.38461538461538464)
        gd = G.GenericDomain(mf.default_context(), space, msh.coords, msh.cp_ids, fields, wf, [])
        ref = None
        for min_itp in (16, 8, 16, 8):
            _lib.lib.mfem_debug_set_mesh_stage_min_itp(min_itp)
            gd.K_linear_func()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                gd.K_linear_func()
            e1.record()
            torch.cuda.synchronize()
            K = gd.K_linear.cpu().numpy()
            if ref is None:
                ref = K
            print(f"{shape} itp {space.itp} itg {space.itg} nel {msh.nel} fields {fields} staged from {min_itp:2d} nodes: K_linear_func {e0.elapsed_time(e1) / 3:8.3f} ms"
                  f"  max rel diff {np.abs(K - ref).max() / np.abs(ref).max():.1e}", flush=True)
        _lib.lib.mfem_debug_set_mesh_stage_min_itp(16)
        del gd
