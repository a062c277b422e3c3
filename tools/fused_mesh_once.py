"""hex-20 elasticity brick N^3 through the fused unstructured assembly (row-owner form), a few K_linear_func calls: for
rocprofv3 --kernel-trace --stats / --pmc."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metafem_jl_amd as mf
from metafem_jl_amd import element, generic as G, mesh as pm, physics
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
mode = sys.argv[2] if len(sys.argv) > 2 else "rows"
space = element.classical_space(3, "Serendipity", 2, 5)
vert, conn = pm.make_Brick((1.0, 1.0, 1.0), (N, N, N))
msh = pm.mesh_Classical(vert, conn, space)
fac = pm.get_BoundaryMesh(msh)
f = fac.select(np.abs(fac.centroid[:, 0]) < 1e-9)
gd = G.GenericDomain(mf.default_context(), space, msh.coords, msh.cp_ids, 3, physics.elasticity_domain(3, 0.5769, 0.3846),
                     [(f.element_ID, f.element_eindex, physics.penalty([0, 1, 2], 1000.0))], fused=mode != "ops", row_owner=mode == "rows")
gd.update_Time()
for _ in range(4):
    gd.K_linear_func()
torch.cuda.synchronize()
print("nel", msh.nel, "nnz", gd.A.nnz)
