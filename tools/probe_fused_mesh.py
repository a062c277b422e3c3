"""On-box probe: K_linear_func through the fused unstructured assembly (mfem_mesh_assemble_*) against the stored-table
operator path (mfem_op_kval_batch) on the reference's example meshes (stress concentration 3-D: hex-20 elasticity; pikachu:
tet-10 thermal) and on a larger hex-20 brick.  Prints ms per K_linear_func (atomics; colour batches in brackets)."""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import metafem_jl_amd as mf
from metafem_jl_amd import element, generic as G, mesh as pm, physics

GOLD = os.path.join(ROOT, "tests", "golden")


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def case(name, space, msh, n_fields, dom, bnd):
    out = {"case": name, "nel": int(msh.nel), "ncp": int(msh.ncp)}
    ref = None
    gd = G.GenericDomain(mf.default_context(), space, msh.coords, msh.cp_ids, n_fields, dom, bnd, fused=True, row_owner=True)
    gd.controlpoints["s"] = torch.full((msh.ncp,), 1600.0, dtype=torch.float64, device="cuda")
    gd.update_Time()
    out["fused_row_owner_ms"] = round(timeit(gd.K_linear_func), 3)
    ref = gd.K_linear.clone()
    gd.K_linear_func()
    out["row_owner_bitwise_reproducible"] = bool(torch.equal(ref, gd.K_linear)) if not bnd or True else None
    del gd
    for colours in (None, "auto"):
        for fused in (True, False):
            gd = G.GenericDomain(mf.default_context(), space, msh.coords, msh.cp_ids, n_fields, dom, bnd, element_colours=colours, fused=fused,
                                 row_owner=False)
            gd.controlpoints["s"] = torch.full((msh.ncp,), 1600.0, dtype=torch.float64, device="cuda")
            gd.update_Time()
            ms = timeit(gd.K_linear_func)
            K = gd.K_linear.clone()
            if ref is None:
                ref = K
            out[("fused" if fused else "operators") + ("_coloured" if colours else "_atomics") + "_ms"] = round(ms, 3)
            out.setdefault("max_rel_diff", 0.0)
            out["max_rel_diff"] = max(out["max_rel_diff"], float((K - ref).abs().max() / ref.abs().max()))
            if colours and fused:
                out["colours"] = len(gd.groups[0].colour_offsets) - 1
            del gd
    out["speedup_row_owner_vs_operators_atomics"] = round(out["operators_atomics_ms"] / out["fused_row_owner_ms"], 2)
    out["speedup_atomics"] = round(out["operators_atomics_ms"] / out["fused_atomics_ms"], 2)
    out["speedup_coloured"] = round(out["operators_coloured_ms"] / out["fused_coloured_ms"], 2)
    print(json.dumps(out), flush=True)


E, nu = 210e9, 0.3
lam, mu = E * nu / ((1 + nu) * (1 - 2 * nu)), E / (2 * (1 + nu))
z = np.load(os.path.join(GOLD, "stress_concentration_3d.npz"))
space = element.classical_space(3, "Serendipity", 2, 5)
msh = pm.mesh_Classical(z["vert"], z["conn"].astype(np.int64), space)
fac = pm.get_BoundaryMesh(msh)
c = fac.centroid
bnd = []
for d in range(3):
    f = fac.select(np.abs(c[:, d]) < 0.05)
    bnd.append((f.element_ID, f.element_eindex, physics.penalty([d], 10000 * E / 25.0)))
case("stress concentration 3-D (hex-20 elasticity, examples/linear_elasticity/stress_concentration/3D_Mesh.inp)", space, msh, 3,
     physics.elasticity_domain(3, lam, mu), bnd)

z = np.load(os.path.join(GOLD, "pikachu_tet10.npz"))
space = element.classical_space(3, "Serendipity", 2, 5, shape="SIMPLEX")
msh = pm.mesh_Classical(z["vert"] / 100.0, z["conn"].astype(np.int64), space)
fac = pm.get_BoundaryMesh(msh)
case("pikachu (tet-10 thermal, examples/thermal_conduction/3D_COMSOL_Mesh.mphtxt)", space, msh, 1, physics.thermal_domain(3, 0.6),
     [(fac.element_ID, fac.element_eindex, physics.thermal_convection(25.0, 293.15))])

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
space = element.classical_space(3, "Serendipity", 2, 5)
vert, conn = pm.make_Brick((1.0, 1.0, 1.0), (N, N, N))
msh = pm.mesh_Classical(vert, conn, space)
fac = pm.get_BoundaryMesh(msh)
f = fac.select(np.abs(fac.centroid[:, 0]) < 1e-9)
case(f"hex-20 elasticity brick {N}^3", space, msh, 3, physics.elasticity_domain(3, 0.5769, 0.3846),
     [(f.element_ID, f.element_eindex, physics.penalty([0, 1, 2], 1000.0))])
