"""The unstructured leg of bench.py on a TET-10 mesh (a brick cut into tetrahedra, serendipity order 2 = the 10-node tetrahedron of the reference's
tet examples): K_linear_func + K_nonlinear_func + 200 SpMV-equivalent steps of idrs!(8) with Pr_Jacobi!.  usage: tet10_leg.py [n = 64] [fields = 1,3]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import bench_legs as L  # noqa: E402
import metafem_jl_amd as mf  # noqa: E402
from metafem_jl_amd import _lib, element, generic as G, mesh as pm, physics  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
fields = [int(f) for f in (sys.argv[2] if len(sys.argv) > 2 else "1,3").split(",")]
B = L.Bench(bench.parse_args([]))
if os.environ.get("MFEM_TERM_MATRIX"):  # A/B: 0 = the non-staged element kernel walks the term list
    _lib.lib.mfem_debug_set_mesh_term_matrix(int(os.environ["MFEM_TERM_MATRIX"]))
if os.environ.get("MFEM_OP_WAVE_MIN_ITP"):  # A/B: elements from this many nodes take the wave forms of the batched var / res operators (default 16)
    _lib.lib.mfem_debug_set_op_wave_forms(1, int(os.environ["MFEM_OP_WAVE_MIN_ITP"]))
space = element.classical_space(3, "Serendipity", 2, 5, shape="SIMPLEX")
t0 = time.perf_counter()
vert, conn = pm.make_Brick((1.0, 1.0, 1.0), (n, n, n), shape="SIMPLEX")
nel = conn.shape[1]
blk = 512
nb = (nel + blk - 1) // blk
perm = (np.random.default_rng(0x5EED).permutation(nb)[:, None] * blk + np.arange(blk)[None, :]).ravel()
perm = perm[perm < nel]
msh = pm.mesh_Classical(vert, conn[:, perm], space)
fac = pm.get_BoundaryMesh(msh)
print(f"tet-10 mesh: {msh.nel} elements, {msh.ncp} control points, host {time.perf_counter() - t0:.1f} s", flush=True)
for F in fields:
    if F == 1:
        wf = physics.thermal_domain(3, L.K_COND)
        bnd = [(fac.element_ID, fac.element_eindex, physics.thermal_convection(L.H, L.TENV))]
    else:
        wf = physics.elasticity_domain(3, L.LAM, L.MU)
        c = fac.centroid
        wall, top = fac.select(np.abs(c[:, 0]) < 1e-9), fac.select(np.abs(c[:, 1] - 1.0) < 1e-9)
        bnd = [(wall.element_ID, wall.element_eindex, physics.penalty([0, 1, 2], L.TAU)), (top.element_ID, top.element_eindex, physics.traction(3, "sl", rows=[1]))]
    gd = G.GenericDomain(B.ctx, space, msh.coords, msh.cp_ids, F, wf, bnd)
    if F == 1:
        gd.controlpoints["s"] = torch.full((msh.ncp,), L.SRC, dtype=torch.float64, device="cuda")
    else:
        for v in (2, 4, 6):
            gd.controlpoints[f"sl{v}"] = torch.full((msh.ncp,), 1.0 if v == 2 else 0.0, dtype=torch.float64, device="cuda")
    A = gd.A
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]

    def step():
        ev[0].record()
        gd.K_linear_func()
        ev[1].record()
        gd.K_nonlinear_func()
        ev[2].record()
        return mf.iterative_Solve(A, gd.K_total, gd.residue, 1e-300, Sv_func=mf.idrs_, Pr_func=mf.Pr_Jacobi_, maxiter=200, max_pass=1, s=8, fixed_iterations=True)

    step()
    torch.cuda.synchronize()
    read = B.spmv_timer()
    t0 = time.perf_counter()
    k_ms = r_ms = s_ms = 0.0
    spmvs = 0
    for _ in range(3):
        dx, st = step()
        torch.cuda.synchronize()
        k_ms += ev[0].elapsed_time(ev[1]); r_ms += ev[1].elapsed_time(ev[2]); s_ms += st.solve_ms; spmvs += st.spmv_count
    el = time.perf_counter() - t0
    sp_ms, sp_n = read()
    mode = __import__("ctypes").c_int32()
    _lib.check(_lib.lib.mfem_csr_solver_layout(B.ctx._h, A._h, __import__("ctypes").byref(mode), None, None, None))
    byts = __import__("ctypes").c_int64()
    _lib.check(_lib.lib.mfem_csr_solver_layout_bytes(B.ctx._h, A._h, __import__("ctypes").byref(byts)))
    print(f"tet-10 {n}^3 x {F}: n_dof {A.n} nnz {A.nnz} ({A.nnz / A.n:.1f} per row) mode {mode.value} bsell F {int(_lib.lib.mfem_debug_bsell_fields(A._h))}: "
          f"{A.n * spmvs / el:.3e} DOF-updates/s, {el / 3 * 1e3:.1f} ms/step = K {k_ms / 3:.2f} + R {r_ms / 3:.2f} + solve {s_ms / 3:.1f}; "
          f"SpMV {sp_ms / max(sp_n, 1):.4f} ms = {byts.value / (sp_ms / max(sp_n, 1) * 1e-3) / 8e12:.3f} of 8 TB/s on design bytes; final res {st.final_res:.2e} (initial {st.initial_res:.2e})", flush=True)
    del gd, A
