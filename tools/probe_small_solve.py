"""Per-iteration latency of the Krylov bodies on a small system (launch-bound regime): tet-10 thermal, 23 703 DOF."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metafem_jl_amd as mf
from metafem_jl_amd import element, generic as G, mesh as pm
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "pikachu_tet10.npz"))
space = element.classical_space(3, "Serendipity", 2, 5, shape="SIMPLEX")
msh = pm.mesh_Classical(z["vert"] / 100.0, z["conn"].astype(np.int64), space)
fac = pm.get_BoundaryMesh(msh)
k, h, T0 = 0.6, 25.0, 293.15
wf = G.WeakForm()
for d in range(3):
    wf.inner_vars.append((f"T_{d}", 0, 1 + d, 0))
    wf.residues.append(G.ResTerm(0, 1 + d, lambda env, d=d: -k * env[f"T_{d}"]))
    wf.linear_gradients.append(G.GradTerm(0, 1 + d, 0, 1 + d, lambda env: -k))
wf.cp_ext_vars.append(("s", "s", 0))
wf.residues.append(G.ResTerm(0, 0, lambda env: env["s"]))
bw = G.WeakForm(inner_vars=[("T", 0, 0, 0)])
bw.residues.append(G.ResTerm(0, 0, lambda env: h * (T0 - env["T"])))
bw.linear_gradients.append(G.GradTerm(0, 0, 0, 0, lambda env: -h))
gd = G.GenericDomain(mf.default_context(), space, msh.coords, msh.cp_ids, 1, wf, [(fac.element_ID, fac.element_eindex, bw)])
gd.controlpoints["s"] = torch.full((msh.ncp,), 1600.0, dtype=torch.float64, device="cuda")
gd.update_Time(); gd.initialize_dx(); gd.K_linear_func(); gd.update_x_star(); gd.K_nonlinear_func()
print("n", gd.A.n, "nnz", gd.A.nnz)
from metafem_jl_amd import _lib
for graphs, sell in ((0, 0), (1, 0), (1, 1)):
  _lib.lib.mfem_debug_set_graphs(graphs, 0)
  _lib.lib.mfem_debug_set_sell(sell)
  print("hipGraph replay of solver cycles:", "on" if graphs else "off", "| row-sorted sliced ELL:", "on" if sell else "off")
  for name, sv, s in (("cg", mf.cg_, 0), ("bicgstabl(2)", mf.bicgstabl_GS_, 2), ("idrs(8)", mf.idrs_, 8), ("cgs2", mf.cgs2_, 0)):
    Kc = gd.K_total.clone() if sv != mf.cg_ else -gd.K_total
    b = gd.residue if sv != mf.cg_ else -gd.residue
    for rep in range(3):
        x, st = mf.iterative_Solve(gd.A, Kc, b, 1e-10, Sv_func=sv, maxiter=2000, max_pass=10, s=s)
    print(f"  {name}: {st.iterations} iterations, {st.spmv_count} SpMV, {st.solve_ms:.2f} ms -> {1e3 * st.solve_ms / max(st.spmv_count, 1):.1f} us per SpMV-equivalent, converged {st.converged}")
