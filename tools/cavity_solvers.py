import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import metafem_jl_amd as mf
from oracle import cavity, solvers
import test_gpu_generic as tg
od = cavity.build_cavity(40, Cb=8.0)
gd = tg._gpu_domain(mf, od, "Serendipity", 2, 5)
n = od.mesh.ncp
od.controlpoints["u1"], od.controlpoints["u2"] = np.zeros(n), np.zeros(n)
cavity.set_step_parameters(od, 0.1)
for k in ("uw1", "uw2", "taum", "tauc"):
    gd.controlpoints[k] = torch.tensor(od.controlpoints[k], device="cuda")
gd.K_linear_func(); gd.x_star.zero_(); gd.K_nonlinear_func()
b = gd.residue; res0 = mf.normalized_norm(b)
K = gd.K_total
d = mf.jacobi_by_diagonal(gd.A, K)
print("res0", res0, "diag min/max", float(d.min()), float(d.max()), "K abs max", float(K.abs().max()))
ref = solvers.solver_lu_cpu(gd.A.rowptr.cpu().numpy(), gd.A.colidx.cpu().numpy(), K.cpu().numpy(), b.cpu().numpy())
for name, sv, s in (("idrs4", mf.idrs_, 4), ("idrs8", mf.idrs_, 8), ("idrs20", mf.idrs_, 20), ("bicg2", mf.bicgstabl_GS_, 2), ("bicg4", mf.bicgstabl_GS_, 4),
                    ("bicg8", mf.bicgstabl_GS_, 8), ("cgs2", mf.cgs2_, 0)):
    for pr in (mf.Pr_Jacobi_, mf.Pr_Jacobi_colnorm_):
        dx, st = mf.iterative_Solve(gd.A, K, b, 1e-8, Sv_func=sv, Pr_func=pr, maxiter=5000, max_pass=20, s=s)
        err = np.abs(dx.cpu().numpy() - ref).max() / np.abs(ref).max()
        print(f"{name} pr={pr}: passes {st.passes} iters {st.iterations} final_res {st.final_res:.3e} conv {st.converged} err {err:.2e} ms {st.solve_ms:.0f}", flush=True)
