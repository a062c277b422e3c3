"""One short solve per config after a warm-up solve, for rocprofv3 --kernel-trace: prints nothing but markers via tiny axpby launches.
usage: solve_trace.py c2|c3|c4 cg|bicgstabl2|idrs8"""
import sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import metafem_jl_amd as mf
cfg, sv = sys.argv[1], sys.argv[2]
lam, mu = 0.5769230769230769, 0.38461538461538464
if cfg == "c3":
    b = mf.make_Brick((1.0, 1.0, 1.0), (128,) * 3, 1, 3); A = b.pattern(3)
    K = b.assemble_elasticity(A, lam, mu, 1000.0, mf.FACE_BITS["x0"])
elif cfg == "c4":
    b = mf.make_Brick((1.0, 1.0, 1.0), (128,) * 3, 2, 5); A = b.pattern(1)
    K = b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
else:
    b = mf.make_Brick((1.0, 1.0, 1.0), (256,) * 3, 1, 3); A = b.pattern(1)
    K = b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
R = mf.FEM_rand(A.n, 3, 0) - 0.5
S = {"cg": (mf.cg_, 0, 4), "bicgstabl2": (mf.bicgstabl_GS_, 2, 4), "idrs8": (mf.idrs_, 8, 9)}[sv]
for _ in range(2):
    mf.iterative_Solve(A, K, R, 1e-300, Sv_func=S[0], maxiter=S[2], max_pass=1, s=S[1], fixed_iterations=True)
marker = torch.zeros(7, dtype=torch.float64, device="cuda")
mf.axpby_(1.0, marker, 1.0, marker)   # marker: k_axpby with n = 7
_, st = mf.iterative_Solve(A, K, R, 1e-300, Sv_func=S[0], maxiter=S[2], max_pass=1, s=S[1], fixed_iterations=True)
mf.axpby_(1.0, marker, 1.0, marker)
torch.cuda.synchronize()
print("solve_ms", st.solve_ms, "spmvs", st.spmv_count)
