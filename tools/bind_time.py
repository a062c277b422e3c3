"""Per-solve work (layout bind, |diag|, scaling, symmetry check, first / true residual) of mfem_solve on the three big configs:
solve_ms of a short fixed-iteration solve minus its iterations at the loop's rate.  usage: bind_time.py [c2 c3 c4]"""
import sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import metafem_jl_amd as mf

lam, mu = 0.5769230769230769, 0.38461538461538464
which = sys.argv[1:] or ["c2", "c3", "c4"]

def overhead(A, K, R, sv, s, unit):
    def run(it):
        best = 1e9
        for _ in range(3):
            _, st = mf.iterative_Solve(A, K, R, 1e-300, Sv_func=sv, maxiter=it, max_pass=1, s=s, fixed_iterations=True)
            best = min(best, st.solve_ms)
        return best, st.spmv_count
    a, na = run(4 * unit)
    b, nb = run(24 * unit)
    per = (b - a) / (nb - na)
    return a - per * na, per, a, na

for cfg in which:
    if cfg == "c2":
        b = mf.make_Brick((1.0, 1.0, 1.0), (256,) * 3, 1, 3)
        A = b.pattern(1)
        K = b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
        R = mf.FEM_rand(A.n, 3, 0) - 0.5
        runs = [("cg", mf.cg_, 0, 1), ("bicgstabl2", mf.bicgstabl_GS_, 2, 2)]
    elif cfg == "c3":
        b = mf.make_Brick((1.0, 1.0, 1.0), (128,) * 3, 1, 3)
        A = b.pattern(3)
        K = b.assemble_elasticity(A, lam, mu, 1000.0, mf.FACE_BITS["x0"])
        R = mf.FEM_rand(A.n, 3, 0) - 0.5
        runs = [("bicgstabl2", mf.bicgstabl_GS_, 2, 2), ("idrs8", mf.idrs_, 8, 9), ("cg", mf.cg_, 0, 1)]
    else:
        b = mf.make_Brick((1.0, 1.0, 1.0), (128,) * 3, 2, 5)
        A = b.pattern(1)
        K = b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
        R = mf.FEM_rand(A.n, 3, 0) - 0.5
        runs = [("cg", mf.cg_, 0, 1), ("bicgstabl2", mf.bicgstabl_GS_, 2, 2)]
    for name, sv, s, unit in runs:
        ov, per, a, na = overhead(A, K, R, sv, s, unit)
        print(f"{cfg} {name:11s} per-solve work {ov:7.3f} ms   per SpMV-step {per:7.4f} ms   (short solve {a:.3f} ms, {na} SpMVs)")
    del b, A, K, R
    torch.cuda.empty_cache()
