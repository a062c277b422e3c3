"""The reference's own solver settings on the lattice-tile layouts: idrs!(s = 8) and bicgstabl_GS!(l = 2) with Pr_Jacobi! (02_Preconditioner.jl:32-37
defaults) on hex-27 128^3 (C4) and hex-8 elasticity 128^3 (C3): time per SpMV-equivalent step, tiles against the layouts they replaced.
usage: probe_lat_solvers.py [N]"""
import sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import metafem_jl_amd as mf
from metafem_jl_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
lam, mu = 0.5769230769230769, 0.38461538461538464
for kind in ("hex27", "elast"):
    for sv, name, kw in ((mf.idrs_, "idrs!(8)", dict(s=8)), (mf.bicgstabl_GS_, "bicgstabl_GS!(2)", dict(s=2)), (mf.cg_, "cg!", {})):
        res = []
        for lat in (1, 0):
            _lib.lib.mfem_debug_set_lat27(lat); _lib.lib.mfem_debug_set_lat8(lat)
            if kind == "hex27":
                b = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N), 2, 5); A = b.pattern(1)
                K = b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
            else:
                b = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N), 1, 3); A = b.pattern(3)
                K = b.assemble_elasticity(A, lam, mu, 1000.0, mf.FACE_BITS["x0"])
            rhs = mf.FEM_rand(A.n, 5, 0) - 0.5
            def solve(it):
                best, sp = 1e9, 0
                for _ in range(2):
                    _, st = mf.iterative_Solve(A, K, rhs, 1e-300, Sv_func=sv, maxiter=it, max_pass=1, fixed_iterations=True, **kw)
                    best = min(best, st.solve_ms); sp = st.spmv_count
                return best, sp
            (a, sa), (c, sc) = solve(16), solve(64)
            res.append((c - a) / max(sc - sa, 1))
            del b, A, K, rhs
            torch.cuda.empty_cache()
        print(f"{kind} {N}^3 {name:18s}: {res[0]:.4f} ms per SpMV-equivalent on the lattice tiles, {res[1]:.4f} on the layout before ({res[1] / res[0]:.2f} x)", flush=True)
_lib.lib.mfem_debug_set_lat27(1); _lib.lib.mfem_debug_set_lat8(1)
