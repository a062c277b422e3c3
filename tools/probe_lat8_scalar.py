"""hex-8 thermal (one field): idrs!(8) / bicgstabl_GS!(2) with Pr_Jacobi! on the lattice tiles (mode 5, F = 1) against the diagonal-slotted layout with the
scaling folded into its copy (mode 2): time per SpMV-equivalent step.  usage: probe_lat8_scalar.py [N ...]"""
import sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import metafem_jl_amd as mf
from metafem_jl_amd import _lib
for N in [int(a) for a in sys.argv[1:]] or [256]:
    for sv, name, kw in ((mf.idrs_, "idrs!(8)", dict(s=8)), (mf.bicgstabl_GS_, "bicgstabl_GS!(2)", dict(s=2))):
        res = []
        for lat in (1, 0):
            _lib.lib.mfem_debug_set_lat8(lat)
            b = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N), 1, 3); A = b.pattern(1)
            K = b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
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
        print(f"hex-8 thermal {N}^3 {name:18s}: {res[0]:.4f} ms per SpMV-equivalent on the lattice tiles, {res[1]:.4f} on mode 2 ({res[1] / res[0]:.2f} x)", flush=True)
_lib.lib.mfem_debug_set_lat8(1)
