"""Where do the lattice-tile layouts start to pay?  CG (hex-27) / BiCGStab(2) (elasticity) step time with the default size thresholds (small systems:
CSR tile kernel + cycle graphs) against the layouts forced (mfem_debug_set_layout_min_rows(0, 0)).  usage: probe_lat_threshold.py"""
import sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import metafem_jl_amd as mf
from metafem_jl_amd import _lib
lam, mu = 0.5769230769230769, 0.38461538461538464
def step_time(A, K, rhs, sv, kw, a_it, c_it):
    def solve(it):
        best, sp = 1e9, 0
        for _ in range(3):
            _, st = mf.iterative_Solve(A, K, rhs, 1e-300, Sv_func=sv, maxiter=it, max_pass=1, fixed_iterations=True, **kw)
            best = min(best, st.solve_ms); sp = st.spmv_count
        return best, sp
    (a, sa), (c, sc) = solve(a_it), solve(c_it)
    return (c - a) / max(sc - sa, 1)
for kind, sizes in (("hex27", (12, 16, 24, 32, 48)), ("elast", (24, 32, 48, 64))):
    for N in sizes:
        out = []
        for forced in (0, 1):
            _lib.lib.mfem_debug_set_layout_min_rows(0, 0) if forced else _lib.lib.mfem_debug_set_layout_min_rows(262144, 1000000)
            if kind == "hex27":
                b = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N), 2, 5); A = b.pattern(1)
                K = b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
                rhs = mf.FEM_rand(A.n, 5, 0) - 0.5
                t = step_time(A, K, rhs, mf.cg_, {}, 40, 160)
            else:
                b = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N), 1, 3); A = b.pattern(3)
                K = b.assemble_elasticity(A, lam, mu, 1000.0, mf.FACE_BITS["x0"])
                rhs = mf.FEM_rand(A.n, 5, 0) - 0.5
                t = step_time(A, K, rhs, mf.bicgstabl_GS_, dict(s=2), 12, 48)
            out.append(t)
            n = A.n
            del b, A, K, rhs
            torch.cuda.empty_cache()
        print(f"{kind} {N}^3: n = {n}: default thresholds {out[0]*1e3:.1f} us per step, layouts forced {out[1]*1e3:.1f} us", flush=True)
_lib.lib.mfem_debug_set_layout_min_rows(262144, 1000000)
