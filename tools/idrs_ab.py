"""idrs!(8) A/B for profiles/r06_idrs_streams.txt: shadow vectors as generated signs (default, round 6) against streamed U(0,1)
vectors (round 5), fused update + combine against the two kernels -- ms per 200-step solve on configs[1] (256^3 hex-8 thermal) and iterations / time to
1e-8 ||r0|| on C2 / C3 / C4.  usage: idrs_ab.py [quick]"""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import metafem_jl_amd as mf
from metafem_jl_amd import _lib

quick = len(sys.argv) > 1
lam, mu = 0.5769230769230769, 0.38461538461538464


def system(cfg, N):
    if cfg == "c3":
        b = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N), 1, 3)
        A = b.pattern(3)
        K = b.assemble_elasticity(A, lam, mu, 1000.0, mf.FACE_BITS["x0"])
        R = b.residual_elasticity(torch.zeros(A.n, dtype=torch.float64, device="cuda"), lam, mu, 1000.0, mf.FACE_BITS["x0"], mf.FACE_BITS["y1"],
                                  (0.0, 1.0, 0.0, 0.0, 0.0, 0.0))
        return b, A, K, R
    order, itg = (1, 3) if cfg == "c2" else (2, 5)
    b = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N), order, itg)
    A = b.pattern(1)
    K = b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
    s = torch.full((A.n,), 1600.0, dtype=torch.float64, device="cuda")
    R = b.residual_thermal(torch.zeros(A.n, dtype=torch.float64, device="cuda"), 0.6, 25.0, 293.15, 0x3F, s=s)
    return b, A, K, R


MODES = ((0, "signs generated, fused update+combine (default)"), (4, "signs generated, two kernels"), (2, "U(0,1) streamed, fused"),
         (6, "U(0,1) streamed, two kernels (round 5)"))
print("# idrs!(s = 8) + Pr_Jacobi!: 200 fixed steps, ms per solve (median of 5)")
for cfg, N in (("c2", 64 if quick else 256),):
    b, A, K, R = system(cfg, N)
    for bits, name in MODES:
        _lib.lib.mfem_debug_set_idrs(bits)
        ts = []
        for _ in range(6):
            _, st = mf.iterative_Solve(A, K, R, 1e-300, Sv_func=mf.idrs_, Pr_func=mf.Pr_Jacobi_, maxiter=200, max_pass=1, s=8, fixed_iterations=True)
            ts.append(st.solve_ms)
        ts = sorted(ts[1:])
        print(f"{cfg}_{N}  mode {bits}: {ts[len(ts) // 2]:8.2f} ms  ({A.n * st.spmv_count / (ts[len(ts) // 2] * 1e-3):.3e} DOF-updates/s, solve only)   {name}", flush=True)
    del b, A, K, R
    torch.cuda.empty_cache()
print("# time to ||r|| <= 1e-8 ||r0|| (maxiter 5000 per pass, 4 passes), seeds 0x5EED, 1, 2: iterations (passes) ms")
for cfg, N in (("c2", 48 if quick else 256), ("c3", 24 if quick else 128), ("c4", 16 if quick else 128)):
    b, A, K, R = system(cfg, N)
    r0 = mf.normalized_norm(R)
    for bits, name in ((0, "signs"), (2, "U(0,1)")):
        _lib.lib.mfem_debug_set_idrs(bits)
        row = []
        for seed in (0x5EED, 1, 2):
            _, st = mf.iterative_Solve(A, K, R, 1e-8 * r0, Sv_func=mf.idrs_, Pr_func=mf.Pr_Jacobi_, maxiter=5000, max_pass=4, s=8, seed=seed)
            row.append(f"{st.iterations:5d} ({st.passes}) {st.solve_ms:8.1f} ms{'' if st.converged else ' NOT CONVERGED'}")
        print(f"{cfg}_{N}  {name:7s}: " + " | ".join(row), flush=True)
    del b, A, K, R
    torch.cuda.empty_cache()
_lib.lib.mfem_debug_set_idrs(0)
