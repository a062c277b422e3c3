"""Per-solve work of CG (layout copy k_dia_vals + checks) with the copy's fast path for full swept tiles (default), without it (bit 29 of mfem_debug_set_ell:
the software-pipelined general path) and without either (bits 28 + 29), at N^3:
two fixed-iteration solves (16 and 116 iterations), best of 3 -> per-solve work = t16 - 16 * per-iteration.  usage: dia_pipe_time.py N [N ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metafem_jl_amd as mf
from metafem_jl_amd import _lib
for N in [int(a) for a in sys.argv[1:]] or [256]:
    brick = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N))
    A = brick.pattern(1)
    K = brick.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
    b = torch.ones(A.n, dtype=torch.float64, device="cuda")
    def best(its, var):
        t = 1e9
        for _ in range(3):
            x, st = mf.iterative_Solve(A, K, b, 1e-30, Sv_func=mf.cg_, maxiter=its, max_pass=1, fixed_iterations=True, cg_variant=var)
            t = min(t, st.solve_ms)
        return t, x
    for var in (3, 4):
        xs = {}
        for knob, tag in ((0, "fast"), (1 << 29, "pipelined"), (3 << 28, "plain"), (0, "fast"), (1 << 29, "pipelined")):
            _lib.lib.mfem_debug_set_ell(1 | knob)
            (t0, x), (t1, _) = best(16, var), best(116, var)
            per = (t1 - t0) / 100
            xs.setdefault(tag, x)
            print(f"N {N} cg_variant {var} copy {tag:9s}: per iteration {per:.4f} ms, per-solve work {t0 - 16 * per:.3f} ms", flush=True)
        _lib.lib.mfem_debug_set_ell(1)
        print(f"N {N} cg_variant {var}: solutions bitwise equal: {bool(torch.equal(xs['fast'], xs['plain']) and torch.equal(xs['fast'], xs['pipelined']))}", flush=True)
    del brick, A, K, b
    torch.cuda.empty_cache()
