"""Classic PCG recurrence (r carried / z = M^-1 r carried) against the single-reduction (Chronopoulos-Gear) form on one GPU at N^3: time per iteration (the second form is what
every rank runs at N > 1, so its single-GPU cost is the compute side of the weak-scaling efficiency)."""
import sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import metafem_jl_amd as mf
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
b = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N))
A = b.pattern(1)
K = b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
rhs = torch.ones(A.n, dtype=torch.float64, device="cuda")
for var, name in ((1, "classic"), (3, "classic, z carried"), (2, "single reduction"), (1, "classic"), (3, "classic, z carried")):
    mf.iterative_Solve(A, K, rhs, 1e-30, Sv_func=mf.cg_, maxiter=50, max_pass=1, fixed_iterations=True, cg_variant=var)
    xs, st = mf.iterative_Solve(A, K, rhs, 1e-30, Sv_func=mf.cg_, maxiter=200, max_pass=1, fixed_iterations=True, cg_variant=var)
    print(f"N={N} {name}: {st.solve_ms / 200:.4f} ms per iteration, |x| {float(xs.norm()):.12e}", flush=True)
