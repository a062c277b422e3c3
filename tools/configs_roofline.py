"""BASELINE configs[2] (hex-8 elasticity 128^3) and configs[3] (hex-27 thermal 128^3) on one MI355X: assembly, residual, the CSR
kernel behind mul! and the Krylov loop's SpMV, each with its roofline object.  Prints one JSON document (profiles/r02_c3_c4_roofline.json).
HBM peak 8 TB/s, FP64 vector / matrix peak 78.6 TFLOP/s (MI355X_MICROARCH.md)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metafem_jl_amd as mf

HBM, FP64 = 8000.0, 78.6


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def hbm(nbytes, ms, **kw):
    return {"bound": "hbm", "ms": round(ms, 4), "algorithmic_bytes": int(nbytes), "achieved": round(nbytes / ms / 1e6, 1), "peak": HBM,
            "unit": "GB/s", "frac": round(nbytes / ms / 1e6 / HBM, 4), **kw}


out = {}
# ---- C3: hex-8 elasticity 128^3, 3 DOF per node
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
b = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N))
A = b.pattern(3)
lam, mu = 0.5769230769230769, 0.38461538461538464
K = torch.empty(A.nnz, dtype=torch.float64, device="cuda")
c3 = {"n": A.n, "nnz": A.nnz}
ms = timeit(lambda: b.assemble_elasticity(A, lam, mu, 1000.0, mf.FACE_BITS["x0"], out=K))
c3["assemble_matrix"] = {**hbm(A.nnz * 8 + 24 * A.n // 3, ms), "note": "bytes = every CSR value written once + coordinates; the kernel is FP64-VALU / LDS bound (profiles/r02_hex8_kernel_counters.json)"}
xs = 0.01 * (mf.FEM_rand(A.n, 2, 0) - 0.5)
R = torch.empty(A.n, dtype=torch.float64, device="cuda")
ms = timeit(lambda: b.residual_elasticity(xs, lam, mu, 1000.0, mf.FACE_BITS["x0"], mf.FACE_BITS["y1"], (0.0, 1.0, 0.0, 0.0, 0.0, 0.0), out=R))
c3["residual"] = {"ms": round(ms, 4)}
rhs = mf.FEM_rand(A.n, 1, 0) - 0.5
y = torch.empty_like(rhs)
csr_bytes = A.nnz * 12 + A.n * 16 + (A.n + 1) * 8
c3["csr_kernel"] = hbm(csr_bytes, timeit(lambda: mf.mul_(y, A, K, rhs), reps=20), kernel="mfem_spmv_csr (mul!)")
for name, sv, s, it in (("bicgstabl2", mf.bicgstabl_GS_, 2, 40), ("idrs8", mf.idrs_, 8, 45)):
    mf.iterative_Solve(A, K, rhs, 1e-300, Sv_func=sv, maxiter=it, max_pass=1, s=s)
    _, st = mf.iterative_Solve(A, K, rhs, 1e-300, Sv_func=sv, maxiter=it, max_pass=1, s=s)
    c3[name] = {"solve_ms": round(st.solve_ms, 3), "spmv_equivalents": st.spmv_count, "ms_per_spmv_equivalent": round(st.solve_ms / st.spmv_count, 4),
                "csr_equivalent_frac": round(csr_bytes / (st.solve_ms / st.spmv_count) / 1e6 / HBM, 4),
                "note": "whole Krylov step (SpMV on the solver layout + its vector kernels) per SpMV, priced with the CSR bytes of one SpMV"}
out["C3 hex-8 elasticity %d^3" % N] = c3
del b, A, K, xs, R, rhs, y
torch.cuda.empty_cache()
# ---- C4: hex-27 thermal 128^3
b = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N), 2, 5)
A = b.pattern(1)
K = torch.empty(A.nnz, dtype=torch.float64, device="cuda")
c4 = {"n": A.n, "nnz": A.nnz}
ms = timeit(lambda: b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F, out=K), reps=3)
flops = 2 * 27 * 27 * 81 * N ** 3
c4["assemble_matrix"] = {"bound": "mfma", "ms": round(ms, 3), "useful_flop": flops, "achieved": round(flops / ms / 1e9, 2), "peak": FP64, "unit": "TFLOP/s",
                         "frac": round(flops / ms / 1e9 / FP64, 4),
                         "traffic_note": "two passes move 12.2 GB (Ke -> scratch) + 12.2 GB (scratch -> gather) + 8.6 GB (CSR values) = 33 GB: 6.6 ms at 5 TB/s"}
x = mf.FEM_rand(A.n, 1, 0) + 300.0
s = torch.full((A.n,), 1600.0, dtype=torch.float64, device="cuda")
R = torch.empty(A.n, dtype=torch.float64, device="cuda")
c4["residual"] = {"ms": round(timeit(lambda: b.residual_thermal(x, 0.6, 25.0, 293.15, 0x3F, s=s, out=R), reps=3), 3)}
y = torch.empty(A.n, dtype=torch.float64, device="cuda")
csr_bytes = A.nnz * 12 + A.n * 16 + (A.n + 1) * 8
c4["csr_kernel"] = hbm(csr_bytes, timeit(lambda: mf.mul_(y, A, K, x), reps=20), kernel="mfem_spmv_csr (mul!)")
rhs = mf.FEM_rand(A.n, 3, 0) - 0.5
mf.iterative_Solve(A, K, rhs, 1e-300, Sv_func=mf.cg_, maxiter=50, max_pass=1, fixed_iterations=True)
_, st = mf.iterative_Solve(A, K, rhs, 1e-300, Sv_func=mf.cg_, maxiter=50, max_pass=1, fixed_iterations=True)
_, st200 = mf.iterative_Solve(A, K, rhs, 1e-300, Sv_func=mf.cg_, maxiter=200, max_pass=1, fixed_iterations=True)
per_it = (st200.solve_ms - st.solve_ms) / 150  # the loop alone: the per-solve part (layout bind, |diag|, first residual) cancels
c4["cg"] = {"ms_per_iteration": round(per_it, 4), "csr_equivalent_frac_of_one_spmv": round(csr_bytes / per_it / 1e6 / HBM, 4),
            "solve_ms_50_iterations": round(st.solve_ms, 2), "solve_ms_200_iterations": round(st200.solve_ms, 2),
            "note": "CG iteration (SpMV on the row-sorted sliced layout + vector kernels) = (200-iteration solve - 50-iteration solve) / 150; "
                    "the solves themselves include the per-solve layout bind"}
out["C4 hex-27 thermal %d^3" % N] = c4
print(json.dumps(out, indent=1))
