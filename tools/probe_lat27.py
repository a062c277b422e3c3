"""hex-27 thermal matrix (config C4): the symmetric lattice-tile layout (solver layout mode 4, spmv_lat27.hip) against the sliced layout
(mode 3) and the CSR kernel: SpMV equality, CG iteration time (200- minus 50-iteration solve), bind cost, converged solutions.
usage: probe_lat27.py [N ...]      (N = elements per direction; odd shapes: "5x6x9")"""
import sys, ctypes as C, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import metafem_jl_amd as mf
from metafem_jl_amd import _lib

shapes = [a for a in sys.argv[1:]] or ["128"]
for sh in shapes:
    dims = tuple(int(v) for v in sh.split("x")) if "x" in sh else (int(sh),) * 3
    if dims[0] * dims[1] * dims[2] < 100000:
        _lib.lib.mfem_debug_set_layout_min_rows(0, 0)
    res = {}
    for lat in (1, 0):
        _lib.lib.mfem_debug_set_lat27(lat)
        b = mf.make_Brick((1.0, 1.0, 1.0), dims, 2, 5)
        A = b.pattern(1)
        K = b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
        x = mf.FEM_rand(A.n, 3, 0) - 0.5
        y0 = torch.zeros(A.n, dtype=torch.float64, device="cuda"); y1 = torch.full_like(y0, 0.25)
        mf.mul_(y0, A, K, x)
        c0 = int(_lib.lib.mfem_debug_lat27_spmv_count())
        _lib.check(_lib.lib.mfem_spmv_solver_layout(b.ctx._h, A._h, K.data_ptr(), x.data_ptr(), y1.data_ptr(), 2.0, -1.0))
        used = int(_lib.lib.mfem_debug_lat27_spmv_count()) - c0
        err = float((2.0 * y0 - 0.25 - y1).abs().max() / y0.abs().max())
        mode, slots, npad, reg = C.c_int32(), C.c_int32(), C.c_int64(), C.c_int64()
        _lib.check(_lib.lib.mfem_csr_solver_layout(b.ctx._h, A._h, C.byref(mode), C.byref(slots), C.byref(npad), C.byref(reg)))
        byts = C.c_int64(); _lib.check(_lib.lib.mfem_csr_solver_layout_bytes(b.ctx._h, A._h, C.byref(byts)))
        rhs = mf.FEM_rand(A.n, 5, 0) - 0.5
        def solve(it):
            best = 1e9
            for _ in range(2):
                _, st = mf.iterative_Solve(A, K, rhs, 1e-300, Sv_func=mf.cg_, maxiter=it, max_pass=1, fixed_iterations=True)
                best = min(best, st.solve_ms)
            return best
        a, c = solve(50), solve(200)
        xs, st = mf.iterative_Solve(A, K, rhs, 1e-10, Sv_func=mf.cg_, maxiter=4000, max_pass=1)
        res[lat] = xs.clone()
        print(f"{sh}: lat27={lat} mode {mode.value} (layout kernel launches {used}) spmv rel err {err:.1e}  CG iteration {(c - a) / 150:.4f} ms  "
              f"per-solve work {a - 50 * (c - a) / 150:.2f} ms  design bytes {byts.value / 1e9:.3f} GB  converged {st.converged} in {st.iterations} it",
              flush=True)
        del b, A, K, x, y0, y1, rhs
        torch.cuda.empty_cache()
    print(f"{sh}: converged solutions differ by {float((res[1] - res[0]).abs().max() / res[0].abs().max()):.2e}", flush=True)
_lib.lib.mfem_debug_set_lat27(1)
