"""hex-8 elasticity matrix (3 fields, config C3): the symmetric lattice-tile layout (solver layout mode 5, spmv_lat8.hip) against the
diagonal-slotted layout (mode 2) and the CSR kernel: SpMV equality, time per BiCGStab(2) SpMV-equivalent step, per-solve work, converged solutions.
usage: probe_lat8.py [N ...]      (N = elements per direction; other shapes: "5x6x9")"""
import sys, ctypes as C, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import metafem_jl_amd as mf
from metafem_jl_amd import _lib

lam, mu = 0.5769230769230769, 0.38461538461538464
shapes = [a for a in sys.argv[1:]] or ["128"]
for sh in shapes:
    dims = tuple(int(v) for v in sh.split("x")) if "x" in sh else (int(sh),) * 3
    if dims[0] * dims[1] * dims[2] < 200000:
        _lib.lib.mfem_debug_set_layout_min_rows(0, 0)
    res = {}
    for lat in (1, 0):
        _lib.lib.mfem_debug_set_lat8(lat)
        b = mf.make_Brick((1.0, 1.0, 1.0), dims, 1, 3)
        A = b.pattern(3)
        K = b.assemble_elasticity(A, lam, mu, 1000.0, mf.FACE_BITS["x0"])
        x = mf.FEM_rand(A.n, 3, 0) - 0.5
        y0 = torch.zeros(A.n, dtype=torch.float64, device="cuda"); y1 = torch.full_like(y0, 0.25)
        mf.mul_(y0, A, K, x)
        c0 = int(_lib.lib.mfem_debug_lat8_spmv_count())
        _lib.check(_lib.lib.mfem_spmv_solver_layout(b.ctx._h, A._h, K.data_ptr(), x.data_ptr(), y1.data_ptr(), 2.0, -1.0))
        used = int(_lib.lib.mfem_debug_lat8_spmv_count()) - c0
        err = float((2.0 * y0 - 0.25 - y1).abs().max() / y0.abs().max())
        mode, slots, npad, reg = C.c_int32(), C.c_int32(), C.c_int64(), C.c_int64()
        _lib.check(_lib.lib.mfem_csr_solver_layout(b.ctx._h, A._h, C.byref(mode), C.byref(slots), C.byref(npad), C.byref(reg)))
        byts = C.c_int64(); _lib.check(_lib.lib.mfem_csr_solver_layout_bytes(b.ctx._h, A._h, C.byref(byts)))
        rhs = mf.FEM_rand(A.n, 5, 0) - 0.5
        def solve(it):
            best, sp = 1e9, 0
            for _ in range(2):
                _, st = mf.iterative_Solve(A, K, rhs, 1e-300, Sv_func=mf.bicgstabl_GS_, s=2, maxiter=it, max_pass=1, fixed_iterations=True)
                best = min(best, st.solve_ms); sp = st.spmv_count
            return best, sp
        (a, sa), (c, sc) = solve(12), solve(48)
        xs, st = mf.iterative_Solve(A, K, rhs, 1e-10, Sv_func=mf.bicgstabl_GS_, s=2, maxiter=4000, max_pass=4)
        res[lat] = xs.clone()
        per = (c - a) / max(sc - sa, 1)
        print(f"{sh}: lat8={lat} mode {mode.value} (layout kernel launches {used}, asymmetry {_lib.lib.mfem_debug_lat8_asymmetry(A._h):.1e}) spmv rel err {err:.1e}  "
              f"BiCGStab(2) step per SpMV {per:.4f} ms  per-solve work {a - sa * per:.2f} ms  design bytes {byts.value / 1e9:.3f} GB  "
              f"converged {st.converged} in {st.iterations} it", flush=True)
        del b, A, K, x, y0, y1, rhs
        torch.cuda.empty_cache()
    print(f"{sh}: converged solutions differ by {float((res[1] - res[0]).abs().max() / res[0].abs().max()):.2e}", flush=True)
_lib.lib.mfem_debug_set_lat8(1)
