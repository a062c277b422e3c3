"""hex-27 128^3 (config C4): the sliced layout with rows sorted inside lattice regions of R^3 points (mfem_debug_set_sell bits 4-7 = R / 8)
and / or XCD-contiguous block walks (bit 2): SpMV equality with the CSR kernel, CG iteration time (200- minus 50-iteration solve).
usage: probe_sell_regions.py [N]"""
import sys, ctypes as C, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import metafem_jl_amd as mf
from metafem_jl_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for R, xcd in [(0, 0), (32, 0), (32, 1), (16, 1), (64, 1), (64, 0), (0, 1)]:
    _lib.lib.mfem_debug_set_sell(1 | (4 if xcd else 0) | ((R // 8) << 4))
    b = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N), 2, 5)
    A = b.pattern(1)
    K = b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
    x = mf.FEM_rand(A.n, 3, 0) - 0.5
    y0 = torch.zeros(A.n, dtype=torch.float64, device="cuda"); y1 = torch.zeros_like(y0)
    mf.mul_(y0, A, K, x)
    _lib.check(_lib.lib.mfem_spmv_solver_layout(b.ctx._h, A._h, K.data_ptr(), x.data_ptr(), y1.data_ptr(), 1.0, 0.0))
    err = float((y0 - y1).abs().max() / y0.abs().max())
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
    print(f"R={R:3d} xcd={xcd}: mode {mode.value} spmv rel err {err:.1e}  CG iteration {(c - a) / 150:.4f} ms  (50-it solve {a:.2f} ms)  layout bytes {byts.value / 1e9:.3f} GB", flush=True)
    del b, A, K, x, y0, y1, rhs
    torch.cuda.empty_cache()
_lib.lib.mfem_debug_set_sell(1)
