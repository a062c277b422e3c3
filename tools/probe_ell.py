"""CG per-iteration time with the slot-major (ELL) SpMV inside mfem_solve vs the CSR tile kernel, hex-8 thermal N^3."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metafem_jl_amd as mf
from metafem_jl_amd import _lib
lib = _lib.lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
brick = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N))
A = brick.pattern(1)
K = brick.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
b = torch.ones(A.n, dtype=torch.float64, device="cuda")
bytes_spmv = A.nnz * 12 + A.n * 16 + (A.n + 1) * 8
def read():
    tot, cnt = C.c_double(), C.c_int64()
    _lib.check(lib.mfem_prof_spmv_read(brick.ctx._h, C.byref(tot), C.byref(cnt), 1))
    return tot.value / max(cnt.value, 1)
res = {}
names = {0: "1 row x9", 1: "2 rows x9", 2: "1 row x27", 3: "2 rows x27", 4: "1 row x3", 5: "2 rows x3", 6: "2 rows x1", 7: "2 rows x2",
         8: "2 rows x4", 9: "2 rows x5", 10: "2 rows x6", 11: "4 rows x1 nt", 12: "4 rows x1", 13: "2 rows pipelined"}
cfgs = [0] + [1 | (6 << 4) | (m << 8) | (b << 24) for b in (0, 1, 2, 3) for m in (6, 8)]
lib.mfem_debug_set_graphs(0, 0)
for ell in cfgs:
    lib.mfem_debug_set_ell(ell)
    _lib.check(lib.mfem_prof_spmv_enable(brick.ctx._h, 0))
    x, st = mf.iterative_Solve(A, K, b, 1e-30, Sv_func=mf.cg_, maxiter=200, max_pass=1, fixed_iterations=True)
    x, st = mf.iterative_Solve(A, K, b, 1e-30, Sv_func=mf.cg_, maxiter=200, max_pass=1, fixed_iterations=True)
    _lib.check(lib.mfem_prof_spmv_enable(brick.ctx._h, 1)); read()
    x2, st2 = mf.iterative_Solve(A, K, b, 1e-30, Sv_func=mf.cg_, maxiter=200, max_pass=1, fixed_iterations=True)
    ms = read()
    res[ell & 1] = x
    print(f"ell={ell & 1} dia={0 if ell & 2 or not ell & 1 else 1} variant {names[(ell >> 4) & 15]} dia-variant {(ell >> 16) & 15} block-code {(ell >> 24) & 3} grid x{(ell >> 8) & 255}: solve {st.solve_ms:.1f} ms for 200 CG iterations = {st.solve_ms/200:.4f} ms/it | SpMV in CG {ms:.4f} ms = {bytes_spmv/ms/1e6:.0f} GB/s "
          f"({bytes_spmv/ms/1e6/80:.1f}% of 8 TB/s)", flush=True)
print("max rel diff of the 200-iteration iterate:", float((res[0] - res[1]).abs().max() / res[0].abs().max()))
