"""CG iterations at N^3 with a chosen mfem_debug_set_ell knob (0 = default: wave-private patch sweep; 0x800000 = workgroup-tile sweep;
0x400000 = plain diagonal-slotted kernel): for rocprofv3 --kernel-trace --stats / --pmc passes.  usage: symp_once.py [knob] [N] [iters]"""
import sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import metafem_jl_amd as mf
from metafem_jl_amd import _lib
knob = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
its = int(sys.argv[3]) if len(sys.argv) > 3 else 20
b = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N))
A = b.pattern(1)
K = b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
rhs = torch.ones(A.n, dtype=torch.float64, device="cuda")
_lib.lib.mfem_debug_set_ell(1 | knob)
xs, st = mf.iterative_Solve(A, K, rhs, 1e-30, Sv_func=mf.cg_, maxiter=its, max_pass=1, fixed_iterations=True)
torch.cuda.synchronize()
print("n", A.n, "ms/it", st.solve_ms / its)
