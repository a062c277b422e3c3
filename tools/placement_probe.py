"""Does the time of the 512^3 CG iteration depend on where the solver's workspace lands?  Several trials in ONE process: the workspace base is placed at
align_up(raw, align) + offset (mfem_debug_set_ws_placement), a fresh context per trial.  usage: placement_probe.py [N]"""
import ctypes as C, gc, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metafem_jl_amd as mf
from metafem_jl_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512



MB = 1 << 20
trials = [(0, 0)] + [(1 << 30, k * 2 * MB) for k in (0, 1, 2, 3, 4, 8, 16, 32, 64, 128, 256)] + [(1 << 30, k * 4096) for k in (1, 2, 16, 64)] + [(0, 0)]
brickctx = None
for align, off in trials:
    _lib.lib.mfem_debug_set_ws_placement(align, off)
    ctx = mf.Context(0)
    brick = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N), ctx=ctx)
    A = brick.pattern(1)
    K = brick.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
    b = torch.ones(A.n, dtype=torch.float64, device="cuda")
    best = 1e9
    for _ in range(2):
        _, st = mf.iterative_Solve(A, K, b, 1e-30, Sv_func=mf.cg_, maxiter=64, max_pass=1, fixed_iterations=True)
        best = min(best, st.solve_ms)
    ws = _lib.lib.mfem_debug_ws_address(ctx._h)
    print(f"align {align:#x} offset {off:#10x}: {best / 64:.4f} ms/it  ws at {ws:#x}  K at {K.data_ptr():#x}", flush=True)
    del brick, A, K, b
    ctx.close()
    del ctx
    gc.collect()
    torch.cuda.empty_cache()
_lib.lib.mfem_debug_set_ws_placement(0, 0)
