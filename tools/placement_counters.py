"""Workload of the placement experiment (round 4): the 512^3 headline matrix, ONE capped CG solve with the workspace placement
trial ON -- the trial times a warm + two solver SpMVs (k_spmv_symp<0>) on each of up to three allocations of the 45 GB workspace, so one process shows
both kinds of backing memory.  Under `rocprofv3 --kernel-trace --pmc ...` every one of those dispatches gets a duration and counter values
(tools/run_placement_counters.sh; summary: tools/placement_counters_summary.py).  usage: placement_counters.py [N]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import metafem_jl_amd as mf
from metafem_jl_amd import _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
_lib.lib.mfem_debug_set_ws_trial(1)
b = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N))
A = b.pattern(1)
K = b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
s = torch.full((A.n,), 1600.0, dtype=torch.float64, device="cuda")
R = b.residual_thermal(torch.zeros(A.n, dtype=torch.float64, device="cuda"), 0.6, 25.0, 293.15, 0x3F, s=s)
del s
x, st = mf.iterative_Solve(A, K, R, 1e-300, Sv_func=mf.cg_, maxiter=6, max_pass=1, fixed_iterations=True)
torch.cuda.synchronize()
log = (C.c_double * 4)()
_lib.lib.mfem_debug_ws_trial_log(mf.default_context()._h, log)
print("TRIAL ms_for_two_spmvs", [round(v, 4) for v in log], "ws", hex(_lib.lib.mfem_debug_ws_address(mf.default_context()._h)), flush=True)
