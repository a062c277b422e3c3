"""One workload for the rocprofv3 --pmc passes of round 3 (tools/run_pmc_r03.sh): build the matrix of a BASELINE config, run its
assembly kernels, a few launches of the CSR kernel behind mul!, one short solve on the solver layout and a calibration kernel of
known bytes (mfem_axpby: 2 vectors read, 1 written).
usage: pmc_leg.py c2_256 | c2_512 | c3_128 | c4_128 | ref_idrs8_256 | nitsche_c2_256 | nitsche_c4_128 [launches]"""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0] if "/" in __file__ else "..")
import metafem_jl_amd as mf

leg = sys.argv[1]
L = int(sys.argv[2]) if len(sys.argv) > 2 else 4
# round 5: the reference's own solver / boundary-condition legs of bench.py (REF_LEGS): ref_idrs8_256, nitsche_c2_256, nitsche_c4_128
REF = {"ref_idrs8": ("c2", "idrs8", False), "nitsche_c2": ("c2", "bicgstabl2", True), "nitsche_c4": ("c4", "bicgstabl2", True)}
cfg, N = leg.rsplit("_", 1)
N = int(N)
lam, mu = 0.5769230769230769, 0.38461538461538464
if cfg in ("u20_1", "u20_3", "tet10_1", "tet10_3"):
    # round 6: the unstructured hex-20 legs of bench.py (u20_thermal_96 / u20_elasticity_96): the same mesh and domain as bench_legs.Bench.unstructured_leg
    import bench
    import bench_legs as BL
    from metafem_jl_amd import generic as G, physics
    import numpy as np

    F = int(cfg[-1])
    Bn = BL.Bench(bench.parse_args([]))
    space, msh, fac = Bn.unstructured_mesh(N, shape="SIMPLEX" if cfg.startswith("tet10") else "CUBE")
    if F == 1:
        gd = G.GenericDomain(Bn.ctx, space, msh.coords, msh.cp_ids, 1, physics.thermal_domain(3, 0.6), [(fac.element_ID, fac.element_eindex, physics.thermal_convection(25.0, 293.15))])
        gd.controlpoints["s"] = torch.full((msh.ncp,), 1600.0, dtype=torch.float64, device="cuda")
    else:
        c = fac.centroid
        wall, top = fac.select(np.abs(c[:, 0]) < 1e-9), fac.select(np.abs(c[:, 1] - 1.0) < 1e-9)
        gd = G.GenericDomain(Bn.ctx, space, msh.coords, msh.cp_ids, 3, physics.elasticity_domain(3, lam, mu),
                             [(wall.element_ID, wall.element_eindex, physics.penalty([0, 1, 2], 1000.0)), (top.element_ID, top.element_eindex, physics.traction(3, "sl", rows=[1]))])
        for v in (2, 4, 6):
            gd.controlpoints[f"sl{v}"] = torch.full((msh.ncp,), 1.0 if v == 2 else 0.0, dtype=torch.float64, device="cuda")
    for _ in range(2):
        gd.K_linear_func()
        gd.K_nonlinear_func()
    A, K, R = gd.A, gd.K_total, gd.residue
    solve = lambda: mf.iterative_Solve(A, K, R, 1e-300, Sv_func=mf.idrs_, Pr_func=mf.Pr_Jacobi_, maxiter=27, max_pass=1, s=8, fixed_iterations=True)
elif cfg in REF:
    base, solver, nitsche = REF[cfg]
    b = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N), 1 if base == "c2" else 2, 3 if base == "c2" else 5)
    A = b.pattern(1)
    fixed = mf.FACE_BITS["x0"] if nitsche else 0
    fix = dict(fixed_faces=fixed, h_penalty=1000.0 if fixed else 0.0, Tw=1173.15 if fixed else 0.0)
    K = b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F & ~fixed, **fix)
    s = torch.full((A.n,), 1600.0, dtype=torch.float64, device="cuda")
    R = b.residual_thermal(torch.zeros(A.n, dtype=torch.float64, device="cuda"), 0.6, 25.0, 293.15, 0x3F & ~fixed, s=s, **fix)
    if solver == "idrs8":
        solve = lambda: mf.iterative_Solve(A, K, R, 1e-300, Sv_func=mf.idrs_, maxiter=27, max_pass=1, s=8, fixed_iterations=True)
    else:
        solve = lambda: mf.iterative_Solve(A, K, R, 1e-300, Sv_func=mf.bicgstabl_GS_, maxiter=12, max_pass=1, s=2, fixed_iterations=True)
elif cfg == "c2":
    b = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N), 1, 3)
    A = b.pattern(1)
    K = b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
    s = torch.full((A.n,), 1600.0, dtype=torch.float64, device="cuda")
    R = b.residual_thermal(torch.zeros(A.n, dtype=torch.float64, device="cuda"), 0.6, 25.0, 293.15, 0x3F, s=s)
    solve = lambda: mf.iterative_Solve(A, K, R, 1e-300, Sv_func=mf.cg_, maxiter=20, max_pass=1, fixed_iterations=True)
elif cfg == "c3":
    b = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N), 1, 3)
    A = b.pattern(3)
    K = b.assemble_elasticity(A, lam, mu, 1000.0, mf.FACE_BITS["x0"])
    R = b.residual_elasticity(torch.zeros(A.n, dtype=torch.float64, device="cuda"), lam, mu, 1000.0, mf.FACE_BITS["x0"], mf.FACE_BITS["y1"],
                              (0.0, 1.0, 0.0, 0.0, 0.0, 0.0))
    solve = lambda: mf.iterative_Solve(A, K, R, 1e-300, Sv_func=mf.bicgstabl_GS_, maxiter=12, max_pass=1, s=2, fixed_iterations=True)
else:
    b = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N), 2, 5)
    A = b.pattern(1)
    K = b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
    s = torch.full((A.n,), 1600.0, dtype=torch.float64, device="cuda")
    R = b.residual_thermal(torch.zeros(A.n, dtype=torch.float64, device="cuda"), 0.6, 25.0, 293.15, 0x3F, s=s)
    solve = lambda: mf.iterative_Solve(A, K, R, 1e-300, Sv_func=mf.cg_, maxiter=20, max_pass=1, fixed_iterations=True)
x = mf.FEM_rand(A.n, 3, 0)
y = torch.zeros(A.n, dtype=torch.float64, device="cuda")
for _ in range(L):
    mf.mul_(y, A, K, x)
for _ in range(3):
    mf.axpby_(0.5, x, 0.25, y)  # calibration: 2 n doubles read, n written
solve()
# C3 / C4: the same solve once more on the layouts the lattice tiles replaced (modes 2 / 3), so that one pass prices both kernels
from metafem_jl_amd import _lib
if cfg == "c4":
    # the two-pass MFMA assembly (what a mesh with a non-affine element takes: this mesh is assembled without it, k_hex27_direct) and the CG iteration as
    # SpMV pass 1 + pass 2 + update (this solve fused pass 2 into the update), so that the counters of those kernels exist too
    _lib.lib.mfem_debug_set_hex27(1 << 9)
    for _ in range(2):
        b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F, out=K)
    _lib.lib.mfem_debug_set_hex27(0)
    _lib.lib.mfem_debug_set_lat27(1 | 4)
    solve()
    _lib.lib.mfem_debug_set_lat27(1)
if cfg in REF:
    # the same solve without the skew remainder (a nonsymmetric K then takes the layouts that read every entry: round 4's path) and without the tiles
    _lib.lib.mfem_debug_set_remainder(0)
    _lib.lib.mfem_debug_set_lat27(0)
    _lib.lib.mfem_debug_set_lat8(0)
    solve()
    _lib.lib.mfem_debug_set_remainder(1)
    _lib.lib.mfem_debug_set_lat27(1)
    _lib.lib.mfem_debug_set_lat8(1)
if cfg in ("c3", "c4"):
    _lib.lib.mfem_debug_set_lat27(0)
    _lib.lib.mfem_debug_set_lat8(0)
    solve()
    _lib.lib.mfem_debug_set_lat27(1)
    _lib.lib.mfem_debug_set_lat8(1)
torch.cuda.synchronize()
byts, cols = A.spmv_bytes()
print("LEG", leg, "n", A.n, "nnz", A.nnz, "csr_design_bytes", byts, "cols_read", cols)
