"""On-box probe of the GENERIC path at scale (device geometry + pattern build + S3 operators), hex-8 thermal N^3 and hex-20
elasticity: per-phase timings with torch events."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metafem_jl_amd as mf
from metafem_jl_amd import element, generic as G, mesh as pm

def timeit(fn, reps=3, warm=1):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

def thermal_wf(dim, k):
    wf = G.WeakForm()
    for d in range(dim):
        wf.inner_vars.append((f"T_{d}", 0, 1 + d, 0))
        wf.residues.append(G.ResTerm(0, 1 + d, lambda env, d=d: -k * env[f"T_{d}"]))
        wf.linear_gradients.append(G.GradTerm(0, 1 + d, 0, 1 + d, lambda env: -k))
    wf.cp_ext_vars.append(("s", "s", 0))
    wf.residues.append(G.ResTerm(0, 0, lambda env: env["s"]))
    return wf

def conv_wf(h, Tenv):
    wf = G.WeakForm(inner_vars=[("T", 0, 0, 0)])
    wf.residues.append(G.ResTerm(0, 0, lambda env: h * (Tenv - env["T"])))
    wf.linear_gradients.append(G.GradTerm(0, 0, 0, 0, lambda env: -h))
    return wf

def elasticity_wf(dim, lam, mu):
    wf = G.WeakForm()
    for i in range(dim):
        for j in range(dim):
            wf.inner_vars.append((f"d{i}_{j}", i, 1 + j, 0))
    def sigma(env, i, j):
        s = mu * (env[f"d{i}_{j}"] + env[f"d{j}_{i}"])
        if i == j:
            s = s + lam * sum(env[f"d{m}_{m}"] for m in range(dim))
        return s
    for i in range(dim):
        for j in range(dim):
            wf.residues.append(G.ResTerm(i, 1 + j, lambda env, i=i, j=j: -sigma(env, i, j)))
            for kk in range(dim):
                for l in range(dim):
                    c = (lam if (i == j and kk == l) else 0.0) + mu * ((i == kk and j == l) + (i == l and j == kk))
                    if c != 0.0:
                        wf.linear_gradients.append(G.GradTerm(i, 1 + j, kk, 1 + l, lambda env, c=c: -c))
    return wf

def run(name, N, itp_type, order, itg, n_fields, dom_wf, bnd_wf):
    t0 = time.perf_counter()
    space = element.classical_space(3, itp_type, order, itg)
    vert, conn = pm.make_Brick((1.0, 1.0, 1.0), (N, N, N))
    msh = pm.mesh_Classical(vert, conn, space)
    fac = pm.get_BoundaryMesh(msh)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    gd = G.GenericDomain(mf.default_context(), space, msh.coords, msh.cp_ids, n_fields, dom_wf, [(fac.element_ID, fac.element_eindex, bnd_wf)])
    torch.cuda.synchronize(); t_setup = time.perf_counter() - t0
    gd.controlpoints["s"] = torch.full((msh.ncp,), 1600.0, dtype=torch.float64, device="cuda")
    print(f"{name}: nel {msh.nel} ncp {msh.ncp} n {gd.basicfield_size} nnz {gd.A.nnz} | host mesh {t_host*1e3:.0f} ms, device setup (geometry + pattern) {t_setup*1e3:.0f} ms", flush=True)
    gd.update_Time()
    ms_lin = timeit(gd.K_linear_func)
    gd.update_x_star()
    ms_non = timeit(gd.K_nonlinear_func)
    print(f"  K_linear_func {ms_lin:.2f} ms ({len(dom_wf.linear_gradients)} + {len(bnd_wf.linear_gradients)} _Kval launches) | K_nonlinear_func {ms_non:.2f} ms "
          f"({len(dom_wf.inner_vars)} _Var + {len(dom_wf.residues)} _Res domain launches)", flush=True)
    return gd

N = int(sys.argv[1]) if len(sys.argv) > 1 else 96
run(f"hex-8 thermal {N}^3 (generic)", N, "Lagrange", 1, 3, 1, thermal_wf(3, 0.6), conv_wf(25.0, 293.15))
brick = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N))
A = brick.pattern(1)
K = torch.empty(A.nnz, dtype=torch.float64, device="cuda")
print(f"  fused hex-8 thermal assembly at the same size: {timeit(lambda: brick.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F, out=K)):.2f} ms", flush=True)
M = max(8, N // 3)
E, nu = 1.0, 0.3
lam, mu = E * nu / ((1 + nu) * (1 - 2 * nu)), E / (2 * (1 + nu))
pen = G.WeakForm(inner_vars=[(f"d{i}", i, 0, 0) for i in range(3)])
for i in range(3):
    pen.residues.append(G.ResTerm(i, 0, lambda env, i=i: -1000.0 * env[f"d{i}"]))
    pen.linear_gradients.append(G.GradTerm(i, 0, i, 0, lambda env: -1000.0))
run(f"hex-20 elasticity {M}^3 (generic)", M, "Serendipity", 2, 5, 3, elasticity_wf(3, lam, mu), pen)
