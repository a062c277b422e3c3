#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/ (run in the BUILD container, where
/root/reference is mounted; the GPU box has neither /root/reference nor a need to rerun this).

1. ceramic_strip_T.npz -- DATA taken from the reference's own committed result file
   examples/thermal_conduction/2D_Ceramic_Strip.vtk (POINTS 2521 + SCALARS T): coordinates and
   temperatures only, no source text.  This is the pin of the oracle.
2. oracle_*.npz -- outputs of the oracle itself on small meshes (K, R, x, reference tables) used as
   regression vectors by the CPU suite and as fixed-size golden vectors by the GPU suite.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import fem, mesh as om, problems, reference_element as re_, solvers, vtk  # noqa: E402

REF = "/root/reference"


def strip():
    pts, sc = vtk.read_vtk_points_scalars(os.path.join(REF, "examples/thermal_conduction/2D_Ceramic_Strip.vtk"))
    np.savez_compressed(os.path.join(HERE, "ceramic_strip_T.npz"), xy=pts[:, :2], T=sc["T"],
                        featool_y=np.array([0.0001, 0.001, 0.002, 0.003, 0.004, 0.005, 0.006, 0.007, 0.008, 0.009, 0.0099]),
                        featool_T=np.array([1086.84, 1086, 1082.73, 1077.63, 1070.24, 1060.78, 1048.83, 1034.63, 1017.81,
                                            998.843, 979.249]))  # examples/thermal_conduction/2D_Script.jl:95-96


def cavity():
    """DATA from the reference's committed result examples/incompressible_flow/lid_driven_cavity_flow/2D_Cavity_Flow.vtk
    (POINTS 4961, SCALARS u1/u2/p; Re = 1000 via solver_LU_CPU, 2D_Script.jl:188-215) + its Ghia et al. table."""
    import csv

    base = os.path.join(REF, "examples/incompressible_flow/lid_driven_cavity_flow")
    pts, sc = vtk.read_vtk_points_scalars(os.path.join(base, "2D_Cavity_Flow.vtk"))
    g = [(float(r["y"]), float(r["u"])) for r in csv.DictReader(open(os.path.join(base, "Ghia_Re1000.csv")))]
    np.savez_compressed(os.path.join(HERE, "cavity_flow_Re1000.npz"), xy=pts[:, :2], u1=sc["u1"], u2=sc["u2"], p=sc["p"],
                        ghia_y=np.array([a for a, _ in g]), ghia_u=np.array([b for _, b in g]))


def cantilever():
    """DATA from the reference's committed result examples/linear_elasticity/cantilever/3D_Cantilever.vtk
    (POINTS 1865 = hex-20 20x4x4, SCALARS d1/d2/d3: the last load case of 3D_Script.jl:139-141)."""
    pts, sc = vtk.read_vtk_points_scalars(os.path.join(REF, "examples/linear_elasticity/cantilever/3D_Cantilever.vtk"))
    np.savez_compressed(os.path.join(HERE, "cantilever_hex20.npz"), xyz=pts, d1=sc["d1"], d2=sc["d2"], d3=sc["d3"])


def stress_concentration():
    """DATA from examples/linear_elasticity/stress_concentration: the Abaqus meshes 2D_Mesh.inp / 3D_Mesh.inp (first-order
    CUBE connectivity, read with oracle.readers.read_inp) and the committed results 2D_MetaFEM.vtk / 3D_MetaFEM.vtk."""
    from oracle import readers

    base = os.path.join(REF, "examples/linear_elasticity/stress_concentration")
    for dim in (2, 3):
        vert, conn = readers.read_inp(os.path.join(base, f"{dim}D_Mesh.inp"))
        pts, sc = vtk.read_vtk_points_scalars(os.path.join(base, f"{dim}D_MetaFEM.vtk"))
        np.savez_compressed(os.path.join(HERE, f"stress_concentration_{dim}d.npz"), vert=vert, conn=conn.astype(np.int32),
                            xyz=pts[:, :dim], **{f"d{i + 1}": sc[f"d{i + 1}"] for i in range(dim)})


def pikachu():
    """DATA from examples/thermal_conduction: 3D_COMSOL_Mesh.mphtxt (3405 vertices, 15334 tetrahedra; oracle.readers.read_mphtxt)
    and the committed tet-10 result 3D_MetaFEM_Result.vtk (23703 points, SCALARS T) of 3D_Script.jl."""
    from oracle import readers

    base = os.path.join(REF, "examples/thermal_conduction")
    vert, conn = readers.read_mphtxt(os.path.join(base, "3D_COMSOL_Mesh.mphtxt"))
    pts, sc = vtk.read_vtk_points_scalars(os.path.join(base, "3D_MetaFEM_Result.vtk"))
    np.savez_compressed(os.path.join(HERE, "pikachu_tet10.npz"), vert=vert, conn=conn.astype(np.int32), xyz=pts, T=sc["T"])


def line_samples():
    """DATA from the reference's committed Paraview line samples (CSV): the columns its scripts plot.
    thermal_conduction/MetaFEM_a.csv, MetaFEM_b.csv (3D_Script.jl:73-74): T along two vertical lines of the tet-10 result;
    stress_concentration/{2D,3D}_MetaFEM_{x,y}.csv (3D_Script.jl:93-94): d1..d3 along the symmetry lines of the plate with a hole;
    cylinder_flow/MetaFEM_y2.csv, MetaFEM_y3.csv (3D_MetaFEM_Script.jl:122-123): p, u1, u2, u3 along two lines through the channel --
    the only reference-produced numbers of a run with Pl_func = Pl_Jacobi (:90), + the COMSOL mesh of that example."""
    import csv

    from oracle import readers

    def read(path, cols):
        rows = list(csv.DictReader(open(path)))
        pts = np.array([[float(r[f"Points:{d}"]) for d in range(3)] for r in rows])
        mask = np.array([int(float(r["vtkValidPointMask"])) for r in rows], dtype=np.int8)
        vals = {c: np.array([float(r[c]) if r[c] != "nan" else np.nan for r in rows]) for c in cols}
        return pts, mask, vals

    out = {}
    base = os.path.join(REF, "examples/thermal_conduction")
    for tag in ("a", "b"):
        pts, mask, v = read(os.path.join(base, f"MetaFEM_{tag}.csv"), ["T"])
        out.update({f"thermal_{tag}_pts": pts, f"thermal_{tag}_mask": mask, f"thermal_{tag}_T": v["T"]})
    base = os.path.join(REF, "examples/linear_elasticity/stress_concentration")
    for dim in (2, 3):
        for tag in ("x", "y"):
            pts, mask, v = read(os.path.join(base, f"{dim}D_MetaFEM_{tag}.csv"), [f"d{i + 1}" for i in range(dim)])
            out.update({f"stress{dim}d_{tag}_pts": pts, f"stress{dim}d_{tag}_mask": mask})
            out.update({f"stress{dim}d_{tag}_d{i + 1}": v[f"d{i + 1}"] for i in range(dim)})
    base = os.path.join(REF, "examples/incompressible_flow/cylinder_flow")
    for tag in ("y2", "y3"):
        pts, mask, v = read(os.path.join(base, f"MetaFEM_{tag}.csv"), ["p", "u1", "u2", "u3"])
        out.update({f"cylinder_{tag}_pts": pts, f"cylinder_{tag}_mask": mask})
        out.update({f"cylinder_{tag}_{c}": v[c] for c in ("p", "u1", "u2", "u3")})
    np.savez_compressed(os.path.join(HERE, "line_samples.npz"), **out)
    vert, conn = readers.read_mphtxt(os.path.join(base, "3D_COMSOL_Mesh.mphtxt"))
    np.savez_compressed(os.path.join(HERE, "cylinder_mesh.npz"), vert=vert, conn=conn.astype(np.int32))


def cylinder_oracle(precomputed=None):
    """Oracle output for pin 9 (about 25 minutes of numpy: six Newton steps on 164 808 unknowns, not something the CPU suite can
    repeat): oracle/cylinder.py runs examples/incompressible_flow/cylinder_flow/3D_MetaFEM_Script.jl with the script's own solver
    (idrs!, s = 8, Pl_func = Pl_Jacobi) and the fields p, u1, u2, u3 are sampled at the points of the reference's MetaFEM_y2.csv /
    MetaFEM_y3.csv (oracle/sampling.py).  Stored: the samples, the Newton history, the nodal solution (float32, for the GPU comparison)."""
    from oracle import cylinder, reference_element as re_
    from oracle.sampling import Sampler

    zm = np.load(os.path.join(HERE, "cylinder_mesh.npz"))
    zl = np.load(os.path.join(HERE, "line_samples.npz"))
    if precomputed is None:
        dom, hist = cylinder.run(zm["vert"], zm["conn"].astype(np.int64))
        x, coords, cp_ids = dom.x, dom.mesh.coords, dom.mesh.cp_ids
    else:  # (x, coords, cp_ids, hist) of an earlier run of the same function
        x, coords, cp_ids, hist = precomputed
    import types

    disc = re_.initialize_classical_element(3, "SIMPLEX", 2, 1, 6, itp_type="Serendipity")
    S = Sampler(types.SimpleNamespace(coords=coords, cp_ids=cp_ids, nel=cp_ids.shape[1]), disc)
    n = coords.shape[0]
    fields = {"p": x[:n], "u1": x[n:2 * n], "u2": x[2 * n:3 * n], "u3": x[3 * n:4 * n]}
    out = {"newton_history": np.array(hist), "x_f32": x.astype(np.float32)}
    for tag in ("y2", "y3"):
        got, valid = S.sample(fields, zl[f"cylinder_{tag}_pts"], tol=1e-5)
        out[f"{tag}_valid"] = valid
        for k, v in got.items():
            out[f"{tag}_{k}"] = v
    np.savez_compressed(os.path.join(HERE, "oracle_cylinder_lines.npz"), **out)


def neo_hookean_oracle(steps=3):
    """Oracle output for the hyperelastic tensile test (examples/hyper_elasticity/static_Neo_Hookean.jl; oracle/hyperelastic.py): mean
    elongation of the right face after the first `steps` load steps of each of the script's three material setups (LU solves, Newton to
    the script's 1e-5).  About 12 s per load step."""
    from oracle import hyperelastic as he

    dom = he.build()
    lu = lambda d: solvers.solver_lu_cpu(d.pattern.rowptr, d.pattern.colidx, d.K_total, d.residue)
    out = {}
    for s, (mu, lam, _, sig) in enumerate([(1e6, 1e6, 10, 4e5), (1e6, 2e8, 40, 1e5), (2e6, 2e8, 80, 1e5)]):  # :88
        d1s, P1s, hists = he.run_setup(dom, mu, lam, steps, sig, linear_solver=lu)
        out[f"d1s_{s}"], out[f"P1s_{s}"] = d1s, P1s
        out[f"newton_last_{s}"] = np.array([h[-1] for h in hists])
    np.savez_compressed(os.path.join(HERE, "oracle_neo_hookean.npz"), **out)


def mooney_rivlin_oracle(steps=3):
    """The same for examples/hyper_elasticity/static_Mooney_Rivlin.jl (W of :48-52, setups of :94)."""
    from oracle import hyperelastic as he

    dom = he.build(model="mooney_rivlin")
    lu = lambda d: solvers.solver_lu_cpu(d.pattern.rowptr, d.pattern.colidx, d.K_total, d.residue)
    out = {}
    for s, (C10, C01, lam, _, sig) in enumerate([(1e6, 1e6, 1e8, 30, 4e5), (1e6, 5e6, 1e8, 30, 5e5), (5e6, 1e6, 1e8, 40, 10e5)]):
        d1s, P1s, hists = he.run_setup(dom, 0, 0, steps, sig, linear_solver=lu, materials=dict(C10=C10, C01=C01, lam=lam))
        out[f"d1s_{s}"], out[f"P1s_{s}"] = d1s, P1s
        out[f"newton_last_{s}"] = np.array([h[-1] for h in hists])
    np.savez_compressed(os.path.join(HERE, "oracle_mooney_rivlin.npz"), **out)


def j2_plasticity_oracle():
    """Oracle output for examples/hypo_elastic_plasticity/J2Plasticity.jl (oracle/plasticity.py): the mean elongation of the right face after
    every load of the script's three load histories / hardening setups (pseudo-time relaxation to max |d1_t| < 1e-4 per load, LU solves, Newton
    with max_iter = 3 to the script's 1e-3), the pseudo-time steps each load took, and the script's own d1_analytical (:226-228) beside them --
    numbers, not source.  About 2.5 minutes."""
    from oracle import plasticity as pl

    dom = pl.build()
    out = {}
    for g in range(3):
        d1, cnt = pl.run_group(dom, pl.S_TEST_GROUPS[g], pl.EB_GROUPS[g], pl.EP_GROUPS[g], linear_solver=pl.lu)
        out[f"d1_{g}"], out[f"steps_{g}"] = d1, np.array(cnt)
        out[f"s_tests_{g}"], out[f"d1_analytical_{g}"] = np.array(pl.S_TEST_GROUPS[g], dtype=float), pl.D1_ANALYTICAL[g]
    np.savez_compressed(os.path.join(HERE, "oracle_j2_plasticity.npz"), **out)


def tables():
    out = {}
    for name, args in {"quad8": (2, "CUBE", 2, 1, 5, "Serendipity"), "hex8": (3, "CUBE", 1, 1, 3, "Lagrange"),
                       "hex27": (3, "CUBE", 2, 1, 5, "Lagrange"), "hex20": (3, "CUBE", 2, 1, 5, "Serendipity")}.items():
        d = re_.initialize_classical_element(*args[:5], itp_type=args[5])
        out[f"{name}_ref"] = d.ref_itp_vals
        out[f"{name}_w"] = d.itg_weight
        out[f"{name}_pos"] = d.itp_pos
        for f, (r, t) in enumerate(zip(d.bdy_ref_itp_vals, d.bdy_tangent_directions)):
            out[f"{name}_bref{f}"] = r
            out[f"{name}_btan{f}"] = t
    np.savez_compressed(os.path.join(HERE, "oracle_tables.npz"), **out)


def thermal_hex8():
    x, n = (1.0, 1.5, 0.75), (4, 4, 4)
    disc = re_.initialize_classical_element(3, "CUBE", 1, 1, 3)
    msh = om.lattice_mesh(x, n, disc)
    fac = om.boundary_facets_structured(x, n, 3)
    od = fem.FEMDomain(msh, disc, 1, problems.thermal_domain(3, 0.6), [(fac, problems.thermal_convection(25.0, 293.15))])
    od.controlpoints["s"] = np.full(msh.ncp, 1600.0)
    od.converge_tol = 1e-9
    od.linear_solver = lambda d: solvers.solver_lu_cpu(d.pattern.rowptr, d.pattern.colidx, d.K_total, d.residue)
    od.update_time()
    od.K_linear_func()
    od.update_x_star()
    od.K_nonlinear_func()
    K, R0 = od.K_linear.copy(), od.residue.copy()
    od.t = 0.0
    od.update_one_step()
    np.savez_compressed(os.path.join(HERE, "oracle_thermal_hex8_4x4x4.npz"), x=np.array(x), n=np.array(n), rowptr=od.pattern.rowptr,
                        colidx=od.pattern.colidx, K=K, R0=R0, T=od.x)


def elasticity_hex8():
    x, n = (3.0, 1.0, 1.0), (3, 2, 2)
    E, nu = 1.0, 0.3
    lam, mu = E * nu / ((1 + nu) * (1 - 2 * nu)), E / (2 * (1 + nu))
    disc = re_.initialize_classical_element(3, "CUBE", 1, 1, 3)
    msh = om.lattice_mesh(x, n, disc)
    fac = om.boundary_facets_structured(x, n, 3)
    fixed = fac.select(fac.element_eindex == 4)  # x = 0
    load = fac.select(fac.element_eindex == 3)  # y = L
    sig = [[0.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 0.0]]
    od = fem.FEMDomain(msh, disc, 3, problems.elasticity_domain(3, lam, mu),
                       [(fixed, problems.elasticity_penalty(3, 1000.0 * E)), (load, problems.elasticity_traction(3, sig))])
    od.converge_tol = 1e-10
    od.linear_solver = lambda d: solvers.solver_lu_cpu(d.pattern.rowptr, d.pattern.colidx, d.K_total, d.residue)
    od.update_time()
    od.K_linear_func()
    od.update_x_star()
    od.K_nonlinear_func()
    K, R0 = od.K_linear.copy(), od.residue.copy()
    od.t = 0.0
    od.update_one_step()
    np.savez_compressed(os.path.join(HERE, "oracle_elasticity_hex8_3x2x2.npz"), x=np.array(x), n=np.array(n), lam=lam, mu=mu,
                        tau=1000.0 * E, rowptr=od.pattern.rowptr, colidx=od.pattern.colidx, K=K, R0=R0, d=od.x)


def thermal_hex27():
    x, n = (1.0, 1.0, 1.0), (2, 2, 2)
    disc = re_.initialize_classical_element(3, "CUBE", 2, 1, 5)
    msh = om.lattice_mesh(x, n, disc)
    fac = om.boundary_facets_structured(x, n, 3)
    od = fem.FEMDomain(msh, disc, 1, problems.thermal_domain(3, 0.6), [(fac, problems.thermal_convection(25.0, 293.15))])
    od.controlpoints["s"] = np.full(msh.ncp, 1600.0)
    od.converge_tol = 1e-9
    od.linear_solver = lambda d: solvers.solver_lu_cpu(d.pattern.rowptr, d.pattern.colidx, d.K_total, d.residue)
    od.update_time()
    od.K_linear_func()
    od.update_x_star()
    od.K_nonlinear_func()
    K, R0 = od.K_linear.copy(), od.residue.copy()
    od.t = 0.0
    od.update_one_step()
    np.savez_compressed(os.path.join(HERE, "oracle_thermal_hex27_2x2x2.npz"), x=np.array(x), n=np.array(n), rowptr=od.pattern.rowptr,
                        colidx=od.pattern.colidx, K=K, R0=R0, T=od.x)


def c_header():
    """The 4x4x4 hex-8 thermal fixture as C arrays for tests/c_abi_smoke.c (a plain-C consumer of include/metafem_mi355x.h):
    the same numbers as oracle_thermal_hex8_4x4x4.npz, data only."""
    d = np.load(os.path.join(HERE, "oracle_thermal_hex8_4x4x4.npz"))

    def arr(ctype, name, a, fmt):
        body = ",\n  ".join(", ".join(fmt(v) for v in a[i:i + 6]) for i in range(0, len(a), 6))
        return f"static const {ctype} {name}[{len(a)}] = {{\n  {body}\n}};\n"

    f = lambda v: float(v).hex()  # exact binary64 values (C99 hexadecimal floating constants)
    i = lambda v: str(int(v))
    with open(os.path.join(HERE, "oracle_thermal_hex8_4x4x4.h"), "w") as out:
        out.write("/* generated by tests/golden/make_golden.py::c_header from oracle_thermal_hex8_4x4x4.npz (oracle output: 4x4x4 hex-8\n"
                  " * Robin thermal problem, k = 0.6, h = 25, Tenv = 293.15, s = 1600; K in CSR order, R0 = residual at x* = 0, T = the\n"
                  " * field after one update_OneStep!).  Data only. */\n")
        out.write(f"#define GOLD_N {len(d['T'])}\n#define GOLD_NNZ {len(d['K'])}\n")
        out.write(arr("double", "gold_size", d["x"], f))
        out.write(arr("int", "gold_num", d["n"], i))
        out.write(arr("long long", "gold_rowptr", d["rowptr"], i))
        out.write(arr("int", "gold_colidx", d["colidx"], i))
        out.write(arr("double", "gold_K", d["K"], f))
        out.write(arr("double", "gold_R0", d["R0"], f))
        out.write(arr("double", "gold_T", d["T"], f))


if __name__ == "__main__":
    if os.path.isdir(REF):
        strip()
        cavity()
        cantilever()
        stress_concentration()
        pikachu()
        line_samples()
    tables()
    thermal_hex8()
    elasticity_hex8()
    thermal_hex27()
    c_header()
    neo_hookean_oracle()
    mooney_rivlin_oracle()
    j2_plasticity_oracle()
    if "--cylinder" in sys.argv:  # 25 minutes: only on request
        cylinder_oracle()
    print("fixtures written to", HERE)
