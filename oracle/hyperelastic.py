"""Uniaxial tensile test on a Neo-Hookean bar (oracle; test infrastructure only): examples/hyper_elasticity/static_Neo_Hookean.jl --
make_Brick 40 x 4 x 4 -> hex-20 serendipity (itg_order 5), F = I + grad d, W = mu/2 (tr C - 3 - 2 ln J) + lam/2 (J - 1)^2 (:45-50),
WF_domain = -Bilinear(F{i,j}, P{i,j}) with P = dW/dF (:50,54), left face fixed by penalty (:55), nominal traction Pl{1,1} on the right face
(:56), load stepping with update_OneStep!(max_iter = 7) and the script's solver bicgstabl_GS!(s = 4, maxiter = 3000, max_pass = 10) (:80).
The script compares the mean elongation of the right face with the closed form uniaxial_Neo_Hookean (:123) -- a known answer held by the
reference for a NONLINEAR problem solved with bicgstabl_GS!.

The reference's symbolic layer differentiates W (d(W, F{i,j}), :50) and then the residual (variation -> gradients); sympy does the same here:
P_ij = dW/dF_ij, A_ijkl = dP_ij/dF_kl, lambdified with common-subexpression elimination."""
from __future__ import annotations

import numpy as np
import sympy as sp

from . import fem, mesh as om, reference_element as re_, solvers
from .fem import AssembleWeakform, GradTerm, ResTerm

INNER_INFOS = [("d1", 0, 0), ("d2", 1, 0), ("d3", 2, 0)]


def uniaxial_neo_hookean(l1, lam, mu):  # (keyword order of the script: l1, lam, mu)
    """static_Neo_Hookean.jl:123 -- nominal stress of the uniaxial state at stretch l1."""
    return mu * l1 + ((lam * mu * (l1 - 1)) / (mu + lam * l1) - mu) / l1


def uniaxial_mooney_rivlin(l1, C10, C01, lam):
    """static_Mooney_Rivlin.jl:125-126 -- Jac(l1, ...) and the nominal stress mooney_Rivlin(l1, ...) of the uniaxial state."""
    a = 2 * C10 + 2 * C01 * l1 ** 2 + lam * l1
    Jac = l1 * (-a + np.sqrt(a ** 2 + 8 * C01 * (lam + 2 * C10 + 4 * C01))) / (4 * C01)
    return 2 * C10 * l1 + 4 * C01 * Jac + (lam * (Jac - 1) - 2 * C10 - 4 * C01) / l1


MODELS = {  # model -> names of its material GLOBAL_VARs, in the order of the lambdified arguments
    "neo_hookean": ("mu", "lam"),
    "mooney_rivlin": ("C10", "C01", "lam"),
}


def _tensors(model: str = "neo_hookean"):
    F = sp.Matrix(3, 3, lambda i, j: sp.Symbol(f"F{i}{j}", real=True))
    mats = sp.symbols(" ".join(MODELS[model]), positive=True)
    J = F.det()
    if model == "neo_hookean":
        mu, lam = mats
        C = F.T * F
        W = mu / 2 * (C.trace() - 3 - 2 * sp.log(J)) + lam / 2 * (J - 1) ** 2  # static_Neo_Hookean.jl:49
    else:
        C10, C01, lam = mats
        B = F * F.T                                                            # static_Mooney_Rivlin.jl:48-52
        I1 = B.trace()
        I2 = (I1 ** 2 - (B * B).trace()) / 2
        W = C10 * (I1 - 3 - 2 * sp.log(J)) + C01 * (I2 - 3 - 4 * sp.log(J)) + lam / 2 * (J - 1) ** 2
    Fs = [F[i, j] for i in range(3) for j in range(3)]
    P = [sp.diff(W, f) for f in Fs]
    A = [[sp.diff(p, f) for f in Fs] for p in P]
    args = Fs + list(mats)
    mods = [{"log": _log}, "numpy"]  # the term functions are data shared with the GPU tests: they must accept torch tensors too
    fP = sp.lambdify(args, P, mods, cse=True)
    fA = sp.lambdify(args, [a for row in A for a in row], mods, cse=True)
    return fP, fA


def _log(x):
    if isinstance(x, np.ndarray) or np.isscalar(x):
        return np.log(x)
    return x.log()  # torch tensor


_CACHE = {}


def domain_weakform(params: dict, model: str = "neo_hookean") -> AssembleWeakform:
    """-Bilinear(F{i,j}, P{i,j}): the dual word F{i,j} = delta{i,j} + d{i;j} varies as d{i;j}.  `params` (the model's material constants)
    is read at call time: GLOBAL_VARs of the script (physics.global_vars, :92-94)."""
    if model not in _CACHE:
        _CACHE[model] = _tensors(model)
    fP, fA = _CACHE[model]
    mat_names = MODELS[model]
    wf = AssembleWeakform()
    wf.inner_vars = [(f"d{i + 1}__d{j}", i, 1 + j, 0) for i in range(3) for j in range(3)]

    def Fargs(env):
        out = []
        for i in range(3):
            for j in range(3):
                g = env[f"d{i + 1}__d{j}"]
                out.append(g + 1.0 if i == j else g)
        return out + [params[k] for k in mat_names]

    def evaluate(env):  # one evaluation of P and A per environment (the 90 term functions share it; cached IN the environment)
        m = env.get("__hyperelastic")
        if m is None:
            a = Fargs(env)
            m = {"P": fP(*a), "A": fA(*a)}
            env["__hyperelastic"] = m
        return m

    for i in range(3):
        for j in range(3):
            wf.residues.append(ResTerm(i, 1 + j, lambda env, q=3 * i + j: -evaluate(env)["P"][q]))
            for k in range(3):
                for l in range(3):
                    wf.nonlinear_gradients.append(GradTerm(i, 1 + j, k, 1 + l, lambda env, q=(3 * i + j) * 9 + 3 * k + l: -evaluate(env)["A"][q]))
    return wf


def fixed_weakform(params: dict) -> AssembleWeakform:
    """tau_b * Bilinear(d{i}, dw{i} - d{i}) with dw = 0 (:55)."""
    wf = AssembleWeakform(inner_vars=[(f"d{i + 1}", i, 0, 0) for i in range(3)])
    for i in range(3):
        wf.residues.append(ResTerm(i, 0, lambda env, i=i: params["tau"] * (0.0 - env[f"d{i + 1}"])))
        wf.linear_gradients.append(GradTerm(i, 0, i, 0, lambda env: -params["tau"]))
    return wf


def load_weakform() -> AssembleWeakform:
    """Bilinear(d{i}, Pl{i,j} * n{j}) with only Pl{1,1} set by the script (:56, 103)."""
    wf = AssembleWeakform()
    wf.cp_ext_vars = [("Pl1", "Pl1", 0)]
    wf.normals = [("n0", 0)]
    wf.residues.append(ResTerm(0, 0, lambda env: env["Pl1"] * env["n0"]))
    return wf


def build(e_number: int = 4, LW_ratio: int = 10, L_box: float = 1.0, model: str = "neo_hookean"):
    """:7-78 (the two scripts differ in W and its constants only)."""
    size = (L_box * LW_ratio, L_box, L_box)
    disc = re_.initialize_classical_element(3, "CUBE", 2, 1, 5, itp_type="Serendipity")  # :69
    vert, conn = om.make_brick(size, (e_number * LW_ratio, e_number, e_number))
    msh = om.mesh_classical(vert, conn, disc)
    fac = om.boundary_facets(msh)
    err = L_box / e_number * 0.01
    c = fac.centroid
    left, right = np.abs(c[:, 0]) < err, np.abs(c[:, 0] - size[0]) < err
    params = dict(mu=1e6, lam=1e6, C10=1e6, C01=1e6, tau=1e9, L=size[0])
    dom = fem.FEMDomain(msh, disc, 3, domain_weakform(params, model), [(fac.select(left), fixed_weakform(params)), (fac.select(right), load_weakform())])
    dom.converge_tol = 1e-5  # :86
    dom.params = params
    dx = L_box / e_number
    dom.right_cps = np.nonzero(np.abs(msh.coords[:, 0] - size[0]) < 0.25 * dx)[0]  # :83-84
    dom.controlpoints["Pl1"] = np.zeros(msh.ncp)
    return dom


def solver_of_the_script(dom):
    """:80 -- bicgstabl_GS!, s = 4, maxiter = 3000, max_pass = 10."""
    return solvers.iterative_solve(dom.pattern.rowptr, dom.pattern.colidx, dom.K_total, dom.residue, dom.converge_tol,
                                   Sv_func=solvers.bicgstabl_gs, maxiter=3000, max_pass=10, s=4)


def run_setup(dom, mu: float, lam: float, total_steps: int, sigma_step: float, linear_solver=None, max_iter: int = 7, materials=None):
    """One entry of `setups` (:88-112): returns (elongations d1s, nominal stresses P1s, Newton histories).  `materials` (dict) replaces
    (mu, lam) for the Mooney-Rivlin script: C10, C01, lam."""
    P = dom.params
    if materials is None:
        materials = dict(mu=mu, lam=lam)
    P.update(materials)
    P["tau"] = 1000 * max(materials.values()) / 1.0  # :92-94 / Mooney :100 (L_box = 1)
    dom.linear_solver = linear_solver or solver_of_the_script
    dom.x[:] = 0.0  # :96-99
    d1s, P1s, hists = [], [], []
    n = dom.mesh.ncp
    for i in range(1, total_steps + 1):
        load = sigma_step * i
        dom.controlpoints["Pl1"] = np.full(n, load)
        hists.append(dom.update_one_step(max_iter=max_iter))  # :105
        d1s.append(dom.x[:n][dom.right_cps].sum() / (P["L"] * dom.right_cps.size))  # :108
        P1s.append(load)
    return np.array(d1s), np.array(P1s), hists
