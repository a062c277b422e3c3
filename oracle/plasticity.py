"""Uniaxial loading of a J2-flow hypoelastic-plastic bar (oracle; test infrastructure only):
examples/hypo_elastic_plasticity/J2Plasticity.jl -- make_Brick 10 x 4 x 4 -> hex-20 serendipity, itg_order 5 (:9-14, :73), left face fixed by
penalty, nominal traction sl{1,1} on the right face (:60-61), TWO time levels (d{i;t}, d{i;t,t}: viscous damping + inertia, :59; dt = 1 with the
dissipative generalised-alpha of FEM_Domain, :244) -- a load step is relaxed in pseudo-time until max |d1_t| < 1e-4 (:279-286) --, update_OneStep!
(max_iter = 3) with the script's solver bicgstabl_GS!(s = 8, maxiter = 2000, max_pass = 20) (:218-219), and a USER FUNCTION in the coefficient stage:
the plastic strain `ep` is an INTEGRATION_POINT_VAR defined as  ep{i,j} = strain_updater(e{1,1}, e{1,2}, e{1,3}, e{2,2}, e{2,3}, e{3,3})  (:52-55).

What the generator makes of that (src/symbolics/08_Tensor.jl:172-183, src/solver/02_LocalAssembly.jl:8,49; src/solver/05_CodeGenerator.jl:15-50):
  * where an expression meets an integration-point symbol, the generated code evaluates its definition ONCE per updater call and binds all six
    Voigt components:  (ep1_1, ep2_2, ep3_3, ep2_3, ep1_3, ep1_2) = Main.strain_updater(e1_1, ...)  (Voigt order of 03_Word.jl:36-37);
  * the words inside the definition (d{i;j}) are inner variables of the residual, but `ep` is EXTERNAL for the variation: the gradient of
    sigma = 2 mu (e - ep) + lam tr(e - ep) delta is the elastic one, constant -> linear gradients (K_linear); the plastic strain acts through
    the residual only.

The script holds its own answer: d1_analytical (:226-228), the elongation of a 1-D bar of length 10 under the load history s_test_groups for
three hardening setups (isotropic Ep = E/2; mixed Eb = Ep = E/4; kinematic Eb = E/2) -- the only reference-held numbers for
max_time_level = 2 and for bicgstabl_GS! with s = 8.
"""
from __future__ import annotations

import numpy as np

from . import fem, mesh as om, reference_element as re_, solvers
from .cantilever import VOIGT
from .fem import AssembleWeakform, GradTerm, ResTerm

INNER_INFOS = [(f"d{i + 1}{suffix}", i, td) for td, suffix in enumerate(("", "_t", "_tt")) for i in range(3)]  # :251-259, dessemble_X!
# (i, j) of Voigt component v = 0..5 (03_Word.jl:37: (1,1) (2,2) (3,3) (2,3) (1,3) (1,2)) and the component of (i, j)
INV_VOIGT = ((0, 0), (1, 1), (2, 2), (1, 2), (0, 2), (0, 1))
VID = [[VOIGT[3][i][j] - 1 for j in range(3)] for i in range(3)]

S_TEST_GROUPS = [[40, 80, 100, 120, 140, 180, 200, 180, 100, 0, -80, -180, -200, -220, -240, -200, -100],
                 [40, 80, 100, 120, 140, 180, 200, 180, 100, 0, -80, -100, -120, -160, -100],
                 [40, 80, 100, 120, 140, 180, 200, 180, 140, 100, 80, 40, 0, -40, -100, -80, -20]]  # :222-224
D1_ANALYTICAL = [np.array([4, 8, 10, 16, 22, 34, 40, 38, 30, 20, 12, 2, 0, -6, -12, -8, 2]) * 1e-3,
                 np.array([4, 8, 10, 16, 22, 34, 40, 38, 30, 20, 12, 10, 4, -8, -2]) * 1e-3,
                 np.array([4, 8, 10, 16, 22, 34, 40, 38, 34, 30, 28, 24, 20, 8, -10, -8, -2]) * 1e-3]  # :226-228
EY = 100e3
EB_GROUPS = [0.0, EY / 4, EY / 2]  # :230
EP_GROUPS = [EY / 2, EY / 4, 0.0]  # :231


def _zeros_like(a):
    return a * 0.0  # numpy array or torch tensor


def _sqrt(a):
    return np.sqrt(a) if isinstance(a, np.ndarray) else a.sqrt()


def _where(c, a, b):
    return np.where(c, a, b) if isinstance(a, np.ndarray) else a.where(c, b)


class MaterialState:
    """:84-117 -- the state of the return mapping at the integration points (arrays [itg, elements]; numpy here, torch tensors in the GPU test):
    committed ep, b (back stress), Y (yield stress) and their trial values *_eval of the last evaluation."""

    def __init__(self, zeros, Y_initial, lam, mu, Eb, Ep, f_res):
        self.ep = [zeros() for _ in range(6)]
        self.b = [zeros() for _ in range(6)]
        self.Y = zeros() + Y_initial
        self.ep_eval = [zeros() for _ in range(6)]
        self.b_eval = [zeros() for _ in range(6)]
        self.Y_eval = zeros() + Y_initial
        self.lam, self.mu, self.Eb, self.Ep, self.f_res = lam, mu, Eb, Ep, f_res
        self.Y_initial = Y_initial
        self._zeros = zeros
        self.calls = 0
        self.yielded_calls = 0

    def reset(self, Eb, Ep):
        """:264-270."""
        for i in range(6):
            self.ep[i] = self._zeros()
            self.b[i] = self._zeros()
        self.Y = self._zeros() + self.Y_initial
        self.Eb, self.Ep = Eb, Ep

    def __call__(self, e11, e12, e13, e22, e23, e33):
        """:119-123 + assemble_strain :126-135 + iterate_stress! :150-202.  Returns the six trial plastic strains in Voigt order."""
        self.calls += 1
        e = [None] * 6
        e[VID[0][0]], e[VID[1][1]], e[VID[2][2]] = e11, e22, e33
        e[VID[0][1]], e[VID[0][2]], e[VID[1][2]] = e12, e13, e23
        lam, mu, Eb, Ep = self.lam, self.mu, self.Eb, self.Ep
        e_eval = [e[i] - self.ep[i] for i in range(6)]
        # estimate_stress :137-148
        tr = e_eval[VID[0][0]] + e_eval[VID[1][1]] + e_eval[VID[2][2]]
        sig = [2 * mu * e_eval[i] for i in range(6)]
        for i in range(3):
            sig[VID[i][i]] = sig[VID[i][i]] + lam * tr
        s = [sig[i] - self.b[i] for i in range(6)]
        skk3 = (s[VID[0][0]] + s[VID[1][1]] + s[VID[2][2]]) / 3
        for i in range(3):
            s[VID[i][i]] = s[VID[i][i]] - skk3  # now dev(s)
        # :176-178 sums over ALL nine (i, j): the off-diagonal components count twice (Frobenius norm of the tensor).  The script accumulates
        # into FEM_buffer (uninitialised memory, :165); a zero start is what the arithmetic means
        s2 = _zeros_like(s[0])
        for i in range(3):
            for j in range(3):
                s2 = s2 + s[VID[i][j]] * s[VID[i][j]]
        smag = _sqrt(s2)
        f = np.sqrt(3 / 2) * smag - self.Y  # :183
        yielded = f > self.f_res            # :184
        if bool(yielded.any()):
            self.yielded_calls += 1
        # (0 / 0 where the deviator vanishes and nothing yields: the direction is not used there)
        safe = _where(yielded, smag, _zeros_like(smag) + 1.0)
        lp = _where(yielded, np.sqrt(3 / 2) * f / (3 * mu + Eb + Ep), _zeros_like(f))  # :186
        for i in range(6):
            n_i = s[i] / safe
            self.ep_eval[i] = self.ep[i] + n_i * lp                      # :189 (lp = 0 where not yielded: :154-156)
            self.b_eval[i] = self.b[i] + (2 / 3 * Eb) * n_i * lp         # :190
        self.Y_eval = self.Y + (np.sqrt(2 / 3) * Ep) * lp                # :192
        return self.ep_eval

    def update_states(self):
        """update_States! :204-211."""
        for i in range(6):
            self.ep[i] = self.ep_eval[i]
            self.b[i] = self.b_eval[i]
        self.Y = self.Y_eval


def domain_weakform(params: dict, state_of) -> AssembleWeakform:
    """WF_domain = Bilinear(d{i;j}, sigma{i,j}) + Bilinear(d{i}, rho * (c * d{i;t} + d{i;t,t}))  (:52-59), in the script's sign.
    state_of(env) -> the MaterialState to call (the term functions of one updater call share its ONE evaluation, cached in the environment)."""
    lam, mu, rho, c = params["lam"], params["mu"], params["rho"], params["c"]
    wf = AssembleWeakform()
    for i in range(3):
        for j in range(3):
            wf.inner_vars.append((f"d{i}_{j}", i, 1 + j, 0))
        wf.inner_vars.append((f"d{i}_t", i, 0, 1))
        wf.inner_vars.append((f"d{i}_tt", i, 0, 2))

    def strain(env, i, j):
        return (env[f"d{i}_{j}"] + env[f"d{j}_{i}"]) / 2

    def ep(env):  # (ep1_1, ..., ep1_2) = Main.strain_updater(e1_1, e1_2, e1_3, e2_2, e2_3, e3_3): once per updater call
        v = env.get("__ep")
        if v is None:
            v = state_of(env)(strain(env, 0, 0), strain(env, 0, 1), strain(env, 0, 2), strain(env, 1, 1), strain(env, 1, 2), strain(env, 2, 2))
            env["__ep"] = v
        return v

    def sigma(env, i, j):
        p = ep(env)
        s = 2 * mu * (strain(env, i, j) - p[VID[i][j]])
        if i == j and lam != 0.0:
            s = s + lam * sum(strain(env, m, m) - p[VID[m][m]] for m in range(3))
        return s

    for i in range(3):
        for j in range(3):
            wf.residues.append(ResTerm(i, 1 + j, lambda env, i=i, j=j: sigma(env, i, j)))
            for k in range(3):
                for l in range(3):
                    cc = (lam if (i == j and k == l) else 0.0) + mu * ((i == k and j == l) + (i == l and j == k))
                    if cc != 0.0:
                        wf.linear_gradients.append(GradTerm(i, 1 + j, k, 1 + l, lambda env, cc=cc: cc))
        wf.residues.append(ResTerm(i, 0, lambda env, i=i: rho * (c * env[f"d{i}_t"] + env[f"d{i}_tt"])))
        wf.linear_gradients.append(GradTerm(i, 0, i, 0, lambda env: rho * c, td_order=1))
        wf.linear_gradients.append(GradTerm(i, 0, i, 0, lambda env: rho, td_order=2))
    return wf


def fixed_weakform(params: dict) -> AssembleWeakform:
    """WF_fixed_bdy = tau_b * Bilinear(d{i}, d{i} - dw{i}) with dw = 0 (:60; the script never sets dw)."""
    tau = params["tau"]
    wf = AssembleWeakform(inner_vars=[(f"d{i}", i, 0, 0) for i in range(3)])
    for i in range(3):
        wf.residues.append(ResTerm(i, 0, lambda env, i=i: tau * env[f"d{i}"]))
        wf.linear_gradients.append(GradTerm(i, 0, i, 0, lambda env: tau))
    return wf


def load_weakform() -> AssembleWeakform:
    """WF_right_bdy = Bilinear(d{i}, -sl{i,j} * n{j}) (:61); the script sets sl{1,1} only (:273) -- a CONTROLPOINT_VAR symmetric tensor, Voigt id 1."""
    wf = AssembleWeakform()
    wf.cp_ext_vars = [("sl1", "sl1", 0)]
    wf.normals = [("n0", 0)]
    wf.residues.append(ResTerm(0, 0, lambda env: -env["sl1"] * env["n0"]))
    return wf


def build(e_number: int = 4, LW_ratio: int = 10, L_box: float = 1.0):
    """:8-81, :213-219, :234-244."""
    size = (L_box * LW_ratio, L_box, L_box)
    disc = re_.initialize_classical_element(3, "CUBE", 2, 1, 5, itp_type="Serendipity")  # :73
    vert, conn = om.make_brick(size, (int(e_number * LW_ratio / 4), e_number, e_number))   # :10
    msh = om.mesh_classical(vert, conn, disc)
    fac = om.boundary_facets(msh)
    err = L_box / e_number * 0.01
    cen = fac.centroid
    left, right = np.abs(cen[:, 0]) < err, np.abs(cen[:, 0] - size[0]) < err
    nu = 0.0
    params = dict(rho=1e3, c=2.0, lam=EY * nu / ((1 + nu) * (1 - 2 * nu)), mu=EY / (2 * (1 + nu)), tau=1000 * EY / L_box ** 2, L=size[0])  # :44-50
    nitg = disc.itg_weight.size if hasattr(disc, "itg_weight") else 27
    state = MaterialState(lambda: np.zeros((nitg, msh.nel)), 100.0, params["lam"], params["mu"], 0.0, EY / 2, 1.0)  # :213-217
    dom = fem.FEMDomain(msh, disc, 3, domain_weakform(params, lambda env: state),
                        [(fac.select(left), fixed_weakform(params)), (fac.select(right), load_weakform())], max_time_level=2)
    dom.converge_tol = 1e-3  # :219
    dom.dt = 1.0             # :244
    dom.params, dom.state = params, state
    dx = L_box / e_number
    dom.right_cps = np.nonzero(np.abs(msh.coords[:, 0] - size[0]) < 0.25 * dx)[0]  # :241
    dom.controlpoints["sl1"] = np.zeros(msh.ncp)
    return dom


def solver_of_the_script(dom):
    """:218 -- bicgstabl_GS!, s = 8, maxiter = 2000, max_pass = 20."""
    return solvers.iterative_solve(dom.pattern.rowptr, dom.pattern.colidx, dom.K_total, dom.residue, dom.converge_tol,
                                   Sv_func=solvers.bicgstabl_gs, maxiter=2000, max_pass=20, s=8)


def lu(dom):
    return solvers.solver_lu_cpu(dom.pattern.rowptr, dom.pattern.colidx, dom.K_total, dom.residue)


def run_group(dom, s_tests, Eb, Ep, linear_solver=None, max_pseudo_steps: int = 400):
    """One entry of zip(s_test_groups, Eb_groups, Ep_groups) (:246-291): returns (d1 per load, pseudo-time steps per load)."""
    n = dom.mesh.ncp
    dom.linear_solver = linear_solver or solver_of_the_script
    dom.x[:] = 0.0          # cpts.d* .= 0 ; assemble_X! (:251-262)
    dom.t = 0.0
    dom.state.reset(Eb, Ep)
    d1s, counts = [], []
    for s in s_tests:
        dom.controlpoints["sl1"] = np.full(n, float(s))  # :273
        counter = 0
        while True:
            counter += 1
            dom.update_one_step(max_iter=3)               # :277
            dom.state.update_states()                     # :281
            N = dom.basicfield_size
            umax = np.abs(dom.x[N:N + n]).max()           # max |d1_t| (:282; dessemble_X! :278)
            if umax < 1e-4 or counter >= max_pseudo_steps:
                d1s.append(dom.x[:n][dom.right_cps].sum() / dom.right_cps.size)  # :286
                counts.append(counter)
                break
    return np.array(d1s), counts
