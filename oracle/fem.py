"""FEM driver: assembled weak forms, generated-updater restatement, Newton / generalised-alpha
(oracle; test infrastructure only).

Restates
  * AssembleWeakform lists               -- src/solver/02_LocalAssembly.jl:30-58,83-112
  * what compile_Updater_GPU emits       -- src/solver/05_CodeGenerator.jl:1-154 (terms), :156-258 (body), :265-291
  * GlobalField / DOF layout             -- src/solver/01_Types.jl:110-132; 03_GlobalAssembly.jl:6-75
  * GeneralAlpha / update_OneStep!       -- src/solver/04_Time_Domain.jl:1-80

The reference's symbolic front end (src/symbolics) stays Julia; here its PRODUCT for a
weak form -- the residual / linear-gradient / nonlinear-gradient term lists -- is written
out by hand in problems.py, and this module replays the call sequence the code generator
would emit for those lists.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import operators as ops
from .geometry import ElementGeometry, FacetGeometry, update_basic_boundary, update_basic_elements
from .mesh import ClassicalMesh, Facets
from .reference_element import ClassicalDiscretization
from .solvers import normalized_norm

Env = Dict[str, np.ndarray]


@dataclass
class ResTerm:
    dual_pos: int
    dual_s: int  # 0 value, 1+d derivative
    fn: Callable[[Env], np.ndarray]


@dataclass
class GradTerm:
    dual_pos: int
    dual_s: int
    base_pos: int
    base_s: int
    fn: Callable[[Env], np.ndarray]
    td_order: int = 0


@dataclass
class AssembleWeakform:
    """02_LocalAssembly.jl:30-58."""

    inner_vars: List[Tuple[str, int, int, int]] = field(default_factory=list)  # (name, basic_pos, s, td_order)
    cp_ext_vars: List[Tuple[str, str, int]] = field(default_factory=list)  # (name, controlpoint sym, s)
    normals: List[Tuple[str, int]] = field(default_factory=list)  # (name, component)
    residues: List[ResTerm] = field(default_factory=list)
    linear_gradients: List[GradTerm] = field(default_factory=list)
    nonlinear_gradients: List[GradTerm] = field(default_factory=list)

    def sparse_positions(self):
        return {(g.dual_pos, g.base_pos) for g in self.linear_gradients + self.nonlinear_gradients}


@dataclass
class GeneralAlpha:
    """04_Time_Domain.jl:1-7; FEM_Domain builds it with dissipative=true (01_Types.jl:168)."""

    alpha_params: Tuple[float, ...] = (1.0, 1.0, 1.0)
    gamma_params: Tuple[float, ...] = (1.0, 1.0)
    beta_params: np.ndarray = field(default_factory=lambda: np.zeros(0))
    K_params: np.ndarray = field(default_factory=lambda: np.zeros(0))


class FEMDomain:
    """One workpiece + its GlobalField (01_Types.jl:110-169)."""

    def __init__(self, mesh: ClassicalMesh, disc: ClassicalDiscretization, n_fields: int,
                 domain_wf: AssembleWeakform, boundaries: Sequence[Tuple[Facets, AssembleWeakform]],
                 max_time_level: int = 0, dissipative: bool = True):
        self.mesh, self.disc, self.n_fields = mesh, disc, n_fields
        self.domain_wf = domain_wf
        self.boundaries = list(boundaries)
        self.max_time_level = max_time_level
        self.controlpoints: Dict[str, np.ndarray] = {}
        self.time = GeneralAlpha(gamma_params=(1.0, 1.0) if dissipative else (0.5, 0.5))
        self.t, self.dt = 0.0, 1.0
        self.converge_tol = 1e-6
        self.linear_solver: Optional[Callable] = None
        self.log: List[str] = []
        # update_Mesh (2_Interface.jl:98-108)
        self.elgeo: ElementGeometry = update_basic_elements(mesh, disc)
        self.fgeo: List[FacetGeometry] = [update_basic_boundary(mesh, disc, f) for f, _ in self.boundaries]
        # assemble_Global_Variables! (03_GlobalAssembly.jl:6-37)
        self.variable_size = mesh.ncp
        self.basicfield_size = n_fields * mesh.ncp
        nglob = (max_time_level + 1) * self.basicfield_size
        self.x, self.dx, self.x_star = np.zeros(nglob), np.zeros(nglob), np.zeros(nglob)
        self.residue = np.zeros(self.basicfield_size)
        blocks = set(domain_wf.sparse_positions())
        for _, wf in self.boundaries:
            blocks |= wf.sparse_positions()
        self.pattern = ops.assemble_sparse_id(mesh.cp_ids, mesh.ncp, sorted(blocks))
        self.K_linear = np.zeros(self.pattern.nnz)
        self.K_total = np.zeros(self.pattern.nnz)
        self._slots = {b: self.pattern.sparse_ids_by_el(b) for b in self.pattern.blocks}

    # -- assemble_X! / dessemble_X! (03_GlobalAssembly.jl:44-75) -------------------------
    def assemble_x(self, inner_infos: Sequence[Tuple[str, int, int]]):
        for sym, basic_pos, td in inner_infos:
            o = basic_pos * self.variable_size + td * self.basicfield_size
            self.x[o:o + self.variable_size] = self.controlpoints[sym]

    def dessemble_x(self, inner_infos: Sequence[Tuple[str, int, int]]):
        for sym, basic_pos, td in inner_infos:
            o = basic_pos * self.variable_size + td * self.basicfield_size
            self.controlpoints[sym] = self.x[o:o + self.variable_size].copy()

    # -- generated updater bodies (05_CodeGenerator.jl:156-258) -----------------------------
    def _parts(self):
        nel = self.mesh.nel
        ids = np.arange(nel)
        yield self.domain_wf, self.elgeo.integral_vals, self.elgeo.integral_weights, ids, ids, None
        for (facets, wf), fg in zip(self.boundaries, self.fgeo):
            fid = np.arange(len(facets))
            yield wf, fg.integral_vals, fg.integral_weights, fid, facets.element_ID, fg

    def _declare_extervars(self, wf: AssembleWeakform, vals, host, el, fg, env: Env):
        """declare_Extervar_GPU (05_CodeGenerator.jl:15-50)."""
        for name, sym, s in wf.cp_ext_vars:
            env[name] = ops.var_basic(vals, s, 0, self.mesh.cp_ids, self.controlpoints[sym], host, el)
        for name, comp in wf.normals:
            env[name] = fg.normal_directions[:, comp, :][:, host]
        env["t"], env["dt"] = self.t, self.dt

    def K_linear_func(self):
        """update_K_Linear_<id> (gen_K_Linear_GPU :52-91)."""
        self.K_linear[:] = 0.0
        for wf, vals, w, host, el, fg in self._parts():
            if not wf.linear_gradients:
                continue
            env: Env = {}
            self._declare_extervars(wf, vals, host, el, fg, env)
            for g in wf.linear_gradients:
                v = g.fn(env) * self.time.K_params[g.td_order] * w[:, host]
                ops.kval_basic(vals, g.dual_s, g.base_s, v, self._slots[(g.dual_pos, g.base_pos)], 0,
                               self.K_linear, host, el)

    def K_nonlinear_func(self):
        """update_K_NonLinear_<id> (gen_Res_K_NonLinear_GPU :93-154; prologue :282-283)."""
        self.residue[:] = 0.0
        self.K_total[:] = self.K_linear
        for wf, vals, w, host, el, fg in self._parts():
            env: Env = {}
            for name, basic_pos, s, td in wf.inner_vars:  # declare_Innervar_GPU :1-13
                shift = td * self.basicfield_size + basic_pos * self.variable_size
                env[name] = ops.var_basic(vals, s, shift, self.mesh.cp_ids, self.x_star, host, el)
            self._declare_extervars(wf, vals, host, el, fg, env)
            for r in wf.residues:
                v = r.fn(env) * w[:, host]
                ops.res_basic(vals, r.dual_s, v, r.dual_pos * self.variable_size, self.mesh.cp_ids,
                              self.residue, host, el)
            for g in wf.nonlinear_gradients:
                v = g.fn(env) * self.time.K_params[g.td_order] * w[:, host]
                ops.kval_basic(vals, g.dual_s, g.base_s, v, self._slots[(g.dual_pos, g.base_pos)], 0,
                               self.K_total, host, el)

    # -- time domain (04_Time_Domain.jl) ------------------------------------------------
    def update_time(self):
        """:10-18."""
        L = self.max_time_level
        self.t += self.dt
        g = self.time.gamma_params
        prod_gamma = np.array([np.prod(g[:i]) for i in range(L + 1)])
        dt_params = np.array([self.dt ** i for i in range(L + 1)])
        self.time.beta_params = 1.0 / (prod_gamma * dt_params)
        self.time.K_params = np.array(self.time.alpha_params[:L + 1]) * self.time.beta_params

    def initialize_dx(self):
        """:20-30."""
        n = self.basicfield_size
        self.dx[:] = 0.0
        for lvl in range(self.max_time_level, 0, -1):
            lo, hi = slice((lvl - 1) * n, lvl * n), slice(lvl * n, (lvl + 1) * n)
            self.dx[lo] = self.dt * (self.x[hi] + self.time.gamma_params[lvl - 1] * self.dx[hi])

    def update_dx(self, delta_x):
        """:32-39."""
        n = self.basicfield_size
        for lvl in range(self.max_time_level + 1):
            self.dx[lvl * n:(lvl + 1) * n] += self.time.beta_params[lvl] * delta_x

    def update_x_star(self):
        """:41-49."""
        n = self.basicfield_size
        self.x_star[:] = self.x
        for lvl in range(self.max_time_level + 1):
            sl = slice(lvl * n, (lvl + 1) * n)
            self.x_star[sl] += self.time.alpha_params[lvl] * self.dx[sl]

    def update_one_step(self, max_iter: int = 4) -> List[float]:
        """update_OneStep! (:59-80).  Returns the residual history."""
        self.update_time()
        self.initialize_dx()
        self.K_linear_func()
        counter = -1
        hist = []
        while True:
            self.update_x_star()
            self.K_nonlinear_func()
            res = normalized_norm(self.residue)
            counter += 1
            hist.append(res)
            if res < self.converge_tol or counter > max_iter:
                break
            delta_x = self.linear_solver(self)
            self.update_dx(-delta_x)
        self.x += self.dx
        return hist
