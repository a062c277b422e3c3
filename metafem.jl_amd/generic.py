"""Generic (unstructured, any weak form) path of the MI355X backend: the host-side mirror of what
`compile_Updater_GPU` emits and `update_OneStep!` drives in the reference, executed with the C-ABI kernels.

  update_Mesh                         -> mfem_update_basic_elements / mfem_update_basic_boundary   (4_Update_Integrator.jl)
  assemble_Global_Variables!          -> mfem_pattern_build (+ vectors)                             (03_GlobalAssembly.jl:6-140)
  update_K_Linear_<id>                -> mfem_op_var (externals), coefficient broadcasts, mfem_op_kval   (05_CodeGenerator.jl:52-91)
  update_K_NonLinear_<id>             -> mfem_op_var, broadcasts, mfem_op_res, mfem_op_kval               (:93-154, 282-283)
  update_OneStep!                     -> Newton loop                                                  (04_Time_Domain.jl:59-80)

The coefficient expressions of a weak form (`vals = @. expr * K_params * w[:, ids]`, 05_CodeGenerator.jl:75,110,136)
are elementwise broadcasts over [itg, n_items] arrays; the reference evaluates them as Julia GPU broadcasts and so
does this mirror with torch elementwise ops on device tensors -- no kernel of the hot path is written in torch.
In the Julia integration those callables are the generated `@.` expressions.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import check, lib


@dataclass
class ResTerm:
    dual_pos: int
    dual_s: int  # 0 value, 1 + d = d/dx_d
    fn: Callable


@dataclass
class GradTerm:
    dual_pos: int
    dual_s: int
    base_pos: int
    base_s: int
    fn: Callable
    td_order: int = 0


@dataclass
class WeakForm:
    """AssembleWeakform (02_LocalAssembly.jl:30-58): what the symbolic layer hands to the code generator."""
    inner_vars: List[Tuple[str, int, int, int]] = field(default_factory=list)  # (name, basic_pos, s, td_order)
    cp_ext_vars: List[Tuple[str, str, int]] = field(default_factory=list)  # (name, controlpoint symbol, s)
    normals: List[Tuple[str, int]] = field(default_factory=list)  # (name, component)
    residues: List[ResTerm] = field(default_factory=list)
    linear_gradients: List[GradTerm] = field(default_factory=list)
    nonlinear_gradients: List[GradTerm] = field(default_factory=list)

    def sparse_positions(self):
        return {(g.dual_pos, g.base_pos) for g in self.linear_gradients + self.nonlinear_gradients}


class _NeedsEnv(Exception):
    pass


class _NoEnv(dict):
    """Probe environment: a coefficient function that reads -- or merely asks about -- any inner variable / external / normal
    is not a constant (every way of looking into the mapping raises, including `'x' in env`, iteration and len)."""

    def _needs(self, *a, **k):
        raise _NeedsEnv(a[0] if a else "env")

    __getitem__ = get = __contains__ = __iter__ = __len__ = keys = values = items = setdefault = pop = _needs

    def __bool__(self):
        raise _NeedsEnv("env")


def constant_coefficient(fn) -> Optional[float]:
    """The value of a term's coefficient if it does not depend on the integration point, else None."""
    try:
        v = fn(_NoEnv())
    except _NeedsEnv:
        return None
    except Exception:
        return None
    if torch.is_tensor(v) or isinstance(v, np.ndarray):
        return None
    try:
        return float(v)
    except Exception:
        return None


class _Group:
    """Integration hosts of one launch family (the elements, or the facets of one boundary group)."""

    def __init__(self, vals, weights, host_ids, el_ids, itg, normals=None, colour_offsets=None, facet_el=None, facet_eidx=None):
        self.vals, self.weights, self.host_ids, self.el_ids, self.itg = vals, weights, host_ids, el_ids, itg
        self.normals, self.colour_offsets = normals, colour_offsets
        self.facet_el, self.facet_eidx = facet_el, facet_eidx  # boundary groups: element / local face id of every facet
        self.n = el_ids.numel()


class GenericDomain:
    """FEM_Domain with one workpiece + GlobalField; static (max_time_level = 0) or generalised-alpha transient."""

    def __init__(self, ctx, space, coords: np.ndarray, cp_ids: np.ndarray, n_fields: int, domain_wf: WeakForm,
                 boundaries: Sequence[Tuple[np.ndarray, np.ndarray, WeakForm]],
                 element_colours: Optional[np.ndarray] = None, max_time_level: int = 0, dissipative: bool = True,
                 batched: bool = True, fused: bool = True, row_owner: bool = True):
        """coords [ncp, dim]; cp_ids [itp, nel] 0-based (controlpoint_IDs in basis order); boundaries =
        [(element_ID[nf], element_eindex[nf] 0-based local face ids, WeakForm)].  element_colours (optional):
        a colour per element such that same-colour elements share no control point -> atomics-free scatter with a fixed
        summation order; "auto" computes one with mesh.colour_Elements (elements and boundary facets); without it the
        operators use FP64 atomics like the reference (order of the additions, hence the last bits, not reproducible)."""
        self.ctx, self.space, self.n_fields = ctx, space, n_fields
        # batched = True: one mfem_op_*_batch launch per integration domain; False: one launch per term, the literal
        # call sequence of the reference's generated updaters (kept for parity tests of the single-term seam)
        self.batched = batched
        # fused = True: linear-gradient terms with constant coefficients go through mfem_mesh_assemble_elements / _facets
        # (geometry on the fly, all such terms of a domain in one launch); the other terms keep the operator path
        self.fused = fused
        # row_owner = True: the fused element assembly runs in its row-owner form (mfem_mesh_assemble_elements_rows: element
        # matrices -> scratch -> one wave per CSR row; no atomics, fixed summation order); False: scatter through the slot table
        self.row_owner = row_owner
        dev = f"cuda:{ctx.device}"
        self.dev = dev
        dim = space.dim
        itp, nel = cp_ids.shape
        ncp = coords.shape[0]
        self.ncp, self.nel, self.itp, self.dim = ncp, nel, itp, dim
        self.domain_wf = domain_wf
        self.bwfs = [b[2] for b in boundaries]
        f64 = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
        i32 = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.int32, device=dev)
        self.coords = f64(coords.T)  # SoA x1|x2|x3
        self.cp = i32(cp_ids.T + 1)  # (nel, itp) C order == [itp, nel] column-major, 1-based like the reference
        # ---- update_Mesh
        nsd = 1 + dim
        ref = f64(space.ref_itp_vals.ravel(order="F"))
        self._ref, self._itgw = ref, f64(space.itg_weight)
        # adjacency of the row-owner assembly: for every control point the (element * itp + local id) pairs, ascending
        flat = np.ascontiguousarray(cp_ids.T).ravel()
        self._adj = i32(np.argsort(flat, kind="stable"))
        self._adj_ptr = torch.tensor(np.concatenate([[0], np.cumsum(np.bincount(flat, minlength=ncp))]), dtype=torch.int64, device=dev)
        vals = torch.empty(space.itg * itp * nsd * nel, dtype=torch.float64, device=dev)
        w = torch.empty(space.itg * nel, dtype=torch.float64, device=dev)
        check(lib.mfem_update_basic_elements(ctx._h, dim, space.itg, itp, nel, ncp, ref.data_ptr(), f64(space.itg_weight).data_ptr(),
                                             self.coords.data_ptr(), self.cp.data_ptr(), 1, vals.data_ptr(), w.data_ptr()))
        auto_colours = isinstance(element_colours, str)
        if auto_colours:
            if element_colours != "auto":
                raise ValueError("element_colours: an array, None or 'auto'")
            from .mesh import colour_Elements

            element_colours = colour_Elements(cp_ids)
        if element_colours is not None:
            order = np.argsort(element_colours, kind="stable")
            offs = np.concatenate([[0], np.cumsum(np.bincount(element_colours, minlength=int(element_colours.max()) + 1))])
        else:
            order, offs = np.arange(nel), None
        ids = i32(order + 1)
        self.groups = [_Group(vals, w, ids, ids, space.itg, colour_offsets=offs)]
        nface = space.bdy_ref_itp_vals.shape[0]
        bref = f64(np.concatenate([space.bdy_ref_itp_vals[f].ravel(order="F") for f in range(nface)]))
        bw = f64(space.bdy_itg_weights.ravel())
        btan = f64(np.concatenate([space.bdy_tangent_directions[f].ravel(order="F") for f in range(nface)]))
        self._bref, self._bw, self._btan, self._nface = bref, bw, btan, nface
        for el, eidx, _ in boundaries:
            nf = len(el)
            fv = torch.empty(space.itg_b * itp * nsd * nf, dtype=torch.float64, device=dev)
            fw = torch.empty(space.itg_b * nf, dtype=torch.float64, device=dev)
            fn = torch.empty(space.itg_b * dim * nf, dtype=torch.float64, device=dev)
            eld, eid = i32(np.asarray(el) + 1), i32(np.asarray(eidx) + 1)
            check(lib.mfem_update_basic_boundary(ctx._h, dim, space.itg_b, itp, nface, nf, ncp, bref.data_ptr(), bw.data_ptr(),
                                                 btan.data_ptr(), self.coords.data_ptr(), self.cp.data_ptr(), eld.data_ptr(),
                                                 eid.data_ptr(), 1, fv.data_ptr(), fw.data_ptr(), fn.data_ptr()))
            if auto_colours and nf > 0:
                # boundary operators touch every node of the host element (05_CodeGenerator.jl:175-189): colour the facets
                # by their hosts' node sets (two facets of one element conflict automatically)
                fcol = colour_Elements(cp_ids[:, np.asarray(el)])
                forder = np.argsort(fcol, kind="stable")
                foffs = np.concatenate([[0], np.cumsum(np.bincount(fcol, minlength=int(fcol.max()) + 1))])
                host = i32(forder + 1)
                self.groups.append(_Group(fv, fw, host, i32(np.asarray(el)[forder] + 1), space.itg_b,
                                          normals=fn.view(nf, dim, space.itg_b), colour_offsets=foffs, facet_el=eld, facet_eidx=eid))
                continue
            host = i32(np.arange(nf) + 1)
            self.groups.append(_Group(fv, fw, host, eld, space.itg_b, normals=fn.view(nf, dim, space.itg_b), facet_el=eld,
                                      facet_eidx=eid))
        # ---- assemble_Global_Variables!
        from . import assemble_SparseID  # late import: package root defines it

        self.variable_size = ncp
        self.basicfield_size = n_fields * ncp
        self.A, self.slots = assemble_SparseID(self.cp, ncp, n_fields=n_fields, index_base=1, ctx=ctx)
        n = self.basicfield_size
        z = lambda m: torch.zeros(m, dtype=torch.float64, device=dev)
        # x, dx, x_star hold max_time_level + 1 blocks of basicfield_size (03_GlobalAssembly.jl:27-31)
        self.max_time_level = max_time_level
        nglob = (max_time_level + 1) * n
        self.x, self.dx, self.x_star, self.residue = z(nglob), z(nglob), z(nglob), z(n)
        self.K_linear = z(self.A.nnz)
        # K_total = K_linear + the nonlinear gradient terms (05_CodeGenerator.jl:282-283).  A form without nonlinear gradient terms never adds anything: its
        # K_total IS K_linear (the same storage; the solver reads K, it does not scale it in place) -- no second nnz-sized array (15 GB for hex-20
        # elasticity at 96^3), no copy per Newton step
        self._K_total_aliases = not any(wf.nonlinear_gradients for wf in [domain_wf] + [b[-1] for b in boundaries])
        self.K_total = self.K_linear if self._K_total_aliases else z(self.A.nnz)
        self.controlpoints: Dict[str, torch.Tensor] = {}
        self.converge_tol = 1e-6
        # GeneralAlpha (04_Time_Domain.jl:1-7); FEM_Domain builds it with dissipative = true (01_Types.jl:168)
        self.alpha_params = (1.0, 1.0, 1.0)
        self.gamma_params = (1.0, 1.0) if dissipative else (0.5, 0.5)
        self.beta_params = [1.0]
        self.K_params = [1.0]  # static: alpha_0 * beta_0 (04_Time_Domain.jl:13-17)
        self.t, self.dt = 0.0, 1.0
        self.linear_solver: Optional[Callable] = None
        self.history: List[float] = []

    # -- assemble_X! / dessemble_X! (03_GlobalAssembly.jl:44-75)
    def assemble_X(self, infos):
        for sym, pos, td in infos:
            o = pos * self.ncp + td * self.basicfield_size
            self.x[o:o + self.ncp] = self.controlpoints[sym]

    def dessemble_X(self, infos):
        for sym, pos, td in infos:
            o = pos * self.ncp + td * self.basicfield_size
            self.controlpoints[sym] = self.x[o:o + self.ncp].clone()

    # -- operator wrappers ------------------------------------------------------------------------
    def _layout(self, g: _Group, colours: bool):
        offs = g.colour_offsets if colours else None
        if offs is None:
            return _lib.OpLayout(g.itg, self.itp, 1 + self.dim, g.weights.numel() // g.itg, 1, 0, None), None
        arr = (C.c_int64 * len(offs))(*[int(v) for v in offs])
        return _lib.OpLayout(g.itg, self.itp, 1 + self.dim, g.weights.numel() // g.itg, 1, len(offs) - 1, arr), arr

    def _var(self, g: _Group, s: int, shift: int, x: torch.Tensor) -> torch.Tensor:
        tgt = torch.zeros(g.n * g.itg, dtype=torch.float64, device=self.dev)  # FEM_buffer zeros (05_CodeGenerator.jl:4)
        L, _k = self._layout(g, False)
        check(lib.mfem_op_var(self.ctx._h, C.byref(L), g.vals.data_ptr(), s, shift, self.cp.data_ptr(), x.data_ptr(), tgt.data_ptr(),
                              g.host_ids.data_ptr(), g.el_ids.data_ptr(), g.n))
        return tgt.view(g.n, g.itg)

    def _w(self, g: _Group) -> torch.Tensor:
        """local_integral_weights[:, local_itg_hostIDs] as [n_items, itg]."""
        return g.weights.view(-1, g.itg)[(g.host_ids - 1).long()]

    def _kval(self, g: _Group, t: GradTerm, vals: torch.Tensor, K: torch.Tensor):
        u = t.dual_pos * self.n_fields + t.base_pos
        L, _k = self._layout(g, True)
        check(lib.mfem_op_kval(self.ctx._h, C.byref(L), g.vals.data_ptr(), t.dual_s, t.base_s, vals.data_ptr(),
                               self.slots[u].data_ptr(), 0, K.data_ptr(), g.host_ids.data_ptr(), g.el_ids.data_ptr(), g.n))

    def _res(self, g: _Group, t: ResTerm, vals: torch.Tensor):
        L, _k = self._layout(g, True)
        check(lib.mfem_op_res(self.ctx._h, C.byref(L), g.vals.data_ptr(), t.dual_s, vals.data_ptr(), t.dual_pos * self.ncp,
                              self.cp.data_ptr(), self.residue.data_ptr(), g.host_ids.data_ptr(), g.el_ids.data_ptr(), g.n))

    # -- batched operator wrappers (mfem_op_*_batch): all terms of one integration domain per launch ------------------
    def _var_many(self, g: _Group, words) -> List[torch.Tensor]:
        """words: [(sd, shift, x tensor)] -> list of [n_items, itg] tensors."""
        out: List[torch.Tensor] = []
        L, _k = self._layout(g, False)
        for c0 in range(0, len(words), _lib.MAX_BATCH_TERMS):
            chunk = words[c0:c0 + _lib.MAX_BATCH_TERMS]
            tgt = torch.empty((len(chunk), g.n, g.itg), dtype=torch.float64, device=self.dev)
            terms = (_lib.VarBatchTerm * len(chunk))(*[_lib.VarBatchTerm(sd, 0, shift, x.data_ptr()) for sd, shift, x in chunk])
            check(lib.mfem_op_var_batch(self.ctx._h, C.byref(L), g.vals.data_ptr(), len(chunk), terms, self.cp.data_ptr(),
                                        tgt.data_ptr(), g.host_ids.data_ptr(), g.el_ids.data_ptr(), g.n))
            out += [tgt[i] for i in range(len(chunk))]
        return out

    def _kval_many(self, g: _Group, terms, env, w, K: torch.Tensor):
        """terms: GradTerms; coefficient tensors are evaluated here, stacked term-major, sorted by sparse block."""
        if not terms:
            return
        order = sorted(range(len(terms)), key=lambda i: terms[i].dual_pos * self.n_fields + terms[i].base_pos)
        L, _k = self._layout(g, True)
        stride = self.nel * self.itp * self.itp
        for c0 in range(0, len(order), _lib.MAX_BATCH_TERMS):
            ids = order[c0:c0 + _lib.MAX_BATCH_TERMS]
            vals = torch.stack([self._vals(terms[i].fn, env, w, self.K_params[terms[i].td_order]) for i in ids])
            arr = (_lib.KvalTerm * len(ids))(*[_lib.KvalTerm(terms[i].dual_s, terms[i].base_s,
                                                             terms[i].dual_pos * self.n_fields + terms[i].base_pos, 0) for i in ids])
            check(lib.mfem_op_kval_batch(self.ctx._h, C.byref(L), g.vals.data_ptr(), len(ids), arr, vals.data_ptr(),
                                         self.slots.data_ptr(), stride, 0, K.data_ptr(), g.host_ids.data_ptr(),
                                         g.el_ids.data_ptr(), g.n))

    def _res_many(self, g: _Group, terms, env, w):
        if not terms:
            return
        order = sorted(range(len(terms)), key=lambda i: terms[i].dual_pos)
        L, _k = self._layout(g, True)
        for c0 in range(0, len(order), _lib.MAX_BATCH_TERMS):
            ids = order[c0:c0 + _lib.MAX_BATCH_TERMS]
            vals = torch.stack([self._vals(terms[i].fn, env, w) for i in ids])
            arr = (_lib.ResBatchTerm * len(ids))(*[_lib.ResBatchTerm(terms[i].dual_s, 0, terms[i].dual_pos * self.ncp) for i in ids])
            check(lib.mfem_op_res_batch(self.ctx._h, C.byref(L), g.vals.data_ptr(), len(ids), arr, vals.data_ptr(),
                                        self.cp.data_ptr(), self.residue.data_ptr(), g.host_ids.data_ptr(), g.el_ids.data_ptr(), g.n))

    def _externals(self, wf: WeakForm, g: _Group, env: dict):
        if self.batched and wf.cp_ext_vars:
            tg = self._var_many(g, [(s, 0, self.controlpoints[sym]) for _, sym, s in wf.cp_ext_vars])
            for (name, _, _), t in zip(wf.cp_ext_vars, tg):
                env[name] = t
        else:
            for name, sym, s in wf.cp_ext_vars:  # declare_Extervar_GPU (05_CodeGenerator.jl:15-50)
                env[name] = self._var(g, s, 0, self.controlpoints[sym])
        for name, comp in wf.normals:
            env[name] = g.normals[:, comp, :][(g.host_ids - 1).long()]
        env["t"], env["dt"] = self.t, self.dt

    def _vals(self, fn, env, w, scale=1.0) -> torch.Tensor:
        v = fn(env)
        if not torch.is_tensor(v):
            v = torch.full_like(w, float(v))
        return (v * scale * w).contiguous()

    def _parts(self):
        yield self.domain_wf, self.groups[0]
        for wf, g in zip(self.bwfs, self.groups[1:]):
            yield wf, g

    def _row_ranks(self) -> torch.Tensor:
        """Column ranks of the row-owner assembly (mfem_mesh_row_ranks), built on first use."""
        if getattr(self, "_ranks", None) is None:
            ranks = torch.empty(self.nel * self.itp * self.itp, dtype=torch.int16, device=self.dev)
            rc = lib.mfem_mesh_row_ranks(self.ctx._h, self.itp, self.nel, self.ncp, self.n_fields, self.A._h, self._adj_ptr.data_ptr(),
                                         self._adj.data_ptr(), self.cp.data_ptr(), 1, ranks.data_ptr())
            if rc == -3:  # MFEM_ERR_UNSUPPORTED: an element lists a control point twice -> the scatter form from now on
                self.row_owner = False
                return None
            check(rc)
            self._ranks = ranks
        return self._ranks

    def _assemble_const(self, g: _Group, cterms, K: torch.Tensor):
        """cterms: [(GradTerm, coefficient)] -> one fused launch per colour (mfem_mesh_assemble_elements / _facets)."""
        cterms = sorted(cterms, key=lambda tc: tc[0].dual_pos * self.n_fields + tc[0].base_pos)
        offs = g.colour_offsets
        ncol = 0 if offs is None else len(offs) - 1
        carr = None if offs is None else (C.c_int64 * len(offs))(*[int(v) for v in offs])
        stride = self.nel * self.itp * self.itp
        for c0 in range(0, len(cterms), _lib.MAX_BATCH_TERMS):
            chunk = cterms[c0:c0 + _lib.MAX_BATCH_TERMS]
            arr = (_lib.ConstTerm * len(chunk))(*[_lib.ConstTerm(t.dual_s, t.base_s, t.dual_pos * self.n_fields + t.base_pos, 0,
                                                                 c * self.K_params[t.td_order]) for t, c in chunk])
            if g.facet_el is None and self.row_owner and self._row_ranks() is not None:
                fresh = getattr(self, "_K_fresh", False) and K is self.K_linear and self.nel > 0
                fn = lib.mfem_mesh_assemble_elements_rows_set if fresh else lib.mfem_mesh_assemble_elements_rows
                rc = fn(self.ctx._h, self.dim, self.space.itg, self.itp, self.nel, self.ncp,
                                                          self._ref.data_ptr(), self._itgw.data_ptr(), self.coords.data_ptr(),
                                                          self.cp.data_ptr(), 1, len(chunk), arr, self.n_fields, self.A._h,
                                                          self._adj_ptr.data_ptr(), self._adj.data_ptr(), self._row_ranks().data_ptr(),
                                                          K.data_ptr())
                if rc == 0:
                    if fresh:
                        self._K_fresh = False
                    continue
                if rc != -3:  # MFEM_ERR_UNSUPPORTED (row too long / scratch too large) falls back to the scatter form
                    check(rc)
            self._K_started(K)
            if g.facet_el is None:
                check(lib.mfem_mesh_assemble_elements(self.ctx._h, self.dim, self.space.itg, self.itp, self.nel, self.ncp,
                                                      self._ref.data_ptr(), self._itgw.data_ptr(), self.coords.data_ptr(),
                                                      self.cp.data_ptr(), 1, len(chunk), arr, self.slots.data_ptr(), stride,
                                                      K.data_ptr(), g.host_ids.data_ptr(), g.n, ncol, carr))
            else:
                check(lib.mfem_mesh_assemble_facets(self.ctx._h, self.dim, self.space.itg_b, self.itp, self._nface,
                                                    g.facet_el.numel(), self.ncp, self._bref.data_ptr(), self._bw.data_ptr(),
                                                    self._btan.data_ptr(), self.coords.data_ptr(), self.cp.data_ptr(),
                                                    g.facet_el.data_ptr(), g.facet_eidx.data_ptr(), 1, len(chunk), arr,
                                                    self.slots.data_ptr(), stride, K.data_ptr(), g.host_ids.data_ptr(), g.n, ncol,
                                                    carr))

    # -- generated updater bodies -------------------------------------------------------------------
    def K_linear_func(self):
        # K_linear starts from zero (05_CodeGenerator.jl:282).  When the first thing it receives is the row-owner element assembly, that launch WRITES
        # every row instead (mfem_mesh_assemble_elements_rows_set): no memset, no read of the zeros.
        self._K_fresh = True
        for wf, g in self._parts():
            if not wf.linear_gradients:
                continue
            terms = wf.linear_gradients
            if self.fused:
                coefs = [constant_coefficient(t.fn) for t in terms]
                cterms = [(t, c) for t, c in zip(terms, coefs) if c is not None]
                terms = [t for t, c in zip(terms, coefs) if c is None]
                if cterms:
                    self._assemble_const(g, cterms, self.K_linear)
                if not terms:
                    continue
            self._K_started(self.K_linear)
            env: dict = {}
            self._externals(wf, g, env)
            w = self._w(g)
            if self.batched:
                self._kval_many(g, terms, env, w, self.K_linear)
                continue
            for t in terms:
                self._kval(g, t, self._vals(t.fn, env, w, self.K_params[t.td_order]), self.K_linear)
        self._K_started(self.K_linear)

    def _K_started(self, K: torch.Tensor):
        """Zero K unless something has been put into it since K_linear_func began."""
        if getattr(self, "_K_fresh", False):
            K.zero_()
            self._K_fresh = False

    def K_nonlinear_func(self):
        self.residue.zero_()
        if not self._K_total_aliases:
            self.K_total.copy_(self.K_linear)  # 05_CodeGenerator.jl:282-283
        for wf, g in self._parts():
            env: dict = {}
            if self.batched:
                tg = self._var_many(g, [(s, td * self.basicfield_size + pos * self.ncp, self.x_star) for _, pos, s, td in wf.inner_vars])
                for (name, _, _, _), t in zip(wf.inner_vars, tg):
                    env[name] = t
                self._externals(wf, g, env)
                w = self._w(g)
                self._res_many(g, wf.residues, env, w)
                self._kval_many(g, wf.nonlinear_gradients, env, w, self.K_total)
                continue
            for name, pos, s, td in wf.inner_vars:  # declare_Innervar_GPU (:1-13)
                env[name] = self._var(g, s, td * self.basicfield_size + pos * self.ncp, self.x_star)
            self._externals(wf, g, env)
            w = self._w(g)
            for t in wf.residues:
                self._res(g, t, self._vals(t.fn, env, w))
            for t in wf.nonlinear_gradients:
                self._kval(g, t, self._vals(t.fn, env, w, self.K_params[t.td_order]), self.K_total)

    # -- time domain (04_Time_Domain.jl:9-49) ---------------------------------------------------------
    def update_Time(self):
        L = self.max_time_level
        self.t += self.dt
        prod_gamma = [float(np.prod(self.gamma_params[:i])) for i in range(L + 1)]
        self.beta_params = [1.0 / (prod_gamma[i] * self.dt ** i) for i in range(L + 1)]
        self.K_params = [self.alpha_params[i] * self.beta_params[i] for i in range(L + 1)]

    def _level(self, v: torch.Tensor, lvl: int) -> torch.Tensor:
        n = self.basicfield_size
        return v[lvl * n:(lvl + 1) * n]

    def initialize_dx(self):
        self.dx.zero_()
        for lvl in range(self.max_time_level, 0, -1):  # predictor: dx[l-1] = dt (x[l] + gamma_l dx[l])
            torch.add(self._level(self.x, lvl), self._level(self.dx, lvl), alpha=self.gamma_params[lvl - 1],
                      out=self._level(self.dx, lvl - 1))
            self._level(self.dx, lvl - 1).mul_(self.dt)

    def update_dx(self, delta_x: torch.Tensor):
        for lvl in range(self.max_time_level + 1):
            self._level(self.dx, lvl).add_(delta_x, alpha=self.beta_params[lvl])

    def update_x_star(self):
        for lvl in range(self.max_time_level + 1):
            torch.add(self._level(self.x, lvl), self._level(self.dx, lvl), alpha=self.alpha_params[lvl],
                      out=self._level(self.x_star, lvl))

    def update_OneStep(self, max_iter: int = 4):
        """update_OneStep! (04_Time_Domain.jl:59-80)."""
        from . import normalized_norm

        self.update_Time()
        self.initialize_dx()
        self.K_linear_func()
        counter = -1
        self.history = []
        while True:
            self.update_x_star()
            self.K_nonlinear_func()
            res = normalized_norm(self.residue, self.ctx)
            counter += 1
            self.history.append(res)
            if res < self.converge_tol or counter > max_iter:
                break
            delta_x = self.linear_solver(self)
            self.update_dx(-delta_x)
        self.x += self.dx
        return self.history
