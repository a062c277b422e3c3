"""Weak form -> AssembleWeakform via symbolic differentiation (oracle; test infrastructure only).

Restates what the reference's symbolic layer hands to the code generator for an arbitrary weak form:
  build_WeakForm / group by dual word        src/symbolics/10_WeakForm.jl:51-124
  variation -> gradient terms                 src/symbolics/09_Differentiation.jl:1-112
  residues / linear / nonlinear gradients     src/solver/02_LocalAssembly.jl:30-58
  explicit_max_sd_order                       src/solver/05_CodeGenerator.jl:6,66,105,127
The reference does this with its own rule-based CAS; here sympy differentiates the same expressions.  A weak form
is given as a list of ``Bilinear(dual_word, expr)`` pairs with the dummy indices already expanded by the caller.

Words: ``W.val(field)`` and ``W.d(field, j)`` (first spatial derivative, 0-based j) are inner-variable words;
``W.ext(name)`` control-point externals (CONTROLPOINT_VAR); ``W.n(j)`` the facet normal.  Words with more than
``max_sd_order`` spatial derivatives evaluate to 0, exactly like the generated code leaves their buffers at zero.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np
import sympy as sp

from .fem import AssembleWeakform, GradTerm, ResTerm


class Words:
    def __init__(self, dim: int, fields: Sequence[str]):
        self.dim = dim
        self.fields = sorted(fields)  # basic_vars sorted by symbol (02_LocalAssembly.jl:93-94)
        self.pos = {f: i for i, f in enumerate(self.fields)}
        self._inner: Dict[str, Tuple[int, int]] = {}
        self._ext: Dict[str, str] = {}
        self._nrm: Dict[str, int] = {}

    def val(self, f: str) -> sp.Symbol:
        name = f"{f}"
        self._inner[name] = (self.pos[f], 0)
        return sp.Symbol(name, real=True)

    def d(self, f: str, j: int) -> sp.Symbol:
        name = f"{f}__d{j}"
        self._inner[name] = (self.pos[f], 1 + j)
        return sp.Symbol(name, real=True)

    def dd(self, f: str, j: int, k: int):
        """Second derivative: dropped by explicit_max_sd_order = 1 (evaluates to zero)."""
        return sp.Integer(0)

    def ext(self, name: str) -> sp.Symbol:
        self._ext[f"ext__{name}"] = name
        return sp.Symbol(f"ext__{name}", real=True)

    def n(self, j: int) -> sp.Symbol:
        self._nrm[f"n__{j}"] = j
        return sp.Symbol(f"n__{j}", real=True)


def assemble(W: Words, bilinears: List[Tuple[sp.Symbol, sp.Expr]]) -> AssembleWeakform:
    """Group Bilinear(dual, expr) by dual word; residual = expr, gradients = d expr / d inner word."""
    by_dual: Dict[str, sp.Expr] = {}
    for dual, expr in bilinears:
        by_dual[dual.name] = by_dual.get(dual.name, 0) + sp.sympify(expr)
    wf = AssembleWeakform()
    used_inner, used_ext, used_n = set(), set(), set()

    def compile_fn(expr):
        expr = sp.simplify(expr)
        syms = sorted(expr.free_symbols, key=lambda s: s.name)
        for s in syms:
            if s.name in W._inner:
                used_inner.add(s.name)
            elif s.name in W._ext:
                used_ext.add(s.name)
            elif s.name in W._nrm:
                used_n.add(s.name)
            else:
                raise KeyError(s.name)
        f = sp.lambdify(syms, expr, "numpy")
        names = [s.name for s in syms]
        return lambda env, f=f, names=names: f(*[env[k] for k in names])

    for dname, expr in by_dual.items():
        dpos, ds = W._inner[dname]
        expr = sp.expand(expr)
        if expr == 0:
            continue
        wf.residues.append(ResTerm(dpos, ds, compile_fn(expr)))
        for vname, (vpos, vs) in list(W._inner.items()):
            v = sp.Symbol(vname, real=True)
            if not expr.has(v):
                continue
            coeff = sp.diff(expr, v)
            if coeff == 0:
                continue
            nonlinear = any(s.name in W._inner for s in coeff.free_symbols)
            term = GradTerm(dpos, ds, vpos, vs, compile_fn(coeff))
            (wf.nonlinear_gradients if nonlinear else wf.linear_gradients).append(term)
    wf.inner_vars = [(nm, W._inner[nm][0], W._inner[nm][1], 0) for nm in sorted(used_inner)]
    wf.cp_ext_vars = [(nm, W._ext[nm], 0) for nm in sorted(used_ext)]
    wf.normals = [(nm, W._nrm[nm]) for nm in sorted(used_n)]
    return wf
