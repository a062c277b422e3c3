"""Reference-element tables (oracle; test infrastructure only).

Restates, for CUBE shapes:
  * Gauss tables on [0,1]      -- src/mesh/spatial_discretization/103_Integrations.jl:1-19
  * boundary quadrature/tangents -- 103_Integrations.jl:21-58
  * Lagrange cube basis          -- 102_Interpolations.jl:3-39
  * Serendipity cube basis       -- 102_Interpolations.jl:69-110
  * evaluate_Itp_Funcs           -- 01_Classical_DIscretization.jl:83-98

All tables use the reference cell [0,1]^dim, first coordinate fastest
(SURVEY.md A3/A4/A5).  Arrays are numpy, indexed [q, a, o1, ..., od] exactly as
``ref_itp_vals`` in the reference (0-based here).
"""
from __future__ import annotations

import itertools
import math
from dataclasses import dataclass, field
from typing import List, Tuple

import numpy as np

from .polynomial import Poly, prod

# 103_Integrations.jl:3-12 --------------------------------------------------
_GP = (
    (0.0,),
    (-1.0 / math.sqrt(3.0), 1.0 / math.sqrt(3.0)),
    (-math.sqrt(3.0 / 5.0), 0.0, math.sqrt(3.0 / 5.0)),
    (
        -math.sqrt(3.0 / 7.0 + 2.0 / 7.0 * math.sqrt(6.0 / 5.0)),
        -math.sqrt(3.0 / 7.0 - 2.0 / 7.0 * math.sqrt(6.0 / 5.0)),
        math.sqrt(3.0 / 7.0 - 2.0 / 7.0 * math.sqrt(6.0 / 5.0)),
        math.sqrt(3.0 / 7.0 + 2.0 / 7.0 * math.sqrt(6.0 / 5.0)),
    ),
)
_GW = (
    (2.0,),
    (1.0, 1.0),
    (5.0 / 9.0, 8.0 / 9.0, 5.0 / 9.0),
    (
        (18.0 - math.sqrt(30.0)) / 36.0,
        (18.0 + math.sqrt(30.0)) / 36.0,
        (18.0 + math.sqrt(30.0)) / 36.0,
        (18.0 - math.sqrt(30.0)) / 36.0,
    ),
)
GAUSS_POS_SHIFTED = tuple(tuple(x / 2.0 + 0.5 for x in row) for row in _GP)  # :1,11
GAUSS_W_SHIFTED = tuple(tuple(w / 2.0 for w in row) for row in _GW)  # :2,12


def gauss_order_of(itg_order: int) -> int:
    """103_Integrations.jl:15."""
    return int(math.ceil((itg_order + 1) / 2))


def _product_first_fastest(ranges):
    """Julia ``Iterators.product`` order: FIRST index fastest."""
    for rev in itertools.product(*reversed(list(ranges))):
        yield tuple(reversed(rev))


def domain_integration_cube(itg_order: int, dim: int):
    """103_Integrations.jl:14-19 -> (pos[itg, dim], weight[itg])."""
    g = gauss_order_of(itg_order)
    if dim == 0:
        return np.zeros((1, 0)), np.ones(1)
    P, W = GAUSS_POS_SHIFTED[g - 1], GAUSS_W_SHIFTED[g - 1]
    pos, w = [], []
    for ids in _product_first_fastest([range(g)] * dim):
        pos.append([P[i] for i in ids])
        w.append(math.prod(W[i] for i in ids))
    return np.array(pos, dtype=np.float64).reshape(-1, dim), np.array(w, dtype=np.float64)


CUBE_FACE_IDS = {2: ((4, 2), (1, 3)), 3: ((5, 3), (2, 4), (1, 6))}  # :24-30 (1-based face ids)


def boundary_integration_cube(itg_order: int, dim: int):
    """103_Integrations.jl:21-58.

    Returns lists indexed by (face id - 1): positions [itg_b, dim], weights
    [itg_b], reference tangents [itg_b, dim, dim-1].
    """
    itg_pos, base_w = domain_integration_cube(itg_order, dim - 1)
    face_ids = CUBE_FACE_IDS[dim]
    nq = itg_pos.shape[0]
    nface = 2 * dim
    bdy_pos = [np.zeros((nq, dim)) for _ in range(nface)]
    bdy_tan = [np.zeros((nq, dim, dim - 1)) for _ in range(nface)]
    for normal_dim in range(1, dim + 1):
        tangent_dim = [(i + normal_dim - 1) % dim + 1 for i in range(1, dim)]  # :37
        for outward in (0, 1):
            fid = face_ids[normal_dim - 1][outward] - 1
            raw = np.zeros((nq, dim, dim - 1))
            for i, td in enumerate(tangent_dim):
                raw[:, td - 1, i] = 1.0
            if dim == 2:
                if (outward + normal_dim) != 2:  # :42-43
                    raw *= -1.0
            else:
                if outward == 0:  # :45
                    raw[:, :, 0] *= -1.0
            bdy_tan[fid] = raw
            for q in range(nq):
                p = np.zeros(dim)
                for i, td in enumerate(tangent_dim):
                    p[td - 1] = itg_pos[q, i]
                p[normal_dim - 1] = outward
                bdy_pos[fid][q] = p
    return bdy_pos, [base_w.copy() for _ in range(nface)], bdy_tan


# -- interpolation -------------------------------------------------------------
def lagrange_1d(order: int) -> List[Poly]:
    """102_Interpolations.jl:3-23."""
    pos = [i / order for i in range(order + 1)]
    out = []
    for a in range(order + 1):
        p = Poly.const(1, 1.0)
        for k in range(order + 1):
            if k == a:
                continue
            den = pos[a] - pos[k]
            term = Poly(1, {(1,): 1.0 / den, (0,): -pos[k] / den})
            p = p * term
        out.append(p)
    return out


def _lift(p1: Poly, i: int, dim: int) -> Poly:
    """substitute_Polynomial(p, 1, x_i) (03_Polynomial.jl:116-125) for a 1-D poly."""
    terms = {}
    for (k,), c in p1.terms.items():
        e = [0] * dim
        e[i] = k
        terms[tuple(e)] = terms.get(tuple(e), 0.0) + c
    return Poly(dim, terms)


def interpolation_cube_lagrange(order: int, dim: int):
    """102_Interpolations.jl:30-39 -> (funcs, ref positions [itp, dim]); x fastest."""
    f1 = lagrange_1d(order)
    tmpl = [[_lift(f, i, dim) for f in f1] for i in range(dim)]
    funcs, pos = [], []
    for ids in _product_first_fastest([range(order + 1)] * dim):
        funcs.append(prod([tmpl[i][ids[i]] for i in range(dim)]))
        pos.append([ids[i] / order for i in range(dim)])
    return funcs, np.array(pos, dtype=np.float64)


def interpolation_cube_serendipity(order: int, dim: int):
    """102_Interpolations.jl:69-110 -> (funcs, ref positions)."""
    xs = [Poly.var(dim, i) for i in range(dim)]
    funcs, pos = [], []
    if order <= 2:
        for coors in _product_first_fastest([range(2)] * dim):
            f = prod([(1 - c) - x for c, x in zip(coors, xs)])
            for i in range(1, order):
                s = [1 - 2 * c for c in coors]
                lin = Poly.const(dim, sum(si * ci for si, ci in zip(s, coors)) + i / order)
                for si, x in zip(s, xs):
                    lin = lin - si * x
                f = f * lin
            f = f / f(coors)
            funcs.append(f)
            pos.append([float(c) for c in coors])
    elif order == 3:
        for coors in _product_first_fastest([range(2)] * dim):
            sq = Poly.const(dim, -((1 / 6) ** 2 + (dim - 1) * (1 / 2) ** 2))
            for x in xs:
                sh = x - 0.5
                sq = sq + sh * sh
            f = prod([(1 - c) - x for c, x in zip(coors, xs)]) * sq
            f = f / f(coors)
            funcs.append(f)
            pos.append([float(c) for c in coors])
    else:
        raise ValueError("Undefined serendipity order")
    for ed in range(dim):
        minor = [i for i in range(dim) if i != ed]
        for mc in _product_first_fastest([range(2)] * (dim - 1)):
            base = prod([(1 - c) - xs[m] for c, m in zip(mc, minor)]) if minor else Poly.const(dim, 1.0)
            for ip in range(1, order):
                f = prod([xs[ed] - (i / order) for i in range(order + 1) if i != ip]) * base
                coor = [ip / order] * dim
                for c, m in zip(mc, minor):
                    coor[m] = float(c)
                f = f / f(coor)
                funcs.append(f)
                pos.append(coor)
    return funcs, np.array(pos, dtype=np.float64)


def evaluate_itp_funcs(funcs: List[Poly], max_sd_order: int, itg_pos: np.ndarray) -> np.ndarray:
    """01_Classical_DIscretization.jl:83-98 -> ref_itp_vals[q, a, o1..od]."""
    dim = funcs[0].dim
    nq, na = itg_pos.shape[0], len(funcs)
    out = np.zeros((nq, na) + (max_sd_order + 1,) * dim)
    for orders in itertools.product(range(max_sd_order + 1), repeat=dim):
        for a, f in enumerate(funcs):
            df = f.derivative(orders)
            for q in range(nq):
                out[(q, a) + orders] = df(itg_pos[q])
    return out


@dataclass
class ClassicalDiscretization:
    """Mirror of ``Classical_Discretization`` (01_Classical_DIscretization.jl:15-33)."""

    dim: int
    shape: str
    itp_type: str
    itp_order: int
    max_sd_order: int
    itg_order: int
    itp_func_num: int
    itg_func_num: int
    bdy_itg_func_num: int
    itp_funcs: List[Poly]
    itp_pos: np.ndarray  # [itp, dim] reference coordinates of the nodes
    itg_pos: np.ndarray
    itg_weight: np.ndarray  # [itg]
    ref_itp_vals: np.ndarray  # [itg, itp, (sd+1,)*dim]
    bdy_itg_pos: List[np.ndarray] = field(default_factory=list)
    bdy_itg_weights: List[np.ndarray] = field(default_factory=list)
    bdy_tangent_directions: List[np.ndarray] = field(default_factory=list)
    bdy_ref_itp_vals: List[np.ndarray] = field(default_factory=list)


def initialize_classical_element(dim: int, shape: str, itp_order: int, max_sd_order: int, itg_order: int,
                                 itp_type: str = "Lagrange") -> ClassicalDiscretization:
    """01_Classical_DIscretization.jl:35-81 (CUBE; SIMPLEX lives in simplex.py)."""
    if shape == "CUBE":
        if itp_type == "Lagrange":
            funcs, pos = interpolation_cube_lagrange(itp_order, dim)
        elif itp_type == "Serendipity":
            funcs, pos = interpolation_cube_serendipity(itp_order, dim)
        else:
            raise ValueError(itp_type)
        itg_pos, itg_w = domain_integration_cube(itg_order, dim)
        bpos, bw, btan = boundary_integration_cube(itg_order, dim)
    elif shape == "SIMPLEX":
        from . import simplex

        funcs, pos = simplex.interpolation_simplex_lagrange(itp_order, dim)
        itg_pos, itg_w = simplex.domain_integration_simplex(itg_order, dim)
        bpos, bw, btan = simplex.boundary_integration_simplex(itg_order, dim)
    else:
        raise ValueError(shape)
    ref = evaluate_itp_funcs(funcs, max_sd_order, itg_pos)
    bref = [evaluate_itp_funcs(funcs, max_sd_order, p) for p in bpos]
    return ClassicalDiscretization(
        dim=dim, shape=shape, itp_type=itp_type, itp_order=itp_order, max_sd_order=max_sd_order,
        itg_order=itg_order, itp_func_num=len(funcs), itg_func_num=len(itg_w), bdy_itg_func_num=len(bw[0]),
        itp_funcs=funcs, itp_pos=pos, itg_pos=itg_pos, itg_weight=itg_w, ref_itp_vals=ref,
        bdy_itg_pos=bpos, bdy_itg_weights=bw, bdy_tangent_directions=btan, bdy_ref_itp_vals=bref)


def sd_index(dim: int, sd_ids=()) -> Tuple[int, ...]:
    """``sd_ids_To_sd_IDs`` (4_Update_Integrator.jl:124), 0-based orders.

    ``sd_ids`` = repeated 0-based dimension numbers, e.g. (0, 0, 2) -> orders (2, 0, 1).
    """
    return tuple(sum(1 for s in sd_ids if s == i) for i in range(dim))
