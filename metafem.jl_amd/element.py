"""Reference-element tables of the product (host, one-off): what `mesh_Classical` computes before anything touches the
device (reference src/mesh/spatial_discretization: 102_Interpolations.jl:3-39,69-110 shape functions,
103_Integrations.jl:1-58 Gauss tables / boundary quadrature / reference tangents, 01_Classical_DIscretization.jl:83-98
evaluate_Itp_Funcs).  The reference builds the shape functions with a polynomial algebra; every cube basis it supports
is a product of linear factors, so here a basis function is stored as (scale, [(alpha, beta), ...]) meaning
scale * prod_k (alpha_k . x + beta_k), from which values and first derivatives follow by the product rule.

Conventions: reference cell [0,1]^dim; Gauss points/weights shifted to [0,1]; tensor orders with the
FIRST coordinate fastest; Lagrange nodes in tensor order; serendipity nodes = corners (tensor order) then mid-edge
nodes by edge direction; local face ids 2-D (4 2; 1 3), 3-D (5 3; 2 4; 1 6) = (low, high) per normal dimension.
"""
from __future__ import annotations

import itertools
import math
from dataclasses import dataclass, field
from typing import List, Tuple

import numpy as np

_GP = ((0.0,), (-1.0 / math.sqrt(3.0), 1.0 / math.sqrt(3.0)), (-math.sqrt(3.0 / 5.0), 0.0, math.sqrt(3.0 / 5.0)),
       (-math.sqrt(3.0 / 7.0 + 2.0 / 7.0 * math.sqrt(6.0 / 5.0)), -math.sqrt(3.0 / 7.0 - 2.0 / 7.0 * math.sqrt(6.0 / 5.0)),
        math.sqrt(3.0 / 7.0 - 2.0 / 7.0 * math.sqrt(6.0 / 5.0)), math.sqrt(3.0 / 7.0 + 2.0 / 7.0 * math.sqrt(6.0 / 5.0))))
_GW = ((2.0,), (1.0, 1.0), (5.0 / 9.0, 8.0 / 9.0, 5.0 / 9.0),
       ((18.0 - math.sqrt(30.0)) / 36.0, (18.0 + math.sqrt(30.0)) / 36.0, (18.0 + math.sqrt(30.0)) / 36.0,
        (18.0 - math.sqrt(30.0)) / 36.0))
CUBE_FACE_IDS = {2: ((4, 2), (1, 3)), 3: ((5, 3), (2, 4), (1, 6))}


def _first_fastest(ranges):
    for rev in itertools.product(*reversed(list(ranges))):
        yield tuple(reversed(rev))


def gauss_cube(itg_order: int, dim: int):
    g = int(math.ceil((itg_order + 1) / 2))
    if dim == 0:
        return np.zeros((1, 0)), np.ones(1)
    P = [x / 2.0 + 0.5 for x in _GP[g - 1]]
    W = [w / 2.0 for w in _GW[g - 1]]
    pos = [[P[i] for i in ids] for ids in _first_fastest([range(g)] * dim)]
    w = [math.prod(W[i] for i in ids) for ids in _first_fastest([range(g)] * dim)]
    return np.array(pos, dtype=np.float64).reshape(-1, dim), np.array(w, dtype=np.float64)


Factor = Tuple[np.ndarray, float]


@dataclass
class Basis:
    scale: float
    factors: List[Factor]

    def value(self, x) -> float:
        v = self.scale
        for a, b in self.factors:
            v *= float(a @ x) + b
        return v

    def grad(self, x) -> np.ndarray:
        vals = [float(a @ x) + b for a, b in self.factors]
        g = np.zeros(len(x))
        for k, (a, _) in enumerate(self.factors):
            p = self.scale
            for l, v in enumerate(vals):
                if l != k:
                    p *= v
            g += p * a
        return g


def _axis(dim, i):
    e = np.zeros(dim)
    e[i] = 1.0
    return e


def lagrange_cube(order: int, dim: int):
    nodes = [i / order for i in range(order + 1)]
    funcs, pos = [], []
    for ids in _first_fastest([range(order + 1)] * dim):
        fac, scale = [], 1.0
        for d, a in enumerate(ids):
            for k in range(order + 1):
                if k != a:
                    fac.append((_axis(dim, d), -nodes[k]))
                    scale /= (nodes[a] - nodes[k])
        funcs.append(Basis(scale, fac))
        pos.append([nodes[a] for a in ids])
    return funcs, np.array(pos)


def serendipity_cube(order: int, dim: int):
    if order != 2:
        raise NotImplementedError("serendipity order 2 (quad-8 / hex-20) is what the reference's examples use")
    funcs, pos = [], []
    for c in _first_fastest([range(2)] * dim):
        fac = [(-_axis(dim, i), float(1 - c[i])) for i in range(dim)]  # (1 - c_i) - x_i
        s = np.array([1.0 - 2.0 * ci for ci in c])
        fac.append((-s, float(s @ np.array(c, dtype=float)) + 0.5))  # s.c + 1/2 - s.x
        b = Basis(1.0, fac)
        b.scale = 1.0 / b.value(np.array(c, dtype=float))
        funcs.append(b)
        pos.append([float(v) for v in c])
    for ed in range(dim):
        minor = [i for i in range(dim) if i != ed]
        for mc in _first_fastest([range(2)] * (dim - 1)):
            fac = [(_axis(dim, ed), 0.0), (_axis(dim, ed), -1.0)]  # x_ed (x_ed - 1): zero at both ends, node at 1/2
            for m, cm in zip(minor, mc):
                fac.append((-_axis(dim, m), float(1 - cm)))
            coor = np.full(dim, 0.5)
            for m, cm in zip(minor, mc):
                coor[m] = float(cm)
            b = Basis(1.0, fac)
            b.scale = 1.0 / b.value(coor)
            funcs.append(b)
            pos.append(list(coor))
    return funcs, np.array(pos)


def _tables(funcs, pts, dim):
    """ref[q, a, s]: s = 0 value, 1 + m = d/dxi_m  (the max_sd_order = 1 slice of ref_itp_vals)."""
    out = np.zeros((len(pts), len(funcs), 1 + dim))
    for q, x in enumerate(pts):
        for a, f in enumerate(funcs):
            out[q, a, 0] = f.value(x)
            out[q, a, 1:] = f.grad(x)
    return out


@dataclass
class ClassicalSpace:
    """Mirror of Classical_Discretization (01_Classical_DIscretization.jl:15-33), first-derivative tables only."""
    dim: int
    itp_type: str
    itp_order: int
    itg_order: int
    itp_pos: np.ndarray
    itg_weight: np.ndarray
    ref_itp_vals: np.ndarray  # [itg, itp, 1+dim]
    bdy_itg_weights: np.ndarray  # [nface, itg_b]
    bdy_ref_itp_vals: np.ndarray  # [nface, itg_b, itp, 1+dim]
    bdy_tangent_directions: np.ndarray  # [nface, itg_b, dim, dim-1]
    funcs: list = field(default_factory=list, repr=False)

    @property
    def itp(self):
        return self.ref_itp_vals.shape[1]

    @property
    def itg(self):
        return self.ref_itp_vals.shape[0]

    @property
    def itg_b(self):
        return self.bdy_itg_weights.shape[1]


def classical_space(dim: int, itp_type: str = "Lagrange", itp_order: int = 1, itg_order: int = 3) -> ClassicalSpace:
    """initialize_Classical_Element(dim, :CUBE, itp_order, max_sd_order = 1, itg_order; itp_type)."""
    if itp_type == "Lagrange":
        funcs, pos = lagrange_cube(itp_order, dim)
    elif itp_type == "Serendipity":
        funcs, pos = serendipity_cube(itp_order, dim)
    else:
        raise ValueError(itp_type)
    qp, qw = gauss_cube(itg_order, dim)
    ref = _tables(funcs, qp, dim)
    fp, fw = gauss_cube(itg_order, dim - 1)
    nq, nface = fp.shape[0], 2 * dim
    bref = np.zeros((nface, nq, len(funcs), 1 + dim))
    btan = np.zeros((nface, nq, dim, dim - 1))
    bw = np.tile(fw, (nface, 1))
    for nd in range(1, dim + 1):
        tdim = [(i + nd - 1) % dim + 1 for i in range(1, dim)]  # 103_Integrations.jl:37
        for outward in (0, 1):
            f = CUBE_FACE_IDS[dim][nd - 1][outward] - 1
            for i, td in enumerate(tdim):
                btan[f, :, td - 1, i] = 1.0
            if dim == 2:
                if (outward + nd) != 2:
                    btan[f] *= -1.0
            elif outward == 0:
                btan[f, :, :, 0] *= -1.0
            pts = np.zeros((nq, dim))
            for i, td in enumerate(tdim):
                pts[:, td - 1] = fp[:, i]
            pts[:, nd - 1] = outward
            bref[f] = _tables(funcs, pts, dim)
    return ClassicalSpace(dim, itp_type, itp_order, itg_order, pos, qw, ref, bw, bref, btan, funcs)
