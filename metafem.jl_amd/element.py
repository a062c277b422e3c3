"""Reference-element tables of the product (host, one-off): what `mesh_Classical` computes before anything touches the
device (reference src/mesh/spatial_discretization: 102_Interpolations.jl:3-39,69-110 shape functions,
103_Integrations.jl:1-58 Gauss tables / boundary quadrature / reference tangents, 01_Classical_DIscretization.jl:83-98
evaluate_Itp_Funcs).  The reference builds the shape functions with a polynomial algebra; every cube basis it supports
is a product of linear factors, so here a basis function is stored as (scale, [(alpha, beta), ...]) meaning
scale * prod_k (alpha_k . x + beta_k), from which values and first derivatives follow by the product rule.

SIMPLEX elements (triangle / tetrahedron, `init_Interpolation_Simplex_Lagrange` 102_Interpolations.jl:46-62, Gauss rules
103_Integrations.jl:60-241) use the same representation: every Lagrange simplex basis is a product of barycentric
linear factors.

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
    if not 1 <= g <= len(_GP):
        raise ValueError(f"itg_order {itg_order} needs {g} Gauss points; tables hold 1..{len(_GP)} (103_Integrations.jl:3-9)")
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
    shape: str = "CUBE"

    @property
    def itp(self):
        return self.ref_itp_vals.shape[1]

    @property
    def itg(self):
        return self.ref_itp_vals.shape[0]

    @property
    def itg_b(self):
        return self.bdy_itg_weights.shape[1]


# ---- SIMPLEX (103_Integrations.jl:60-80): symmetric rules given by their orbit generators -------------------------
_TRI_POS = (((0.10128650732345633880098736191512383,), (0.47014206410511508977044120951344760,), ()),
            ((0.06308901449150222834033160287081916,), (0.24928674517091042129163855310701908,),
             (0.05314504984481694735324967163139815, 0.31035245103378440541660773395655215)),
            ((), (0.17056930775176020662229350149146450,), (0.05054722831703097545842355059659895,),
             (0.45929258829272315602881551449416932,),
             (0.26311282963463811342178578628464359, 0.00839477740995760533721383453929445)))
_TRI_W = ((0.12593918054482715259568394550018133, 0.13239415278850618073764938783315200, 9.0 / 40.0),
          (0.05084490637020681692093680910686898, 0.11678627572637936602528961138557944,
           0.08285107561837357519355345642044245),
          (0.14431560767778716825109111048906462, 0.10321737053471825028179155029212903,
           0.03245849762319808031092592834178060, 0.09509163426728462479389610438858432,
           0.02723031417443499426484469007390892))
_TET_POS = (((0.31088591926330060979734573376345783,), (0.09273525031089122640232391373703061,),
             (-0.04550370412564964949188052627933943,)),
            ((0.21460287125915202928883921938628499,), (0.04067395853461135311557944895641006,),
             (0.32233789014227551034399447076249213,),
             (0.06366100187501752529923552760572698, 0.60300566479164914136743113906093969)),
            ((0.03967542307038990126507132953938949,), (0.31448780069809631378416056269714830,),
             (0.10198669306270330000000000000000000,), (0.18420369694919151227594641734890918,),
             (-0.06343628775453989240514123870189827,),
             (0.02169016206772800480266248262493018, 0.71993192203946593588943495335273478),
             (0.20448008063679571424133557487274534, 0.58057719012880922417539817139062041)))
_TET_W = ((0.11268792571801585079918565233328633, 0.07349304311636194954371020548632750,
           0.04254602077708146643806942812025744),
          (0.03992275025816749209969062755747998, 0.01007721105532064294801323744593686,
           0.05535718154365472209515327785372602, 27.0 / 560.0),
          (0.00639714777990232132145142033517302, 0.04019044802096617248816115847981783,
           0.02430797550477032117486910877192260, 0.05485889241369744046692412399039144,
           0.03571961223409918246495096899661762, 0.00718319069785253940945110521980376,
           0.01637218194531911754093813975611913))
SIMPLEX_FACE_VERTS = {2: ((0, 1), (1, 2), (2, 0)), 3: ((0, 1, 2), (0, 1, 3), (3, 1, 2), (0, 2, 3))}


def _simplex_bary(itg_order: int, nv: int):
    """Barycentric points (origin weight first) + weights summing to 1; orbits expanded in the reference's order."""
    rid = 0 if itg_order <= 5 else 1 if itg_order <= 6 else 2 if itg_order <= 8 else None
    if rid is None:
        raise ValueError("Wrong integral order")
    gens, ws = (_TRI_POS, _TRI_W) if nv == 3 else (_TET_POS, _TET_W)
    pts, wts = [], []
    for g, w in zip(gens[rid], ws[rid]):
        orbit = []
        if len(g) == 0:
            orbit.append((1.0 / nv,) * nv)
        elif len(g) == 1 and g[0] >= 0:
            a = g[0]
            orbit += [tuple(1 - (nv - 1) * a if k == i else a for k in range(nv)) for i in range(nv)]
        elif len(g) == 1:  # tetrahedron only: two entries 1/2 - b, two entries b
            b = -g[0]
            for i, j in _first_fastest([range(4)] * 2):
                if i < j:
                    orbit.append(tuple(0.5 - b if k in (i, j) else b for k in range(4)))
        elif nv == 3:
            src = (g[0], g[1], 1.0 - g[0] - g[1])
            orbit += [(src[i], src[j], src[3 - i - j]) for i, j in _first_fastest([range(3)] * 2) if i != j]
        else:
            a, b = g
            c = 1 - 2 * a - b
            for i, j in _first_fastest([range(4)] * 2):
                if i != j:
                    orbit.append(tuple(b if k == i else c if k == j else a for k in range(4)))
        pts += orbit
        wts += [w] * len(orbit)
    return np.array(pts, dtype=np.float64), np.array(wts, dtype=np.float64)


def gauss_simplex(itg_order: int, dim: int):
    b, w = _simplex_bary(itg_order, dim + 1)
    return b[:, 1:].copy(), w / math.factorial(dim)


def lagrange_simplex(order: int, dim: int):
    lam = [(_axis(dim, i), 0.0) for i in range(dim)] + [(-np.ones(dim), 1.0)]  # barycentric coordinates, origin last
    funcs, pos = [], []
    for ip in _first_fastest([range(order + 1)] * dim):
        rest = order - sum(ip)
        if rest < 0:
            continue
        fac, scale = [], 1.0
        for (al, be), m in zip(lam, list(ip) + [rest]):
            for k in range(m):  # Phi_m(l) = prod_{k<m} (p l - k) / (m - k)
                fac.append((order * al, order * be - k))
                scale /= (m - k)
        funcs.append(Basis(scale, fac))
        pos.append([i / order for i in ip])
    return funcs, np.array(pos)


def _simplex_boundary(itg_order: int, dim: int):
    V = np.vstack([np.zeros(dim), np.eye(dim)])
    if dim == 2:
        p1, w1 = gauss_cube(itg_order, 1)
        a = p1[:, 0]
        pos = [np.outer(1 - a, V[s]) + np.outer(a, V[e]) for s, e in SIMPLEX_FACE_VERTS[2]]
        ws = [w1.copy(), w1 * math.sqrt(2.0), w1.copy()]
        tans = [np.array([[1.0], [0.0]]), np.array([[-1.0], [1.0]]) / math.sqrt(2.0), np.array([[0.0], [-1.0]])]
    else:
        b, w = _simplex_bary(itg_order, 3)
        pos = [b[:, [0]] * V[c0] + b[:, [1]] * V[c1] + b[:, [2]] * V[c2] for c0, c1, c2 in SIMPLEX_FACE_VERTS[3]]
        ws = [w * 0.5, w * 0.5, w * 0.5 * math.sqrt(3.0), w * 0.5]
        r2, r6 = math.sqrt(2.0), math.sqrt(6.0)
        tans = [np.array([[-1.0, 0.0], [0.0, 1.0], [0.0, 0.0]]), np.array([[0.0, 1.0], [0.0, 0.0], [-1.0, 0.0]]),
                np.array([[-1.0 / r2, -1.0 / r6], [1.0 / r2, -1.0 / r6], [0.0, 2.0 / r6]]),
                np.array([[0.0, 0.0], [-1.0, 0.0], [0.0, 1.0]])]
    return pos, ws, tans


def _simplex_space(dim, itp_type, itp_order, itg_order) -> ClassicalSpace:
    funcs, pos = lagrange_simplex(itp_order, dim)  # :Serendipity on a SIMPLEX is the same basis (01_Classical_DIscretization.jl:72)
    qp, qw = gauss_simplex(itg_order, dim)
    bpos, bws, tans = _simplex_boundary(itg_order, dim)
    nq = bpos[0].shape[0]
    bref = np.stack([_tables(funcs, p, dim) for p in bpos])
    btan = np.stack([np.tile(t[None], (nq, 1, 1)) for t in tans])
    return ClassicalSpace(dim, itp_type, itp_order, itg_order, pos, qw, _tables(funcs, qp, dim), np.stack(bws), bref, btan, funcs,
                          shape="SIMPLEX")


def classical_space(dim: int, itp_type: str = "Lagrange", itp_order: int = 1, itg_order: int = 3,
                    shape: str = "CUBE") -> ClassicalSpace:
    """initialize_Classical_Element(dim, shape, itp_order, max_sd_order = 1, itg_order; itp_type)."""
    if shape == "SIMPLEX":
        return _simplex_space(dim, itp_type, itp_order, itg_order)
    if shape != "CUBE":
        raise ValueError(shape)
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
