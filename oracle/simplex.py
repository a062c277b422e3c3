"""SIMPLEX reference tables (oracle; test infrastructure only).

Restates
  init_Interpolation_Simplex_Lagrange                 spatial_discretization/102_Interpolations.jl:46-62
  GAUSS_POINT_{POS,WEIGHT}_{TRIANGLE,TETRAHEDRON}      spatial_discretization/103_Integrations.jl:60-80
  _init_Integration_Triangle_Gauss / Tetrahedron       :82-118, :149-201
  init_Domain_Integration_{Triangle,Tetrahedron}_Gauss :120-123, :203-206
  init_Boundary_Integration_{Triangle,Tetrahedron}_Gauss :125-147, :208-241
The reference's `itp_type = :Serendipity` on a SIMPLEX is the same Lagrange basis (01_Classical_DIscretization.jl:60-73).

Reference simplex: vertices (0,0[,0]), (1,0[,0]), (0,1[,0])[, (0,0,1)]; barycentric tuples are stored (l1, l2, l3[, l4])
with l1 the weight of the ORIGIN vertex and the cartesian position = (l2, l3[, l4]).
Local faces: triangle 1: v1-v2 (y = 0), 2: v2-v3 (slanted), 3: v3-v1 (x = 0);
tetrahedron 1: v1 v2 v3 (z = 0), 2: v1 v2 v4 (y = 0), 3: v4 v2 v3 (slanted), 4: v1 v3 v4 (x = 0)
(ref_geometry/002_Initialization.jl:1-8).
"""
from __future__ import annotations

import itertools
import math
from typing import List

import numpy as np

from .polynomial import Poly, prod
from .reference_element import _product_first_fastest, domain_integration_cube, gauss_order_of, lagrange_1d

# 103_Integrations.jl:60-66 -- (a,) = the 3 points (1-2a, a, a) and rotations; (a, b) = all 6 orderings of (a, b, 1-a-b)
TRI_POS = (((0.10128650732345633880098736191512383,), (0.47014206410511508977044120951344760,), ()),
           ((0.06308901449150222834033160287081916,), (0.24928674517091042129163855310701908,),
            (0.05314504984481694735324967163139815, 0.31035245103378440541660773395655215)),
           ((), (0.17056930775176020662229350149146450,), (0.05054722831703097545842355059659895,),
            (0.45929258829272315602881551449416932,),
            (0.26311282963463811342178578628464359, 0.00839477740995760533721383453929445)))
TRI_W = ((0.12593918054482715259568394550018133, 0.13239415278850618073764938783315200, 9.0 / 40.0),
         (0.05084490637020681692093680910686898, 0.11678627572637936602528961138557944,
          0.08285107561837357519355345642044245),
         (0.14431560767778716825109111048906462, 0.10321737053471825028179155029212903,
          0.03245849762319808031092592834178060, 0.09509163426728462479389610438858432,
          0.02723031417443499426484469007390892))
# :69-80 -- (a,) a >= 0: (1-3a, a, a, a) and rotations; a < 0: the 6 edge-midpoint-like points with b = -a;
# (a, b): 12 points with two entries a, one b, one 1-2a-b
TET_POS = (((0.31088591926330060979734573376345783,), (0.09273525031089122640232391373703061,),
            (-0.04550370412564964949188052627933943,)),
           ((0.21460287125915202928883921938628499,), (0.04067395853461135311557944895641006,),
            (0.32233789014227551034399447076249213,),
            (0.06366100187501752529923552760572698, 0.60300566479164914136743113906093969)),
           ((0.03967542307038990126507132953938949,), (0.31448780069809631378416056269714830,),
            (0.10198669306270330000000000000000000,), (0.18420369694919151227594641734890918,),
            (-0.06343628775453989240514123870189827,),
            (0.02169016206772800480266248262493018, 0.71993192203946593588943495335273478),
            (0.20448008063679571424133557487274534, 0.58057719012880922417539817139062041)))
TET_W = ((0.11268792571801585079918565233328633, 0.07349304311636194954371020548632750,
          0.04254602077708146643806942812025744),
         (0.03992275025816749209969062755747998, 0.01007721105532064294801323744593686,
          0.05535718154365472209515327785372602, 27.0 / 560.0),
         (0.00639714777990232132145142033517302, 0.04019044802096617248816115847981783,
          0.02430797550477032117486910877192260, 0.05485889241369744046692412399039144,
          0.03571961223409918246495096899661762, 0.00718319069785253940945110521980376,
          0.01637218194531911754093813975611913))


def _rule_id(itg_order: int) -> int:
    if itg_order <= 5:
        return 0
    if itg_order <= 6:
        return 1
    if itg_order <= 8:
        return 2
    raise ValueError("Wrong integral order")


def _basis_tup(i: int, n: int, wi: float, wo: float):
    return tuple(wi if k == i else wo for k in range(n))


def _triangle_bary(itg_order: int):
    """:82-118 -> barycentric triples and weights (sum of weights = 1)."""
    rid = _rule_id(itg_order)
    pts, ws = [], []
    for pos, w in zip(TRI_POS[rid], TRI_W[rid]):
        if len(pos) == 0:
            pts.append((1 / 3,) * 3)
            ws.append(w)
        elif len(pos) == 1:
            a = pos[0]
            for i in range(3):
                pts.append(_basis_tup(i, 3, 1 - 2 * a, a))
                ws.append(w)
        else:
            src = (pos[0], pos[1], 1.0 - sum(pos))
            for i, j in _product_first_fastest([range(3)] * 2):
                if i == j:
                    continue
                k = 3 - i - j
                pts.append((src[i], src[j], src[k]))
                ws.append(w)
    return np.array(pts, dtype=np.float64), np.array(ws, dtype=np.float64)


def _tetrahedron_bary(itg_order: int):
    """:149-201."""
    rid = _rule_id(itg_order)
    pts, ws = [], []
    for pos, w in zip(TET_POS[rid], TET_W[rid]):
        if len(pos) == 0:
            pts.append((0.25,) * 4)
            ws.append(w)
        elif len(pos) == 1:
            a = pos[0]
            if a >= 0:
                for i in range(4):
                    pts.append(_basis_tup(i, 4, 1 - 3 * a, a))
                    ws.append(w)
            else:
                b = -a
                for i, j in _product_first_fastest([range(4)] * 2):
                    if i >= j:
                        continue
                    src = [b] * 4
                    src[i] = src[j] = 0.5 - b
                    pts.append(tuple(src))
                    ws.append(w)
        elif len(pos) == 2:
            a, b = pos
            c = 1 - 2 * a - b
            for i, j in _product_first_fastest([range(4)] * 2):
                if i == j:
                    continue
                src = [a] * 4
                src[i] = b
                src[j] = c
                pts.append(tuple(src))
                ws.append(w)
        else:
            src = tuple(pos) + (1.0 - sum(pos),)
            for i, j, k in _product_first_fastest([range(4)] * 3):
                if i == j or i == k or j == k:
                    continue
                l = 6 - i - j - k
                pts.append((src[i], src[j], src[k], src[l]))
                ws.append(w)
    return np.array(pts, dtype=np.float64), np.array(ws, dtype=np.float64)


def domain_integration_simplex(itg_order: int, dim: int):
    """:120-123 / :203-206 -> (itg_pos[itg, dim], itg_weight[itg]); weights sum to the simplex volume 1/dim!."""
    if dim == 2:
        b, w = _triangle_bary(itg_order)
        return b[:, 1:3].copy(), w / 2.0
    if dim == 3:
        b, w = _tetrahedron_bary(itg_order)
        return b[:, 1:4].copy(), w / 6.0
    raise ValueError("Wrong dimension")


def boundary_integration_simplex(itg_order: int, dim: int):
    """:125-147 / :208-241 -> per local face: positions, weights, unit reference tangents [itg_b, dim, dim-1]."""
    if dim == 2:
        p1, w1 = domain_integration_cube(itg_order, 1)
        a = p1[:, 0]
        V = np.array([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0]])
        ends = [(0, 1), (1, 2), (2, 0)]
        pos = [np.outer(1 - a, V[s]) + np.outer(a, V[e]) for s, e in ends]
        ws = [w1.copy(), w1 * math.sqrt(2.0), w1.copy()]
        tans = [np.array([1.0, 0.0]), np.array([-1.0, 1.0]) / math.sqrt(2.0), np.array([0.0, -1.0])]
        tan = [np.tile(t.reshape(1, 2, 1), (len(a), 1, 1)) for t in tans]
        return pos, ws, tan
    if dim == 3:
        b, w = _triangle_bary(itg_order)
        V = np.array([[0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]])
        corners = [(0, 1, 2), (0, 1, 3), (3, 1, 2), (0, 2, 3)]
        pos = [b[:, [0]] * V[c0] + b[:, [1]] * V[c1] + b[:, [2]] * V[c2] for c0, c1, c2 in corners]
        ws = [w * 0.5 for _ in range(4)]
        ws[2] = ws[2] * math.sqrt(3.0)
        t = [(np.array([-1.0, 0.0, 0.0]), np.array([0.0, 1.0, 0.0])),
             (np.array([0.0, 0.0, -1.0]), np.array([1.0, 0.0, 0.0])),
             (np.array([-1.0, 1.0, 0.0]) / math.sqrt(2.0), np.array([-1.0, -1.0, 2.0]) / math.sqrt(6.0)),
             (np.array([0.0, -1.0, 0.0]), np.array([0.0, 0.0, 1.0]))]
        tan = [np.tile(np.stack([t1, t2], axis=1).reshape(1, 3, 2), (len(w), 1, 1)) for t1, t2 in t]
        return pos, ws, tan
    raise ValueError("Wrong dimension")


def _substitute_1d(p1: Poly, target: Poly) -> Poly:
    """substitute_Polynomial(p, 1, target) for a 1-D polynomial p (03_Polynomial.jl:116-125)."""
    dim = target.dim
    out = Poly(dim, {})
    for (k,), c in p1.terms.items():
        term = Poly.const(dim, c)
        for _ in range(k):
            term = term * target
        out = out + term
    return out


def interpolation_simplex_lagrange(order: int, dim: int):
    """:46-62 -> (funcs, ref positions [itp, dim]); lattice enumeration with the first coordinate fastest."""
    f1: List[Poly] = [Poly.const(1, 1.0)]
    for m in range(1, order + 1):
        last = lagrange_1d(m)[-1]
        f1.append(_substitute_1d(last, Poly(1, {(1,): order / m})))
    vol = [Poly.var(dim, i) for i in range(dim)]
    lastc = Poly.const(dim, 1.0)
    for i in range(dim):
        lastc = lastc - Poly.var(dim, i)
    vol.append(lastc)
    tmpl = [[_substitute_1d(f, vol[i]) for f in f1] for i in range(dim + 1)]
    funcs, pos = [], []
    for ip in _product_first_fastest([range(order + 1)] * dim):
        rest = order - sum(ip)
        if rest < 0:
            continue
        funcs.append(prod([tmpl[i][ip[i]] for i in range(dim)]) * tmpl[dim][rest])
        pos.append([ip[i] / order for i in range(dim)])
    return funcs, np.array(pos, dtype=np.float64)
