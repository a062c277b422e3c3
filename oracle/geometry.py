"""Per-element / per-facet geometry update (oracle; test infrastructure only).

Restates src/mesh/unstructured_mesh/4_Update_Integrator.jl:
  update_BasicElements_{2,3}D :2-33, inv_Jac_2D :77-88, inv_Jac_3D :90-121,
  update_Basic_itgval_1_{2,3}D :125-154 (first-order push-forward),
  update_BasicBoundary_{2,3}D :35-75, tangents :163-196, normals :198-227.

Physical basis tables are stored as ``integral_vals[q, a, s, e]`` with s = 0 the
value and s = 1 + d the derivative d/dx_d  (the reference stores the full
(sd+1)^dim hyper-cube; with max_sd_order = 1 only these dim+1 slots are ever written
or read -- 05_CodeGenerator.jl:6,66,105,127).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .mesh import ClassicalMesh, Facets
from .reference_element import ClassicalDiscretization


def _first_derivs(ref: np.ndarray, dim: int) -> np.ndarray:
    """ref[q, a, o1..od] -> [dim(X), q, a] with X the differentiated reference dim."""
    out = []
    for X in range(dim):
        idx = [0] * dim
        idx[X] = 1
        out.append(ref[(slice(None), slice(None)) + tuple(idx)])
    return np.stack(out)


def _value(ref: np.ndarray, dim: int) -> np.ndarray:
    return ref[(slice(None), slice(None)) + (0,) * dim]


def inv_jac(J: np.ndarray):
    """J[i, X, q, e] -> det[q, e], Jinv[m, s, q, e]; explicit adjugate formulas of
    inv_Jac_2D/3D (4_Update_Integrator.jl:77-121), same operation order."""
    dim = J.shape[0]
    if dim == 2:
        det = J[0, 0] * J[1, 1] - J[0, 1] * J[1, 0]
        inv = np.empty_like(J)
        inv[0, 0] = J[1, 1] / det
        inv[0, 1] = -J[0, 1] / det
        inv[1, 0] = -J[1, 0] / det
        inv[1, 1] = J[0, 0] / det
        return det, inv
    j = J
    det = (j[0, 0] * j[1, 1] * j[2, 2] - j[0, 0] * j[1, 2] * j[2, 1] - j[0, 1] * j[1, 0] * j[2, 2]
           + j[0, 1] * j[1, 2] * j[2, 0] + j[0, 2] * j[1, 0] * j[2, 1] - j[0, 2] * j[1, 1] * j[2, 0])
    inv = np.empty_like(J)
    inv[0, 0] = (j[1, 1] * j[2, 2] - j[1, 2] * j[2, 1]) / det
    inv[0, 1] = (j[0, 2] * j[2, 1] - j[0, 1] * j[2, 2]) / det
    inv[0, 2] = (j[0, 1] * j[1, 2] - j[1, 1] * j[0, 2]) / det
    inv[1, 0] = (j[1, 2] * j[2, 0] - j[2, 2] * j[1, 0]) / det
    inv[1, 1] = (j[0, 0] * j[2, 2] - j[0, 2] * j[2, 0]) / det
    inv[1, 2] = (j[0, 2] * j[1, 0] - j[0, 0] * j[1, 2]) / det
    inv[2, 0] = (j[1, 0] * j[2, 1] - j[1, 1] * j[2, 0]) / det
    inv[2, 1] = (j[0, 1] * j[2, 0] - j[2, 1] * j[0, 0]) / det
    inv[2, 2] = (j[0, 0] * j[1, 1] - j[1, 0] * j[0, 1]) / det
    return det, inv


def _push_forward(ref: np.ndarray, Jinv: np.ndarray, dim: int) -> np.ndarray:
    """integral_vals[q, a, s, e]; dN_a/dx_s = sum_m dN_a/dxi_m * Jinv[m, s] (:133-142)."""
    nq, na = ref.shape[:2]
    ne = Jinv.shape[-1]
    d = _first_derivs(ref, dim)  # [m, q, a]
    vals = np.empty((nq, na, dim + 1, ne))
    vals[:, :, 0, :] = _value(ref, dim)[:, :, None]
    for s in range(dim):
        acc = np.zeros((nq, na, ne))
        for m in range(dim):
            acc += d[m][:, :, None] * Jinv[m, s][:, None, :]
        vals[:, :, 1 + s, :] = acc
    return vals


@dataclass
class ElementGeometry:
    jacobian: np.ndarray  # [i, X, q, e]
    inverse_jacobian: np.ndarray  # [m, s, q, e]
    dets: np.ndarray  # [q, e]
    integral_vals: np.ndarray  # [q, a, 1+dim, e]
    integral_weights: np.ndarray  # [q, e]


def update_basic_elements(mesh: ClassicalMesh, disc: ClassicalDiscretization) -> ElementGeometry:
    """update_BasicElements_{2,3}D (4_Update_Integrator.jl:2-33)."""
    dim = disc.dim
    X = mesh.coords[mesh.cp_ids]  # [a, e, i]
    d = _first_derivs(disc.ref_itp_vals, dim)  # [X, q, a]
    J = np.einsum("Xqa,aei->iXqe", d, X)
    det, Jinv = inv_jac(J)
    vals = _push_forward(disc.ref_itp_vals, Jinv, dim)
    w = disc.itg_weight[:, None] * det
    return ElementGeometry(J, Jinv, det, vals, w)


@dataclass
class FacetGeometry:
    integral_vals: np.ndarray  # [q_b, a, 1+dim, f]  (all itp nodes of the HOST element)
    integral_weights: np.ndarray  # [q_b, f]
    normal_directions: np.ndarray  # [q_b, dim, f]
    tangent_directions: np.ndarray  # [q_b, dim, dim-1, f]
    bdy_dets: np.ndarray  # [q_b, f]


def update_basic_boundary(mesh: ClassicalMesh, disc: ClassicalDiscretization, facets: Facets) -> FacetGeometry:
    """update_BasicBoundary_{2,3}D (4_Update_Integrator.jl:35-75) for the given facets."""
    dim = disc.dim
    nf = len(facets)
    nqb, na = disc.bdy_itg_func_num, disc.itp_func_num
    vals = np.zeros((nqb, na, dim + 1, nf))
    wts = np.zeros((nqb, nf))
    nrm = np.zeros((nqb, dim, nf))
    tan = np.zeros((nqb, dim, dim - 1, nf))
    bdet = np.zeros((nqb, nf))
    for eindex in range(2 * dim if disc.shape == "CUBE" else dim + 1):
        sel = np.nonzero(facets.element_eindex == eindex)[0]
        if sel.size == 0:
            continue
        ref = disc.bdy_ref_itp_vals[eindex]
        X = mesh.coords[mesh.cp_ids[:, facets.element_ID[sel]]]  # [a, f, i]
        d = _first_derivs(ref, dim)
        J = np.einsum("Xqa,afi->iXqf", d, X)
        _, Jinv = inv_jac(J)
        bt = disc.bdy_tangent_directions[eindex]  # [q, X, t]
        t = np.einsum("iXqf,qXt->qitf", J, bt)  # update_Basic_Tangent (:163-196)
        if dim == 2:
            t1, t2 = t[:, 0, 0, :], t[:, 1, 0, :]
            ld = np.sqrt(t1 ** 2.0 + t2 ** 2.0)
            n = np.stack([t2 / ld, -t1 / ld], axis=1)  # :198-208
        else:
            t11, t21, t31 = t[:, 0, 0, :], t[:, 1, 0, :], t[:, 2, 0, :]
            t12, t22, t32 = t[:, 0, 1, :], t[:, 1, 1, :], t[:, 2, 1, :]
            rn1 = t21 * t32 - t31 * t22
            rn2 = -t11 * t32 + t31 * t12
            rn3 = t11 * t22 - t21 * t12
            ld = np.sqrt(rn1 ** 2.0 + rn2 ** 2.0 + rn3 ** 2.0)
            n = np.stack([rn1 / ld, rn2 / ld, rn3 / ld], axis=1)  # :210-226
        vals[..., sel] = _push_forward(ref, Jinv, dim)
        wts[:, sel] = disc.bdy_itg_weights[eindex][:, None] * ld
        nrm[..., sel] = n
        tan[..., sel] = t
        bdet[:, sel] = ld
    return FacetGeometry(vals, wts, nrm, tan, bdet)
