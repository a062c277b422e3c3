"""Mesh generators and control-point layout (oracle; test infrastructure only).

Restates
  * make_Square / make_Brick (CUBE)  -- src/mesh/ref_geometry/201_Helper_TM.jl:7-51
  * the element-structure maps        -- src/mesh/spatial_discretization/101_Structures.jl:1-92,198-247
  * allocate_Basic_WP_Mesh_2D/3D      -- src/mesh/unstructured_mesh/3_InitializeMesh.jl:1-163
  * boundary-face detection           -- src/mesh/ref_geometry/002_Initialization.jl:277-289

All ids are 0-based here (the reference is 1-based).  The reference numbers
edge/face/interior control points in GPU-hash-table order (SURVEY.md §4 caveat),
so numbering beyond the vertices is not reproducible; this restatement numbers
them in first-appearance order and every comparison is done by coordinates.
"""
from __future__ import annotations

import itertools
from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np

from .reference_element import CUBE_FACE_IDS, ClassicalDiscretization

# 101_Structures.jl:4,36-37,202,228 (1-based in the reference): connectivity vertex j -> tensor basis id
VERTEX_CP_IDS = {2: (0, 1, 3, 2), 3: (0, 1, 3, 2, 4, 5, 7, 6)}


def simplex_split(cube_conn: np.ndarray, n) -> np.ndarray:
    """The :SIMPLEX branches of make_Square / make_Brick (201_Helper_TM.jl:20-29, 55-76), written as the loops they are."""
    ncell = cube_conn.shape[1]
    if cube_conn.shape[0] == 4:
        out = np.zeros((3, 2 * ncell), dtype=np.int64)
        for e in range(ncell):
            out[:, e] = cube_conn[[0, 1, 3], e]
            out[:, e + ncell] = cube_conn[[2, 3, 1], e]  # "OK but all minus" (:27)
        return out
    out = np.zeros((4, 5 * ncell), dtype=np.int64)
    fwd = [[1, 2, 4, 5], [3, 4, 2, 7], [8, 7, 5, 4], [6, 5, 7, 2], [4, 7, 5, 2]]
    bwd = [[5, 8, 6, 1], [2, 1, 6, 3], [7, 6, 8, 3], [4, 1, 3, 8], [1, 3, 8, 6]]
    e = 0
    for i in range(1, n[0] + 1):
        for j in range(1, n[1] + 1):
            for k in range(1, n[2] + 1):
                pick = fwd if (i + j + k) % 2 == 1 else bwd
                for t in range(5):
                    out[:, e + t * ncell] = cube_conn[[v - 1 for v in pick[t]], e]
                e += 1
    return out


def make_square(x: Tuple[float, float], n: Tuple[int, int]):
    """201_Helper_TM.jl:7-33 (CUBE).  -> coors[2, nv], connections[4, nel] (0-based)."""
    dx = (x[0] / n[0], x[1] / n[1])
    i, j = np.meshgrid(np.arange(n[0] + 1), np.arange(n[1] + 1), indexing="ij")
    coors = np.stack([dx[0] * i.ravel(), dx[1] * j.ravel()]).astype(np.float64)
    ie, je = np.meshgrid(np.arange(1, n[0] + 1), np.arange(1, n[1] + 1), indexing="ij")
    ie, je = ie.ravel(), je.ravel()
    s = n[1] + 1
    conn = np.stack([(ie - 1) * s + je, ie * s + je, ie * s + je + 1, (ie - 1) * s + je + 1]) - 1
    return coors, conn.astype(np.int64)


def make_brick(x: Tuple[float, float, float], n: Tuple[int, int, int]):
    """201_Helper_TM.jl:36-51 (CUBE).  Node id = i*(n2+1)*(n3+1) + j*(n3+1) + k, k fastest."""
    dx = tuple(x[d] / n[d] for d in range(3))
    i, j, k = np.meshgrid(np.arange(n[0] + 1), np.arange(n[1] + 1), np.arange(n[2] + 1), indexing="ij")
    coors = np.stack([dx[0] * i.ravel(), dx[1] * j.ravel(), dx[2] * k.ravel()]).astype(np.float64)
    ie, je, ke = np.meshgrid(np.arange(1, n[0] + 1), np.arange(1, n[1] + 1), np.arange(1, n[2] + 1), indexing="ij")
    ie, je, ke = ie.ravel(), je.ravel(), ke.ravel()
    s2, s3 = (n[1] + 1) * (n[2] + 1), (n[2] + 1)
    conn = np.stack([
        (ie - 1) * s2 + (je - 1) * s3 + ke,
        ie * s2 + (je - 1) * s3 + ke,
        ie * s2 + je * s3 + ke,
        (ie - 1) * s2 + je * s3 + ke,
        (ie - 1) * s2 + (je - 1) * s3 + ke + 1,
        ie * s2 + (je - 1) * s3 + ke + 1,
        ie * s2 + je * s3 + ke + 1,
        (ie - 1) * s2 + je * s3 + ke + 1,
    ]) - 1
    return coors, conn.astype(np.int64)


@dataclass
class ClassicalMesh:
    dim: int
    coords: np.ndarray  # [ncp, dim]
    cp_ids: np.ndarray  # [itp, nel]  == elements.controlpoint_IDs (basis order)
    vert_conn: np.ndarray  # [2^dim, nel] the first-order connectivity (counter-clockwise)
    n_vertices: int

    @property
    def ncp(self) -> int:
        return self.coords.shape[0]

    @property
    def nel(self) -> int:
        return self.cp_ids.shape[1]


def _corner_to_conn_index(dim: int) -> Dict[Tuple[int, ...], int]:
    inv = {}
    for j, t in enumerate(VERTEX_CP_IDS[dim]):
        c = tuple((t >> d) & 1 for d in range(dim))
        inv[c] = j
    return inv


def mesh_classical(vert: np.ndarray, conn: np.ndarray, disc: ClassicalDiscretization) -> ClassicalMesh:
    """Control points + ``controlpoint_IDs`` for a CUBE mesh (3_InitializeMesh.jl:1-163).

    Every basis node is attached to the lowest-dimensional entity (vertex / segment /
    face / block) containing its reference position; nodes on shared entities are
    shared between elements.  Positions are the multilinear images of the entity's
    vertices (``segment_cp_pos``/``face_cp_pos``/``block_cp_pos``).
    """
    dim = disc.dim
    if disc.shape == "SIMPLEX":
        return _mesh_classical_simplex(vert, conn, disc)
    assert disc.shape == "CUBE"
    nel = conn.shape[1]
    nv = vert.shape[1]
    corner_idx = _corner_to_conn_index(dim)
    itp = disc.itp_func_num
    cp_ids = np.zeros((itp, nel), dtype=np.int64)
    coords: List[np.ndarray] = [vert[:, v].copy() for v in range(nv)]
    table: Dict[tuple, int] = {}
    eps = 1e-12
    # per basis node: free dims, entity corner list and multilinear weights
    node_info = []
    for a in range(itp):
        xi = disc.itp_pos[a]
        free = [d for d in range(dim) if eps < xi[d] < 1 - eps]
        corners, weights = [], []
        for fc in itertools.product((0, 1), repeat=len(free)):
            c = [int(round(xi[d])) for d in range(dim)]
            w = 1.0
            for d, b in zip(free, fc):
                c[d] = b
                w *= xi[d] if b else (1.0 - xi[d])
            corners.append(corner_idx[tuple(c)])
            weights.append(w)
        node_info.append((free, corners, np.array(weights)))
    for e in range(nel):
        ev = conn[:, e]
        for a in range(itp):
            free, corners, weights = node_info[a]
            gv = [int(ev[c]) for c in corners]
            if not free:
                cp_ids[a, e] = gv[0]
                continue
            if len(free) == 1:
                lo, hi = (gv[0], gv[1]) if gv[0] < gv[1] else (gv[1], gv[0])
                t = weights[1] if gv[0] < gv[1] else weights[0]  # fraction measured from the smaller id
                key = (lo, hi, round(float(t), 9))
            elif len(free) == dim:
                key = ("b", e, a)
            else:
                if disc.itp_order > 2:
                    raise NotImplementedError("TO DO face control point matching (3_InitializeMesh.jl:133)")
                key = tuple(sorted(gv))
            cid = table.get(key)
            if cid is None:
                cid = len(coords)
                table[key] = cid
                coords.append(sum(w * vert[:, v] for w, v in zip(weights, gv)))
            cp_ids[a, e] = cid
    return ClassicalMesh(dim=dim, coords=np.array(coords, dtype=np.float64), cp_ids=cp_ids,
                         vert_conn=conn.copy(), n_vertices=nv)


def _mesh_classical_simplex(vert: np.ndarray, conn: np.ndarray, disc: ClassicalDiscretization) -> ClassicalMesh:
    """SIMPLEX control points (3_InitializeMesh.jl:1-163 with init_Structure_Triangle/Tetrahedron_Lagrange,
    101_Structures.jl:93-196): connectivity vertex j is reference vertex j (origin, e1, e2[, e3]); a lattice node
    belongs to the entity spanned by the vertices with non-zero barycentric weight."""
    dim, itp = disc.dim, disc.itp_func_num
    nel, nv = conn.shape[1], vert.shape[1]
    cp_ids = np.zeros((itp, nel), dtype=np.int64)
    coords: List[np.ndarray] = [vert[:, v].copy() for v in range(nv)]
    table: Dict[tuple, int] = {}
    info = []
    for a in range(itp):
        xi = disc.itp_pos[a]
        bary = np.concatenate([[1.0 - xi.sum()], xi])
        nz = [v for v in range(dim + 1) if bary[v] > 1e-12]
        info.append((nz, bary[nz]))
    for e in range(nel):
        ev = conn[:, e]
        for a in range(itp):
            nz, w = info[a]
            gv = [int(ev[v]) for v in nz]
            if len(nz) == 1:
                cp_ids[a, e] = gv[0]
                continue
            if len(nz) == 2:
                lo, hi = (gv[0], gv[1]) if gv[0] < gv[1] else (gv[1], gv[0])
                t = w[1] if gv[0] < gv[1] else w[0]
                key = (lo, hi, round(float(t), 9))
            elif len(nz) == dim + 1:
                key = ("b", e, a)
            else:
                if disc.itp_order > 3:
                    raise NotImplementedError("face control point matching (3_InitializeMesh.jl:133)")
                key = tuple(sorted(gv))
            cid = table.get(key)
            if cid is None:
                cid = len(coords)
                table[key] = cid
                coords.append(sum(wi * vert[:, v] for wi, v in zip(w, gv)))
            cp_ids[a, e] = cid
    return ClassicalMesh(dim=dim, coords=np.array(coords, dtype=np.float64), cp_ids=cp_ids, vert_conn=conn.copy(),
                         n_vertices=nv)


SIMPLEX_FACE_VERTS = {2: ((0, 1), (1, 2), (2, 0)), 3: ((0, 1, 2), (0, 1, 3), (3, 1, 2), (0, 2, 3))}  # 002_Initialization.jl:1-7


def lattice_mesh(x, n, disc: ClassicalDiscretization) -> ClassicalMesh:
    """Structured brick/square whose control points are numbered as a lattice.

    Lagrange order p on n elements/side -> (p*n+1)^dim lattice, id = sum_d i_d * stride_d
    with the LAST dimension fastest, i.e. make_Brick's vertex ordering
    (201_Helper_TM.jl:36-41) applied to the refined lattice.  This is the numbering the
    product's structured fast path uses; connectivity is in basis (tensor) order.
    Equivalent to ``mesh_classical(make_brick(...))`` up to a control-point permutation.
    """
    dim = disc.dim
    assert disc.itp_type == "Lagrange"
    p = disc.itp_order
    m = [p * n[d] + 1 for d in range(dim)]
    strides = [int(np.prod(m[d + 1:])) for d in range(dim)]
    grids = np.meshgrid(*[np.arange(m[d]) for d in range(dim)], indexing="ij")
    coords = np.stack([g.ravel() * (x[d] / (p * n[d])) for d, g in enumerate(grids)], axis=1).astype(np.float64)
    eg = np.meshgrid(*[np.arange(n[d]) for d in range(dim)], indexing="ij")
    e0 = sum(eg[d].ravel() * p * strides[d] for d in range(dim))
    itp = disc.itp_func_num
    cp_ids = np.zeros((itp, e0.size), dtype=np.int64)
    for a in range(itp):
        off = sum(int(round(disc.itp_pos[a, d] * p)) * strides[d] for d in range(dim))
        cp_ids[a] = e0 + off
    # first-order connectivity (counter-clockwise) for completeness
    vc = np.zeros((2 ** dim, e0.size), dtype=np.int64)
    for j, t in enumerate(VERTEX_CP_IDS[dim]):
        off = sum(((t >> d) & 1) * p * strides[d] for d in range(dim))
        vc[j] = e0 + off
    return ClassicalMesh(dim=dim, coords=coords, cp_ids=cp_ids, vert_conn=vc, n_vertices=coords.shape[0])


@dataclass
class Facets:
    """Boundary facets bound to their host element (3_InitializeMesh.jl:165-178)."""

    element_ID: np.ndarray  # [nf]
    element_eindex: np.ndarray  # [nf] 0-based local face id (reference eindex - 1)
    centroid: np.ndarray  # [nf, dim]

    def select(self, mask) -> "Facets":
        return Facets(self.element_ID[mask], self.element_eindex[mask], self.centroid[mask])

    def __len__(self):
        return len(self.element_ID)


def boundary_facets(mesh: ClassicalMesh) -> Facets:
    """Faces (3-D) / segments (2-D) used by exactly one element: ``get_BoundaryMesh``
    (002_Initialization.jl:277-289), with the host-element local face index of
    ``specify_eindex``."""
    dim = mesh.dim
    corner_idx = _corner_to_conn_index(dim)
    face_corners = {}
    simplex = mesh.vert_conn.shape[0] == dim + 1
    if simplex:
        face_corners = {fid: list(vs) for fid, vs in enumerate(SIMPLEX_FACE_VERTS[dim])}
    for nd in range(0 if simplex else dim):
        for outward in (0, 1):
            fid = CUBE_FACE_IDS[dim][nd][outward] - 1
            cs = []
            for c in itertools.product((0, 1), repeat=dim):
                if c[nd] == outward:
                    cs.append(corner_idx[c])
            face_corners[fid] = cs
    conn = mesh.vert_conn
    nel = conn.shape[1]
    keys = {}
    for fid, cs in face_corners.items():
        fv = np.sort(conn[cs, :], axis=0)  # [nvf, nel]
        for e in range(nel):
            k = tuple(fv[:, e])
            keys.setdefault(k, []).append((e, fid))
    els, eidx, cen = [], [], []
    for k, lst in keys.items():
        if len(lst) == 1:
            e, fid = lst[0]
            els.append(e)
            eidx.append(fid)
            cen.append(mesh.coords[list(k)].mean(axis=0))
    order = np.lexsort((np.array(eidx), np.array(els)))
    return Facets(np.array(els, dtype=np.int64)[order], np.array(eidx, dtype=np.int64)[order],
                  np.array(cen, dtype=np.float64)[order])


def boundary_facets_structured(x, n, dim: int) -> Facets:
    """Vectorised boundary facets of make_Square/make_Brick meshes (element order i-outer, last dim fastest)."""
    eg = np.meshgrid(*[np.arange(n[d]) for d in range(dim)], indexing="ij")
    eid = np.arange(int(np.prod(n))).reshape([n[d] for d in range(dim)])
    els, eidx, cen = [], [], []
    for nd in range(dim):
        for outward in (0, 1):
            fid = CUBE_FACE_IDS[dim][nd][outward] - 1
            sl = [slice(None)] * dim
            sl[nd] = (n[nd] - 1) if outward else 0
            e = eid[tuple(sl)].ravel()
            c = np.stack([(eg[d][tuple(sl)].ravel() + 0.5) * (x[d] / n[d]) for d in range(dim)], axis=1)
            c[:, nd] = float(outward) * x[nd]
            els.append(e)
            eidx.append(np.full(e.shape, fid))
            cen.append(c)
    return Facets(np.concatenate(els), np.concatenate(eidx), np.concatenate(cen))
