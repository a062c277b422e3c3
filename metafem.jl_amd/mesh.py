"""Host-side mesh entry points either side of the hot path (SURVEY.md §8 f n3): what a user script calls before
`update_Mesh` / `assemble_Global_Variables!` hand arrays to the device.

  read_Mesh (.inp, .mphtxt)                    src/mesh/ref_geometry/100_File_Reader.jl:8-19, 101_Read_INP.jl, 102_Read_MPHTXT.jl
  make_Square / make_Brick (:CUBE, :SIMPLEX)   src/mesh/ref_geometry/201_Helper_TM.jl:7-77
  get_BoundaryMesh (+ specify_eindex)          src/mesh/ref_geometry/002_Initialization.jl:277-289
  mesh_Classical (control points of an         src/mesh/unstructured_mesh/3_InitializeMesh.jl:1-178 with the element
    unstructured first-order mesh)               structures of spatial_discretization/101_Structures.jl

The reference builds vertices/segments/faces/blocks tables with GPU hash tables and then allocates control points per
entity.  Here the same sharing rule -- a basis node belongs to the lowest-dimensional entity containing it, entities are
identified by their vertex ids -- is evaluated with array sorts (np.unique on vertex-id tuples), all elements at once.
Control-point numbering beyond the vertices is therefore "sorted entity key" order; the reference's own numbering
there is hash-table order and not reproducible, so every consumer addresses nodes through
`controlpoint_IDs`, never by position.  All ids are 0-based.
"""
from __future__ import annotations

import re
from dataclasses import dataclass
from typing import Tuple

import numpy as np

from .element import CUBE_FACE_IDS, SIMPLEX_FACE_VERTS, ClassicalSpace

VERTEX_CP_IDS = {2: (0, 1, 3, 2), 3: (0, 1, 3, 2, 4, 5, 7, 6)}  # connectivity vertex j -> tensor corner (101_Structures.jl:36-37)


# ---- readers ---------------------------------------------------------------------------------------------------------
def _read_inp(path: str):
    """First *Node block + first *Element block; node labels are replaced by their position in the file."""
    vids = coors = conn = None
    with open(path) as fh:
        lines = fh.read().split("\n")
    i, n = 0, len(lines)
    while i < n:
        ln = lines[i]
        if not ln.startswith("*") or ln.startswith("**"):
            i += 1
            continue
        while re.search(r", *$", lines[i]):  # keyword line continued on the next line
            i += 1
        key = ln.split(",")[0].strip().upper()
        i += 1
        rows = []
        while i < n and lines[i] != "" and not (lines[i].startswith("*") and not lines[i].startswith("**")):
            if not lines[i].startswith("**"):
                rows.append(lines[i])
            i += 1
        if key == "*NODE" and vids is None:
            tab = np.array([[float(t) for t in r.split(",") if t.strip()] for r in rows])
            vids, coors = tab[:, 0].astype(np.int64), np.ascontiguousarray(tab[:, 1:].T)
        elif key == "*ELEMENT" and conn is None:
            conn = np.array([[int(t) for t in r.split(",") if t.strip()] for r in rows], dtype=np.int64)[:, 1:].T
        if vids is not None and conn is not None:
            local = np.zeros(vids.max() + 1, dtype=np.int64)
            local[vids] = np.arange(vids.size)
            return coors, local[conn]
    raise ValueError(f"{path}: no *Node / *Element block")


def _read_mphtxt(path: str):
    """The 'number of mesh points' block and the first 'number of elements' block after it."""
    with open(path) as fh:
        lines = [ln.strip() for ln in fh]
    lines = [ln for ln in lines if ln and not ln.startswith("#")]
    coors, start, i = None, 0, 0
    while i < len(lines):
        tok = lines[i].split(" ")
        if tok[2:6] == ["number", "of", "mesh", "points"]:
            nv, start = int(tok[0]), int(lines[i + 1].split(" ")[0])
            coors = np.array([ln.split() for ln in lines[i + 2:i + 2 + nv]], dtype=np.float64).T
            i += 2 + nv
        elif tok[2:5] == ["number", "of", "elements"]:
            ne = int(tok[0])
            conn = np.array([ln.split() for ln in lines[i + 1:i + 1 + ne]], dtype=np.int64).T
            i += 1 + ne
            if coors is not None:
                return np.ascontiguousarray(coors), conn - start
        else:
            i += 1
    raise ValueError(f"{path}: no mesh points / elements block")


def read_Mesh(filename: str) -> Tuple[np.ndarray, np.ndarray]:
    """-> (coors[dim, nv], connections[nv_per_el, nel]); connections 0-based."""
    ext = filename.rsplit(".", 1)[-1].lower()
    if ext == "inp":
        return _read_inp(filename)
    if ext == "mphtxt":
        return _read_mphtxt(filename)
    raise ValueError("Undefined file type")


# ---- generators ------------------------------------------------------------------------------------------------------
def make_Square(x, n, shape: str = "CUBE"):
    dx = (x[0] / n[0], x[1] / n[1])
    i, j = np.meshgrid(np.arange(n[0] + 1), np.arange(n[1] + 1), indexing="ij")
    coors = np.stack([dx[0] * i.ravel(), dx[1] * j.ravel()]).astype(np.float64)
    ie, je = [a.ravel() for a in np.meshgrid(np.arange(n[0]), np.arange(n[1]), indexing="ij")]
    s = n[1] + 1
    cube = np.stack([ie * s + je, (ie + 1) * s + je, (ie + 1) * s + je + 1, ie * s + je + 1]).astype(np.int64)
    if shape == "CUBE":
        return coors, cube
    if shape == "SIMPLEX":  # two triangles per cell, the second with the reversed orientation kept by the reference (:27)
        return coors, np.concatenate([cube[[0, 1, 3]], cube[[2, 3, 1]]], axis=1)
    raise ValueError(shape)


def make_Brick(x, n, shape: str = "CUBE"):
    dx = tuple(x[d] / n[d] for d in range(3))
    g = np.meshgrid(*[np.arange(n[d] + 1) for d in range(3)], indexing="ij")
    coors = np.stack([dx[d] * g[d].ravel() for d in range(3)]).astype(np.float64)
    ie, je, ke = [a.ravel() for a in np.meshgrid(*[np.arange(n[d]) for d in range(3)], indexing="ij")]
    s2, s3 = (n[1] + 1) * (n[2] + 1), n[2] + 1
    b = ie * s2 + je * s3 + ke
    cube = np.stack([b, b + s2, b + s2 + s3, b + s3, b + 1, b + s2 + 1, b + s2 + s3 + 1, b + s3 + 1]).astype(np.int64)
    if shape == "CUBE":
        return coors, cube
    if shape != "SIMPLEX":
        raise ValueError(shape)
    # five tetrahedra per cell, alternating orientation with the parity of the 1-based (i + j + k) (:57-76)
    odd = ((ie + je + ke + 3) % 2) == 1
    fwd = ((0, 1, 3, 4), (2, 3, 1, 6), (7, 6, 4, 3), (5, 4, 6, 1), (3, 6, 4, 1))
    bwd = ((4, 7, 5, 0), (1, 0, 5, 2), (6, 5, 7, 2), (3, 0, 2, 7), (0, 2, 7, 5))
    ncell = b.size
    conn = np.zeros((4, 5 * ncell), dtype=np.int64)
    for t in range(5):
        sel = np.where(odd[None, :], cube[list(fwd[t])], cube[list(bwd[t])])
        conn[:, t * ncell:(t + 1) * ncell] = sel
    return coors, conn


def _unique_rows(K: np.ndarray, return_counts: bool = False):
    """np.unique(K, axis=0, return_index=True, return_inverse=True[, return_counts]) for integer rows >= -1: the rows are packed into ONE int64 key
    (mixed radix, first column most significant: the same lexicographic order) when the radix product fits -- 10-20x faster than the
    structured-dtype sort numpy uses for axis = 0 on the multi-million-row keys of a 96^3 mesh; otherwise numpy's own path."""
    K = np.asarray(K)
    lo = int(K.min()) if K.size else 0
    radix = [int(K[:, c].max()) - lo + 1 for c in range(K.shape[1])] if K.size else []
    total = 1
    for r in radix:
        total *= r
    if K.size == 0 or lo < -1 or total >= 2 ** 62:
        return np.unique(K, axis=0, return_index=True, return_inverse=True, return_counts=return_counts)
    key = np.zeros(K.shape[0], dtype=np.int64)
    for c, r in enumerate(radix):
        key = key * r + (K[:, c] - lo)
    out = np.unique(key, return_index=True, return_inverse=True, return_counts=return_counts)
    return (K[out[1]],) + tuple(out[1:])


# ---- control points --------------------------------------------------------------------------------------------------
@dataclass
class ClassicalMesh:
    dim: int
    shape: str
    coords: np.ndarray  # [ncp, dim]
    cp_ids: np.ndarray  # [itp, nel]  elements.controlpoint_IDs in basis order
    vert_conn: np.ndarray  # [nv_per_el, nel]
    n_vertices: int

    @property
    def ncp(self) -> int:
        return self.coords.shape[0]

    @property
    def nel(self) -> int:
        return self.cp_ids.shape[1]


def _node_entities(space: ClassicalSpace):
    """Per basis node: the local connectivity slots of the entity's vertices and their interpolation weights."""
    dim, out = space.dim, []
    if space.shape == "SIMPLEX":
        for xi in space.itp_pos:
            bary = np.concatenate([[1.0 - xi.sum()], xi])
            nz = [v for v in range(dim + 1) if bary[v] > 1e-12]
            out.append((nz, bary[nz]))
        return out
    corner_slot = {}
    for j, t in enumerate(VERTEX_CP_IDS[dim]):
        corner_slot[tuple((t >> d) & 1 for d in range(dim))] = j
    for xi in space.itp_pos:
        free = [d for d in range(dim) if 1e-12 < xi[d] < 1 - 1e-12]
        slots, w = [], []
        for bits in np.ndindex(*([2] * len(free))):
            c = [int(round(v)) for v in xi]
            wt = 1.0
            for d, bit in zip(free, bits):
                c[d] = bit
                wt *= xi[d] if bit else 1.0 - xi[d]
            slots.append(corner_slot[tuple(c)])
            w.append(wt)
        out.append((slots, np.array(w)))
    return out


def mesh_Classical(vert: np.ndarray, conn: np.ndarray, space: ClassicalSpace) -> ClassicalMesh:
    dim, itp = space.dim, space.itp
    nv, nel = vert.shape[1], conn.shape[1]
    nvpe = dim + 1 if space.shape == "SIMPLEX" else 2 ** dim
    if conn.shape[0] != nvpe:
        raise ValueError(f"connections have {conn.shape[0]} vertices per element, {space.shape} in {dim}-D needs {nvpe}")
    ents = _node_entities(space)
    cp_ids = np.zeros((itp, nel), dtype=np.int64)
    blocks = [np.ascontiguousarray(vert.T)]
    next_id = nv
    q = 4 * 9 * 5 * 7  # fractions along an edge become exact integers for orders 1..4 (and any Lagrange lattice up to 9)
    by_kind = {"edge": [], "face": [], "block": []}
    for a, (slots, w) in enumerate(ents):
        if len(slots) == 1:
            cp_ids[a] = conn[slots[0]]
        elif len(slots) == 2:
            by_kind["edge"].append(a)
        elif len(slots) == nvpe:
            by_kind["block"].append(a)
        else:
            by_kind["face"].append(a)
    if by_kind["edge"]:
        keys, pos, owner = [], [], []
        for a in by_kind["edge"]:
            slots, w = ents[a]
            g0, g1 = conn[slots[0]], conn[slots[1]]
            swap = g0 > g1
            lo, hi = np.where(swap, g1, g0), np.where(swap, g0, g1)
            t = np.where(swap, w[0], w[1])  # fraction measured from the smaller vertex id
            keys.append(np.stack([lo, hi, np.rint(t * q).astype(np.int64)], axis=1))
            pos.append(w[0] * vert[:, g0].T + w[1] * vert[:, g1].T)
            owner.append(a)
        K = np.concatenate(keys)
        uniq, first, inv = _unique_rows(K)
        inv = inv.reshape(-1)
        blocks.append(np.concatenate(pos)[first])
        for k, a in enumerate(owner):
            cp_ids[a] = next_id + inv[k * nel:(k + 1) * nel]
        next_id += uniq.shape[0]
    if by_kind["face"]:
        per_face = {}
        for a in by_kind["face"]:
            per_face.setdefault(tuple(sorted(ents[a][0])), []).append(a)
        if any(len(v) > 1 for v in per_face.values()):
            raise NotImplementedError("several control points per face need the face orientation (3_InitializeMesh.jl:133)")
        keys, pos, owner = [], [], []
        width = max(len(k) for k in per_face)
        for slots_key, (a,) in per_face.items():
            slots, w = ents[a]
            g = np.sort(conn[list(slots)], axis=0).T
            if g.shape[1] < width:
                g = np.concatenate([g, np.full((nel, width - g.shape[1]), -1, dtype=np.int64)], axis=1)
            keys.append(g)
            pos.append(sum(wi * vert[:, conn[s]].T for wi, s in zip(w, slots)))
            owner.append(a)
        K = np.concatenate(keys)
        uniq, first, inv = _unique_rows(K)
        inv = inv.reshape(-1)
        blocks.append(np.concatenate(pos)[first])
        for k, a in enumerate(owner):
            cp_ids[a] = next_id + inv[k * nel:(k + 1) * nel]
        next_id += uniq.shape[0]
    for a in by_kind["block"]:
        slots, w = ents[a]
        blocks.append(sum(wi * vert[:, conn[s]].T for wi, s in zip(w, slots)))
        cp_ids[a] = next_id + np.arange(nel)
        next_id += nel
    return ClassicalMesh(dim, space.shape, np.concatenate(blocks).astype(np.float64), cp_ids, conn.copy(), nv)


# ---- boundary --------------------------------------------------------------------------------------------------------
@dataclass
class Facets:
    element_ID: np.ndarray  # [nf] host element
    element_eindex: np.ndarray  # [nf] 0-based local face id of the host (reference eindex - 1)
    centroid: np.ndarray  # [nf, dim] mean of the facet's vertices

    def select(self, mask) -> "Facets":
        return Facets(self.element_ID[mask], self.element_eindex[mask], self.centroid[mask])

    def __len__(self):
        return len(self.element_ID)


def _face_slots(dim: int, shape: str):
    if shape == "SIMPLEX":
        return {f: list(v) for f, v in enumerate(SIMPLEX_FACE_VERTS[dim])}
    corner_slot = {}
    for j, t in enumerate(VERTEX_CP_IDS[dim]):
        corner_slot[tuple((t >> d) & 1 for d in range(dim))] = j
    out = {}
    for nd in range(dim):
        for outward in (0, 1):
            fid = CUBE_FACE_IDS[dim][nd][outward] - 1
            out[fid] = [corner_slot[c] for c in np.ndindex(*([2] * dim)) if c[nd] == outward]
    return out


def get_BoundaryMesh(mesh: ClassicalMesh) -> Facets:
    """Facets used by exactly one element, with the host's local face id; ordered by (element, face id)."""
    conn, nel = mesh.vert_conn, mesh.nel
    faces = _face_slots(mesh.dim, mesh.shape)
    fids = sorted(faces)
    K = np.concatenate([np.sort(conn[faces[f]], axis=0).T for f in fids])
    uniq, _, inv, cnt = _unique_rows(K, return_counts=True)
    single = cnt[inv.reshape(-1)] == 1
    el = np.tile(np.arange(nel), len(fids))[single]
    fi = np.repeat(np.array(fids), nel)[single]
    cen = mesh.coords[K[single]].mean(axis=1)
    order = np.lexsort((fi, el))
    return Facets(el[order], fi[order], cen[order])


# ---- colouring -------------------------------------------------------------------------------------------------------
def colour_Elements(cp_ids: np.ndarray, seed: int = 0x5EED) -> np.ndarray:
    """Greedy element colouring of an unstructured mesh: elements of one colour share no control point, so the S3 operators
    (and any scatter into K / residue) can accumulate without atomics and with a fixed summation order -- the
    colour-partitioned ordering of SURVEY.md §8 (b) for meshes without lattice structure (structured bricks use the
    closed-form parity colouring).  Each colour is a MAXIMAL independent set grown by rounds of random-priority
    selection (an element joins when it holds the highest priority at every one of its nodes), all elements at once.
    Returns colour[nel] (int64, 0-based, colours in order of creation)."""
    itp, nel = cp_ids.shape
    ncp = int(cp_ids.max()) + 1
    prio = np.random.default_rng(seed).permutation(nel).astype(np.int64)
    colour = np.full(nel, -1, dtype=np.int64)
    c = 0
    while (colour < 0).any():
        blocked = np.zeros(ncp, dtype=bool)
        avail = colour < 0
        while True:
            cand = avail & ~blocked[cp_ids].any(axis=0)
            ids = np.nonzero(cand)[0]
            if ids.size == 0:
                break
            nodemax = np.full(ncp, -1, dtype=np.int64)
            np.maximum.at(nodemax, cp_ids[:, ids].ravel(), np.tile(prio[ids], itp))
            sel = ids[(nodemax[cp_ids[:, ids]] == prio[ids][None, :]).all(axis=0)]
            colour[sel] = c
            blocked[cp_ids[:, sel].ravel()] = True
            avail[sel] = False
        c += 1
    return colour
