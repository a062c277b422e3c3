"""Legacy-ASCII VTK reader/writer for golden fields (oracle; test infrastructure only).

Format written by the reference: src/mesh/unstructured_mesh/5_VTK.jl:7-158 --
``POINTS n float`` (values printed as Float64), ``CELLS``, ``CELL_TYPES``, then
``POINT_DATA n`` with one ``SCALARS <sym> float 1`` + ``LOOKUP_TABLE default`` block per
local inner variable.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np


def read_vtk_points_scalars(path: str) -> Tuple[np.ndarray, Dict[str, np.ndarray]]:
    with open(path, "r") as fh:
        lines = fh.read().split("\n")
    pts = None
    scal: Dict[str, np.ndarray] = {}
    i = 0
    npts = 0
    while i < len(lines):
        tok = lines[i].split()
        if tok and tok[0] == "POINTS":
            npts = int(tok[1])
            vals = []
            i += 1
            while len(vals) < 3 * npts:
                vals.extend(float(v) for v in lines[i].split())
                i += 1
            pts = np.array(vals[:3 * npts]).reshape(npts, 3)
            continue
        if tok and tok[0] == "SCALARS":
            name = tok[1]
            i += 2  # skip LOOKUP_TABLE
            vals = []
            while len(vals) < npts:
                vals.extend(float(v) for v in lines[i].split())
                i += 1
            scal[name] = np.array(vals[:npts])
            continue
        i += 1
    return pts, scal


def write_vtk_points_scalars(path: str, pts: np.ndarray, scalars: Dict[str, np.ndarray]) -> None:
    """Point cloud + scalars in the reference's block layout (cells omitted: VTK_VERTEX)."""
    n = pts.shape[0]
    p3 = np.zeros((n, 3))
    p3[:, :pts.shape[1]] = pts
    with open(path, "w") as fh:
        fh.write("# vtk DataFile Version 2.0\nmetafem\nASCII\nDATASET UNSTRUCTURED_GRID\n")
        fh.write(f"POINTS {n} float\n")
        for r in p3:
            fh.write(f"{r[0]!r} {r[1]!r} {r[2]!r}\n")
        fh.write(f"CELLS {n} {2 * n}\n")
        for i in range(n):
            fh.write(f"1 {i}\n")
        fh.write(f"CELL_TYPES {n}\n" + "1\n" * n)
        fh.write(f"POINT_DATA {n}\n")
        for k, v in scalars.items():
            fh.write(f"SCALARS {k} float 1\nLOOKUP_TABLE default\n")
            for x in v:
                fh.write(f"{float(x)!r}\n")
