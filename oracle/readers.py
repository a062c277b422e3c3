"""External mesh readers (oracle; test infrastructure only).

Restates
  read_Mesh                 src/mesh/ref_geometry/100_File_Reader.jl:8-19
  read_INP                  src/mesh/ref_geometry/101_Read_INP.jl:1-57   (Abaqus first-order mesh: first *Node and first
                            *Element block; node labels compacted to file order)
  read_MPHTXT               src/mesh/ref_geometry/102_Read_MPHTXT.jl     (COMSOL mesh: the highest-dimensional element type)
Returns (coors[dim, nv], connections[nvpe, nel] 0-based), the pair construct_TotalMesh consumes.
"""
from __future__ import annotations

import re
from typing import List, Tuple

import numpy as np


def read_inp(path: str) -> Tuple[np.ndarray, np.ndarray]:
    lines = [ln.rstrip("\n") for ln in open(path)]
    i, n = 0, len(lines)
    vids = coors = el_vids = None

    def block(start: int):
        rows, j = [], start
        while j < n:
            ln = lines[j]
            if ln.startswith("**"):  # is_Comment_INP (:1)
                j += 1
                continue
            if ln == "" or ln.startswith("*"):  # block_Finished (:6)
                break
            rows.append([t for t in ln.strip().split(",") if t.strip() != ""])
            j += 1
        return rows, j

    while i < n:
        ln = lines[i]
        if ln.startswith("**") or not ln.startswith("*"):
            i += 1
            continue
        while re.search(r", *$", lines[i]):  # has_Nextline_INP (:3): a keyword line continued over several lines
            i += 1
        key = ln.split(",")[0].strip().upper()
        rows, i = block(i + 1)
        if key == "*NODE" and vids is None:
            vids = np.array([int(r[0]) for r in rows])
            coors = np.array([[float(v) for v in r[1:]] for r in rows], dtype=np.float64).T
        elif key == "*ELEMENT" and el_vids is None:
            el_vids = np.array([[int(v) for v in r[1:]] for r in rows]).T
        if vids is not None and el_vids is not None:
            local = np.zeros(vids.max() + 1, dtype=np.int64)
            local[vids] = np.arange(vids.size)  # :51-54 node labels -> positions in the file
            return coors, local[el_vids]
    raise ValueError("no *Node / *Element block found")


def read_mesh(path: str):
    ext = path.rsplit(".", 1)[-1].lower()
    if ext == "inp":
        return read_inp(path)
    if ext == "mphtxt":
        return read_mphtxt(path)
    raise ValueError("Undefined file type")


def read_mphtxt(path: str):
    """102_Read_MPHTXT.jl:4-45: the '<n> # number of mesh points' block (preceded by the lowest vertex index) and the
    FIRST '<n> # number of elements' block after it; blank lines and '#' lines are skipped."""
    lines = [ln.strip() for ln in open(path)]
    lines = [ln for ln in lines if ln and not ln.startswith("#")]  # is_Comment_MPHTXT
    coors = conn = None
    start_vid = 0
    i = 0
    while i < len(lines):
        tok = lines[i].split(" ")
        if len(tok) >= 6 and tok[2:6] == ["number", "of", "mesh", "points"]:
            nvert = int(tok[0])
            start_vid = int(lines[i + 1].split(" ")[0])
            coors = np.array([[float(v) for v in lines[i + 2 + k].split()] for k in range(nvert)], dtype=np.float64).T
            i += 2 + nvert
            continue
        if len(tok) >= 5 and tok[2:5] == ["number", "of", "elements"]:
            nel = int(tok[0])
            conn = np.array([[int(v) for v in lines[i + 1 + k].split()] for k in range(nel)], dtype=np.int64).T
            i += 1 + nel
            if coors is not None:
                return coors, conn - start_vid
            continue
        i += 1
    raise ValueError("no mesh points / elements block found")
