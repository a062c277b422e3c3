"""write_VTK (reference src/mesh/unstructured_mesh/5_VTK.jl:7-158): the on-disk format after the hot path -- legacy ASCII
`DATASET UNSTRUCTURED_GRID`, control points as POINTS (x scale [+ shift field]), one cell per element with the
quadratic VTK cell the basis maps onto, then POINT_DATA with one `SCALARS <sym> float 1` block per inner variable.

The node-order tables translate the basis (tensor / lattice) order of `controlpoint_IDs` to VTK's corner-then-mid-edge
order; interior/face nodes of the full Lagrange cubes have no slot in VTK's 8-/20-node cells and are left out of the
cell (they remain in POINTS), exactly like the reference."""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

# (dim, shape, itp_type, order) -> (VTK cell type, basis ids (0-based) in VTK node order)   5_VTK.jl:30-116
_CELLS = {
    (2, "CUBE", "Lagrange", 1): (9, (0, 1, 3, 2)),
    (2, "CUBE", "Lagrange", 2): (23, (0, 2, 8, 6, 1, 5, 7, 3)),
    (2, "CUBE", "Serendipity", 1): (9, (0, 1, 3, 2)),
    (2, "CUBE", "Serendipity", 2): (23, (0, 1, 3, 2, 4, 7, 5, 6)),
    (2, "SIMPLEX", None, 1): (5, (0, 1, 2)),
    (2, "SIMPLEX", None, 2): (22, (0, 2, 5, 1, 4, 3)),
    (3, "CUBE", "Lagrange", 1): (12, (0, 1, 3, 2, 4, 5, 7, 6)),
    (3, "CUBE", "Lagrange", 2): (25, (0, 2, 8, 6, 18, 20, 26, 24, 1, 5, 7, 3, 19, 23, 25, 21, 9, 11, 17, 15)),
    (3, "CUBE", "Serendipity", 1): (12, (0, 1, 3, 2, 4, 5, 7, 6)),
    (3, "CUBE", "Serendipity", 2): (25, (0, 1, 3, 2, 4, 5, 7, 6, 8, 13, 9, 12, 10, 15, 11, 14, 16, 17, 19, 18)),
    (3, "SIMPLEX", None, 1): (10, (0, 1, 2, 3)),
    (3, "SIMPLEX", None, 2): (24, (0, 2, 5, 9, 1, 4, 3, 6, 7, 8)),
}


def vtk_cell(dim: int, shape: str, itp_type: str, itp_order: int):
    key = (dim, shape, None if shape == "SIMPLEX" else itp_type, itp_order)
    if key not in _CELLS:
        raise ValueError(f"write_VTK: no VTK cell for {shape} {itp_type} order {itp_order} in {dim}-D")
    return _CELLS[key]


def write_VTK(fname: str, coords: np.ndarray, cp_ids: np.ndarray, space, fields: Dict[str, np.ndarray], scale: float = 1.0,
              shift: Optional[np.ndarray] = None) -> None:
    """coords [ncp, dim]; cp_ids [itp, nel] 0-based (controlpoint_IDs in basis order); fields: name -> nodal array
    (local_innervar_infos order); shift [ncp, dim] = the `shift_sym` displacement added before scaling."""
    ncp, dim = coords.shape
    cell_type, order = vtk_cell(dim, space.shape, space.itp_type, space.itp_order)
    xs = np.zeros((ncp, 3))
    xs[:, :dim] = coords if shift is None else coords + shift
    xs *= scale
    cells = cp_ids[list(order)].T  # [nel, nodes]
    nel, m = cells.shape
    with open(fname, "w") as io:
        io.write(f"# vtk DataFile Version 3.0\n{fname}\nASCII\nDATASET UNSTRUCTURED_GRID\n")
        io.write(f"POINTS {ncp} float\n")
        io.write("\n".join(f"{r[0]!r} {r[1]!r} {r[2]!r}" for r in xs.tolist()))
        io.write(f"\nCELLS {nel} {nel * (1 + m)}\n")
        io.write("\n".join(" ".join(map(str, [m] + row)) for row in cells.tolist()))
        io.write(f"\nCELL_TYPES  {nel}\n" + f"{cell_type}\n" * nel)
        io.write(f"POINT_DATA {ncp}\n")
        for sym, vals in fields.items():
            v = np.asarray(vals, dtype=np.float64).reshape(-1)
            if v.size != ncp:
                raise ValueError(f"field {sym}: {v.size} values for {ncp} control points")
            io.write(f"SCALARS {sym} float 1\nLOOKUP_TABLE default\n")
            io.write("\n".join(repr(x) for x in v.tolist()) + "\n")
