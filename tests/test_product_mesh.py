"""Host-side mesh entry points of the product (metafem.jl_amd/mesh.py, element.py SIMPLEX tables) against the oracle's
loop-style restatement of the same reference functions (CPU; no device needed)."""
import numpy as np
import pytest
from scipy.spatial import cKDTree

import metafem_jl_amd  # noqa: F401  (loads the package under its importable alias)
from metafem_jl_amd import element, mesh as pm
from oracle import mesh as om, readers, reference_element as re_

INP = """*Heading
** Job name: tiny Model name: Model-1
*Preprint, echo=NO, model=NO,
 history=NO, contact=NO
**
*Part, name=Part-1
*Node
      1,           4.,           0.
      2,           0.,           0.
     10,           2.,           0.
      4,           4.,           2.
      5,           0.,           2.
      7,           2.,          2.5
*Element, type=CPS4R
  1, 2, 10, 7, 5
  2, 10, 1, 4, 7
*Nset, nset=Set-3, generate
 1, 7, 1
*End Part
"""

MPHTXT = """# Created by COMSOL Multiphysics.

# Major & minor version
0 1
1 # number of tags
# Tags
5 mesh1
1 # number of types
# Types
3 obj

# --------- Object 0 ----------

0 0 1
4 Mesh # class
4 # version
3 # sdim
5 # number of mesh points
1 # lowest mesh point index

# Mesh point coordinates
0 0 0
1 0 0
0 1 0
0 0 1
1 1 1

1 # number of element types

# Type #0

3 tet # type name


4 # number of nodes per element
2 # number of elements
# Elements
1 2 3 4
2 3 4 5

2 # number of geometric entity indices
# Geometric entity indices
1
1
"""


def test_read_mesh_inp_and_mphtxt(tmp_path):
    a, b = tmp_path / "m.inp", tmp_path / "m.mphtxt"
    a.write_text(INP)
    b.write_text(MPHTXT)
    for path in (str(a), str(b)):
        vo, co = readers.read_mesh(path)
        vp, cp = pm.read_Mesh(path)
        assert np.array_equal(vo, vp) and np.array_equal(co, cp)
    v, c = pm.read_Mesh(str(a))
    assert v.shape == (2, 6) and c.tolist() == [[1, 2], [2, 0], [5, 3], [4, 5]]  # labels 1,2,10,4,5,7 -> file positions
    v, c = pm.read_Mesh(str(b))
    assert v.shape == (3, 5) and c.T.tolist() == [[0, 1, 2, 3], [1, 2, 3, 4]]
    with pytest.raises(ValueError):
        pm.read_Mesh(str(tmp_path / "m.xyz"))


@pytest.mark.parametrize("shape", ["CUBE", "SIMPLEX"])
def test_generators_match_the_loop_restatement(shape):
    v2, c2 = pm.make_Square((2.0, 1.0), (3, 2), shape)
    vo, co = om.make_square((2.0, 1.0), (3, 2))
    assert np.array_equal(v2, vo) and np.array_equal(c2, co if shape == "CUBE" else om.simplex_split(co, (3, 2)))
    v3, c3 = pm.make_Brick((1.0, 2.0, 1.5), (2, 3, 2), shape)
    vo, co = om.make_brick((1.0, 2.0, 1.5), (2, 3, 2))
    assert np.array_equal(v3, vo) and np.array_equal(c3, co if shape == "CUBE" else om.simplex_split(co, (2, 3, 2)))
    if shape == "SIMPLEX":  # the 5-tet split fills every cell: volumes add up
        p = v3.T[c3]
        vol = np.abs(np.einsum("ei,ei->e", np.cross(p[1] - p[0], p[2] - p[0]), p[3] - p[0])) / 6
        assert np.isclose(vol.sum(), 3.0)


CASES = [("CUBE", 2, "Serendipity", 2), ("CUBE", 2, "Lagrange", 2), ("CUBE", 3, "Serendipity", 2), ("CUBE", 3, "Lagrange", 2),
         ("CUBE", 3, "Lagrange", 1), ("SIMPLEX", 2, "Serendipity", 2), ("SIMPLEX", 3, "Serendipity", 2), ("SIMPLEX", 3, "Lagrange", 3),
         ("SIMPLEX", 2, "Lagrange", 1)]


@pytest.mark.parametrize("shape,dim,itp_type,order", CASES)
def test_mesh_classical_and_boundary_match_oracle(shape, dim, itp_type, order):
    n = (3, 2) if dim == 2 else (2, 3, 2)
    x = (2.0, 1.0) if dim == 2 else (1.0, 2.0, 1.5)
    vert, conn = (pm.make_Square if dim == 2 else pm.make_Brick)(x, n, shape)
    rng = np.random.default_rng(3)
    vert = vert + 0.05 * rng.standard_normal(vert.shape)  # generic positions: coordinates identify control points
    perm = rng.permutation(conn.shape[1])
    conn = conn[:, perm]
    space = element.classical_space(dim, itp_type, order, 5, shape=shape)
    disc = re_.initialize_classical_element(dim, shape, order, 1, 5, itp_type=itp_type)
    mp = pm.mesh_Classical(vert, conn, space)
    mo = om.mesh_classical(vert, conn, disc)
    assert mp.ncp == mo.ncp and mp.cp_ids.shape == mo.cp_ids.shape
    assert np.array_equal(mp.cp_ids[:, 0] < mp.n_vertices, mo.cp_ids[:, 0] < mo.n_vertices)
    d, idx = cKDTree(mo.coords).query(mp.coords)
    assert d.max() < 1e-12 and np.unique(idx).size == mp.ncp  # same set of control points
    assert np.array_equal(idx[mp.cp_ids], mo.cp_ids)  # same element -> node incidence in basis order
    fp, fo = pm.get_BoundaryMesh(mp), om.boundary_facets(mo)
    assert np.array_equal(fp.element_ID, fo.element_ID) and np.array_equal(fp.element_eindex, fo.element_eindex)
    assert np.allclose(fp.centroid, fo.centroid, atol=1e-14)


VTK_EDGES = {23: ((0, 1), (1, 2), (2, 3), (3, 0)), 22: ((0, 1), (1, 2), (2, 0)),
             25: ((0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)),
             24: ((0, 1), (1, 2), (2, 0), (0, 3), (1, 3), (2, 3))}


@pytest.mark.parametrize("shape,dim,itp_type,order", [c for c in CASES if c[3] <= 2])
def test_write_vtk_layout_and_node_order(tmp_path, shape, dim, itp_type, order):
    """Reads back through the oracle's reader (the one that parses the reference's committed VTKs) and checks VTK's
    corner / mid-edge convention geometrically: mid-edge node k of a quadratic cell is the midpoint of VTK edge k."""
    from metafem_jl_amd import vtk as pv
    from oracle import vtk as ov

    n = (3, 2) if dim == 2 else (2, 2, 2)
    x = (2.0, 1.0) if dim == 2 else (1.0, 2.0, 1.5)
    vert, conn = (pm.make_Square if dim == 2 else pm.make_Brick)(x, n, shape)
    space = element.classical_space(dim, itp_type, order, 5, shape=shape)
    msh = pm.mesh_Classical(vert, conn, space)
    rng = np.random.default_rng(1)
    fields = {"T": rng.standard_normal(msh.ncp), "d2": rng.standard_normal(msh.ncp) * 1e-9}
    path = str(tmp_path / "out.vtk")
    pv.write_VTK(path, msh.coords, msh.cp_ids, space, fields, scale=100.0)
    pts, sc = ov.read_vtk_points_scalars(path)
    assert np.array_equal(pts[:, :dim], msh.coords * 100.0) and list(sc) == ["T", "d2"]
    assert all(np.array_equal(sc[k], fields[k]) for k in fields)  # repr round-trips Float64 exactly
    lines = open(path).read().split("\n")
    assert lines[0] == "# vtk DataFile Version 3.0" and lines[2:4] == ["ASCII", "DATASET UNSTRUCTURED_GRID"]
    c0 = lines.index(next(l for l in lines if l.startswith("CELLS")))
    cell_type, nodes = pv.vtk_cell(dim, shape, itp_type, order)
    assert lines[c0].split() == ["CELLS", str(msh.nel), str(msh.nel * (1 + len(nodes)))]
    cells = np.array([[int(t) for t in lines[c0 + 1 + e].split()] for e in range(msh.nel)])
    assert np.all(cells[:, 0] == len(nodes)) and lines[c0 + 1 + msh.nel].split() == ["CELL_TYPES", str(msh.nel)]
    assert set(lines[c0 + 2 + msh.nel:c0 + 2 + 2 * msh.nel]) == {str(cell_type)}
    if cell_type in VTK_EDGES:
        ncorner = len(nodes) - len(VTK_EDGES[cell_type])
        for k, (a, b) in enumerate(VTK_EDGES[cell_type]):
            mid = 0.5 * (msh.coords[cells[:, 1 + a]] + msh.coords[cells[:, 1 + b]])
            assert np.allclose(msh.coords[cells[:, 1 + ncorner + k]], mid, atol=1e-13)


@pytest.mark.parametrize("shape,dim,itp_type,order", [("CUBE", 3, "Serendipity", 2), ("SIMPLEX", 3, "Serendipity", 2), ("CUBE", 2, "Lagrange", 2)])
def test_element_colouring_is_conflict_free_and_compact(shape, dim, itp_type, order):
    n = (6, 5) if dim == 2 else (4, 3, 3)
    vert, conn = (pm.make_Square if dim == 2 else pm.make_Brick)((1.0,) * dim, n, shape)
    space = element.classical_space(dim, itp_type, order, 5, shape=shape)
    msh = pm.mesh_Classical(vert, conn, space)
    col = pm.colour_Elements(msh.cp_ids)
    assert col.min() == 0 and (col >= 0).all()
    for c in range(col.max() + 1):
        nodes = msh.cp_ids[:, col == c].ravel()
        assert np.unique(nodes).size == nodes.size  # no control point twice within a colour
    # a colour cannot be smaller than the largest number of elements around one node; greedy stays within 2x of that bound
    deg = np.bincount(msh.cp_ids.ravel()).max()
    assert deg <= col.max() + 1 <= 2 * deg + 2
    assert np.array_equal(col, pm.colour_Elements(msh.cp_ids))  # seeded: reproducible


def test_constant_coefficient_probe_traps_every_look_into_the_environment(mf):
    """generic.constant_coefficient decides which terms take the fused constant-coefficient assembly: a coefficient function that
    reads, tests membership in, iterates or measures the environment is NOT a constant."""
    from metafem_jl_amd.generic import constant_coefficient

    assert constant_coefficient(lambda env: 0.6) == 0.6
    assert constant_coefficient(lambda env: -2 * 3.5) == -7.0
    assert constant_coefficient(lambda env: env["T"] * 2.0) is None
    assert constant_coefficient(lambda env: env.get("T", 1.0)) is None
    assert constant_coefficient(lambda env: 1.0 if "x" in env else 2.0) is None
    assert constant_coefficient(lambda env: float(len(env))) is None
    assert constant_coefficient(lambda env: 1.0 if env else 2.0) is None
    assert constant_coefficient(lambda env: sum(1.0 for _ in env)) is None
    assert constant_coefficient(lambda env: float(len(list(env.keys())))) is None
