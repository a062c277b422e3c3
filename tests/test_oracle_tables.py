"""Oracle self-checks of the reference-element tables (CPU)."""
import itertools
import os

import numpy as np
import pytest

from oracle import reference_element as re_

GOLD = os.path.join(os.path.dirname(__file__), "golden")

ELEMENTS = {"quad8": (2, "CUBE", 2, 1, 5, "Serendipity"), "hex8": (3, "CUBE", 1, 1, 3, "Lagrange"),
            "hex27": (3, "CUBE", 2, 1, 5, "Lagrange"), "hex20": (3, "CUBE", 2, 1, 5, "Serendipity"),
            "quad4": (2, "CUBE", 1, 1, 3, "Lagrange"), "quad9": (2, "CUBE", 2, 1, 5, "Lagrange")}


def _disc(name):
    a = ELEMENTS[name]
    return re_.initialize_classical_element(*a[:5], itp_type=a[5])


@pytest.mark.parametrize("name", list(ELEMENTS))
def test_partition_of_unity_and_weights(name):
    d = _disc(name)
    val = d.ref_itp_vals[(slice(None), slice(None)) + (0,) * d.dim]
    assert np.allclose(val.sum(axis=1), 1.0, atol=1e-13)
    for m in range(d.dim):
        idx = [0] * d.dim
        idx[m] = 1
        assert np.allclose(d.ref_itp_vals[(slice(None), slice(None)) + tuple(idx)].sum(axis=1), 0.0, atol=1e-12)
    assert abs(d.itg_weight.sum() - 1.0) < 1e-14  # reference cell is [0,1]^d (103_Integrations.jl:1-2)
    for w in d.bdy_itg_weights:
        assert abs(w.sum() - 1.0) < 1e-14


@pytest.mark.parametrize("name", list(ELEMENTS))
def test_kronecker_delta_at_nodes(name):
    d = _disc(name)
    M = np.array([[f(p) for f in d.itp_funcs] for p in d.itp_pos])
    assert np.allclose(M, np.eye(d.itp_func_num), atol=1e-12)


def test_node_counts_and_orders():
    assert _disc("quad8").itp_func_num == 8 and _disc("quad8").itg_func_num == 9
    assert _disc("hex8").itp_func_num == 8 and _disc("hex8").itg_func_num == 8
    assert _disc("hex27").itp_func_num == 27 and _disc("hex27").itg_func_num == 27
    assert _disc("hex20").itp_func_num == 20
    # gauss_order = ceil((itg_order+1)/2) (103_Integrations.jl:15)
    assert [re_.gauss_order_of(k) for k in range(8)] == [1, 1, 2, 2, 3, 3, 4, 4]


def test_hex8_is_tensor_ordered_x_fastest():
    d = _disc("hex8")
    expect = np.array([[(b >> k) & 1 for k in range(3)] for b in range(8)], dtype=float)
    assert np.array_equal(d.itp_pos, expect)
    # Gauss points: first coordinate fastest
    g = re_.GAUSS_POS_SHIFTED[1]
    assert np.allclose(d.itg_pos[1], [g[1], g[0], g[0]])


def test_quad8_serendipity_closed_form():
    d = _disc("quad8")
    x, y = 0.3, 0.6
    assert abs(d.itp_funcs[0]((x, y)) - (1 - x) * (1 - y) * (1 - 2 * x - 2 * y)) < 1e-14
    assert np.allclose(d.itp_pos[4:], [[0.5, 0], [0.5, 1], [0, 0.5], [1, 0.5]])


@pytest.mark.parametrize("dim", [2, 3])
def test_reference_tangents_give_outward_normals(dim):
    """t1 x t2 (3-D) / (t2, -t1) (2-D) of the REFERENCE tangents is the outward axis (103_Integrations.jl:40-47)."""
    pos, w, tan = re_.boundary_integration_cube(3, dim)
    for nd in range(dim):
        for outward in (0, 1):
            f = re_.CUBE_FACE_IDS[dim][nd][outward] - 1
            t = tan[f][0]
            if dim == 3:
                n = np.cross(t[:, 0], t[:, 1])
            else:
                n = np.array([t[1, 0], -t[0, 0]])
            e = np.zeros(dim)
            e[nd] = 1.0 if outward else -1.0
            assert np.allclose(n, e)
            assert np.allclose(pos[f][:, nd], outward)


def test_tables_match_committed_fixture():
    z = np.load(os.path.join(GOLD, "oracle_tables.npz"))
    for name in ("quad8", "hex8", "hex27", "hex20"):
        d = _disc(name)
        assert np.array_equal(d.ref_itp_vals, z[f"{name}_ref"])
        assert np.array_equal(d.itg_weight, z[f"{name}_w"])


def test_polynomial_check_clear_threshold():
    from oracle.polynomial import Poly

    p = Poly(1, {(1,): 1.0, (0,): 5e-9}) * Poly.const(1, 1.0)
    assert (0,) not in p.terms  # |c| < 1e-8 is dropped (03_Polynomial.jl:61-75)
    assert Poly(2, {(2, 1): 3.0}).derivative((1, 1)).terms == {(1, 0): 6.0}
