"""Host-side reference tables of the product (metafem.jl_amd/element.py) equal the oracle's polynomial-algebra tables (CPU)."""
import numpy as np
import pytest

from oracle import reference_element as re_

CASES = [(2, "Lagrange", 1, 3), (2, "Lagrange", 2, 5), (2, "Serendipity", 2, 5), (3, "Lagrange", 1, 3), (3, "Lagrange", 2, 5),
         (3, "Serendipity", 2, 5), (3, "Lagrange", 1, 7), (3, "Lagrange", 2, 1)]


@pytest.mark.parametrize("dim,itp_type,order,itg", CASES)
def test_space_tables_match_oracle(mf, dim, itp_type, order, itg):
    from metafem_jl_amd import element

    sp = element.classical_space(dim, itp_type, order, itg)
    od = re_.initialize_classical_element(dim, "CUBE", order, 1, itg, itp_type=itp_type)
    assert np.allclose(sp.itp_pos, od.itp_pos, atol=1e-15)
    assert np.allclose(sp.itg_weight, od.itg_weight, rtol=1e-15)

    def first(ref):
        out = [ref[(slice(None), slice(None)) + (0,) * dim]]
        for m in range(dim):
            idx = [0] * dim
            idx[m] = 1
            out.append(ref[(slice(None), slice(None)) + tuple(idx)])
        return np.stack(out, axis=2)

    assert np.allclose(sp.ref_itp_vals, first(od.ref_itp_vals), atol=2e-14)
    for f in range(2 * dim):
        assert np.allclose(sp.bdy_ref_itp_vals[f], first(od.bdy_ref_itp_vals[f]), atol=2e-14)
        assert np.array_equal(sp.bdy_tangent_directions[f], od.bdy_tangent_directions[f])
        assert np.allclose(sp.bdy_itg_weights[f], od.bdy_itg_weights[f], rtol=1e-15)
