"""GPU parity: generic element operators with the reference signatures (_Var_Basic / _Kval_Basic / _Res_Basic)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(kind):
    from oracle import fem, mesh as om, problems, reference_element as re_

    if kind == "hex8":
        disc = re_.initialize_classical_element(3, "CUBE", 1, 1, 3)
        n = (4, 3, 5)
        msh = om.lattice_mesh((1.0, 1.0, 1.0), n, disc)
    elif kind == "hex27":
        disc = re_.initialize_classical_element(3, "CUBE", 2, 1, 5)
        n = (3, 2, 2)
        msh = om.lattice_mesh((1.0, 1.0, 1.0), n, disc)
    else:  # quad8 serendipity, unstructured numbering
        disc = re_.initialize_classical_element(2, "CUBE", 2, 1, 5, itp_type="Serendipity")
        n = (6, 4)
        vert, conn = om.make_square((2.0, 1.0), n)
        msh = om.mesh_classical(vert, conn, disc)
    c = msh.coords
    msh.coords = c + 0.02 * np.sin(3 * c[:, ::-1])
    fac = om.boundary_facets(msh) if kind == "quad8" else om.boundary_facets_structured((1.0, 1.0, 1.0), n, 3)
    dim = disc.dim
    od = fem.FEMDomain(msh, disc, 1, problems.thermal_domain(dim, 0.6), [(fac, problems.thermal_convection(25.0, 293.15))])
    # parity colouring of the structured element grid: same-colour elements share no node
    grid = np.stack(np.meshgrid(*[np.arange(k) for k in n], indexing="ij"), axis=-1).reshape(-1, dim)
    colour = sum((grid[:, d] % 2) << d for d in range(dim))
    return od, colour


def _dev(a, dtype=None):
    import torch

    return torch.tensor(np.ascontiguousarray(a), dtype=dtype, device="cuda")


@pytest.mark.parametrize("kind", ["hex8", "hex27", "quad8"])
@pytest.mark.parametrize("base", [0, 1])
def test_domain_operators(mf, kind, base):
    import torch
    from oracle import operators as oo

    od, colour = _setup(kind)
    g = od.elgeo
    itg, itp, nsd, nel = g.integral_vals.shape
    rng = np.random.default_rng(0)
    N = _dev(g.integral_vals.ravel(order="F"))
    cp = _dev(od.mesh.cp_ids.ravel(order="F") + base, torch.int32)
    slots_np = od.pattern.sparse_ids_by_el((0, 0))
    slots = _dev(slots_np.ravel(order="F") + base, torch.int32)
    order = np.argsort(colour, kind="stable")
    offs = np.concatenate([[0], np.cumsum(np.bincount(colour, minlength=colour.max() + 1))])
    el = _dev(order + base, torch.int32)
    dims = (itg, itp, nsd, nel)
    x = rng.standard_normal(od.mesh.ncp)
    vals = rng.standard_normal((itg, nel))
    nnz, ncp = od.pattern.nnz, od.mesh.ncp
    for sd in range(nsd):
        # _Var_Basic (accumulates into a non-zero target)
        t0 = rng.standard_normal((itg, nel))
        ref = t0 + oo.var_basic(g.integral_vals, sd, 0, od.mesh.cp_ids, x, order, order)
        tgt = _dev(t0.ravel(order="F"))
        mf._Var_Basic(N, sd, 0, cp, _dev(x), tgt, el, el, dims=dims, index_base=base)
        assert np.allclose(tgt.cpu().numpy().reshape(nel, itg).T, ref, rtol=1e-13, atol=1e-13)
        for offsets in (None, offs):  # FP64 atomics (reference behaviour) and colour batches
            K0 = rng.standard_normal(nnz)
            Kref = K0.copy()
            oo.kval_basic(g.integral_vals, sd, (sd + 1) % nsd, vals[:, order], slots_np, 0, Kref, order, order)
            Kd = _dev(K0)
            mf._Kval_Basic(N, sd, (sd + 1) % nsd, _dev(vals[:, order].ravel(order="F")), slots, 0, Kd, el, el, dims=dims,
                           index_base=base, colour_offsets=offsets)
            assert np.max(np.abs(Kd.cpu().numpy() - Kref)) <= 1e-12 * np.abs(Kref).max()
            R0 = rng.standard_normal(ncp)
            Rref = R0.copy()
            oo.res_basic(g.integral_vals, sd, vals[:, order], 0, od.mesh.cp_ids, Rref, order, order)
            Rd = _dev(R0)
            mf._Res_Basic(N, sd, _dev(vals[:, order].ravel(order="F")), 0, cp, Rd, el, el, dims=dims, index_base=base,
                          colour_offsets=offsets)
            assert np.max(np.abs(Rd.cpu().numpy() - Rref)) <= 1e-12 * np.abs(Rref).max()


def test_facet_operators_and_field_shift(mf):
    """Boundary launch: itg_hostIDs = facet ids, elIDs = host elements (05_CodeGenerator.jl:175-189); shifts select a field block."""
    import torch
    from oracle import operators as oo

    od, _ = _setup("hex8")
    fg, facets = od.fgeo[0], od.boundaries[0][0]
    itg, itp, nsd, nf = fg.integral_vals.shape
    N = _dev(fg.integral_vals.ravel(order="F"))
    rng = np.random.default_rng(4)
    vals = rng.standard_normal((itg, nf))
    host = np.arange(nf)
    ncp = od.mesh.ncp
    shift = ncp  # second field of a 2-field vector
    R0 = np.zeros(2 * ncp)
    Rref = R0.copy()
    oo.res_basic(fg.integral_vals, 0, vals, shift, od.mesh.cp_ids, Rref, host, facets.element_ID)
    Rd = _dev(R0)
    mf._Res_Basic(N, 0, _dev(vals.ravel(order="F")), shift, _dev(od.mesh.cp_ids.ravel(order="F") + 1, torch.int32), Rd,
                  _dev(host + 1, torch.int32), _dev(facets.element_ID + 1, torch.int32), dims=(itg, itp, nsd, nf), index_base=1)
    assert np.max(np.abs(Rd.cpu().numpy() - Rref)) <= 1e-12 * np.abs(Rref).max()
    assert np.all(Rd.cpu().numpy()[:ncp] == 0.0)


def test_operator_argument_errors(mf):
    import torch

    z = torch.zeros(8, dtype=torch.float64, device="cuda")
    i = torch.zeros(8, dtype=torch.int32, device="cuda")
    with pytest.raises(mf.MetaFEMError):
        mf._Var_Basic(z, 5, 0, i, z, z, i, i, dims=(2, 2, 2, 1), index_base=0)  # sd out of range
    with pytest.raises(mf.MetaFEMError):
        mf._Res_Basic(z, 0, z, 0, i, z, i, i, dims=(2, 2, 2, 1), index_base=3)
