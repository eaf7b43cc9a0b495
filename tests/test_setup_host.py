"""Product host setup (dune-ddm_amd/setup_host.py, problem.py) vs the oracle's literal
message-passing restatement: integer maps bit-exact, matrices and POU bit-exact."""
import numpy as np
import pytest

from oracle import setup_oracle as so


def _check(ddm, N, P, overlap, pou_type, shrink, kappa=None):
    from dune_ddm_amd import synth
    from dune_ddm_amd.problem import build_structured
    grid = synth.StructuredPoisson(N, P, kappa)
    dec = build_structured(grid, overlap=overlap, pou_type=pou_type, shrink=shrink, neumann=True)
    subs = grid.subdomains()
    ranks, ext = so.make_overlapping_communication(subs, overlap)
    for sd, r, e in zip(dec.subs, ranks, ext):
        assert sd.n_o == r.n_o and sd.n == r.n
        assert (sd.glob == np.array(r.glob)).all()                     # local numbering incl. arrival order
        assert (sd.owner_ovlp.astype(bool) == np.array(r.owner)).all()
        assert (dec.meta["ext_boundary"][sd.id] == e).all()
    Adir, dm = so.overlapping_matrix(ranks, subs, [s.dirichlet for s in subs])
    for sd, A, m in zip(dec.subs, Adir, dm):
        assert (sd.A_dir.indptr == A.indptr).all() and (sd.A_dir.indices == A.indices).all()
        assert (sd.A_dir.data == A.data).all()
        assert (sd.dirichlet_ovlp == m).all()
    for kind, mine in (("all_to_all", dec.ovlp_all), ("owner_to_all", dec.ovlp_owner)):
        ref = so.interface_pairs(ranks, kind)
        assert set(ref) == set(mine)
        for k in ref:
            assert (ref[k][0] == mine[k][0]).all() and (ref[k][1] == mine[k][1]).all()
    pou, bmask = so.partition_of_unity(ranks, Adir, pou_type, shrink, overlap)
    for sd, w in zip(dec.subs, pou):
        assert (sd.pou == w).all()                                     # bit-exact
    if bmask is None:
        bmask = so.identify_boundary(ranks, Adir)
    if True:
        for sd, b in zip(dec.subs, bmask):
            assert (dec.meta["boundary"][sd.id] == b).all()
        s = so.add_vector(ranks, pou)                                  # is_pou (examples/poisson.cc:141-156)
        assert max(np.abs(v - 1).max() for v in s) < 1e-10
        # Neumann region = graph distance <= 2*overlap by the reference's Gauss-Seidel sweeps
        reg = so.neumann_region_masks(Adir, bmask, overlap)
        for sd, rg in zip(dec.subs, reg):
            Bn = sd.B_neu
            rows_with_entries = np.diff(Bn.indptr) > 0
            assert (rows_with_entries == rg).all()
            # A_neu: same pattern as A_dir, symmetric, zero row sums away from Dirichlet rows/cols
            assert (sd.A_neu.indptr == sd.A_dir.indptr).all() and (sd.A_neu.indices == sd.A_dir.indices).all()
            assert abs(sd.A_neu - sd.A_neu.T).max() == 0 and abs(Bn - Bn.T).max() == 0
    return dec


def test_3d_2x2x2_overlap2_distance(ddm):
    _check(ddm, (11, 11, 11), (2, 2, 2), 2, "distance", 0)


def test_3d_uneven_overlap1_standard(ddm):
    _check(ddm, (10, 9, 8), (3, 2, 1), 1, "standard", 0)


def test_2d_overlap3_shrink(ddm):
    _check(ddm, (23, 19), (2, 2), 3, "distance", 1)


def test_3d_trivial_pou_and_contrast(ddm):
    from dune_ddm_amd import synth
    k = synth.islands_kappa((8, 8, 8), contrast=1e6, period=4, width=2)
    _check(ddm, (9, 9, 9), (2, 1, 2), 2, "trivial", 0, k)


def test_invalid_parameters(ddm):
    from dune_ddm_amd import synth
    from dune_ddm_amd.problem import build_structured
    grid = synth.StructuredPoisson((9, 9), (2, 2))
    with pytest.raises(ValueError):
        build_structured(grid, overlap=0)
    with pytest.raises(ValueError):
        build_structured(grid, overlap=2, pou_type="distance", shrink=2)      # pou.hh:184
    with pytest.raises(ValueError):
        build_structured(grid, overlap=2, pou_type="bogus")                   # pou.hh:176
