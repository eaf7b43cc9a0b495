"""Overlapping Neumann matrices by assembly interception (dune_ddm_amd/neumann_assembly.py, the mirror of
examples/assemblewrapper.hh:182-367 + examples/pdelab_helper.hh:113-436):
  * against the oracle's literal per-element restatement (oracle/neumann_oracle.py): triples and matrices bit for bit;
  * against the DEFINITION the procedure implements -- the sum of the element matrices over the elements inside the region -- which is
    what dune_ddm_amd.synth assembles directly and the benchmark uses (build_structured(..., neumann=True)).
CPU only.  The reference holds no golden data for this path (parity unpinned against the reference itself; pinned by the definition)."""
import numpy as np
import pytest
import scipy.sparse as sp


def _same(A, B):
    A, B = sp.csr_matrix(A), sp.csr_matrix(B)
    A.sort_indices()
    B.sort_indices()
    return A.shape == B.shape and np.array_equal(A.indptr, B.indptr) and np.array_equal(A.indices, B.indices) and np.array_equal(A.data, B.data)


@pytest.mark.parametrize("N,P,overlap", [((9, 8), (2, 2), 1), ((13, 11), (3, 2), 2), ((7, 6, 6), (2, 1, 2), 1), ((9, 9, 9), (2, 2, 2), 2)])
def test_interception_gives_the_region_matrices(ddm, N, P, overlap):
    from dune_ddm_amd import synth
    from dune_ddm_amd.neumann_assembly import assemble_overlapping_matrices
    from dune_ddm_amd.problem import build_structured
    kappa = synth.islands_kappa(tuple(n - 1 for n in N), 1e3, 3, 1)              # integer-valued: every sum is exact
    grid = synth.StructuredPoisson(N, P, kappa)
    got = assemble_overlapping_matrices(grid, overlap, "all", "overlap")
    dec = build_structured(grid, overlap=overlap, neumann=True)                    # the direct element sums
    for m, sd in zip(got, dec.subs):
        assert _same(m.A_dir, sd.A_dir)
        assert np.array_equal(m.boundary_dst <= 2 * overlap, sd.boundary_dist <= 2 * overlap)
        assert _same(m.A_neu, sd.A_neu)
        assert _same(m.B_neu, sd.B_neu)
        assert any(len(t.val) for r, t in m.triples_sent.items() if r >= 0)       # corrections really travelled


@pytest.mark.parametrize("N,P,overlap,regions", [((9, 8), (2, 2), 1, ("all", "overlap")), ((8, 7, 6), (2, 1, 2), 1, ("all", "overlap")),
                                                 ((11, 9), (2, 2), 2, ("all", "all")), ((12, 10), (2, 2), 1, ("overlap", "overlap"))])
def test_against_the_oracle_restatement(ddm, N, P, overlap, regions):
    from dune_ddm_amd import synth
    from dune_ddm_amd.neumann_assembly import assemble_overlapping_matrices
    from oracle import neumann_oracle as no
    rng = np.random.default_rng(4)
    kappa = rng.random(tuple(n - 1 for n in N)[::-1]) + 0.5                       # non-integer: the ORDER of the additions matters
    grid = synth.StructuredPoisson(N, P, kappa)
    got = assemble_overlapping_matrices(grid, overlap, *regions)
    subs = grid.subdomains()
    elements = []
    for r in range(grid.nranks):
        dofs, Ke = grid.elements(r)
        elements.append([(dofs[e], Ke[e]) for e in range(len(dofs))])
    want = no.assemble_overlapping_matrices(subs, elements, [s.dirichlet for s in subs], overlap, *regions)
    for m, (A_dir, A_neu, B_neu, dm, dst, triples) in zip(got, want):
        assert np.array_equal(m.dirichlet_mask_ovlp > 0, dm > 0)
        assert np.array_equal(np.minimum(m.boundary_dst, 4 * overlap + 2), np.minimum(dst, 4 * overlap + 2))
        for rank, t in m.triples_sent.items():
            ref = [x for x in triples[rank] if x[2] != 0.0 or True]
            nz = {(int(r), int(c)): v for r, c, v in zip(t.row, t.col, t.val)}
            for (r, c, v) in ref:                                                  # the oracle also lists untouched pattern entries (0.0)
                assert nz.get((r, c), 0.0) == v
            assert set(nz) <= {(r, c) for r, c, _ in ref}
        # the matrices: same stored pattern; values to rounding (synth sums the element stencils of A_dir in another order than the
        # oracle's message passing sums the ranks' additive matrices -- the corrections themselves were compared exactly above)
        for mine, ref in ((m.A_dir, A_dir), (m.A_neu, A_neu), (m.B_neu, B_neu)):
            mine, ref = sp.csr_matrix(mine), sp.csr_matrix(ref)
            mine.sort_indices()
            ref.sort_indices()
            assert np.array_equal(mine.indptr, ref.indptr) and np.array_equal(mine.indices, ref.indices)
            assert np.allclose(mine.data, ref.data, rtol=1e-13, atol=1e-13)


def test_against_the_oracle_exactly_with_integer_coefficients(ddm):
    from dune_ddm_amd import synth
    from dune_ddm_amd.neumann_assembly import assemble_overlapping_matrices
    from oracle import neumann_oracle as no
    N, P, overlap = (10, 9), (2, 2), 2
    grid = synth.StructuredPoisson(N, P, synth.islands_kappa(tuple(n - 1 for n in N), 64, 3, 1))
    got = assemble_overlapping_matrices(grid, overlap, "all", "overlap")
    subs = grid.subdomains()
    elements = []
    for r in range(grid.nranks):
        dofs, Ke = grid.elements(r)
        elements.append([(dofs[e], Ke[e]) for e in range(len(dofs))])
    want = no.assemble_overlapping_matrices(subs, elements, [s.dirichlet for s in subs], overlap, "all", "overlap")
    for m, (A_dir, A_neu, B_neu, dm, dst, triples) in zip(got, want):
        assert _same(m.A_dir, A_dir) and _same(m.A_neu, A_neu) and _same(m.B_neu, B_neu)


def test_rejects_what_the_reference_rejects(ddm):
    from dune_ddm_amd import synth
    from dune_ddm_amd.neumann_assembly import assemble_overlapping_matrices
    grid = synth.StructuredPoisson((7, 7), (2, 1))
    with pytest.raises(NotImplementedError, match="only supported if the first is NeumannRegion::All"):
        assemble_overlapping_matrices(grid, 1, "overlap", "extended_overlap")
    with pytest.raises(NotImplementedError):
        assemble_overlapping_matrices(grid, 1, "all", "ring")
