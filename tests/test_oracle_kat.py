"""Pins the oracle against every golden vector the reference's own tests hold for this path
(SURVEY.md 8c items 1-3)."""
import numpy as np
import scipy.sparse as sp

from oracle import apply_oracle as ao
from oracle import setup_oracle as so
from tests import kat_data as kd


def _to_global(M, glob, n=9):
    G = np.zeros((n, n))
    C = sp.coo_matrix(M)
    for i, j, v in zip(C.row, C.col, C.data):
        G[glob[i], glob[j]] += v
    return G


def test_overlap6_reproduces_full_matrix_on_rank0():
    # test_galerkin_coarse_matrix.cc:196-212 -- Frobenius difference <= 1e-16
    subs = kd.chain()
    ranks, _ = so.make_overlapping_communication(subs, 6)
    Aovlp, _ = so.overlapping_matrix(ranks, subs)
    assert ranks[0].n == 9
    G = _to_global(Aovlp[0], ranks[0].glob)
    assert np.linalg.norm(G - kd.A_GLOBAL) <= 1e-16
    # every rank ends with the full chain; each global id has exactly one owner
    owners = np.zeros(9, dtype=int)
    for r in ranks:
        assert sorted(r.glob) == list(range(9))
        for g, o in zip(r.glob, r.owner):
            owners[g] += int(o)
    assert (owners == 1).all()


def test_overlap1_index_sets_and_pou_sum():
    subs = kd.chain()
    ranks, ext = so.make_overlapping_communication(subs, 1)
    # arrival order: neighbours in ascending rank (this is what the POU layout :219-247 assumes)
    assert [r.glob for r in ranks] == [[0, 1, 2, 3], [2, 3, 4, 1, 5], [4, 5, 6, 3, 7], [6, 7, 8, 5]]
    assert [list(np.nonzero(m)[0]) for m in ext] == [[3], [3, 4], [3, 4], [3]]
    pou = [np.array(kd.POU[r]) for r in range(4)]
    s = so.add_vector(ranks, pou)                     # :251-263
    for v in s:
        assert (v == 1).all()


def _comm(ranks):
    return ao.Comm(len(ranks), so.interface_pairs(ranks, "owner_to_all"), so.interface_pairs(ranks, "all_to_all"),
                   [np.array(r.owner, dtype=np.uint8) for r in ranks])


def test_galerkin_coarse_matrix_golden():
    """test_galerkin_coarse_matrix.cc:216-283 -- ||A0 - expected||_F <= 1e-12.

    The stored 4x4 matrix is the *global* R A R^T for R[p, g] = pou_p[g] on the overlap-1 index sets
    (the stale test drove an older constructor).  The current build_solver
    (galerkin_preconditioner.hh:219-349) only ever multiplies with the *local* overlapping matrix,
    which equals the global product iff the basis vectors vanish on the subdomain boundary
    (SURVEY.md 3.4).  The hand-written POU of the test does not, so the vectors are carried on
    the overlap-2 index sets, extended by zero: there they vanish on the boundary layer and the
    reference algorithm must reproduce the golden values exactly."""
    subs = kd.chain()
    r1, _ = so.make_overlapping_communication(subs, 1)
    pou_by_gid = [{g: kd.POU[r.rank][i] for i, g in enumerate(r.glob)} for r in r1]
    ranks, _ = so.make_overlapping_communication(subs, 2)
    Aovlp, _ = so.overlapping_matrix(ranks, subs)
    ts = [[np.array([pou_by_gid[r.rank].get(g, 0.0) for g in r.glob])] for r in ranks]
    gp = ao.GalerkinPreconditioner([ao.Csr(A) for A in Aovlp], ts, _comm(ranks))
    A0 = gp.a0.toarray()
    assert A0.shape == (4, 4)
    assert np.linalg.norm(A0 - kd.A0_EXPECTED) <= 1e-12
    # exact zeros are dropped from the sparse coarse matrix (helpers.hh:237,253)
    assert gp.a0.nnz == 14
    # and on the overlap-1 sets (vectors non-zero on the boundary) only the diagonal blocks agree
    A1, _ = so.overlapping_matrix(r1, subs)
    g1 = ao.GalerkinPreconditioner([ao.Csr(A) for A in A1], [[np.array(kd.POU[r])] for r in range(4)], _comm(r1))
    assert np.allclose(np.diag(g1.a0.toarray()), np.diag(kd.A0_EXPECTED), rtol=0, atol=1e-12)


def test_gather_matrix_from_rows_layout():
    # tests/test_build_matrix.cc:34-40: one row per rank, A[i][j] = i + j
    size, ncols = 5, 10
    rows = [np.arange(ncols, dtype=float) + r for r in range(size)]
    A = ao.gather_matrix_from_rows_flat(rows, ncols).toarray()
    assert A.shape == (size, ncols)
    for i in range(size):
        for j in range(ncols):
            assert A[i, j] == float(i + j)
    # :49-80 uneven rows per rank (2 on even ranks, 3 on odd), every entry == rank
    slabs = []
    for r in range(size):
        k = 2 if r % 2 == 0 else 3
        slabs.append(np.full(k * ncols, float(r)))             # column-major k x ncols slab
    B = ao.gather_matrix_from_rows_flat(slabs, ncols, clip_tolerance=-1.0).toarray()
    assert B.shape == (sum(2 if r % 2 == 0 else 3 for r in range(size)), ncols)
    row = 0
    for r in range(size):
        for _ in range(2 if r % 2 == 0 else 3):
            assert (B[row] == r).all()
            row += 1


def test_oracle_dense_lu_with_row_interchanges():
    """the coarse solve of the oracle (stand-in for the reference's `coarse_solver` from the dune-istl factory,
    galerkin_preconditioner.hh:338-352): partial pivoting that actually interchanges rows -- zero diagonal blocks, as the Galerkin
    matrix of non-neighbouring subdomains has, and a non-symmetric matrix -- against numpy's LAPACK solve"""
    from oracle import apply_oracle as ao
    rng = np.random.default_rng(0)
    n = 48
    A = np.zeros((n, n))
    for i in range(4):
        for j in range(4):
            if abs(i - j) <= 1:
                A[12 * i:12 * i + 12, 12 * j:12 * j + 12] = rng.standard_normal((12, 12))
    for M in (A @ A.T + 1e-3 * np.eye(n), A + 0.1 * np.eye(n), rng.standard_normal((n, n)), np.array([[0.0, 2.0], [3.0, 1.0]])):
        lu = ao.DenseLU(M)
        assert (lu.piv != np.arange(len(M))).any() or len(M) == n and M is not A     # the cases do pivot
        b = rng.standard_normal(len(M))
        x = lu.solve(b)
        assert np.abs(x - np.linalg.solve(M, b)).max() <= 1e-10 * np.abs(x).max()
