"""Host part of the sparse direct local solver (csrc/sparse_chol_host.hpp through the C ABI, no device needed):
the factor handed to the device engines -- permutation + CSR in the ILU(0) storage convention (unit lower factor, inverse
pivots on the diagonal, D L^T above) -- applied here by sequential substitution must solve the system to rounding
(checked against scipy's SuperLU), be exact on the pattern (L D L^T = P A P^T), and the nested-dissection ordering must
keep the fill of grid problems near the known asymptotics."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spl


def _apply(f, d):
    """x = P^T (L D L^T)^-1 P d with the factor in the ILU(0) convention: v = L^-1 d (unit lower), x_i = dinv_i (v_i - sum_{j>i} u_ij x_j)"""
    perm, rp, ci, lu = f["perm"], f["rowptr"], f["col"], f["lu"]
    n = len(perm)
    F = sp.csr_matrix((lu, ci, rp), shape=(n, n))
    rows = np.repeat(np.arange(n), np.diff(rp))
    Lm = sp.csr_matrix((np.where(ci < rows, lu, 0.0), ci, rp), shape=(n, n)) + sp.eye(n)
    Um = sp.csr_matrix((np.where(ci > rows, lu, 0.0), ci, rp), shape=(n, n)) + sp.diags(1.0 / F.diagonal())
    v = spl.spsolve_triangular(Lm.tocsr(), d[perm], lower=True)
    xp = spl.spsolve_triangular(Um.tocsr(), v, lower=False)
    x = np.empty(n)
    x[perm] = xp
    return x, Lm, Um


def _cases(ddm):
    from dune_ddm_amd import synth
    from dune_ddm_amd.problem import build_structured
    dec = build_structured(synth.StructuredPoisson((11, 10, 9), (2, 1, 1)), overlap=2, neumann=True)
    yield "poisson A_dir", dec.subs[0].A_dir, None
    mats = [sd.A_dir for sd in dec.subs]
    yield "two blocks", sp.block_diag(mats, format="csr"), np.cumsum([0] + [m.shape[0] for m in mats])
    dece = build_structured(synth.StructuredElasticity((12, 2, 3), 2), overlap=1, neumann=True, second_region="all")
    sd = dece.subs[1]
    C = sp.diags(sd.pou) @ sd.A_neu @ sp.diags(sd.pou)
    yield "elasticity A_neu + sigma D A D (floating subdomain)", sp.csr_matrix(sd.A_neu + 1e-3 * C), None
    dg = build_structured(synth.StructuredDG2D((16, 16), (2, 2)), overlap=2, neumann=True)
    yield "DG symmetric part", dg.subs[0].A_neu, None


def test_factor_solves_and_reproduces_matrix(ddm):
    rng = np.random.default_rng(5)
    for name, M, bp in _cases(ddm):
        M = sp.csr_matrix(M)
        M.sort_indices()
        n = M.shape[0]
        f = ddm.chol_host(M, bp)
        assert sorted(f["perm"]) == list(range(n))
        if bp is not None:                                   # the permutation stays inside the blocks
            for b in range(len(bp) - 1):
                assert set(f["perm"][bp[b]:bp[b + 1]]) == set(range(bp[b], bp[b + 1]))
        rows = np.repeat(np.arange(n), np.diff(f["rowptr"]))
        assert all(np.all(np.diff(f["col"][f["rowptr"][i]:f["rowptr"][i + 1]]) > 0) for i in range(0, n, 7))   # sorted columns
        d = rng.standard_normal(n)
        x, Lm, Um = _apply(f, d)
        xs = spl.spsolve(M.tocsc(), d)
        assert np.abs(x - xs).max() <= 1e-9 * np.abs(xs).max(), name
        PAP = M[f["perm"]][:, f["perm"]]
        assert abs(Lm @ Um - PAP).max() <= 1e-12 * abs(M).max(), name
        assert f["nnzL"] == (f["col"] <= rows).sum() and f["flops"] > 0


def test_rejects_indefinite_and_reports_symbolic_only(ddm):
    M = sp.csr_matrix(np.array([[1.0, 2.0], [2.0, 1.0]]))
    with pytest.raises(ddm.DdmError):
        ddm.chol_host(M)
    f = ddm.chol_host(M, numeric=False)                      # ordering + symbolic analysis only
    assert f["lu"] is None and f["nnzL"] == 3


def test_nested_dissection_fill(ddm):
    """regression guard on the ordering quality: entries of L per row on a 2-D 9-point grid (grows like log n: measured 32.5 at
    65^2, 40.3 at 129^2) and on a 3-D 27-point grid (grows like n^(1/3): measured 254 at 21^3) -- a natural-order (banded)
    factor has 65 / 441 entries per row there and grows like n^(1/2) / n^(2/3)"""
    from dune_ddm_amd import synth
    for N, P, per_row in (((65, 65), (1, 1), 40.0), ((129, 129), (1, 1), 50.0), ((21, 21, 21), (1, 1, 1), 300.0)):
        A = synth.StructuredPoisson(N, P).subdomain(0).A
        f = ddm.chol_host(A, numeric=False)
        assert f["nnzL"] < per_row * A.shape[0], (N, f["nnzL"] / A.shape[0])


def test_lu_without_pivoting_for_the_nonsymmetric_dg_operator(ddm):
    """general = True: L U on the pattern of A + A^T for the non-symmetric SIPG / upwind matrix of BASELINE configs[3] (what the
    reference hands to UMFPACK): solves to rounding, L U = P A P^T on the pattern; also a structurally non-symmetric input."""
    from dune_ddm_amd import synth
    from dune_ddm_amd.problem import build_structured
    rng = np.random.default_rng(9)
    dg = build_structured(synth.StructuredDG2D((16, 16), (2, 2)), overlap=2, neumann=False)
    mats = [sd.A_dir for sd in dg.subs[:2]]
    M = sp.block_diag(mats, format="csr")
    M.sort_indices()
    bp = np.cumsum([0] + [m.shape[0] for m in mats])
    T = sp.csr_matrix(M.copy())
    T.data = np.where(rng.random(T.nnz) < 0.2, 0.0, T.data)      # drop entries on one side only
    T.eliminate_zeros()
    T = (T + 50 * sp.eye(T.shape[0])).tocsr()
    T.sort_indices()
    for name, A, b_ptr in (("dg", M, bp), ("structurally non-symmetric", T, bp)):
        assert abs(A - A.T).max() > 1e-3
        f = ddm.chol_host(A, b_ptr, general=True)
        d = rng.standard_normal(A.shape[0])
        x, Lm, Um = _apply(f, d)
        xs = spl.spsolve(A.tocsc(), d)
        assert np.abs(x - xs).max() <= 1e-9 * np.abs(xs).max(), name
        PAP = A[f["perm"]][:, f["perm"]]
        assert abs(Lm @ Um - PAP).max() <= 1e-12 * abs(A).max(), name
