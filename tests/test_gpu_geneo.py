"""-m gpu: the device GenEO builder (block eigensolver, dune-ddm_amd/geneo.py) against the oracle's
literal Spectra restatement: eigenvalues, spanned subspace, and the outer CG iteration count."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _orth(V):
    q, _ = np.linalg.qr(np.asarray(V).T)
    return q


def _sin_largest_angle(U, V):
    """sine of the largest principal angle between span(U) and span(V) (rows = vectors)"""
    Qu, Qv = _orth(U), _orth(V)
    return float(np.linalg.norm(Qu - Qv @ (Qv.T @ Qu), 2))


@pytest.mark.parametrize("m", [24, 8, 6])
def test_multi_rhs_single_precision_sweeps(ddm, m):
    """ddm_ilu0_solve_multi_f32 (the GenEO block iteration's preconditioner application: float factor entries and work block, double
    in / out) against the double solve on 4 independent blocks: relative error of single precision times the growth of the two
    sweeps (here < 2e-5 of the column's largest entry); m = 6 is not a multiple of 4 and must give the double result bit for bit;
    repeated calls (graph replay) give identical results."""
    import torch
    from dune_ddm_amd import synth
    import scipy.sparse as sp
    ctx = ddm.torch_context(0)
    M1 = synth.StructuredPoisson((14, 12, 11), (1, 1, 1)).subdomain(0).A
    M = sp.block_diag([M1] * 4, format="csr")
    n1 = M1.shape[0]
    A = ddm.CsrMatrix(ctx, M)
    F = ddm.Ilu0(ctx, A, block_ptr=[0, n1, 2 * n1, 3 * n1, 4 * n1])
    rng = np.random.default_rng(5)
    D = torch.as_tensor(rng.standard_normal((4 * n1, m))).cuda()
    X64, X32, X32b = torch.empty_like(D), torch.empty_like(D), torch.empty_like(D)
    F.solve_multi(D, X64)
    F.solve_multi(D, X32, single_precision=True)
    F.solve_multi(D, X32, single_precision=True)          # replay of the captured graph
    F.solve_multi(D, X32b, single_precision=True)         # other output block: new capture
    ctx.sync()
    assert torch.equal(X32, X32b)
    if m % 4:
        assert torch.equal(X32, X64)
    else:
        err = (X32 - X64).abs().max(dim=0).values / X64.abs().max(dim=0).values
        assert not torch.equal(X32, X64) and float(err.max()) < 2e-5, err
    ctx.close()


@pytest.mark.parametrize("m", [7, 8, 24])
def test_multi_rhs_kernels(ddm, m):
    """ddm_csr_mm and ddm_ilu0_solve_multi == column-by-column single-vector kernels (bit-exact: same order); m = 8, 24 take the
    four-columns-per-thread SpMM kernel."""
    import torch
    from dune_ddm_amd import synth
    ctx = ddm.torch_context(0)
    M = synth.StructuredPoisson((12, 11, 10), (1, 1, 1)).subdomain(0).A
    A = ddm.CsrMatrix(ctx, M)
    F = ddm.Ilu0(ctx, A)
    n = M.shape[0]
    rng = np.random.default_rng(11)
    X = torch.as_tensor(rng.standard_normal((n, m))).cuda()
    Y = torch.empty_like(X)
    Z = torch.empty_like(X)
    A.mm(X, Y)
    F.solve_multi(X, Z)
    ctx.sync()
    for j in range(m):
        xj = X[:, j].contiguous()
        yj = torch.empty_like(xj)
        zj = torch.empty_like(xj)
        A.mv(xj, yj)
        F.solve(xj, zj)
        ctx.sync()
        assert torch.allclose(Y[:, j], yj, rtol=1e-13, atol=1e-13)
        assert (Z[:, j] == zj).all()
    # ... and against the oracle (CSR product in row order; ILU(0) solve is bit-exact, a5)
    from oracle import apply_oracle as ao
    oA = ao.Csr(M)
    oF = ao.Ilu0(oA)
    Xh = X.cpu().numpy()
    for j in range(m):
        xj = np.ascontiguousarray(Xh[:, j])
        zo, yo = np.empty(n), np.empty(n)
        oF.apply(zo, xj)
        oA.mv(xj, yo)
        assert np.array_equal(Z[:, j].cpu().numpy(), zo)
        assert np.allclose(Y[:, j].cpu().numpy(), yo, rtol=1e-13, atol=1e-13)
    ctx.close()


@pytest.mark.parametrize("N,nev,contrast", [((45, 41, 37), 3, None), ((25, 25, 25), 4, 1e4)])
def test_geneo_eigenpairs_and_iteration_count(ddm, N, nev, contrast):
    # Grids are chosen without symmetry-induced multiple eigenvalues among the wanted ones: the reference's
    # single-vector Lanczos returns only one copy of a multiple eigenvalue (on 41^3 it skips the second
    # copy of 0.905 and returns a decoupled Dirichlet mode lambda = 1 instead), the block method finds both.
    from dune_ddm_amd import synth
    from dune_ddm_amd.geneo import geneo_basis
    from dune_ddm_amd.problem import build_structured
    from dune_ddm_amd.solver import TwoLevelSchwarz
    from oracle import geneo_oracle as go
    from tests.oracle_bridge import oracle_solve
    kappa = None if contrast is None else synth.islands_kappa(tuple(n - 1 for n in N), contrast=contrast, period=6, width=2)
    dec = build_structured(synth.StructuredPoisson(N, (2, 2, 2), kappa), overlap=2, pou_type="distance", neumann=True)
    tl = TwoLevelSchwarz(dec, coarse="none")
    basis, info = geneo_basis(tl, nev=nev, tol=1e-5, return_info=True)
    assert info["converged"]
    obasis = {}
    for sd in dec.subs:
        vecs, lam = go.geneo_basis(sd.A_neu, sd.B_neu, sd.pou, {"nev": nev})
        # the configuration is chosen so that no decoupled Dirichlet mode (lambda = 1/pou^2 >= 1) is wanted
        assert lam.max() < 1.0 - 1e-6
        assert np.allclose(info["eigenvalues"][sd.id], lam, rtol=1e-6)           # eigenvalue error ~ residual^2
        ov = np.array(vecs)
        ov[:, sd.dirichlet_ovlp > 0] = 0.0
        obasis[sd.id] = [v for v in ov]
        assert _sin_largest_angle(basis[sd.id], ov) < 2e-3                        # eigenvector error ~ tol / gap
        assert np.abs(np.linalg.norm(basis[sd.id], axis=1) - 1.0).max() < 1e-12 or (sd.dirichlet_ovlp > 0).any()
    tl.set_coarse_basis(basis)
    tl.rebuild_combined("additive")
    res, hist, x = tl.solve(reduction=1e-10, maxit=500)
    it, conv, hist_o, xo = oracle_solve(dec, reduction=1e-10, maxit=500, coarse=obasis, schwarz_type="standard", mode="additive")
    assert res.converged and conv
    assert abs(res.iterations - it) <= 1, (res.iterations, it)                   # same coarse space up to the eigensolver tolerance
    # and with the SAME (device-computed) basis handed to the oracle the histories agree per iteration
    it2, conv2, hist2, _ = oracle_solve(dec, reduction=1e-10, maxit=500, coarse={s: list(basis[s]) for s in basis}, schwarz_type="standard", mode="additive")
    h2 = np.array(hist2)
    # (absolute floor 4e-12 ||r_0|| instead of 1e-14: the device applies the replicated explicit inverse of the
    #  K x K coarse matrix, the oracle an LU solve; the two differ by cond(R A R^T) * eps in every application.  Observed: <= 1.2e-12
    #  ||r_0||, reached in the last three of 71 iterations; the first 40 iterations agree to 1e-8 relative with a floor of 1e-13)
    assert it2 == res.iterations and (np.abs(hist - h2) <= 1e-8 * h2 + 4e-12 * h2[0]).all()
    assert (np.abs(hist[:40] - h2[:40]) <= 1e-8 * h2[:40] + 1e-13 * h2[0]).all()
    tl.ctx.close()


def test_geneo_nev20_symmetric_grid_multiple_eigenvalues(ddm):
    """BASELINE configs[2]'s eigensolver setting (nev = 20) on a cube split 2 x 2 x 2 -- every subdomain pencil has the symmetries
    of the cube corner, i.e. MULTIPLE eigenvalues among the wanted ones (0.29, 0.5037, 0.5677, ... are double).  57^3 with overlap 1
    is the smallest such instance whose 20 lowest eigenvalues stay below the decoupled Dirichlet unit modes (lambda = 1 / pou^2 >= 1)
    that the library deflates (csrc/geneo.hpp) and the reference would return.  Against the oracle's literal Spectra restatement:
    (i) eigenvalues, all 20 incl. multiplicities, 1e-6 relative; (ii) every oracle eigenvector whose eigenvalue lies strictly below
    the last wanted one is contained in the device span (sine of the angle < 2e-3) -- the 20th eigenvalue (0.97271) is itself double
    and the cut goes through its eigenspace: the single-vector Lanczos returns one vector of it, the block method another;
    (iii) outer CG: iteration count within +-2 with each side's own basis, and per-iteration parity with the same basis."""
    from concurrent.futures import ThreadPoolExecutor
    from dune_ddm_amd import synth
    from dune_ddm_amd.geneo import geneo_basis
    from dune_ddm_amd.problem import build_structured
    from dune_ddm_amd.solver import TwoLevelSchwarz
    from oracle import geneo_oracle as go
    from tests.oracle_bridge import oracle_solve
    nev = 20
    dec = build_structured(synth.StructuredPoisson((57, 57, 57), (2, 2, 2)), overlap=1, pou_type="distance", neumann=True)
    tl = TwoLevelSchwarz(dec, coarse="none")
    basis, info = geneo_basis(tl, nev=nev, tol=1e-5, return_info=True)
    assert info["converged"]
    with ThreadPoolExecutor(8) as ex:
        ores = list(ex.map(lambda sd: go.geneo_basis(sd.A_neu, sd.B_neu, sd.pou, {"nev": nev}), dec.subs))
    obasis = {}
    for sd, (vecs, lam) in zip(dec.subs, ores):
        lam_d = info["eigenvalues"][sd.id]
        assert lam.max() < 1.0 - 1e-6                                                   # no deflated unit mode among the wanted ones
        assert np.allclose(lam_d, lam, rtol=1e-6), (sd.id, lam_d, lam)                  # (i)
        assert np.min(np.diff(lam) / lam[1:]) < 1e-6                                    # the instance does have multiple eigenvalues
        ov = np.array(vecs)
        ov[:, sd.dirichlet_ovlp > 0] = 0.0
        obasis[sd.id] = [v for v in ov]
        Q = _orth(basis[sd.id])
        below = [v / np.linalg.norm(v) for v, l in zip(ov, lam) if l < lam[-1] * (1 - 1e-3)]
        assert len(below) >= nev - 2
        for u in below:
            assert np.linalg.norm(u - Q @ (Q.T @ u)) < 2e-3, sd.id                      # (ii)
    tl.set_coarse_basis(basis)
    tl.rebuild_combined("additive")
    res, hist, x = tl.solve(reduction=1e-10, maxit=500)
    it, conv, hist_o, xo = oracle_solve(dec, reduction=1e-10, maxit=500, coarse=obasis, schwarz_type="standard", mode="additive")
    assert res.converged and conv and abs(res.iterations - it) <= 2, (res.iterations, it)               # (iii)
    it2, conv2, hist2, _ = oracle_solve(dec, reduction=1e-10, maxit=500, coarse={s: list(basis[s]) for s in basis}, schwarz_type="standard", mode="additive")
    h2 = np.array(hist2)
    # same basis on both sides: identical count; per-iteration parity 1e-8 while ||r_k|| > 1e-4 ||r_0||, afterwards CG amplifies the
    # rounding-level differences of the reduction orders (measured 1.6e-1 at the last of 81 iterations here),
    # so the tail is only required to stay within a factor 2
    assert it2 == res.iterations
    early = h2 > 1e-4 * h2[0]
    assert (np.abs(hist - h2)[early] <= 1e-8 * h2[early] + 1e-12 * h2[0]).all(), float(np.max(np.abs(hist - h2)[early] / h2[early]))
    assert (np.abs(np.log(hist / h2)) < np.log(2.0)).all()
    print(f"[geneo nev=20] device {res.iterations} iterations, oracle basis {it}; block iterations {info['iterations']}, direct {info['used_direct']}")
    tl.ctx.close()


def test_geneo_threshold_mode_matches_oracle(ddm):
    """Threshold mode of spectra_gevp_op (dune/ddm/eigensolvers/spectra.hh:157-163, 186-189) through ddm_geneo_basis: start with
    nev = 4, double until the largest computed eigenvalue exceeds the threshold (or nev >= nev_max = 2 nev), keep the eigenvalues below it.
    On the 25^3 islands problem with threshold 0.5 the subdomains keep 3 or 4 vectors, six of the eight only after a doubling
    (eigenvalues: 0.2713 0.2716 0.4026 | 0.8905 ... resp. 0.1767 0.2715 0.4027 0.4425 | 0.7014 ...).  Against the oracle's literal
    restatement (oracle/geneo_oracle.py), subdomain by subdomain: the same NUMBER of vectors, eigenvalues 1e-6, spans 2e-3 -- and
    the independent host check of every returned pair (geneo.host_eigenpair_residuals)."""
    from dune_ddm_amd import synth
    from dune_ddm_amd.geneo import geneo_basis_from_params, host_eigenpair_residuals
    from dune_ddm_amd.problem import build_structured
    from dune_ddm_amd.solver import TwoLevelSchwarz
    from oracle import geneo_oracle as go
    N = (25, 25, 25)
    kappa = synth.islands_kappa(tuple(n - 1 for n in N), contrast=1e4, period=6, width=2)
    dec = build_structured(synth.StructuredPoisson(N, (2, 2, 2), kappa), overlap=2, pou_type="distance", neumann=True)
    tl = TwoLevelSchwarz(dec, coarse="none")
    eig = {"nev": 4, "threshold": 0.5}
    # (no nev_max key: eigensolver_params.hh:23 lets that key overwrite ncv and leaves nev_max UNINITIALISED in the reference; without
    #  it nev_max = 2 nev = 8, so the subdomains whose 4th eigenvalue is below the threshold double once)
    basis, info = geneo_basis_from_params(tl, eig)
    assert info["converged"]
    counts = []
    for sd in dec.subs:
        vecs, lam = go.geneo_basis(sd.A_neu, sd.B_neu, sd.pou, dict(eig))
        k = len(vecs)
        counts.append(k)
        assert info["nconv"][sd.id] == k == basis[sd.id].shape[0], (sd.id, info["nconv"][sd.id], k)
        assert np.allclose(info["eigenvalues"][sd.id][:k], lam[:k], rtol=1e-6)
        assert (lam[:max(k - 1, 1)] < 0.5).all()
        ov = np.array(vecs)
        ov[:, sd.dirichlet_ovlp > 0] = 0.0
        assert _sin_largest_angle(basis[sd.id], ov) < 2e-3
        res, rq = host_eigenpair_residuals(sd, basis[sd.id], info["eigenvalues"][sd.id][:k])
        assert res.max() < 1e-3 and rq.max() < 1e-6, (sd.id, res.max(), rq.max())
    assert min(counts) >= 1 and len(set(counts)) > 1          # the subdomains really keep different numbers of vectors
    assert info["nev"] == 8                                    # and nev was doubled
    tl.ctx.close()


def test_geneo_wide_block_beyond_48_columns(ddm):
    """nev + extra > 48 (round 2 refused it; the reference's threshold mode may double nev up to nev_max, spectra.hh:157-163,186-189): the
    Gram / rotation kernels then work in column panels.  nev = 60 on the subdomains of a 2 x 2 x 2 decomposition of 21 x 19 x 17 (1 300 -
    1 700 rows each), against a DENSE generalized eigen-decomposition on the host (scipy.linalg.eigh of the reciprocal pencil
    C~ x = mu (A + sigma C~) x, C~ = D B D without the Dirichlet rows / columns -- the decoupled Dirichlet unit modes the library deflates,
    DESIGN 5; at these sizes the single-vector Lanczos restatement does not resolve the highly degenerate eigenvalue 1): all 60
    eigenvalues 1e-6, and the independent residual check of every returned pair."""
    import scipy.linalg as sla
    import scipy.sparse as sp
    from dune_ddm_amd import synth
    from dune_ddm_amd.geneo import geneo_basis, host_eigenpair_residuals
    from dune_ddm_amd.problem import build_structured
    from dune_ddm_amd.solver import TwoLevelSchwarz
    nev, sigma = 60, 1e-3
    N = (21, 19, 17)
    kappa = synth.islands_kappa(tuple(n - 1 for n in N), contrast=1e3, period=5, width=2)
    dec = build_structured(synth.StructuredPoisson(N, (2, 2, 2), kappa), overlap=2, pou_type="distance", neumann=True)
    tl = TwoLevelSchwarz(dec, coarse="none")
    basis, info = geneo_basis(tl, nev=nev, tol=1e-6, shift=sigma, return_info=True, maxit=600)
    assert info["converged"] and info["nev"] == nev
    for sd in dec.subs:
        free = (sd.dirichlet_ovlp == 0).astype(float)
        Dp = sp.diags(sd.pou * free)
        C = (Dp @ sp.csr_matrix(sd.B_neu) @ Dp).toarray()
        At = sp.csr_matrix(sd.A_neu).toarray() + sigma * C
        mu = sla.eigh(C, At, eigvals_only=True)[::-1][:nev]
        lam_ref = 1.0 / mu - sigma
        lam_d = info["eigenvalues"][sd.id]
        assert np.allclose(lam_d, lam_ref, rtol=1e-6, atol=1e-9), (sd.id, np.abs(lam_d - lam_ref).max())
        res, rq = host_eigenpair_residuals(sd, basis[sd.id], lam_d)
        assert res.max() < 1e-4 and rq.max() < 1e-7, (sd.id, res.max(), rq.max())
    tl.ctx.close()


def test_geneo_cache_blocked_row_order_is_bit_identical(ddm, monkeypatch):
    """The block products A~ X, C~ X of the eigensolver run in a cache-blocked row order when the subdomain matrices come from a
    structured grid (k_spmm_rowmajor4_tiled; tests/test_row_order.py); every row is computed exactly as in the natural order, so the
    whole eigensolver run -- eigenvalues, basis, iteration count -- must be BIT-identical to a run with DDM_SPMM_NATURAL_ORDER=1."""
    from dune_ddm_amd import synth
    from dune_ddm_amd.geneo import geneo_basis
    from dune_ddm_amd.problem import build_structured, _block_diag
    from dune_ddm_amd.solver import TwoLevelSchwarz
    import scipy.sparse as sp
    dec = build_structured(synth.StructuredPoisson((37, 33, 29), (2, 1, 1)), overlap=2, pou_type="distance", neumann=True)
    found, _ = ddm.row_order_tiled_host(np.concatenate([[0], np.cumsum([sd.n for sd in dec.subs])]), sp.csr_matrix(_block_diag([sd.A_neu for sd in dec.subs])))
    assert found                                                                      # the tiled kernel is what runs below
    runs = []
    for natural in (False, True):
        if natural:
            monkeypatch.setenv("DDM_SPMM_NATURAL_ORDER", "1")
        tl = TwoLevelSchwarz(dec, coarse="none")
        basis, info = geneo_basis(tl, nev=8, tol=1e-6, return_info=True)      # block width 8 + 4 = 12: a multiple of 4, the four-column kernels run
        runs.append((basis, info))
        tl.ctx.close()
    (b0, i0), (b1, i1) = runs
    assert i0["iterations"] == i1["iterations"] and i0["converged"] and i1["converged"]
    for s in b0:
        assert np.array_equal(b0[s], b1[s])
        assert np.array_equal(i0["eigenvalues"][s], i1["eigenvalues"][s])


def test_host_only_matrix_objects(ddm):
    """ddm_csr_create_host: the GenEO inputs A_neu / B_neu are read on the host only; such an object must be refused (DDM_EINVAL with a
    message, not a fault) by every entry point that would touch device arrays."""
    import torch
    from dune_ddm_amd import synth
    ctx = ddm.torch_context(0)
    M = synth.StructuredPoisson((9, 8, 7), (1, 1, 1)).subdomain(0).A
    H = ddm.CsrMatrix(ctx, M, host_only=True)
    x = torch.ones(M.shape[0], dtype=torch.float64, device="cuda")
    y = torch.empty_like(x)
    with pytest.raises(RuntimeError, match="without device arrays"):
        H.mv(x, y)
    X = torch.ones((M.shape[0], 4), dtype=torch.float64, device="cuda")
    with pytest.raises(RuntimeError, match="without device arrays"):
        H.mm(X, torch.empty_like(X))
    V = torch.ones((2, M.shape[0]), dtype=torch.float64, device="cuda")
    with pytest.raises(RuntimeError, match="without device arrays"):
        ddm.galerkin_products(ctx, H, V, V, 0, M.shape[0])
    F = ddm.Ilu0(ctx, H)                                  # ILU(0) factorises on the host and uploads its own schedule: fine
    F.solve(x, y)
    G = ddm.Ilu0(ctx, ddm.CsrMatrix(ctx, M))
    z = torch.empty_like(x)
    G.solve(x, z)
    ctx.sync()
    assert torch.equal(y, z)
    ctx.close()
