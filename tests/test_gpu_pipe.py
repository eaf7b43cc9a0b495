"""-m gpu: the "pipe" triangular-solve engine (chains x tasks, trsv_pipe.hpp) against the oracle's sequential ILU(0)
back-solve.  Same summation order per row => bit-exact, checked as such."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def _blocks(ddm, N, P, kappa=None):
    from dune_ddm_amd import synth
    from dune_ddm_amd.problem import build_structured
    grid = synth.StructuredPoisson(N, P, kappa)
    dec = build_structured(grid, overlap=2, pou_type="distance", shrink=0)
    mats = [sd.A_dir.tocsr() for sd in dec.subs]
    M = sp.block_diag(mats, format="csr")
    M.sort_indices()
    bp = np.concatenate([[0], np.cumsum([m.shape[0] for m in mats])]).astype(np.int64)
    return M, bp


def _oracle_solve(M, bp, d):
    from oracle import apply_oracle as ao
    xo = np.zeros(M.shape[0])
    for b in range(len(bp) - 1):
        r0, r1 = bp[b], bp[b + 1]
        Mb = sp.csr_matrix(M[r0:r1, r0:r1])
        Mb.sort_indices()
        xb = np.zeros(r1 - r0)
        ao.Ilu0(ao.Csr(Mb)).apply(xb, np.ascontiguousarray(d[r0:r1]))
        xo[r0:r1] = xb
    return xo


@pytest.mark.parametrize("spread", ["0", "1"])                  # XCD-local hand-overs / subdomains spread over all XCDs (write-through)
@pytest.mark.parametrize("N,P", [((26, 24, 22), (2, 2, 2)),      # 8 subdomains
                                 ((20, 18, 16), (2, 1, 1)),      # 2 subdomains
                                 ((24, 21, 16), (3, 3, 2)),      # 18 subdomains: XCDs own two or three of them and work on them at once
                                 ((9, 8, 7), (1, 1, 1))])
def test_pipe_solve_bit_exact(ddm, N, P, spread, monkeypatch):
    import torch
    assert torch.cuda.is_available()
    monkeypatch.setenv("DDM_TRSV_MODE", "pipe")
    monkeypatch.setenv("DDM_PIPE_SPREAD", spread)
    M, bp = _blocks(ddm, N, P)
    ctx = ddm.torch_context(0)
    A = ddm.CsrMatrix(ctx, M)
    F = ddm.Ilu0(ctx, A, bp)
    rng = np.random.default_rng(5)
    n = M.shape[0]
    xd = torch.zeros(n, dtype=torch.float64, device="cuda")
    for rep in range(4):                                       # graph replays, epochs, fresh right-hand sides
        d = rng.standard_normal(n)
        dd = torch.as_tensor(d).cuda()
        xd.fill_(float("nan"))
        F.solve(dd, xd)
        ctx.sync()
        assert F.status() == 0
        assert np.array_equal(xd.cpu().numpy(), _oracle_solve(M, bp, d))
    ctx.close()


def test_pipe_stress_many_solves_uneven(ddm, monkeypatch):
    """hand-offs under load: 8 subdomains of different sizes, 40 back-to-back solves with the previous result as input"""
    import torch
    monkeypatch.setenv("DDM_TRSV_MODE", "pipe")
    from dune_ddm_amd import synth
    M, bp = _blocks(ddm, (37, 29, 23), (2, 2, 2), synth.islands_kappa((36, 28, 22), 1e3, 4, 2))
    ctx = ddm.torch_context(0)
    F = ddm.Ilu0(ctx, ddm.CsrMatrix(ctx, M), bp)
    n = M.shape[0]
    rng = np.random.default_rng(9)
    d = rng.standard_normal(n)
    a = torch.as_tensor(d).cuda()
    b = torch.zeros_like(a)
    ref = d.copy()
    for it in range(40):
        F.solve(a, b)
        a, b = b, a
    ctx.sync()
    assert F.status() == 0
    for it in range(40):
        ref = _oracle_solve(M, bp, ref)
    assert np.array_equal(a.cpu().numpy(), ref)
    ctx.close()


def test_pipe_falls_back_for_wide_rows(ddm, monkeypatch):
    """rows wider than the tile format (here a dense-ish band matrix): the pipe builder declines, the object silently uses the
    loader engine and the result is still the oracle's"""
    import torch
    monkeypatch.setenv("DDM_TRSV_MODE", "pipe")
    n = 600
    rng = np.random.default_rng(2)
    B = sp.diags([rng.standard_normal(n - abs(k)) for k in range(-40, 41)], list(range(-40, 41)), format="csr")
    M = sp.csr_matrix(B + B.T + sp.eye(n) * 200.0)
    M.sort_indices()
    bp = np.array([0, n], dtype=np.int64)
    ctx = ddm.torch_context(0)
    F = ddm.Ilu0(ctx, ddm.CsrMatrix(ctx, M), bp)
    d = rng.standard_normal(n)
    x = torch.zeros(n, dtype=torch.float64, device="cuda")
    F.solve(torch.as_tensor(d).cuda(), x)
    ctx.sync()
    assert F.status() == 0
    xo = _oracle_solve(M, bp, d)
    assert np.max(np.abs(x.cpu().numpy() - xo)) <= 1e-12 * np.max(np.abs(xo))
    ctx.close()
