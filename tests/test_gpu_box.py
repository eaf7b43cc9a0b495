"""-m gpu: the "box" triangular-solve engine (structured leading box of every block walked plane by plane, the rows behind it by a nested
factor: trsv_box.hpp) against the oracle's sequential ILU(0) back-solve.  Same summation order per row => bit-exact, checked as such."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu

from tests.test_gpu_pipe import _blocks, _oracle_solve  # noqa: E402


@pytest.mark.parametrize("spread", ["0", "1"])                  # XCD-local hand-overs / placement-independent (write-through)
@pytest.mark.parametrize("N,P", [((26, 24, 22), (2, 2, 2)),      # 8 subdomains, boxes 13 x 12 x 11 .. with shells on three sides
                                 ((20, 18, 16), (2, 1, 1)),      # 2 subdomains
                                 ((44, 40, 18), (4, 3, 1)),      # 12 subdomains: XCDs own one or two of them; shells on up to four sides
                                 ((9, 8, 7), (1, 1, 1)),         # a plain box, no shell
                                 ((24, 21, 16), (3, 3, 2)),      # boxes smaller than their shells: declined, the pipe engine takes the matrix
                                 ((150, 9, 8), (1, 1, 1))])      # nx > 120: declined as well
def test_box_solve_bit_exact(ddm, N, P, spread, monkeypatch):
    import torch
    assert torch.cuda.is_available()
    monkeypatch.setenv("DDM_TRSV_MODE", "box")
    monkeypatch.setenv("DDM_BOX_SPREAD", spread)
    M, bp = _blocks(ddm, N, P)
    ctx = ddm.torch_context(0)
    A = ddm.CsrMatrix(ctx, M)
    F = ddm.Ilu0(ctx, A, bp)
    assert F.engine() == ("pipe" if N[0] > 120 or P == (3, 3, 2) else "box")
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


def test_box_large_lines_and_many_planes(ddm, monkeypatch):
    """lines longer than 64 rows and more than 64 lines per plane (lanes serve two lines), high-contrast coefficient, 40 back-to-back
    solves with the previous result as input"""
    import torch
    monkeypatch.setenv("DDM_TRSV_MODE", "box")
    from dune_ddm_amd import synth
    M, bp = _blocks(ddm, (119, 71, 21), (1, 1, 2), synth.islands_kappa((118, 70, 20), 1e3, 4, 2))
    ctx = ddm.torch_context(0)
    F = ddm.Ilu0(ctx, ddm.CsrMatrix(ctx, M), bp)
    assert F.engine() == "box"
    n = M.shape[0]
    rng = np.random.default_rng(9)
    d = rng.standard_normal(n)
    a = torch.as_tensor(d).cuda()
    b = torch.zeros_like(a)
    ref = d.copy()
    for it in range(6):
        F.solve(a, b)
        a, b = b, a
        ref = _oracle_solve(M, bp, ref)
    ctx.sync()
    assert F.status() == 0
    assert np.array_equal(a.cpu().numpy(), ref)
    ctx.close()
