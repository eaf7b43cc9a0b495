"""-m gpu: the FP64-MFMA contractions of the GenEO block eigensolver (csrc/geneo_kernels.hpp) against an FP64 host reference:
per-subdomain Gram products U^T V and basis rotations U Y of tall-skinny row-major blocks, for ragged subdomain sizes (not
multiples of the 4-row MFMA step, the 16-row slab or the 2048-row chunk), widths that are not multiples of the 16-column tile,
strided views, and asymmetric data (a transposed result must not pass).  Tolerance: 1e-13 relative to sum |a||b| -- the MFMA
sums the same FP64 products in a different order."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("pu,pv", [(72, 72), (24, 24), (7, 19), (78, 26), (132, 132), (1, 3), (204, 204), (300, 68), (130, 150)])   # the last three: column panels
def test_gram_matches_fp64_reference(ddm, pu, pv):
    import torch
    ctx = ddm.torch_context(0)
    rng = np.random.default_rng(pu * 1000 + pv)
    sizes = [4099, 17, 2048, 1, 6150]
    bp = np.concatenate([[0], np.cumsum(sizes)])
    n = int(bp[-1])
    big_u = rng.standard_normal((n, pu + 5))
    big_v = rng.standard_normal((n, pv + 3)) + np.arange(pv + 3)[None, :]      # asymmetric columns
    Ud, Vd = torch.as_tensor(big_u).cuda(), torch.as_tensor(big_v).cuda()
    for U, V, Uh, Vh in ((Ud[:, :pu].contiguous(), Vd[:, :pv].contiguous(), big_u[:, :pu], big_v[:, :pv]),):
        G = ddm.blockvec_gram(ctx, bp, U, V)
        for s in range(len(sizes)):
            ref = Uh[bp[s]:bp[s + 1]].T @ Vh[bp[s]:bp[s + 1]]
            bound = np.abs(Uh[bp[s]:bp[s + 1]]).T @ np.abs(Vh[bp[s]:bp[s + 1]])
            assert np.all(np.abs(G[s] - ref) <= 1e-13 * bound + 1e-300), (s, np.abs(G[s] - ref).max())
    ctx.close()


@pytest.mark.parametrize("p,q", [(72, 48), (72, 24), (24, 24), (5, 3), (132, 44), (9, 17), (204, 136), (90, 50), (300, 100)])   # the last three: output and inner panels
def test_rotate_matches_fp64_reference(ddm, p, q):
    import torch
    ctx = ddm.torch_context(0)
    rng = np.random.default_rng(p * 100 + q)
    sizes = [4099, 17, 2048, 1, 6150, 15]
    bp = np.concatenate([[0], np.cumsum(sizes)])
    n = int(bp[-1])
    Uh = rng.standard_normal((n, p))
    Y = rng.standard_normal((len(sizes), p, q))
    Bh = rng.standard_normal((n, q + 4))
    U = torch.as_tensor(Uh).cuda()
    out = torch.full((n, q + 4), float("nan"), dtype=torch.float64, device="cuda")      # wider than q: a strided destination
    ddm.blockvec_rotate(ctx, bp, U, Y, out)
    o = out.cpu().numpy()
    assert np.isnan(o[:, q:]).all()                                                     # nothing written beyond q columns
    for s in range(len(sizes)):
        rows = slice(bp[s], bp[s + 1])
        ref = Uh[rows] @ Y[s]
        assert np.all(np.abs(o[rows, :q] - ref) <= 1e-13 * (np.abs(Uh[rows]) @ np.abs(Y[s])) + 1e-300)
    base = torch.as_tensor(Bh).cuda()
    ddm.blockvec_rotate(ctx, bp, U, Y, base, base=base)                                 # in place: W <- W - X coef
    b = base.cpu().numpy()
    for s in range(len(sizes)):
        rows = slice(bp[s], bp[s + 1])
        ref = Bh[rows, :q] - Uh[rows] @ Y[s]
        assert np.all(np.abs(b[rows, :q] - ref) <= 1e-13 * (np.abs(Bh[rows, :q]) + np.abs(Uh[rows]) @ np.abs(Y[s])))
    assert np.array_equal(b[:, q:], Bh[:, q:])
    ctx.close()


@pytest.mark.parametrize("p", [72, 48, 80, 33, 5, 96])   # 96: beyond five tiles, two general products
def test_gram2_sym_matches_fp64_reference(ddm, p):
    """the fused pair of symmetric products of the Rayleigh-Ritz step (k_gram2_sym): U^T (M1 U), U^T (M2 U) with symmetric M1, M2 applied
    row-wise per subdomain (here: symmetric tridiagonal-in-rows stand-ins built on the host), against the FP64 host products; the result
    must be exactly symmetric (mirrored upper triangle)"""
    import torch
    ctx = ddm.torch_context(0)
    rng = np.random.default_rng(100 + p)
    sizes = [4099, 17, 2048, 1, 6150]
    bp = np.concatenate([[0], np.cumsum(sizes)])
    n = int(bp[-1])
    Uh = rng.standard_normal((n, p + 3))
    V1h = np.empty((n, p + 3))
    V2h = np.empty((n, p + 3))
    for s in range(len(sizes)):           # V = M U with M symmetric per subdomain => U^T V symmetric up to rounding
        r = slice(bp[s], bp[s + 1])
        d1, d2 = rng.uniform(1, 2, sizes[s]), rng.uniform(0, 1, sizes[s])
        V1h[r] = d1[:, None] * Uh[r]
        V2h[r] = d2[:, None] * Uh[r]
        if sizes[s] > 1:
            e = rng.standard_normal(sizes[s] - 1)
            V1h[r][1:] += e[:, None] * Uh[r][:-1]
            V1h[r][:-1] += e[:, None] * Uh[r][1:]
    U, V1, V2 = (torch.as_tensor(x).cuda() for x in (Uh, V1h, V2h))
    g1, g2 = ddm.blockvec_gram2_sym(ctx, bp, U[:, :p], V1[:, :p], V2[:, :p])     # strided views (ld = p + 3)
    for s in range(len(sizes)):
        r = slice(bp[s], bp[s + 1])
        for g, Vh in ((g1, V1h), (g2, V2h)):
            ref = Uh[r, :p].T @ Vh[r, :p]
            bound = np.abs(Uh[r, :p]).T @ np.abs(Vh[r, :p])
            assert np.all(np.abs(g[s] - ref) <= 2e-13 * bound + 1e-300), (s, np.abs(g[s] - ref).max())
            if p <= 80:
                assert np.array_equal(g[s], g[s].T)
    ctx.close()
