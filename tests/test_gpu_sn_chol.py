"""The sparse direct solver whose numeric factorisation runs ON THE DEVICE (csrc/sn_chol.hpp: supernodal Cholesky, FP64-MFMA updates,
panel solves) against scipy's SuperLU -- the oracle's `DirectSolver` (oracle/apply_oracle.py) is the same splu -- and by
residuals at the sizes SuperLU takes too long for."""
import os
import time

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spl

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device (no CPU fallback exists)"
    return torch


@pytest.fixture()
def device_engine():
    old = os.environ.get("DDM_DIRECT_ENGINE")
    os.environ["DDM_DIRECT_ENGINE"] = "device"
    yield
    if old is None:
        del os.environ["DDM_DIRECT_ENGINE"]
    else:
        os.environ["DDM_DIRECT_ENGINE"] = old


def _blocks(ddm, shape, parts, overlap=2, kappa=None):
    from dune_ddm_amd import synth
    from dune_ddm_amd.problem import RankLocal, build_structured
    dec = build_structured(synth.StructuredPoisson(shape, parts, kappa), overlap=overlap, pou_type="distance")
    rl = RankLocal(dec, 0, 1)
    return dec, rl


@pytest.mark.parametrize("shape,parts", [((17, 16, 15), (2, 2, 2)), ((33, 31, 29), (2, 2, 2)), ((70, 66), (2, 2))])
def test_device_cholesky_matches_superlu(ddm, torch_cuda, device_engine, shape, parts):
    import torch
    dec, rl = _blocks(ddm, shape, parts)
    ctx = ddm.torch_context(0)
    A = ddm.CsrMatrix(ctx, rl.A_dir)
    F = ddm.Ilu0(ctx, A, rl.block_ptr, direct=True)
    assert ctx.lib.ddm_ilu0_is_direct(F.h) == 1
    n = rl.n
    rng = np.random.default_rng(3)
    lus = [spl.splu(sp.csc_matrix(sd.A_dir)) for sd in dec.subs]

    def ref(b):
        out = np.empty_like(b)
        for i, lu in enumerate(lus):
            a, e = int(rl.block_ptr[i]), int(rl.block_ptr[i + 1])
            out[a:e] = lu.solve(b[a:e])
        return out

    # single right-hand side
    b = rng.standard_normal(n)
    x = torch.zeros(n, dtype=torch.float64, device="cuda")
    F.solve(torch.as_tensor(b).cuda(), x)
    ctx.sync()
    assert F.status() == 0
    xr = ref(b)
    assert np.abs(x.cpu().numpy() - xr).max() <= 1e-9 * np.abs(xr).max()
    # blocks of right-hand sides (GenEO: 24; an odd count; the widest allowed)
    for m in (24, 5, 48):
        B = rng.standard_normal((n, m))
        X = torch.zeros((n, m), dtype=torch.float64, device="cuda")
        F.solve_multi(torch.as_tensor(B).cuda(), X)
        ctx.sync()
        Xr = ref(B)
        assert np.abs(X.cpu().numpy() - Xr).max() <= 1e-9 * np.abs(Xr).max(), m
    # bitwise reproducible (round 4: coloured update phases, slot-ordered forward sweeps -- no atomics): a second solve with the same
    # factor AND a second factorisation of the same matrix give the same bits, single vector and block
    x2 = torch.zeros(n, dtype=torch.float64, device="cuda")
    F.solve(torch.as_tensor(b).cuda(), x2)
    ctx.sync()
    assert torch.equal(x2, x)
    F2 = ddm.Ilu0(ctx, A, rl.block_ptr, direct=True)
    x3 = torch.zeros(n, dtype=torch.float64, device="cuda")
    F2.solve(torch.as_tensor(b).cuda(), x3)
    X2 = torch.zeros((n, 48), dtype=torch.float64, device="cuda")
    F2.solve_multi(torch.as_tensor(B).cuda(), X2)
    ctx.sync()
    assert F2.status() == 0
    assert torch.equal(x3, x) and torch.equal(X2, X)
    steps, om = F.refinement()
    assert steps <= 1 and om[steps] < 1e-12, (steps, om)      # SPD: the probe is at rounding level at once (or after one step)
    ctx.close()


def test_device_single_solve_survives_wider_multi_solve(ddm, torch_cuda, device_engine):
    """the single-vector HIP graph holds the backward sweep's scratch pointer; a block solve in between re-allocates that
    scratch for more columns.  The SAME d / x tensors are used before and after, so a stale graph would be replayed as is."""
    import torch
    dec, rl = _blocks(ddm, (21, 20, 19), (2, 2, 2))
    ctx = ddm.torch_context(0)
    F = ddm.Ilu0(ctx, ddm.CsrMatrix(ctx, rl.A_dir), rl.block_ptr, direct=True)
    n = rl.n
    rng = np.random.default_rng(17)
    d = torch.as_tensor(rng.standard_normal(n)).cuda()
    x = torch.zeros(n, dtype=torch.float64, device="cuda")
    F.solve(d, x)
    ctx.sync()
    x0 = x.clone()
    for m in (24, 48):
        B = torch.as_tensor(rng.standard_normal((n, m))).cuda()
        X = torch.zeros((n, m), dtype=torch.float64, device="cuda")
        F.solve_multi(B, X)
        ctx.sync()
        x.zero_()
        F.solve(d, x)                                      # same pointers as the captured graph
        ctx.sync()
        assert F.status() == 0
        assert (x - x0).abs().max().item() <= 1e-12 * x0.abs().max().item()
    ctx.close()


def test_device_cholesky_rejects_indefinite_matrix(ddm, torch_cuda, device_engine):
    dec, rl = _blocks(ddm, (13, 12, 11), (1, 1, 1), overlap=1)
    M = sp.csr_matrix(rl.A_dir).copy()
    M.setdiag(-np.abs(M.diagonal()))
    ctx = ddm.torch_context(0)
    A = ddm.CsrMatrix(ctx, M)
    with pytest.raises(ddm.DdmError) as e:
        ddm.Ilu0(ctx, A, rl.block_ptr, direct=True)
    assert e.value.code == ddm.DDM_ENUMERIC and "positive definite" in str(e.value)
    ctx.close()


def test_schwarz_with_device_cholesky_matches_oracle(ddm, torch_cuda, device_engine):
    """[schwarz.subdomain_solver] type = cholmod (examples/poisson.ini:23) served by the device engine: two-level CG against the
    oracle with exact local solves -- identical iteration count, residual history within the direct-solver tolerance."""
    from dune_ddm_amd.solver import TwoLevelSchwarz
    from tests.oracle_bridge import oracle_solve
    dec, rl = _blocks(ddm, (25, 23, 21), (2, 2, 2))
    tl = TwoLevelSchwarz(dec, coarse="pou", schwarz_type="standard", mode="additive", subdomain_solver="cholmod")
    res, hist, x = tl.solve(reduction=1e-10, maxit=200)
    tl.prec.check_status()
    it, conv, hist_o, xo = oracle_solve(dec, reduction=1e-10, maxit=200, coarse="pou", schwarz_type="standard", mode="additive", local_solver="direct")
    ho = np.asarray(hist_o)
    assert res.converged and conv and res.iterations == it, (res.iterations, it)
    # per-iteration parity while the residual is large; afterwards CG amplifies the rounding-level difference between two exact local
    # solvers (device Cholesky vs SuperLU: 1e-13 per application) like it amplifies a change of summation order
    # (tests/test_oracle_order_sensitivity.py): same curve within a factor 2, same count, same solution
    early = ho >= 1e-4 * ho[0]
    assert early.sum() >= 10 and (np.abs(hist - ho)[early] <= 1e-7 * ho[early] + 1e-11 * ho[0]).all(), float(np.max(np.abs(hist - ho)[early] / ho[early]))
    assert (np.abs(np.log(hist / ho)) < np.log(2.0)).all()
    xr = np.concatenate(xo)
    assert np.abs(x.cpu().numpy() - xr).max() <= 1e-8 * np.abs(xr).max()
    tl.ctx.close()


def test_device_cholesky_64_cubed_subdomain(ddm, torch_cuda, device_engine):
    """A 64^3 3-D subdomain (262 144 rows, 27-point stencil) factorised on the device: the solution satisfies the
    system to 1e-10 (SuperLU needs minutes and tens of GB for this size; parity against it is the 1e-9 test above at 33 x 31 x 29)."""
    import torch
    from dune_ddm_amd import synth
    from dune_ddm_amd.problem import build_structured
    dec = build_structured(synth.StructuredPoisson((64, 64, 64), (1, 1, 1)), overlap=1, pou_type="distance")
    M = sp.csr_matrix(dec.subs[0].A_dir)
    n = M.shape[0]
    assert n == 64 ** 3
    sym = ddm.sn_symbolic_host(M)[0]
    ctx = ddm.torch_context(0)
    A = ddm.CsrMatrix(ctx, M)
    t0 = time.perf_counter()
    F = ddm.Ilu0(ctx, A, [0, n], direct=True)
    ctx.sync()
    t_fac = time.perf_counter() - t0
    rng = np.random.default_rng(5)
    B = rng.standard_normal((n, 24))
    X = torch.zeros((n, 24), dtype=torch.float64, device="cuda")
    Bd = torch.as_tensor(B).cuda()
    F.solve_multi(Bd, X)
    ctx.sync()
    t1 = time.perf_counter()
    for _ in range(5):
        F.solve_multi(Bd, X)
    ctx.sync()
    t_solve = (time.perf_counter() - t1) / 5
    Xh = X.cpu().numpy()
    R = M @ Xh - B
    rel = np.linalg.norm(R, axis=0) / np.linalg.norm(B, axis=0)
    print(f"[sn 64^3] {len(sym['first']) - 1} supernodes, {sym['levels']} levels, {sym['entries'] * 8e-9:.2f} GB of panels, {sym['flops']:.3g} multiply-adds; "
          f"ordering + analysis + numeric factorisation {t_fac:.2f} s; 24-column solve {1e3 * t_solve:.1f} ms; worst relative residual {rel.max():.2e}")
    assert rel.max() <= 1e-10
    assert F.status() == 0
    # single vector: ONE block, so the persistent kernel of the top levels runs in its placement-independent mode (all workgroups
    # one group, write-through hand-overs); timing printed for the record
    bd = Bd[:, 0].contiguous()
    xs = torch.zeros(n, dtype=torch.float64, device="cuda")
    F.solve(bd, xs)
    ctx.sync()
    t2 = time.perf_counter()
    for _ in range(20):
        F.solve(bd, xs)
    ctx.sync()
    t_one = (time.perf_counter() - t2) / 20
    r1 = np.linalg.norm(M @ xs.cpu().numpy() - B[:, 0]) / np.linalg.norm(B[:, 0])
    print(f"[sn 64^3] single-vector solve {1e3 * t_one:.2f} ms, relative residual {r1:.2e}")
    assert r1 <= 1e-10 and F.status() == 0
    ctx.close()


@pytest.mark.parametrize("parts", [(2, 2, 2), (1, 1, 1), (3, 2, 2)])
def test_single_vector_top_kernel_matches_level_kernels(ddm, torch_cuda, device_engine, parts, monkeypatch):
    """The persistent kernels for the top levels of the tree (sn_solve1.hpp; 8 blocks: one XCD per block, 1 block: all workgroups one
    group, 12 blocks: XCDs with two blocks) -- the separators as dense chains with inverted triangles (default) and link by link
    (DDM_SN_CHAINS=0) -- against the level-by-level launches (DDM_SN_TOP_MAX=0): same solution to rounding (the paths split the
    sums differently), each path bitwise reproducible."""
    import torch
    dec, rl = _blocks(ddm, (37, 35, 33), parts)
    n = rl.n
    rng = np.random.default_rng(23)
    d = torch.as_tensor(rng.standard_normal(n)).cuda()
    out = {}
    for top, chains in (("128", "1"), ("128", "0"), ("0", "1")):     # dense chains / link by link / level launches only
        monkeypatch.setenv("DDM_SN_TOP_MAX", top)
        monkeypatch.setenv("DDM_SN_CHAINS", chains)
        ctx = ddm.torch_context(0)
        F = ddm.Ilu0(ctx, ddm.CsrMatrix(ctx, rl.A_dir), rl.block_ptr, direct=True)
        x = torch.zeros(n, dtype=torch.float64, device="cuda")
        y = torch.zeros(n, dtype=torch.float64, device="cuda")
        F.solve(d, x)
        F.solve(d, y)
        ctx.sync()
        assert F.status() == 0 and torch.equal(x, y)
        out[(top, chains)] = x.cpu().numpy()
        ctx.close()
    M = sp.csr_matrix(rl.A_dir)
    ref = out[("0", "1")]
    for key in (("128", "1"), ("128", "0")):
        assert np.abs(out[key] - ref).max() <= 1e-12 * np.abs(ref).max(), key
        assert np.linalg.norm(M @ out[key] - d.cpu().numpy()) <= 1e-11 * np.linalg.norm(d.cpu().numpy()), key


@pytest.mark.parametrize("case", ["dg", "pivoting"])
def test_device_lu_matches_superlu(ddm, torch_cuda, device_engine, case):
    """The L U variant of the device engine (`type = umfpack`, general = 1): non-symmetric values on the symmetric pattern, threshold
    partial pivoting inside the diagonal blocks of the supernodes (UMFPACK's default 0.1).  "dg": the convection-diffusion DG operator
    of BASELINE configs[3] (8 subdomains as diagonal blocks); "pivoting": a 3-D stencil matrix whose diagonal is made tiny in a third
    of the rows, so that the diagonal pivot fails the threshold test and rows ARE exchanged.  Against SuperLU to 1e-9."""
    import torch
    from dune_ddm_amd import synth
    from dune_ddm_amd.problem import RankLocal, build_structured
    rng = np.random.default_rng(11)
    if case == "dg":
        dec = build_structured(synth.StructuredDG2D((48, 48), (4, 2)), overlap=2)
        rl = RankLocal(dec, 0, 1)
        M, bp = sp.csr_matrix(rl.A_dir), rl.block_ptr
    else:
        dec, rl = _blocks(ddm, (19, 18, 17), (1, 1, 1), overlap=1)
        M = sp.csr_matrix(rl.A_dir).tolil()
        n0 = M.shape[0]
        # non-symmetric off-diagonal perturbation + tiny diagonal entries in every third row
        M = sp.csr_matrix(M)
        M.data = M.data * (1.0 + 0.3 * rng.standard_normal(len(M.data)))
        d = M.diagonal()
        d[::3] *= 1e-6
        M.setdiag(d)
        M, bp = sp.csr_matrix(M), np.array([0, n0], dtype=np.int64)
    assert abs(M - M.T).max() > 1e-3 * abs(M).max()
    n = M.shape[0]
    ctx = ddm.torch_context(0)
    A = ddm.CsrMatrix(ctx, M)
    F = ddm.Ilu0(ctx, A, bp, direct=True, general=True)
    lus = [spl.splu(sp.csc_matrix(M[int(bp[i]):int(bp[i + 1]), int(bp[i]):int(bp[i + 1])])) for i in range(len(bp) - 1)]

    def ref(b):
        out = np.empty_like(b)
        for i, lu in enumerate(lus):
            out[int(bp[i]):int(bp[i + 1])] = lu.solve(b[int(bp[i]):int(bp[i + 1])])
        return out

    b = rng.standard_normal(n)
    x = torch.zeros(n, dtype=torch.float64, device="cuda")
    F.solve(torch.as_tensor(b).cuda(), x)
    ctx.sync()
    assert F.status() == 0
    xr = ref(b)
    assert np.abs(x.cpu().numpy() - xr).max() <= 1e-9 * np.abs(xr).max(), float(np.abs(x.cpu().numpy() - xr).max() / np.abs(xr).max())
    B = rng.standard_normal((n, 20))
    X = torch.zeros((n, 20), dtype=torch.float64, device="cuda")
    F.solve_multi(torch.as_tensor(B).cuda(), X)
    ctx.sync()
    Xr = ref(B)
    assert np.abs(X.cpu().numpy() - Xr).max() <= 1e-9 * np.abs(Xr).max()
    res = np.abs(M @ X.cpu().numpy() - B).max() / np.abs(B).max()
    res_ref = np.abs(M @ Xr - B).max() / np.abs(B).max()
    # "pivoting": rows are only exchanged INSIDE the diagonal block of a supernode (static structure), so a tiny diagonal entry whose
    # large partners sit in the rows below still produces element growth (~1e6 here): 5e-9 backward error without refinement (round 3).
    # Round 4: iterative refinement with the stopping rule of dune/ddm/eigensolvers/umfpack.hh:42-129, fixed per factor on a probe
    # right-hand side -- the solves now reach SuperLU's level
    steps, om = F.refinement()
    anorm = abs(M).sum(axis=1).max()
    Xh = X.cpu().numpy()
    omega = np.linalg.norm(M @ Xh - B) / (anorm * np.linalg.norm(Xh) + np.linalg.norm(B))
    print(f"[sn lu {case}] residual {res:.2e} (SuperLU {res_ref:.2e}); refinement steps {steps}, probe backward errors {om[:steps + 1]}, block backward error {omega:.2e}")
    # (om = max(normwise backward error, 1e-2 x componentwise backward error) of the probe right-hand side)
    assert om[steps] <= 1e-12 and omega <= 1e-12, (steps, om, omega)
    if case == "pivoting":
        assert steps >= 1 and om[0] > 1e-13                     # (the unrefined factor is what round 3 shipped: residual 5e-9)
    assert res <= 2e-11, (res, res_ref)
    # the same factor twice: same bits (no atomics)
    F2 = ddm.Ilu0(ctx, A, bp, direct=True, general=True)
    x2 = torch.zeros(n, dtype=torch.float64, device="cuda")
    F2.solve(torch.as_tensor(b).cuda(), x2)
    ctx.sync()
    assert torch.equal(x2, x)
    ctx.close()
