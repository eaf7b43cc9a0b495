"""-m gpu parity tests: the HIP path (through the C ABI of libddm_hip.so) against the CPU oracle
on the same seeded inputs.  FP tolerance is stated per test; integer data is compared exactly."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu

RTOL_VEC = 1e-12      # one kernel application (different summation order only)
RTOL_HIST = 1e-8      # per-iteration ||r_k||: |gpu - oracle| <= RTOL_HIST*||r_k|| + ATOL_HIST*||r_0||
ATOL_HIST = 1e-14     # (rounding differences are amplified by CG once ||r_k|| approaches 1e-10 ||r_0||)


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device (no CPU fallback exists)"
    return torch


def _dev(torch, a):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64)).cuda()


def _relerr(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def _random_csr(rng, n, density=0.01, long_row=None):
    M = sp.random(n, n, density=density, random_state=np.random.RandomState(rng.integers(1 << 30)), format="lil")
    if long_row is not None:
        M[long_row, :] = rng.standard_normal(n)
    M = sp.csr_matrix(M)
    M.data = rng.standard_normal(M.nnz)
    M.sort_indices()
    return M


def test_csr_mv_usmv_random_and_edge_cases(ddm, torch_cuda):
    from oracle import apply_oracle as ao
    torch = torch_cuda
    ctx = ddm.torch_context(0)
    rng = np.random.default_rng(42)
    cases = [_random_csr(rng, 5000, 0.004), _random_csr(rng, 3000, 0.002, long_row=17),   # a row longer than the LDS tile
             sp.csr_matrix((700, 700)),                                                    # empty matrix (ragged: all rows empty)
             sp.eye(1, format="csr")]
    for M in cases:
        n = M.shape[0]
        x = rng.standard_normal(n)
        y0 = rng.standard_normal(n)
        A = ddm.CsrMatrix(ctx, M)
        xd, yd = _dev(torch, x), _dev(torch, y0)
        A.mv(xd, yd)
        ctx.sync()
        yo = np.zeros(n)
        ao.Csr(M).mv(x, yo)
        assert _relerr(yd.cpu().numpy(), yo) < RTOL_VEC or np.abs(yo).max() == 0
        yd = _dev(torch, y0)
        A.usmv(-0.75, xd, yd)
        ctx.sync()
        yo = y0.copy()
        ao.Csr(M).usmv(-0.75, x, yo)
        assert np.max(np.abs(yd.cpu().numpy() - yo)) <= RTOL_VEC * max(1.0, np.abs(yo).max())
    ctx.close()


@pytest.mark.parametrize("trsv_mode", ["box", "pipe", "xcd2", "levels"])
def test_ilu0_factor_and_solve(ddm, torch_cuda, trsv_mode, monkeypatch):
    """the four triangular-solve engines: box (default where the blocks start with a structured box), pipe (any matrix), xcd2 (fallback
    for matrices pipe declines), one launch per level"""
    monkeypatch.setenv("DDM_TRSV_MODE", trsv_mode)
    from dune_ddm_amd import synth
    from oracle import apply_oracle as ao
    torch = torch_cuda
    ctx = ddm.torch_context(0)
    grid = synth.StructuredPoisson((14, 12, 11), (1, 1, 1), synth.islands_kappa((13, 11, 10), 1e4, 4, 2))
    M = grid.subdomain(0).A
    A = ddm.CsrMatrix(ctx, M)
    F = ddm.Ilu0(ctx, A)
    ref = ao.Ilu0(ao.Csr(M))
    assert _relerr(F.factors(), ref.lu) < 1e-13            # same elimination order on the host
    rng = np.random.default_rng(1)
    d = rng.standard_normal(M.shape[0])
    xd = torch.zeros(M.shape[0], dtype=torch.float64, device="cuda")
    dd = _dev(torch, d)
    for _ in range(3):                                     # later calls replay the captured graph (counters re-zeroed)
        xd.fill_(float("nan"))
        F.solve(dd, xd)
    ctx.sync()
    assert F.status() == 0
    xo = np.zeros(M.shape[0])
    ref.apply(xo, d)
    assert _relerr(xd.cpu().numpy(), xo) < RTOL_VEC
    assert F.num_levels(False) > 1 and F.num_levels(True) > 1
    with pytest.raises(ddm.DdmError):
        F.solve(dd, dd)                                    # aliasing is rejected
    # zero pivot is reported, not silently produced
    Z = sp.csr_matrix(np.array([[0.0, 1.0], [1.0, 0.0]]))
    with pytest.raises(ddm.DdmError):
        ddm.Ilu0(ctx, ddm.CsrMatrix(ctx, sp.csr_matrix((np.array([0.0, 1.0, 1.0, 1.0]), np.array([0, 1, 0, 1]), np.array([0, 2, 4])), shape=(2, 2))))
    ctx.close()


def _build(ddm, N, P, overlap=2, pou_type="distance", shrink=0, kappa=None, neumann=False):
    from dune_ddm_amd import synth
    from dune_ddm_amd.problem import build_structured
    grid = synth.StructuredPoisson(N, P, kappa)
    return build_structured(grid, overlap=overlap, pou_type=pou_type, shrink=shrink, neumann=neumann)


def test_operator_dot_and_preconditioner_applies(ddm, torch_cuda):
    """a1, a2, a3, a4, a6, a9 applied once to a random consistent vector."""
    from dune_ddm_amd.solver import TwoLevelSchwarz
    from tests.oracle_bridge import oracle_objects
    torch = torch_cuda
    dec = _build(ddm, (13, 12, 11), (2, 2, 2))
    rng = np.random.default_rng(7)
    xg = rng.standard_normal(dec.nglobal)                  # consistent: same value on every holder
    for stype, mode in (("standard", "additive"), ("restricted", "additive"), ("standard", "multiplicative")):
        tl = TwoLevelSchwarz(dec, coarse="pou", schwarz_type=stype, mode=mode)
        op, sp_, prec, sch, gal = oracle_objects(dec, schwarz_type=stype, mode=mode, coarse="pou")
        xs = [xg[sd.glob[:sd.n_o]] for sd in dec.subs]
        xd = tl.to_device(tl.rl.cat_novlp(xs))
        yd = tl.zeros(tl.rl.n_o)
        tl.op.apply(xd, yd)
        yo = [np.zeros(sd.n_o) for sd in dec.subs]
        op.apply(xs, yo)
        tl.ctx.sync()
        assert _relerr(yd.cpu().numpy(), np.concatenate(yo)) < RTOL_VEC
        tl.op.applyscaleadd(-0.5, xd, yd)
        op.applyscaleadd(-0.5, xs, yo)
        tl.ctx.sync()
        assert _relerr(yd.cpu().numpy(), np.concatenate(yo)) < RTOL_VEC
        assert abs(tl.op.dot(xd, yd) - sp_.dot(xs, yo)) < 1e-12 * abs(sp_.dot(xs, yo))
        assert abs(tl.op.norm(xd) - sp_.norm(xs)) < 1e-13 * sp_.norm(xs)
        # Schwarz / Galerkin / Combined on the (consistent) vector y
        for dev_prec, ora_prec in ((tl.schwarz, sch), (tl.galerkin, gal), (tl.prec, prec)):
            zd = tl.zeros(tl.rl.n_o)
            dev_prec.apply(zd, yd)
            zo = [np.zeros(sd.n_o) for sd in dec.subs]
            ora_prec.apply(zo, [v.copy() for v in yo])
            tl.ctx.sync()
            assert _relerr(zd.cpu().numpy(), np.concatenate(zo)) < 1e-10
        # coarse matrix assembled on the device == oracle's build_solver
        assert _relerr(tl.a0, gal.a0.toarray()) < 1e-12
        tl.ctx.close()


@pytest.mark.parametrize("cfg", [
    dict(N=(17, 17, 17), P=(2, 2, 2), overlap=2, coarse="pou", stype="standard", mode="additive"),
    dict(N=(21, 13, 9), P=(3, 2, 1), overlap=1, coarse="none", stype="standard", mode="additive"),
    dict(N=(40, 33), P=(2, 2), overlap=2, coarse="none", stype="standard", mode="additive"),       # BASELINE config 1 (2-D, one-level)
    # multiplicative combination is not symmetric: CG need not converge, compare 25 iterations of the recurrences
    dict(N=(15, 15, 15), P=(2, 2, 2), overlap=2, coarse="pou", stype="standard", mode="multiplicative", maxit=25),
    dict(N=(20, 20, 20), P=(1, 1, 1), overlap=1, coarse="none", stype="standard", mode="additive"),  # BASELINE config 2 shape: ILU(0)-CG
])
def test_cg_history_matches_oracle(ddm, torch_cuda, cfg):
    """Iteration count identical, per-iteration residual norms within 1e-8 relative."""
    from dune_ddm_amd.solver import TwoLevelSchwarz
    from tests.oracle_bridge import oracle_solve
    dec = _build(ddm, cfg["N"], cfg["P"], overlap=cfg["overlap"])
    tl = TwoLevelSchwarz(dec, coarse=cfg["coarse"], schwarz_type=cfg["stype"], mode=cfg["mode"])
    maxit = cfg.get("maxit", 300)
    res, hist, x = tl.solve(reduction=1e-10, maxit=maxit)
    it, conv, hist_o, xo = oracle_solve(dec, reduction=1e-10, maxit=maxit, coarse=cfg["coarse"], schwarz_type=cfg["stype"], mode=cfg["mode"])
    assert res.iterations == it and bool(res.converged) == conv
    hist_o = np.array(hist_o)
    assert (np.abs(hist - hist_o) <= RTOL_HIST * hist_o + ATOL_HIST * hist_o[0]).all()
    if "maxit" not in cfg:
        assert conv and res.reduction <= 1e-10
        assert _relerr(x.cpu().numpy(), np.concatenate(xo)) < 1e-8
    tl.ctx.close()


def test_halo_exchange_bit_exact(ddm, torch_cuda):
    """copy and add exchanges reproduce the oracle's communication bit for bit (same summation order)."""
    from dune_ddm_amd.problem import RankLocal
    from oracle import apply_oracle as ao
    torch = torch_cuda
    dec = _build(ddm, (12, 11, 10), (2, 2, 2), overlap=2)
    rl = RankLocal(dec)
    ctx = ddm.torch_context(0)
    rng = np.random.default_rng(3)
    ocomm = ao.Comm(dec.nsub, dec.ovlp_owner, dec.ovlp_all, [sd.owner_ovlp for sd in dec.subs])
    for mode, plan, fn in ((ddm.Halo.COPY, rl.plan_ovlp_copy, ocomm.copyOwnerToAll), (ddm.Halo.ADD, rl.plan_ovlp_add, ocomm.addOwnerCopyToAll)):
        vs = [rng.standard_normal(sd.n) for sd in dec.subs]
        h = ddm.Halo(ctx, 7, mode, plan)
        vd = _dev(torch, np.concatenate(vs))
        h.exchange(vd)
        ctx.sync()
        fn(vs)
        assert (vd.cpu().numpy() == np.concatenate(vs)).all()
    ctx.close()


@pytest.mark.parametrize("cfg", [
    # the reference's shipped configuration: restricted additive Schwarz + multiplicative coarse level + GMRES (examples/poisson.ini)
    dict(N=(17, 16, 15), P=(2, 2, 2), overlap=2, stype="restricted", mode="multiplicative", restart=100),
    dict(N=(17, 16, 15), P=(2, 2, 2), overlap=2, stype="restricted", mode="additive", restart=6),       # exercises restarts
    dict(N=(30, 27), P=(2, 2), overlap=1, stype="standard", mode="additive", restart=30),               # TwoLevelSchwarzSolver default (twolevel_schwarz.hh:121-130)
])
def test_gmres_history_matches_oracle(ddm, torch_cuda, cfg):
    """restartedgmressolver: identical iteration count, per-iteration (preconditioned) defect norms within tolerance."""
    from dune_ddm_amd.solver import TwoLevelSchwarz
    from tests.oracle_bridge import oracle_solve
    dec = _build(ddm, cfg["N"], cfg["P"], overlap=cfg["overlap"])
    tl = TwoLevelSchwarz(dec, coarse="pou", schwarz_type=cfg["stype"], mode=cfg["mode"])
    res, hist, x = tl.solve(reduction=1e-10, maxit=200, solver="restartedgmressolver", restart=cfg["restart"])
    it, conv, hist_o, xo = oracle_solve(dec, reduction=1e-10, maxit=200, solver="restartedgmressolver", restart=cfg["restart"], coarse="pou",
                                        schwarz_type=cfg["stype"], mode=cfg["mode"])
    assert res.converged and conv and res.iterations == it
    ho = np.array(hist_o)
    assert (np.abs(hist - ho) <= RTOL_HIST * ho + 1e-12 * ho[0]).all()
    assert _relerr(x.cpu().numpy(), np.concatenate(xo)) < 1e-8
    with pytest.raises(NotImplementedError):
        tl.solve(solver="minressolver")
    tl.ctx.close()


@pytest.mark.parametrize("cfg", [
    # the reference's shipped local solver (examples/poisson.ini:23 `type = cholmod`): sparse Cholesky of A_dir
    dict(kind="poisson", solver="cholmod", stype="restricted", mode="multiplicative", krylov="restartedgmressolver"),
    dict(kind="poisson", solver="direct", stype="standard", mode="additive", krylov="cgsolver"),
    # non-symmetric DG operator: `type = umfpack` -> L U without pivoting; "direct" picks it from the values
    dict(kind="dg", solver="umfpack", stype="standard", mode="additive", krylov="restartedgmressolver"),
    dict(kind="dg", solver="direct", stype="restricted", mode="additive", krylov="restartedgmressolver"),
])
def test_direct_subdomain_solver_matches_oracle(ddm, torch_cuda, cfg):
    """Schwarz with the sparse direct local solver (host factorisation, device triangular solves in the fill-reducing order) against
    the oracle's SchwarzPreconditioner with an exact local solve (scipy SuperLU): identical iteration counts, residual histories
    within the stated tolerance, one preconditioner application to 1e-9."""
    from dune_ddm_amd import synth
    from dune_ddm_amd.problem import build_structured
    from dune_ddm_amd.solver import TwoLevelSchwarz
    from tests.oracle_bridge import oracle_objects, oracle_solve
    if cfg["kind"] == "poisson":
        dec = _build(ddm, (17, 16, 15), (2, 2, 2))
    else:
        dec = build_structured(synth.StructuredDG2D((24, 24), (2, 2)), overlap=2)
    tl = TwoLevelSchwarz(dec, coarse="pou", schwarz_type=cfg["stype"], mode=cfg["mode"], subdomain_solver=cfg["solver"])
    op, sp_, prec, sch, gal = oracle_objects(dec, schwarz_type=cfg["stype"], mode=cfg["mode"], coarse="pou", local_solver="direct")
    rng = np.random.default_rng(3)
    xg = rng.standard_normal(dec.nglobal)
    ys = [xg[sd.glob[:sd.n_o]] for sd in dec.subs]
    zd = tl.zeros(tl.rl.n_o)
    tl.schwarz.apply(zd, tl.to_device(tl.rl.cat_novlp(ys)))
    zo = [np.zeros(sd.n_o) for sd in dec.subs]
    sch.apply(zo, [v.copy() for v in ys])
    tl.ctx.sync()
    tl.schwarz.check_status()
    assert _relerr(zd.cpu().numpy(), np.concatenate(zo)) < 1e-9
    res, hist, x = tl.solve(reduction=1e-10, maxit=200, solver=cfg["krylov"], restart=50)
    it, conv, hist_o, xo = oracle_solve(dec, reduction=1e-10, maxit=200, solver=cfg["krylov"], restart=50, coarse="pou", schwarz_type=cfg["stype"],
                                        mode=cfg["mode"], local_solver="direct")
    ho = np.array(hist_o)
    assert res.converged and conv and res.iterations == it, (res.iterations, it)
    assert (np.abs(hist - ho) <= 1e-7 * ho + 1e-11 * ho[0]).all()
    assert _relerr(x.cpu().numpy(), np.concatenate(xo)) < 1e-7
    tl.ctx.close()


@pytest.mark.parametrize("cfg", [
    dict(kind="poisson", stype="restricted", mode="multiplicative", solver="ilu0"),     # non-symmetric preconditioner
    dict(kind="dg", stype="standard", mode="additive", solver="umfpack"),               # non-symmetric operator (configs[3])
])
def test_bicgstab_history_matches_oracle(ddm, torch_cuda, cfg):
    """[solver] type = bicgstabsolver: dune-istl's BiCGSTAB recurrences on the device against the oracle's restatement: identical
    half-step count, defect norm after every half step within 1e-7 ||r_k|| + 1e-11 ||r_0|| (BiCGSTAB amplifies rounding more than CG)."""
    from dune_ddm_amd import synth
    from dune_ddm_amd.problem import build_structured
    from dune_ddm_amd.solver import TwoLevelSchwarz
    from tests.oracle_bridge import oracle_solve
    dec = _build(ddm, (17, 16, 15), (2, 2, 2)) if cfg["kind"] == "poisson" else build_structured(synth.StructuredDG2D((24, 24), (2, 2)), overlap=2)
    # random (consistent) right-hand side: the load vector of f = 1 lies in the span of the POU coarse space, the multiplicative coarse
    # correction makes the first residual orthogonal to it and <r~, r_1> vanishes -- a genuine BiCGSTAB breakdown (both sides then
    # iterate on rounding noise), not a parity case
    bg = np.random.default_rng(12).standard_normal(dec.nglobal)
    for sd in dec.subs:
        sd.b = np.where(sd.dirichlet_ovlp[:sd.n_o] > 0, 0.0, bg[sd.glob[:sd.n_o]])
    tl = TwoLevelSchwarz(dec, coarse="pou", schwarz_type=cfg["stype"], mode=cfg["mode"], subdomain_solver=cfg["solver"])
    res, hist, x = tl.solve(reduction=1e-9, maxit=200, solver="bicgstabsolver")
    it, conv, hist_o, xo = oracle_solve(dec, reduction=1e-9, maxit=200, solver="bicgstabsolver", coarse="pou", schwarz_type=cfg["stype"], mode=cfg["mode"],
                                        local_solver="ilu0" if cfg["solver"] == "ilu0" else "direct")
    ho = np.array(hist_o)
    assert res.converged and conv and res.iterations == it and len(hist) == len(ho), (res.iterations, it, len(hist), len(ho))
    assert (np.abs(hist - ho) <= 1e-7 * ho + 1e-11 * ho[0]).all(), float(np.max(np.abs(hist - ho) / ho))
    assert _relerr(x.cpu().numpy(), np.concatenate(xo)) < 1e-7
    tl.ctx.close()
