"""-m gpu: the hot-path kernels at BASELINE.json's full size (216^3 = 10 M DoF, 8 overlapping subdomains, ILU(0)),
checked through properties that need no oracle run of that size (the sequential oracle takes minutes there):

* the triangular solve inverts the factors it was given:  L (U x) = d  to rounding, with L, U taken from the library's
  own ILU(0) (whose values are checked against the oracle at small sizes in test_gpu_parity.py / test_gpu_pipe.py);
* it is linear:  solve(a d1 + b d2) = a solve(d1) + b solve(d2)  to rounding;
* repeated solves of one right-hand side are bit-identical (the schedule is deterministic, no atomics in the sums);
* the CSR-stream SpMV is bit-exact against scipy's sequential row sums (same summation order)."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def test_full_size_triangular_solve_and_spmv(ddm):
    import torch
    from dune_ddm_amd import synth
    from dune_ddm_amd.problem import build_structured
    assert torch.cuda.is_available()
    N = 216
    dec = build_structured(synth.StructuredPoisson((N, N, N), (2, 2, 2)), overlap=2, pou_type="distance", shrink=0)
    mats = [sd.A_dir.tocsr() for sd in dec.subs]
    M = sp.block_diag(mats, format="csr")
    M.sort_indices()
    block_ptr = np.concatenate([[0], np.cumsum([m.shape[0] for m in mats])]).astype(np.int64)
    del mats, dec
    n = M.shape[0]
    assert n > 10_000_000 and M.nnz > 250_000_000 and len(block_ptr) == 9
    ctx = ddm.torch_context(0)
    A = ddm.CsrMatrix(ctx, M)
    # --- SpMV, bit-exact (scipy sums a row's products in column order, as k_spmv_stream does) ---
    rng = np.random.default_rng(11)
    xh = rng.standard_normal(n)
    xd = torch.as_tensor(xh).cuda()
    yd = torch.zeros_like(xd)
    A.mv(xd, yd)
    ctx.sync()
    assert np.array_equal(yd.cpu().numpy(), M @ xh)
    # --- ILU(0) of the block-diagonal matrix (blocks = subdomains) ---
    F = ddm.Ilu0(ctx, A, block_ptr)
    d1, d2 = rng.standard_normal(n), rng.standard_normal(n)
    outs = []
    xs = torch.zeros(n, dtype=torch.float64, device="cuda")
    for rhs in (d1, d2, 0.75 * d1 - 1.25 * d2, d1):
        xs.fill_(float("nan"))
        F.solve(torch.as_tensor(rhs).cuda(), xs)
        ctx.sync()
        assert F.status() == 0
        outs.append(xs.cpu().numpy().copy())
    x1, x2, x12, x1b = outs
    assert np.array_equal(x1, x1b)                                                  # deterministic
    scale = max(np.abs(x1).max(), np.abs(x2).max())
    assert np.abs(x12 - (0.75 * x1 - 1.25 * x2)).max() <= 1e-10 * scale             # linear (to rounding)
    # L (U x) = d with the library's factors: multipliers below the diagonal, U above, INVERSE pivots on it
    lu = F.factors()
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(M.indptr))
    cols = M.indices
    low, up, dg = cols < rows, cols > rows, cols == rows
    assert dg.sum() == n
    Ux = sp.csr_matrix((np.where(up, lu, 0.0), M.indices, M.indptr), shape=M.shape) @ x1 + x1 / lu[dg]
    LUx = sp.csr_matrix((np.where(low, lu, 0.0), M.indices, M.indptr), shape=M.shape) @ Ux + Ux
    assert np.abs(LUx - d1).max() <= 1e-10 * np.abs(d1).max()
    ctx.close()



def test_full_length_cg_parity_geneo_96(ddm):
    """BASELINE configs[2] at 96^3 = 884 736 DoF (8 subdomains, overlap 2, ILU(0), GenEO nev = 20 -> K = 160, additive, CG to
    1e-10, examples/poisson.ini:14), the HIP solve against the oracle run to CONVERGENCE (8 host threads) with the same coarse basis.
    What two correct floating-point CG runs with different summation orders can agree on, and what is asserted:
      * while ||r_k|| >= 2e-3 ||r_0|| (the first ~40 iterations): | ||r_k||_hip - ||r_k||_oracle | <= 1e-8 ||r_k||
        (measured 1e-14 at k = 20, 2e-10 at k = 40);
      * afterwards CG amplifies the rounding-level differences (10^6-term reductions in another order, explicit replicated inverse
        of the coarse matrix vs LU).  That this is a property of CG in floating point and not of the HIP path is DEMONSTRATED
        here, not assumed: the oracle is run a second and third time with nothing changed but the order of the additions inside
        its global dot products (descending, pairwise: tests/test_oracle_order_sensitivity.py has the CPU-only version); the
        running maximum of the relative deviation between those ORACLE runs is the envelope two correct implementations drift
        apart by.  Asserted: at every iteration the HIP-vs-oracle deviation is at most ENVELOPE_FACTOR x that envelope
        (floored at 1e-8), and the HIP curve is not an outlier of the family {oracle, oracle-descending, oracle-pairwise};
      * iteration counts equal, or different by ONE because the iterate that sits on the threshold 1e-10 ||r_0|| is just below it
        on one side and just above on the other (measured: 141 HIP / 142 oracle, ||r_141|| = 0.97e-10 vs 1.0e-10 ||r_0||)."""
    from dune_ddm_amd import synth
    from dune_ddm_amd.geneo import geneo_basis
    from dune_ddm_amd.problem import build_structured
    from dune_ddm_amd.solver import TwoLevelSchwarz
    from oracle import apply_oracle as ao
    from tests.oracle_bridge import oracle_solve
    dec = build_structured(synth.StructuredPoisson((96, 96, 96), (2, 2, 2)), overlap=2, pou_type="distance", shrink=0, neumann=True)
    tl = TwoLevelSchwarz(dec, coarse="none")
    basis, info = geneo_basis(tl, nev=20, return_info=True)
    assert info["converged"]
    # the device-built GenEO pairs checked on the host, independently of the device path AND of the oracle (scipy only): all 160
    # pairs satisfy A_neu x = lambda D B_neu D x to the eigensolver tolerance (the oracle below is handed this basis)
    from dune_ddm_amd.geneo import host_eigenpair_residuals
    worst_res = worst_rq = 0.0
    for sd in dec.subs:
        r_, q_ = host_eigenpair_residuals(sd, basis[sd.id], info["eigenvalues"][sd.id])
        worst_res, worst_rq = max(worst_res, float(r_.max())), max(worst_rq, float(q_.max()))
    print(f"[96^3 GenEO] host check of the 160 device eigenpairs: worst ||A x - lambda C x|| / ||lambda C x|| = {worst_res:.2e}, worst Rayleigh-quotient mismatch {worst_rq:.2e}")
    assert worst_res < 1e-4 and worst_rq < 1e-8
    tl.set_coarse_basis(basis)
    tl.rebuild_combined("additive")
    res, hist, x = tl.solve(reduction=1e-10, maxit=1000)
    tl.prec.check_status()
    ao.set_threads(8)
    try:
        it, conv, hist_o, xo = oracle_solve(dec, reduction=1e-10, maxit=1000, coarse={s: list(basis[s]) for s in basis}, schwarz_type="standard", mode="additive")
        variants = {}
        for order in (1, 2):           # the same oracle, only the summation order of the global dots changed
            ao.set_dot_order(order)
            it_v, conv_v, hist_v, _ = oracle_solve(dec, reduction=1e-10, maxit=1000, coarse={s: list(basis[s]) for s in basis}, schwarz_type="standard", mode="additive")
            variants[order] = (it_v, conv_v, np.asarray(hist_v))
    finally:
        ao.set_dot_order(0)
        ao.set_threads(1)
    ho, hh = np.asarray(hist_o), np.asarray(hist)
    assert res.converged and conv, (res.converged, conv)
    if res.iterations != it:           # the ONLY allowed difference: one iteration, because the last common iterate straddles the threshold
        k_last = min(len(ho), len(hh)) - 1
        assert abs(res.iterations - it) == 1 and min(hh[k_last], ho[k_last]) <= 1e-10 * ho[0] < max(hh[k_last], ho[k_last]), (res.iterations, it, hh[k_last] / ho[0], ho[k_last] / ho[0])
    m = min([len(ho), len(hh)] + [len(v[2]) for v in variants.values()])
    dev = np.abs(hh[:m] - ho[:m])
    rel = dev / ho[:m]
    env = np.zeros(m)
    for order, (it_v, conv_v, hv) in variants.items():
        assert conv_v and abs(it_v - it) <= 1
        env = np.maximum(env, np.maximum.accumulate(np.abs(hv[:m] - ho[:m]) / ho[:m]))
    print("[96^3 GenEO] k, r_k/r_0, |dr|/r_k hip-vs-oracle, envelope oracle-vs-oracle(other summation order):",
          [(k, f"{ho[k] / ho[0]:.1e}", f"{rel[k]:.1e}", f"{env[k]:.1e}") for k in range(0, m, 10)])
    early = ho[:m] >= 2e-3 * ho[0]
    assert early.sum() >= 30 and (dev[early] <= 1e-8 * ho[:m][early]).all(), float(np.max(dev[early] / ho[:m][early]))
    ENVELOPE_FACTOR = 10.0            # measured: 0.70 and 2.59 in two runs (the HIP run deviates about as much from the oracle as the oracle from its re-ordered self)
    bound = np.maximum(1e-8, ENVELOPE_FACTOR * env)
    worst = int(np.argmax(rel / bound))
    print(f"[96^3 GenEO] max over k of (hip-vs-oracle deviation) / max(1e-8, envelope): {np.max(rel / np.maximum(1e-8, env)):.2f} at k = {int(np.argmax(rel / np.maximum(1e-8, env)))}")
    assert (rel <= bound).all(), (worst, float(rel[worst]), float(env[worst]))
    assert env.max() >= 1e-4           # the oracle runs themselves drift apart by orders of magnitude: the effect is CG's
    if res.iterations != it:   # the last common iterate sits on the threshold
        k = m - 1
        assert min(hh[k], ho[k]) <= 1e-10 * ho[0] < max(hh[k], ho[k])
    xh, xr = x.cpu().numpy(), np.concatenate(xo)
    assert np.max(np.abs(xh - xr)) <= 1e-8 * np.max(np.abs(xr))                 # the SOLUTIONS agree far better than the residual curves
    print(f"[96^3 GenEO] iterations hip {res.iterations} / oracle {it}; max |dr| / r_k = {np.max(dev / ho[:m]):.2e}; last common iterate r_k / r_0: "
          f"hip {hh[m - 1] / ho[0]:.3e} oracle {ho[m - 1] / ho[0]:.3e}; solution difference {np.max(np.abs(xh - xr)) / np.max(np.abs(xr)):.1e}; engine {tl.schwarz.engine()}")
    tl.ctx.close()


def _true_residual(dec, G, b_glob, tl, x, backward=False):
    """|| b - G x || / || b || with the GLOBAL matrix assembled on the host (x: rank-local consistent device vector);
    backward=True: the normwise backward error || b - G x ||_inf / (|| G ||_inf || x ||_inf + || b ||_inf)"""
    xg = np.zeros(dec.nglobal)
    xh = x.cpu().numpy()
    for sd in dec.subs:
        xg[sd.glob[:sd.n_o]] = xh[tl.rl.off_o[sd.id]:tl.rl.off_o[sd.id] + sd.n_o]
    r = b_glob - G @ xg
    if backward:
        return float(np.abs(r).max() / (abs(G).sum(axis=1).max() * np.abs(xg).max() + np.abs(b_glob).max()))
    return float(np.linalg.norm(r) / np.linalg.norm(b_glob))


def test_full_size_dg_convection_diffusion(ddm):
    """BASELINE configs[3] at the size the metric names: Q1-DG on 512^2 cells = 1 048 576 DoF, 8 subdomains (4 x 2), overlap 2,
    checkerboard coefficient 1e-6 / 1, b = (1/3, 1) (examples/pdelab_example.ini, convection_diffusion_coefficient.lua).
    No oracle run at this size; checked through properties: the non-symmetric SpMV is bit-exact against scipy; the ILU(0) solve
    inverts its own factors; with the reference's shipped local solver choice (`type = umfpack` -> sparse L U) the one-level
    restarted GMRES converges in a handful of iterations and the TRUE residual of the global system (host assembly) confirms it;
    the GenEO eigenproblems (symmetric part) converge with the exact preconditioner and the two-level solve converges too."""
    import torch
    from dune_ddm_amd import synth
    from dune_ddm_amd.geneo import geneo_basis
    from dune_ddm_amd.problem import build_structured
    from dune_ddm_amd.solver import TwoLevelSchwarz
    grid = synth.StructuredDG2D((512, 512), (4, 2))
    dec = build_structured(grid, overlap=2, pou_type="distance", shrink=0, neumann=True)
    assert dec.nglobal == 1_048_576 and dec.nsub == 8
    tl = TwoLevelSchwarz(dec, coarse="none", schwarz_type="standard", mode="additive", subdomain_solver="umfpack")
    rng = np.random.default_rng(2)
    xh = rng.standard_normal(tl.rl.n_o)
    xd, yd = tl.to_device(xh), tl.zeros(tl.rl.n_o)
    tl.A.mv(xd, yd)
    tl.ctx.sync()
    assert np.array_equal(yd.cpu().numpy(), tl.rl.A @ xh)                     # row sums in column order on both sides
    b_glob = np.zeros(dec.nglobal)
    for sd in dec.subs:
        b_glob[sd.glob[:sd.n_o]] = sd.b
    res, hist, x = tl.solve(reduction=1e-8, maxit=200, solver="restartedgmressolver", restart=50)
    tl.prec.check_status()
    assert res.converged and res.iterations <= 25, res.iterations
    assert _true_residual(dec, grid.G, b_glob, tl, x) < 1e-6
    basis, info = geneo_basis(tl, nev=16, tol=1e-5, return_info=True)
    assert info["converged"] and info["used_direct"] and info["iterations"] <= 40
    tl.set_coarse_basis(basis)
    tl.rebuild_combined("additive")
    res2, hist2, x2 = tl.solve(reduction=1e-8, maxit=300, solver="restartedgmressolver", restart=50)
    assert res2.converged and _true_residual(dec, grid.G, b_glob, tl, x2) < 1e-6
    # ILU(0) (the hand-tuned engines' input): solve inverts the library's own factors
    F = ddm.Ilu0(tl.ctx, tl.A_dir, tl.rl.block_ptr)
    n = tl.rl.n
    d = rng.standard_normal(n)
    xs = torch.zeros(n, dtype=torch.float64, device="cuda")
    F.solve(torch.as_tensor(d).cuda(), xs)
    tl.ctx.sync()
    assert F.status() == 0
    M = tl.rl.A_dir
    lu = F.factors()
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(M.indptr))
    low, up, dg = M.indices < rows, M.indices > rows, M.indices == rows
    x1 = xs.cpu().numpy()
    Ux = sp.csr_matrix((np.where(up, lu, 0.0), M.indices, M.indptr), shape=M.shape) @ x1 + x1 / lu[dg]
    LUx = sp.csr_matrix((np.where(low, lu, 0.0), M.indices, M.indptr), shape=M.shape) @ Ux + Ux
    assert np.abs(LUx - d).max() <= 1e-9 * np.abs(d).max()
    print(f"[DG 512^2] one-level GMRES(umfpack) {res.iterations} its, two-level GenEO {res2.iterations} its, GenEO {info['iterations']} block iterations; ILU(0) engine {F.engine()}")
    tl.ctx.close()


def test_full_size_elasticity(ddm):
    """BASELINE configs[4]: P1 elasticity on the 160 x 16 x 24 simplex box (one global refinement of examples/linearelasticity.cc:39-41:
    205 275 DoF), steel / rubber coefficient, 8 subdomains, overlap 1, GenEO nev = 12 (B = A_neu), restricted Schwarz with the
    shipped `type = cholmod` local solver, multiplicative coarse level, restarted GMRES (examples/linearelasticity.ini).  Properties:
    GenEO converges (exact preconditioner) and finds the six rigid-body modes of a floating subdomain (eigenvalues ~ 0); the
    two-level solve converges in far fewer iterations than its reduction target allows and the TRUE residual of the globally
    assembled system confirms it; rows have 45 entries (wider than the ILU(0) pipe format: engine reported)."""
    from dune_ddm_amd import synth
    from dune_ddm_amd.geneo import geneo_basis
    from dune_ddm_amd.problem import build_structured
    from dune_ddm_amd.solver import TwoLevelSchwarz
    grid = synth.StructuredElasticity(refine=1, parts=8)
    dec = build_structured(grid, overlap=1, pou_type="distance", shrink=0, neumann=True, second_region="all")
    assert dec.nglobal == 205_275 and int(np.diff(dec.subs[3].A_dir.indptr).max()) == 45
    tl = TwoLevelSchwarz(dec, coarse="none", schwarz_type="restricted", mode="multiplicative", subdomain_solver="cholmod")
    basis, info = geneo_basis(tl, nev=12, tol=1e-6, return_info=True)
    assert info["converged"] and info["used_direct"]
    # regression bound of the stale-products fix (12-17 block iterations in 48 of 48 runs since; 36-120 or none before)
    assert info["iterations"] <= 25, info["iterations"]
    lam = info["eigenvalues"][3]                                              # a floating subdomain
    assert (np.abs(lam[:6]) < 1e-8).all() and lam[6] > 1e-6
    tl.set_coarse_basis(basis)
    tl.rebuild_combined("multiplicative")
    res, hist, x = tl.solve(reduction=1e-6, maxit=300, solver="restartedgmressolver", restart=100)
    tl.prec.check_status()
    assert res.converged and res.iterations <= 60, res.iterations
    G = sp.csr_matrix((dec.nglobal, dec.nglobal))
    b_glob = np.zeros(dec.nglobal)
    for sd in dec.subs:
        Pm = sp.csr_matrix((np.ones(sd.n_o), (np.arange(sd.n_o), sd.glob[:sd.n_o])), shape=(sd.n_o, dec.nglobal))
        G = G + Pm.T @ sd.A @ Pm
        b_glob[sd.glob[:sd.n_o]] = sd.b
    # steel / rubber: entries of G span 1e7 .. 1e11 against a load of O(1), so || b - G x || / || b || is meaningless at a
    # reduction of 1e-6 of the PRECONDITIONED defect (what left-preconditioned GMRES monitors); the backward error is the property
    assert _true_residual(dec, G.tocsr(), b_glob, tl, x, backward=True) < 1e-7
    tl_ilu = TwoLevelSchwarz(dec, coarse="none", schwarz_type="restricted", mode="multiplicative")
    print(f"[elasticity 205k] two-level GMRES(cholmod) {res.iterations} its, GenEO {info['iterations']} block iterations, ILU(0) engine {tl_ilu.schwarz.engine()}")
    tl_ilu.ctx.close()
    tl.ctx.close()
