"""The oracle's Spectra restatement (oracle/geneo_oracle.py) pinned by the properties the
reference's own eigensolver tests check, and cross-checked against ARPACK."""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spl

from oracle import geneo_oracle as go


def _pencil(ddm, N=(13, 13, 13), P=(2, 2, 2), sub=0):
    from dune_ddm_amd import synth
    from dune_ddm_amd.problem import build_structured
    dec = build_structured(synth.StructuredPoisson(N, P), overlap=2, pou_type="distance", neumann=True)
    sd = dec.subs[sub]
    return dec, sd, go.scale_matrix_with_pou(sd.B_neu, sd.pou)


def test_simple_random_is_the_lehmer_lcg():
    v = go.simple_random_vec(3, 0)          # seed 0 -> state 1 (SimpleRandom.h:91-95)
    assert np.allclose(v, [16807 / 2147483647 - 0.5, 282475249 / 2147483647 - 0.5, 1622650073 / 2147483647 - 0.5], rtol=0, atol=1e-16)


def test_params_quirks():
    p = go.EigensolverParams({"nev": 8})
    assert (p.nev, p.ncv, p.nev_max, p.shift, p.tolerance, p.threshold) == (8, 16, 16, 1e-3, 1e-5, -0.5)
    q = go.EigensolverParams({"nev": 8, "nev_max": 40})     # eigensolver_params.hh:23: the key overwrites ncv
    assert q.ncv == 40 and q.nev_max is None


def test_irlm_matches_arpack_and_is_B_orthonormal(ddm):
    dec, sd, C = _pencil(ddm)
    lam, X, solver = go.spectra_gevp(sd.A_neu, C, go.EigensolverParams({"nev": 8}))
    w = np.sort(spl.eigsh(sd.A_neu.tocsc(), k=8, M=C.tocsc(), sigma=1e-3, which="LM", return_eigenvectors=False, tol=1e-12))
    assert np.allclose(lam, w, rtol=1e-8)
    assert (np.diff(lam) >= -1e-12).all()                                  # sorted ascending (spectra.hh:138)
    G = X.T @ (C @ X)
    assert np.abs(G - np.eye(8)).max() < 1e-8                              # test_eigensolver.cc:103-114
    R = sd.A_neu @ X - (C @ X) * lam[None, :]
    assert np.abs(R).max() < 1e-4 * np.abs(sd.A_neu @ X).max()
    # Lanczos relation OP V - V H - f e_m^T = 0 (test_lanczos_step.cc:239-260), OP = (A - sigma C)^-1 C
    lu = spl.splu((sd.A_neu - 1e-3 * C).tocsc())
    V, H, f = solver.fac.V, solver.fac.H, solver.fac.f
    E = lu.solve(C @ V) - V @ H
    E[:, -1] -= f
    assert np.abs(E).max() < 1e-8 * max(1.0, np.abs(H).max())


def test_threshold_mode_and_basis_finalisation(ddm):
    dec, sd, C = _pencil(ddm, N=(17, 17, 17))
    basis, lam = go.geneo_basis(sd.A_neu, sd.B_neu, sd.pou, {"nev": 4, "threshold": 1.0})
    assert len(basis) >= 1 and len(basis) == len(lam)
    assert (lam[:-1] < 1.0).all()                                          # spectra.hh:157-163: kept eigenvalues are below the threshold
    for v in basis:
        assert abs(np.linalg.norm(v) - 1.0) < 1e-14                        # coarse_spaces.hh:55-60
        assert (v[sd.pou == 0] == 0).all()                                 # vanishes on the subdomain boundary
