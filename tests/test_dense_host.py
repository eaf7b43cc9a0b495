"""Host logic of the GenEO Rayleigh-Ritz step (csrc/dense_host.hpp through the C ABI, no device needed): the symmetric
eigensolver against numpy, and the rank-revealing Rayleigh-Ritz against scipy's generalised eigensolver, also on a
rank-deficient basis (duplicated and zero directions) where a Cholesky-based orthonormalisation breaks down."""
import ctypes

import numpy as np
import scipy.linalg as sl


def _hp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def test_sym_eig_matches_numpy(ddm):
    lib = ddm.load_library()
    rng = np.random.default_rng(3)
    for n in (1, 2, 5, 24, 72, 132):
        A = rng.standard_normal((n, n))
        A = A + A.T
        if n > 10:
            A[3] = A[:, 3] = 0.0            # an isolated zero row / column
            A[5, 7] = A[7, 5] = 1e8         # widely varying scales
        V = np.ascontiguousarray(A.copy())
        w = np.empty(n)
        assert lib.ddm_dense_sym_eig_host(n, _hp(V), _hp(w)) == 0
        assert np.allclose(w, np.linalg.eigvalsh(A), rtol=1e-12, atol=1e-12 * np.abs(A).max())
        assert np.abs(V.T @ V - np.eye(n)).max() < 1e-12
        assert np.abs(A @ V - V * w[None, :]).max() < 1e-11 * max(1.0, np.abs(A).max())


def test_rayleigh_ritz_full_rank_and_deficient(ddm):
    lib = ddm.load_library()
    rng = np.random.default_rng(4)
    n, p, keep = 300, 36, 12
    M = rng.standard_normal((n, n))
    A = M @ M.T + n * np.eye(n)
    Cm = rng.standard_normal((n, n))
    C = Cm @ Cm.T
    S = rng.standard_normal((n, p))
    for deficient in (False, True):
        if deficient:
            S[:, 5] = S[:, 4] * (1 + 1e-13)          # numerically dependent directions
            S[:, 9] = 0.0                             # a dropped (zero) direction
        gA, gC = np.ascontiguousarray(S.T @ A @ S), np.ascontiguousarray(S.T @ C @ S)
        mu = np.empty(keep)
        Y = np.empty((p, keep))
        r = lib.ddm_dense_rayleigh_ritz_host(p, _hp(gA), _hp(gC), keep, 1e-11, _hp(mu), _hp(Y))
        assert r == (p - 2 if deficient else p)
        Q, _ = np.linalg.qr(S[:, [j for j in range(p) if not (deficient and j in (5, 9))]])
        w = sl.eigh(Q.T @ C @ Q, Q.T @ A @ Q, eigvals_only=True)[::-1][:keep]
        assert np.allclose(mu, w, rtol=1e-8)
        X = S @ Y
        assert np.abs(X.T @ A @ X - np.eye(keep)).max() < 1e-7
        assert np.abs(X.T @ C @ X - np.diag(mu)).max() < 1e-7 * np.abs(mu).max()
