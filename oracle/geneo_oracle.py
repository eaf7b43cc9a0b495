"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of the coarse-basis builder half of the hot path:
GenEOCoarseSpace::setup_geneo_impl -> solve_gevp -> spectra_gevp_op -> Spectra's
SymGEigsShiftSolver (shift-invert implicitly restarted Lanczos in the B inner product).
numpy / scipy; small problems only.  Citations are relative to /root/reference.

Third-party pieces that are absent from the snapshot and what stands in for them here:
  * UMFPACK LU of A - sigma*B (eigensolvers/spectra.hh:42-62)  -> scipy.sparse.linalg.splu
  * Eigen's dense kernels inside Spectra (tridiagonal QR / eigen decomposition)
    -> numpy.linalg.qr normalised to the Givens convention (positive diagonal, det Q = +1) and
       numpy.linalg.eigh.  Ritz vectors are therefore equal up to sign.
Parity status: the reference stores no GenEO eigenvalues or basis vectors anywhere
=> "parity unpinned" by reference fixtures; this restatement is pinned by properties the
reference's own tests use (test_eigensolver.cc:103-114 B-orthonormality / agreement < 1e-8,
test_lanczos_step.cc:239-260 Lanczos relation < 1e-8) and cross-checked against ARPACK
(scipy eigsh, shift-invert) in tests/test_oracle_geneo.py.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spl


# ---- Util/SimpleRandom.h:30-64, 80-120 --------------------------------------------------------
def _next_long_rand(seed):
    m_a, m_max = 16807, 2147483647
    lo = m_a * (seed & 0xFFFF)
    hi = m_a * (seed >> 16)
    lo += (hi & 0x7FFF) << 16
    if lo > m_max:
        lo &= m_max
        lo += 1
    lo += hi >> 15
    if lo > m_max:
        lo &= m_max
        lo += 1
    return lo


def simple_random_vec(n, init_seed):
    """SimpleRandom<double>(init_seed).random_vec(n): U(-0.5, 0.5) from a Lehmer LCG."""
    m_max = 2147483647
    state = (init_seed & m_max) if init_seed else 1
    out = np.empty(n)
    for i in range(n):
        state = _next_long_rand(state)
        out[i] = state / m_max - 0.5
    return out


# ---- eigensolver_params.hh:8-62 --------------------------------------------------------------
class EigensolverParams:
    def __init__(self, ptree=None):
        ptree = dict(ptree or {})
        self.type = "Spectra"
        self.nev = 16
        self.maxit = 1000
        self.seed = 1
        self.blocksize = 8
        self.tolerance = 1e-5
        self.shift = 1e-3
        self.threshold = -0.5
        self.nev_max = None
        if "nev" in ptree:
            self.nev = int(ptree["nev"])
        self.ncv = int(ptree["ncv"]) if "ncv" in ptree else 2 * self.nev               # :21-22
        if "nev_max" in ptree:
            self.ncv = int(ptree["nev_max"])     # sic: the key overwrites ncv (:23); nev_max stays unset
        else:
            self.nev_max = 2 * self.nev                                                 # :24
        for k, cast in (("maxit", int), ("tolerance", float), ("shift", float), ("seed", int), ("blocksize", int), ("threshold", float)):
            if k in ptree:
                setattr(self, k, cast(ptree[k]))
        if "type" in ptree and ptree["type"] != "Spectra":
            raise NotImplementedError("Unknown eigensolver type '" + str(ptree["type"]) + "'")   # :35


# ---- Spectra ------------------------------------------------------------------------------------
EPS = np.finfo(np.float64).eps
NEAR_0 = np.finfo(np.float64).tiny * 10.0


class _ArnoldiOp:
    """MatOp/internal/ArnoldiOp.h:69-91 with SymGEigsShiftInvertOp.h:86-90."""

    def __init__(self, solve, B):
        self.solve, self.B = solve, B
        self.n = B.shape[0]
        self.nmatop = 0

    def perform_op(self, x):                     # y = (A - sigma B)^-1 B x
        self.nmatop += 1
        return self.solve(self.B @ x)

    def inner(self, x, y):
        return float(x @ (self.B @ y))

    def adjoint(self, V, y):
        return V.T @ (self.B @ y)

    def norm(self, x):
        return float(np.sqrt(self.inner(x, x)))


def _givens_qr(T):
    """TridiagQR: QR by Givens rotations => R has a non-negative diagonal (except possibly the
    last entry) and det Q = +1."""
    Q, R = np.linalg.qr(T)
    n = T.shape[0]
    for i in range(n - 1):
        if R[i, i] < 0:
            Q[:, i] = -Q[:, i]
            R[i, :] = -R[i, :]
    if np.linalg.det(Q) < 0:
        Q[:, n - 1] = -Q[:, n - 1]
        R[n - 1, :] = -R[n - 1, :]
    return Q, R


class Lanczos:
    """LinAlg/Arnoldi.h + LinAlg/Lanczos.h."""

    def __init__(self, op: _ArnoldiOp, m):
        self.op, self.n, self.m, self.k = op, op.n, m, 0

    def init(self, v0):                                           # Arnoldi.h:134-184
        self.V = np.zeros((self.n, self.m))
        self.H = np.zeros((self.m, self.m))
        if self.op.norm(v0) < NEAR_0:
            raise ValueError("initial residual vector cannot be zero")
        v = self.op.perform_op(v0)
        v = v / self.op.norm(v)
        self.V[:, 0] = v
        w = self.op.perform_op(v)
        self.H[0, 0] = self.op.inner(v, w)
        self.f = w - v * self.H[0, 0]
        if np.abs(self.f).max() < EPS * abs(self.H[0, 0]):
            self.f[:] = 0.0
            self.beta = 0.0
        else:
            self.beta = self.op.norm(self.f)
        self.k = 1

    def expand_basis(self, V, seed):                              # Arnoldi.h:66-118
        for it in range(5):
            if it == 0:
                f = self.op.perform_op(simple_random_vec(self.n, seed + 123 * it))
            else:
                f = simple_random_vec(self.n, seed + 123 * it)
            Vf = self.op.adjoint(V, f)
            f = f - V @ Vf
            fnorm = self.op.norm(f)
            Vf = self.op.adjoint(V, f)
            err = np.abs(Vf).max()
            count = 0
            while count < 3 and err >= EPS * fnorm:
                f = f - V @ Vf
                fnorm = self.op.norm(f)
                Vf = self.op.adjoint(V, f)
                err = np.abs(Vf).max()
                count += 1
            if err < EPS * fnorm:
                break
        self.f, self.beta = f, fnorm

    def factorize_from(self, from_k, to_m):                       # Lanczos.h:62-190
        if to_m <= from_k:
            return
        if from_k > self.k:
            raise ValueError("Lanczos: from_k is larger than the current subspace dimension")
        beta_thresh = EPS * np.sqrt(self.n)
        eps_sqrt = np.sqrt(EPS)
        self.H[:, from_k:] = 0.0
        self.H[from_k:, :from_k] = 0.0
        for i in range(from_k, to_m):
            restart = self.beta < NEAR_0
            if not restart:
                v = self.f / self.beta
                if self.beta < eps_sqrt:
                    restart = abs(self.op.inner(self.V[:, i - 1], v)) > eps_sqrt
            if restart:
                self.expand_basis(self.V[:, :i], 2 * i)
                v = self.f / self.beta
            self.V[:, i] = v
            self.H[i, i - 1] = 0.0 if restart else self.beta
            self.H[i - 1, i] = self.H[i, i - 1]
            w = self.op.perform_op(v)
            if not restart:
                w = w - self.H[i, i - 1] * self.V[:, i - 1]
            self.H[i, i] = self.op.inner(v, w)
            self.f = w - self.H[i, i] * v
            self.beta = self.op.norm(self.f)
            Vs = self.V[:, :i + 1]
            Vf = self.op.adjoint(Vs, self.f)
            err = np.abs(Vf).max()
            count = 0
            while count < 5 and err > EPS * self.beta:
                if self.beta < beta_thresh:
                    self.f[:] = 0.0
                    self.beta = 0.0
                    break
                self.f = self.f - Vs @ Vf
                self.H[i - 1, i] += Vf[i - 1]
                self.H[i, i - 1] = self.H[i - 1, i]
                self.H[i, i] += Vf[i]
                self.beta = self.op.norm(self.f)
                Vf = self.op.adjoint(Vs, self.f)
                err = np.abs(Vf).max()
                count += 1
        self.k = to_m

    def compress_V(self, Q):                                      # Arnoldi.h:310-329 (after m_k was reduced)
        k, m = self.k, self.m
        Vs = np.zeros((self.n, k + 1))
        for i in range(k):
            nnz = m - k + i + 1
            Vs[:, i] = self.V[:, :nnz] @ Q[:nnz, i]
        Vs[:, k] = self.V @ Q[:, k]
        self.V[:, :k + 1] = Vs
        self.f = self.f * Q[m - 1, k - 1] + self.V[:, k] * self.H[k, k - 1]
        self.beta = self.op.norm(self.f)


class SymGEigsShiftSolver:
    """SymGEigsShiftSolver<..., ShiftInvert> on top of HermEigsBase (HermEigsBase.h:104-391)."""

    def __init__(self, solve, B, nev, ncv, sigma):
        n = B.shape[0]
        if nev < 1 or nev > n - 1:
            raise ValueError("nev must satisfy 1 <= nev <= n - 1, n is the size of matrix")
        if ncv <= nev or ncv > n:
            raise ValueError("ncv must satisfy nev < ncv <= n, n is the size of matrix")
        self.op = _ArnoldiOp(solve, B)
        self.n, self.nev, self.ncv, self.sigma = n, nev, ncv, sigma
        self.fac = Lanczos(self.op, ncv)
        self.niter = 0
        self.info = "NotComputed"

    def init(self):                                               # HermEigsBase.h:337-342: SimpleRandom(0)
        self.ritz_val = np.zeros(self.ncv)
        self.ritz_vec = np.zeros((self.ncv, self.nev))
        self.ritz_est = np.zeros(self.ncv)
        self.ritz_conv = np.zeros(self.nev, dtype=bool)
        self.fac.init(simple_random_vec(self.n, 0))

    def _retrieve_ritzpair(self):                                 # :204-224, selection = LargestMagn
        evals, evecs = np.linalg.eigh(self.fac.H)
        ind = np.argsort(-np.abs(evals), kind="stable")
        self.ritz_val = evals[ind]
        self.ritz_est = evecs[self.ncv - 1, ind]
        self.ritz_vec = evecs[:, ind[:self.nev]]

    def _num_converged(self, tol):                                # :158-175
        eps23 = EPS ** (2.0 / 3.0)
        thresh = tol * np.maximum(np.abs(self.ritz_val[:self.nev]), eps23)
        resid = np.abs(self.ritz_est[:self.nev]) * self.fac.beta
        self.ritz_conv = resid < thresh
        return int(self.ritz_conv.sum())

    def _nev_adjusted(self, nconv):                               # :178-201
        nev_new = self.nev + int(np.sum(np.abs(self.ritz_est[self.nev:self.ncv]) < NEAR_0))
        nev_new += min(nconv, (self.ncv - nev_new) // 2)
        if nev_new == 1 and self.ncv >= 6:
            nev_new = self.ncv // 2
        elif nev_new == 1 and self.ncv > 2:
            nev_new = 2
        return min(nev_new, self.ncv - 1)

    def _restart(self, k):                                        # :104-155
        if k >= self.ncv:
            return
        Q = np.eye(self.ncv)
        shifts = sorted(self.ritz_val[self.ncv - (self.ncv - k):], key=lambda v: -abs(v))   # tail(nshift), large first
        for mu in shifts:
            Qi, Ri = _givens_qr(self.fac.H - mu * np.eye(self.ncv))
            Q = Q @ Qi
            self.fac.H = Ri @ Qi + mu * np.eye(self.ncv)                                    # compress_H: Q'HQ
            self.fac.H = np.triu(np.tril(self.fac.H, 1), -1)                               # stays tridiagonal
            self.fac.k -= 1
        self.fac.compress_V(Q)
        self.fac.factorize_from(k, self.ncv)
        self._retrieve_ritzpair()

    def compute(self, maxit=1000, tol=1e-10):                      # :366-391 (+ SymGEigsShiftSolver.h:170-176)
        self.fac.factorize_from(1, self.ncv)
        self._retrieve_ritzpair()
        nconv, i = 0, 0
        for i in range(maxit):
            nconv = self._num_converged(tol)
            if nconv >= self.nev:
                break
            self._restart(self._nev_adjusted(nconv))
        # sort_ritzpair: lambda = 1/nu + sigma, then SmallestAlge
        lam = 1.0 / self.ritz_val[:self.nev] + self.sigma
        ind = np.argsort(lam, kind="stable")
        self.lam = lam[ind]
        self.ritz_vec = self.ritz_vec[:, ind]
        self.ritz_conv = self.ritz_conv[ind]
        self.niter += i + 1
        self.info = "Successful" if nconv >= self.nev else "NotConverging"
        return min(self.nev, nconv)

    def eigenvalues(self):
        return self.lam[self.ritz_conv]

    def eigenvectors(self):
        return self.fac.V @ self.ritz_vec[:, self.ritz_conv]


def spectra_gevp(A, B, params: EigensolverParams):
    """spectra_gevp / spectra_gevp_op (dune/ddm/eigensolvers/spectra.hh:111-254).
    Returns (eigenvalues ascending, eigenvectors as columns, solver)."""
    A = sp.csc_matrix(A)
    B = sp.csr_matrix(B)
    nev, nev_max, shift, tol, threshold = params.nev, params.nev_max, params.shift, params.tolerance, params.threshold
    done = threshold < 0
    ncv = params.ncv
    tries = 3
    lu = spl.splu((A - shift * sp.csc_matrix(B)).tocsc())          # set_shift (:42-62): LU of A - sigma B, no iterative refinement
    while True:
        if ncv <= nev:
            ncv = 2 * nev                                          # :127
        geigs = SymGEigsShiftSolver(lu.solve, B, nev, ncv, shift)  # :130
        geigs.init()                                               # :131
        nconv = geigs.compute(maxit=100, tol=tol)                  # :137-138 (maxit hard-coded 100)
        if geigs.info == "Successful":
            evalues, evecs = geigs.eigenvalues(), geigs.eigenvectors()
            if evalues[nconv - 1] >= threshold or (nev_max is not None and nev >= nev_max):   # :157
                if threshold > 0:                                  # :158-163
                    cnt = 0
                    while cnt < nconv - 1 and evalues[cnt] < threshold:
                        cnt += 1
                    nconv = max(cnt, 1)
                return evalues[:nconv], evecs[:, :nconv], geigs    # :177-182
            if not done:
                nev *= 2                                           # :186-189
        else:                                                      # NotConverging (:191-203)
            if tries != 0:
                tries -= 1
                ncv *= 2
                done = False
                continue
            raise RuntimeError("Computation of eigenvalues failed, not yet converged, no more tries left (MPI_Abort 12)")
        if done:
            raise RuntimeError("eigensolver loop ended without a result")


# ---- coarse_spaces.hh ----------------------------------------------------------------------------
def scale_matrix_with_pou(C, pou):
    """detail::scale_matrix_with_pou (coarse_spaces.hh:74-96): C_ij *= pou_i * pou_j on the stored pattern."""
    C = sp.csr_matrix(C).copy()
    rows = np.repeat(np.arange(C.shape[0]), np.diff(C.indptr))
    C.data = C.data * (pou[rows] * pou[C.indices])
    return C


def finalize_eigenvectors(vecs, pou):
    """detail::finalize_eigenvectors (coarse_spaces.hh:52-61)."""
    out = []
    for v in vecs:
        v = v * pou
        out.append(v * (1.0 / np.sqrt(float(v @ v))))
    return out


def geneo_basis(A_neu, B_neu, pou, eig_ptree=None):
    """GenEOCoarseSpace::setup_geneo_impl (coarse_spaces.hh:319-331).  Returns (basis list, eigenvalues)."""
    if len(pou) != A_neu.shape[0]:
        raise ValueError("The matrix and the partition of unity must have the same size")
    C = scale_matrix_with_pou(B_neu, pou)
    lam, X, _ = spectra_gevp(A_neu, C, EigensolverParams(eig_ptree))
    return finalize_eigenvectors([X[:, j].copy() for j in range(X.shape[1])], pou), lam
