"""GenEO coarse-basis builder on the device.

Reference: GenEOCoarseSpace::setup_geneo_impl (dune/ddm/coarsespaces/coarse_spaces.hh:319-331):
C = D B_neu D, lowest ``nev`` eigenpairs of  A_neu x = lambda C x,  v <- D v / ||D v||_2, then the
caller zeroes the Dirichlet entries (examples/poisson.cc:235-238,282).  The reference solves the
pencil with shift-invert Lanczos (Spectra) on top of a sparse LU of A - sigma*C
(eigensolvers/spectra.hh:28-254).  A sparse direct factorisation is a latency-bound host algorithm;
the MI355X-native design computes the *same invariant subspace* with a block method that needs
only the kernels the hot path already has (SURVEY.md App. A.9 allows exactly this):

  LOBPCG on the reciprocal pencil  C~ x = mu A~ x  (largest mu),  A~ = A_neu + shift*C~  (SPD also
  for floating subdomains; identical eigenvectors, lambda = 1/mu - shift), A~-orthonormal blocks,
  preconditioner T = ILU(0) of A~ applied to all columns at once by the multi-RHS level-scheduled
  triangular solve, products with A~ and C~ as row-major SpMM, Rayleigh-Ritz on [X W P] per
  subdomain.  All subdomains of the rank are iterated in lockstep on their concatenated vectors.

C~ is C with the rows/columns of global Dirichlet DoFs removed: after the symmetric elimination
(examples/pdelab_helper.hh:33-46) those DoFs are decoupled unit eigenvectors (lambda = 1/pou_i^2) that
``zero_at_dirichlet`` turns into zero vectors -- they can only make R A R^T singular, so they are
deflated here.  Whenever the reference's result is usable (no such mode among the wanted ones) the
two agree; tests/test_gpu_geneo.py checks eigenvalues and the spanned subspace against the oracle's
literal Spectra restatement.

Dense block algebra (Gram matrices, small eigenproblems, basis rotations) goes through torch
(rocBLAS / rocSOLVER): plain library GEMMs on the setup path, not part of the timed Krylov loop.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as sl
import scipy.sparse as sp

from . import CsrMatrix, Ilu0
from .problem import _block_diag


def _scale_with_pou(B, pou):
    """detail::scale_matrix_with_pou (coarse_spaces.hh:74-96)"""
    C = sp.csr_matrix(B).copy()
    rows = np.repeat(np.arange(C.shape[0]), np.diff(C.indptr))
    C.data = C.data * (pou[rows] * pou[C.indices])
    return C


def _drop_dofs(C, mask):
    C = sp.csr_matrix(C).copy()
    rows = np.repeat(np.arange(C.shape[0]), np.diff(C.indptr))
    C.data[(mask[rows] > 0) | (mask[C.indices] > 0)] = 0.0
    return C


class EigensolverParams:
    """The keys of ``<prefix>.eigensolver`` with the reference's defaults and quirks
    (dune/ddm/eigensolvers/eigensolver_params.hh:8-62): ``nev_max`` overwrites **ncv** (:23); ``maxit``,
    ``seed``, ``blocksize`` are parsed but unused by the reference driver (spectra.hh:137 hard-codes 100).
    ``ncv`` has no meaning for the block method (the block size is nev + 4)."""

    def __init__(self, ptree=None):
        ptree = dict(ptree or {})
        self.type = ptree.get("type", "Spectra")
        if self.type != "Spectra":
            raise NotImplementedError("Unknown eigensolver type '" + str(self.type) + "'")   # :35
        self.nev = int(ptree.get("nev", 16))
        self.ncv = int(ptree["ncv"]) if "ncv" in ptree else 2 * self.nev
        self.nev_max = None
        if "nev_max" in ptree:
            self.ncv = int(ptree["nev_max"])
        else:
            self.nev_max = 2 * self.nev
        self.maxit = int(ptree.get("maxit", 1000))
        self.tolerance = float(ptree.get("tolerance", 1e-5))
        self.shift = float(ptree.get("shift", 1e-3))
        self.seed = int(ptree.get("seed", 1))
        self.blocksize = int(ptree.get("blocksize", 8))
        self.threshold = float(ptree.get("threshold", -0.5))


def geneo_basis_from_params(tl, eig_ptree=None, **kw):
    """GenEOCoarseSpace driven by the eigensolver sub-tree, incl. the threshold mode of
    spectra_gevp_op (eigensolvers/spectra.hh:157-163,186-189): with threshold > 0 keep the eigenvalues
    below it (at least one) and double nev until the largest computed one exceeds it or nev >= nev_max."""
    p = EigensolverParams(eig_ptree)
    nev = p.nev
    while True:
        basis, info = geneo_basis(tl, nev=nev, tol=p.tolerance, shift=p.shift, return_info=True, **kw)
        if p.threshold <= 0:
            return basis, info
        lam = info["eigenvalues"]
        done = all(l[-1] >= p.threshold for l in lam.values()) or (p.nev_max is not None and nev >= p.nev_max)
        if done:
            out = {}
            for s, vecs in basis.items():
                cnt = 0
                while cnt < len(lam[s]) - 1 and lam[s][cnt] < p.threshold:
                    cnt += 1
                out[s] = vecs[:max(cnt, 1)]
            return out, info
        nev *= 2


def geneo_basis(tl, nev=20, tol=1e-5, shift=1e-3, maxit=400, extra=4, seed=0, verbose=False, return_info=False):
    """Returns {local subdomain id: (k, n_s) ndarray} ready for TwoLevelSchwarz.set_coarse_basis
    (POU-scaled, 2-normalised, zero at Dirichlet DoFs)."""
    torch = tl.torch
    rl, ctx, dev = tl.rl, tl.ctx, tl.dev
    for sd in rl.subs:
        if sd.A_neu is None or sd.B_neu is None:
            raise ValueError("GenEO needs the Neumann matrices (build_structured(..., neumann=True))")
        if sd.pou is None or len(sd.pou) != sd.n:
            raise ValueError("The matrix and the partition of unity must have the same size")     # coarse_spaces.hh:323
    A = _block_diag([sd.A_neu for sd in rl.subs])
    C = _drop_dofs(_scale_with_pou(_block_diag([sd.B_neu for sd in rl.subs]), rl.pou), rl.dirichlet_ovlp)
    At = sp.csr_matrix(A + shift * C)
    At.sort_indices()
    n = rl.n
    m = nev + extra
    dA, dC = CsrMatrix(ctx, At), CsrMatrix(ctx, C)
    T = Ilu0(ctx, dA, rl.block_ptr)
    subs = [(int(rl.block_ptr[i]), int(rl.block_ptr[i + 1])) for i in range(len(rl.subs))]
    free = torch.as_tensor((rl.dirichlet_ovlp == 0).astype(np.float64), device=dev)[:, None]

    def mm(M, X):
        Y = torch.empty_like(X)
        M.mm(X, Y)
        return Y

    def per_sub(fn):
        for (a, b) in subs:
            fn(slice(a, b))

    CH = 4096   # rows per split-K chunk of the tall-skinny products

    def gram(U, V):
        """per-subdomain U^T V (p x q) for tall-skinny row-major blocks.  A plain GEMM call would put the
        whole K = n_s reduction on a handful of workgroups; split it into CH-row chunks (batched GEMM over
        the chunks, then a sum) so that every CU works.  TODO(next round): hand-written FP64 MFMA kernel."""
        out = []
        for (a, b) in subs:
            nfull = (b - a) // CH
            p, q = U.shape[1], V.shape[1]
            G = torch.zeros((p, q), dtype=U.dtype, device=dev)
            if nfull:
                Uc = U[a:a + nfull * CH].view(nfull, CH, p)
                Vc = V[a:a + nfull * CH].view(nfull, CH, q)
                G += torch.bmm(Uc.transpose(1, 2), Vc).sum(dim=0)
            if a + nfull * CH < b:
                G += U[a + nfull * CH:b].T @ V[a + nfull * CH:b]
            out.append(G)
        return torch.stack(out)                                               # (nsub, p, q)

    def rotate(U, Ms):
        out = torch.empty((n, Ms.shape[2]), dtype=U.dtype, device=dev)
        for i, (a, b) in enumerate(subs):
            out[a:b] = U[a:b] @ Ms[i]
        return out

    def a_orthonormalise(blocks, Ablock):
        """makes blocks[0] A~-orthonormal per subdomain (Cholesky QR) and applies the same
        transformation to the other blocks; returns False if a Gram matrix is not SPD.
        The p x p factorisations run on the host (LAPACK): a few hundred KB per iteration."""
        G = gram(blocks[0], Ablock).cpu().numpy()
        T_ = np.empty_like(G)
        for i in range(G.shape[0]):
            try:
                L = np.linalg.cholesky(0.5 * (G[i] + G[i].T))
            except np.linalg.LinAlgError:
                return False
            T_[i] = sl.solve_triangular(L, np.eye(L.shape[0]), lower=True).T      # L^-T
        Td = torch.as_tensor(T_, device=dev)
        for k in range(len(blocks)):
            blocks[k].copy_(rotate(blocks[k], Td))
        return True

    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    X = (torch.rand((n, m), generator=g, dtype=torch.float64) - 0.5).to(dev) * free
    AX = mm(dA, X)
    if not a_orthonormalise([X, AX], AX):
        raise RuntimeError("GenEO: initial block is not A-independent")
    CX = mm(dC, X)

    def small_gevp(gA, gC, keep):
        """host LAPACK: the `keep` largest mu of gC y = mu gA y per subdomain (y^T gA y = 1)"""
        ws, Ys = [], []
        for i in range(gA.shape[0]):
            try:
                w, Z = sl.eigh(0.5 * (gC[i] + gC[i].T), 0.5 * (gA[i] + gA[i].T))
            except (np.linalg.LinAlgError, sl.LinAlgError):
                return None
            ws.append(w[::-1][:keep].copy())
            Ys.append(Z[:, ::-1][:, :keep].copy())
        return torch.as_tensor(np.array(ws), device=dev), np.array(Ys)

    def rayleigh_ritz(S, AS, CS, keep):
        """Rayleigh-Ritz of the pencil (C~, A~) on span[S_0 S_1 ...] per subdomain.  Only the upper block
        triangle of the Gram matrices is computed (tall-skinny products); returns mu and the coefficient
        blocks Y_k (nsub, m, keep) so that the new vectors are sum_k S_k Y_k (no concatenation)."""
        nb = len(S)
        gA = np.zeros((len(subs), nb * m, nb * m))
        gC = np.zeros_like(gA)
        for i in range(nb):
            for j in range(i, nb):
                a_ij = gram(S[i], AS[j]).cpu().numpy()
                c_ij = gram(S[i], CS[j]).cpu().numpy()
                gA[:, i * m:(i + 1) * m, j * m:(j + 1) * m] = a_ij
                gC[:, i * m:(i + 1) * m, j * m:(j + 1) * m] = c_ij
                if j > i:
                    gA[:, j * m:(j + 1) * m, i * m:(i + 1) * m] = a_ij.transpose(0, 2, 1)
                    gC[:, j * m:(j + 1) * m, i * m:(i + 1) * m] = c_ij.transpose(0, 2, 1)
        out = small_gevp(gA, gC, keep)
        if out is None:
            return None
        w, Y = out
        return w, [torch.as_tensor(np.ascontiguousarray(Y[:, k * m:(k + 1) * m, :]), device=dev) for k in range(nb)]

    def combine(blocks, Ys, skip_first=False):
        out = None
        for k, (B_, Y_) in enumerate(zip(blocks, Ys)):
            if skip_first and k == 0:
                continue
            t = rotate(B_, Y_)
            out = t if out is None else out.add_(t)
        return out

    import time as _time
    prof = {}

    def tick(name, t0):
        if verbose:
            torch.cuda.synchronize()
            prof[name] = prof.get(name, 0.0) + _time.perf_counter() - t0
        return _time.perf_counter()

    out = rayleigh_ritz([X], [AX], [CX], m)
    if out is None:
        raise RuntimeError("GenEO: Rayleigh-Ritz on the initial block failed")
    mu, Ys = out
    X, AX, CX = rotate(X, Ys[0]), rotate(AX, Ys[0]), rotate(CX, Ys[0])
    P = AP = CP = None
    info = {"iterations": 0, "converged": False}
    resn = None
    for it in range(maxit):
        R = torch.empty_like(X)
        for i, (a, b) in enumerate(subs):
            R[a:b] = CX[a:b] - AX[a:b] * mu[i][None, :]
        num = torch.stack([R[a:b].norm(dim=0) for (a, b) in subs])
        den = torch.stack([AX[a:b].norm(dim=0) for (a, b) in subs]) * mu.abs()
        resn = (num / den)[:, :nev]
        worst = float(resn.max())
        info["iterations"] = it
        if verbose:
            print(f"[geneo] it {it:3d}  max rel. residual {worst:.3e}  lambda_min {float((1.0 / mu[:, 0]).min() - shift):.5f}", flush=True)
        if worst < tol:
            info["converged"] = True
            break
        t0 = tick("residual", _time.perf_counter()) if verbose else 0.0
        W = torch.empty_like(R)
        T.solve_multi(R, W)                                           # W = T R, all columns at once
        t0 = tick("ilu_multi", t0)
        coef = gram(AX, W)                                            # A~-orthogonalise against X
        W = W - rotate(X, coef)
        t0 = tick("ortho_X", t0)
        AW = mm(dA, W)
        t0 = tick("spmm", t0)
        if not a_orthonormalise([W, AW], AW):
            break
        t0 = tick("ortho_W", t0)
        CW = mm(dC, W)
        t0 = tick("spmm", t0)
        blocks = ([X, W], [AX, AW], [CX, CW])
        if P is not None:
            if a_orthonormalise([P, AP, CP], AP):
                blocks = ([X, W, P], [AX, AW, AP], [CX, CW, CP])
        out = rayleigh_ritz(*blocks, m)
        if out is None and P is not None:                             # ill-conditioned basis: drop P once
            blocks = ([X, W], [AX, AW], [CX, CW])
            out = rayleigh_ritz(*blocks, m)
        if out is None:
            break
        t0 = tick("rayleigh_ritz", t0)
        mu, Ys = out
        S, AS, CS = blocks
        P, AP, CP = combine(S, Ys, True), combine(AS, Ys, True), combine(CS, Ys, True)      # P = [W P] Y_{W,P}
        X, AX, CX = rotate(X, Ys[0]).add_(P), rotate(AX, Ys[0]).add_(AP), rotate(CX, Ys[0]).add_(CP)
        t0 = tick("rotate", t0)
    if verbose:
        print("[geneo] phase seconds:", {k: round(v, 2) for k, v in prof.items()}, flush=True)
    lam = (1.0 / mu[:, :nev] - shift).cpu().numpy()                   # lambda = 1/mu - shift, ascending
    Xh = X[:, :nev].cpu().numpy()
    basis = {}
    for i, sd in enumerate(rl.subs):
        a, b = subs[i]
        vecs = []
        for j in range(nev):                                          # finalize_eigenvectors (coarse_spaces.hh:52-61)
            v = Xh[a:b, j] * sd.pou
            v = v * (1.0 / np.sqrt(float(v @ v)))
            v[sd.dirichlet_ovlp > 0] = 0.0                            # zero_at_dirichlet (poisson.cc:235-238)
            vecs.append(v)
        basis[sd.id] = np.array(vecs)
    info["eigenvalues"] = {sd.id: lam[i] for i, sd in enumerate(rl.subs)}
    info["residuals"] = None if resn is None else resn.cpu().numpy()
    tl.geneo_info = info
    return (basis, info) if return_info else basis
