"""GenEO coarse-basis builder: host-side mirror of the reference's GenEOCoarseSpace
(dune/ddm/coarsespaces/coarse_spaces.hh:286-331) over the C ABI entry ``ddm_geneo_basis``.

All compute is in libddm_hip.so (csrc/geneo.hpp: block eigensolver on the device -- SpMM, multi-RHS triangular solves with the
sparse Cholesky or ILU(0) factor, FP64-MFMA Gram and rotation kernels; p x p projected eigenproblems on the host).  This module
only flattens the rank-local subdomain data into the arguments of that call and maps the keys of the eigensolver sub-tree.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import CsrMatrix, GeneoInfo, GeneoParams, _hp, _np
from .problem import _block_diag


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
    """GenEOCoarseSpace driven by the eigensolver sub-tree, incl. the threshold mode of spectra_gevp_op
    (eigensolvers/spectra.hh:157-163,186-189)."""
    p = EigensolverParams(eig_ptree)
    return geneo_basis(tl, nev=p.nev, tol=p.tolerance, shift=p.shift, threshold=p.threshold,
                       nev_max=p.nev_max if p.nev_max is not None else p.nev, return_info=True, **kw)


def geneo_basis(tl, nev=20, tol=1e-5, shift=1e-3, maxit=400, extra=4, seed=0, verbose=False, return_info=False, threshold=-0.5,
                nev_max=None, preconditioner="auto", max_direct_flops=None, require_convergence=True):
    """Returns {local subdomain id: (k, n_s) ndarray} ready for TwoLevelSchwarz.set_coarse_basis (POU-scaled, 2-normalised, zero at
    Dirichlet DoFs).  preconditioner: "auto" (sparse Cholesky of A_neu + shift C when affordable, else ILU(0)), "ilu0", "cholesky".
    Raises if the eigensolver did not converge (require_convergence=False returns the block it has, flagged in info)."""
    rl, ctx = tl.rl, tl.ctx
    for sd in rl.subs:
        if sd.A_neu is None or sd.B_neu is None:
            raise ValueError("GenEO needs the Neumann matrices (build_structured(..., neumann=True))")
        if sd.pou is None or len(sd.pou) != sd.n:
            raise ValueError("The matrix and the partition of unity must have the same size")     # coarse_spaces.hh:323
    import time
    t0 = time.perf_counter()
    same = all(sd.B_neu is sd.A_neu for sd in rl.subs)
    dA = CsrMatrix(ctx, _block_diag([sd.A_neu for sd in rl.subs]), host_only=True)       # read on the host only (pencil assembly)
    dB = dA if same else CsrMatrix(ctx, _block_diag([sd.B_neu for sd in rl.subs]), host_only=True)
    t1 = time.perf_counter()
    par = GeneoParams()
    ctx.lib.ddm_geneo_params_default(ctypes.byref(par))
    par.nev, par.tolerance, par.shift, par.maxit, par.extra, par.seed = int(nev), float(tol), float(shift), int(maxit), int(extra), int(seed)
    par.threshold = float(threshold)
    par.nev_max = int(nev_max if nev_max is not None else 2 * nev)
    par.preconditioner = {"auto": 0, "ilu0": 1, "cholesky": 2}[preconditioner]
    if max_direct_flops is not None:          # (None: the library's per-rank time / memory budget, ddm_geneo_params_default)
        par.max_direct_flops = float(max_direct_flops)
    par.verbose = int(bool(verbose))
    kmax = max(par.nev, par.nev_max if threshold > 0 else par.nev)
    n, nsub = rl.n, len(rl.subs)
    try:      # page-locked destination: the 1.7 GB basis of the headline size comes down in 0.1 s instead of 2 s into fresh pageable pages
        import torch
        basis = torch.empty((kmax, n), dtype=torch.float64, pin_memory=True).numpy()
    except Exception:
        basis = np.empty((kmax, n), dtype=np.float64)
    nconv = np.zeros(nsub, dtype=np.int32)
    eig = np.zeros((nsub, kmax), dtype=np.float64)
    info_c = GeneoInfo()
    bp = _np(rl.block_ptr, np.int64)
    pou = _np(rl.pou, np.float64)
    dm = _np(rl.dirichlet_ovlp, np.uint8)
    ctx.check(ctx.lib.ddm_geneo_basis(ctx.h, dA.h, dB.h, nsub, _hp(bp), _hp(pou), _hp(dm), ctypes.byref(par), kmax, _hp(basis), _hp(nconv),
                                      _hp(eig), ctypes.byref(info_c)))
    t2 = time.perf_counter()
    if hasattr(tl, "setup_times"):
        tl.setup_times["GenEO inputs (block-diagonal A_neu, B_neu, upload)"] = t1 - t0
        tl.setup_times["GenEO (ddm_geneo_basis incl. basis download)"] = t2 - t1
    k = int(info_c.nev)
    info = {"iterations": int(info_c.iterations), "converged": bool(info_c.converged), "used_direct": bool(info_c.used_direct), "nev": k,
            "worst_residual": float(info_c.worst_residual), "setup_s": float(info_c.setup_s), "iterate_s": float(info_c.iterate_s),
            "direct_flops": float(info_c.direct_flops),
            "eigenvalues": {sd.id: eig[i, :k].copy() for i, sd in enumerate(rl.subs)}, "nconv": {sd.id: int(nconv[i]) for i, sd in enumerate(rl.subs)}}
    tl.geneo_info = info
    if require_convergence and not info["converged"]:
        raise RuntimeError(f"GenEO eigensolver did not converge in {info['iterations']} block iterations (worst residual {info['worst_residual']:.3e})")
    out = {}
    for i, sd in enumerate(rl.subs):
        a, b = int(rl.block_ptr[i]), int(rl.block_ptr[i + 1])
        out[sd.id] = np.ascontiguousarray(basis[:int(nconv[i]), a:b])
    return (out, info) if return_info else out


def host_eigenpair_residuals(sd, vectors, eigenvalues):
    """Independent host check (scipy only, nothing of the device path) of the GenEO pairs of ONE subdomain as ``geneo_basis`` returns
    them: vectors v_j = D x_j / ||D x_j||_2 (D = diag(pou), Dirichlet entries zero) and eigenvalues lambda_j of
    A_neu x = lambda D B_neu D x (coarse_spaces.hh:319-331).  The eigenvector itself is recovered from v: x = v / pou where pou > 0;
    on the rows G with pou = 0 (the subdomain boundary) the pencil's right-hand side vanishes, so those rows of the eigen-equation read
    A_GG x_G = -A_GI x_I and determine x_G (sparse LU of the small boundary block).  Returns for every pair
        || A_neu x - lambda D B_neu D x ||_2 / || lambda D B_neu D x ||_2      (rows with pou > 0; the G rows hold by construction)
    and the relative mismatch of the Rayleigh quotient (x^T A x) / (x^T D B D x) with lambda."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    A = sp.csr_matrix(sd.A_neu)
    B = sp.csr_matrix(sd.B_neu)
    pou = np.asarray(sd.pou, dtype=np.float64)
    V = np.asarray(vectors, dtype=np.float64)
    lam = np.asarray(eigenvalues, dtype=np.float64)[:V.shape[0]]
    I = np.nonzero(pou > 0)[0]
    G = np.nonzero(pou <= 0)[0]
    X = np.zeros((A.shape[0], V.shape[0]))
    X[I] = (V[:, I] / pou[I][None, :]).T
    if len(G):
        AGG = A[G][:, G].tocsc()
        rhs = -(A[G][:, I] @ X[I])
        X[G] = spl.splu(AGG).solve(np.ascontiguousarray(rhs))
    AX = A @ X
    CX = pou[:, None] * (B @ (pou[:, None] * X))
    R = AX - CX * lam[None, :]
    res = np.linalg.norm(R[I], axis=0) / np.linalg.norm(CX * lam[None, :], axis=0)
    rq = np.einsum("ij,ij->j", X, AX) / np.einsum("ij,ij->j", X, CX)
    return res, np.abs(rq - lam) / np.abs(lam)
