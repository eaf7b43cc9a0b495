"""The remaining coarse-space builders of dune/ddm/coarsespaces/coarse_spaces.hh as host-side mirrors over
the C ABI: every eigenproblem runs in ``ddm_geneo_basis`` / ``ddm_msgfem_basis`` and every interior solve in ``ddm_harmonic_*``
(device; csrc/geneo.hpp).  What is left here is index bookkeeping (which DoFs form the ring, its boundary layers, where the ring
eigenvectors go) and the final ``v <- D v / ||D v||`` on the assembled vectors.

  msgfem_basis              MsGFEMCoarseSpace              coarse_spaces.hh:663-831   (default of examples/poisson.ini:36)
  constraint_geneo_basis    ConstraintGenEOCoarseSpace     :394-490  (the snapshot's solve_gevp drops the constraint callback,
                                                           eigensolvers/eigensolvers.hh:27-30, so this *is* GenEO)
  geneo_ring_basis          GenEORingCoarseSpace           :502-648
  msgfem_ring_basis         MsGFEMRingCoarseSpace          :913-1163
  harmonic_extension_basis  HarmonicExtensionCoarseSpace   :1232-1266
  svd_basis                 SVDCoarseSpace                 :1268-1407
  pou_basis                 POUCoarseSpace                 :1175-1231

All return {local subdomain id: (k, n_s) ndarray}, the argument of ``TwoLevelSchwarz.set_coarse_basis``.
"""
from __future__ import annotations

import ctypes

import numpy as np
import scipy.sparse as sp

from . import CsrMatrix, GeneoInfo, GeneoParams, HarmonicExtension, _hp, _np
from .geneo import geneo_basis
from .problem import _block_diag


def _params(ctx, nev, tol, shift, maxit, extra, seed, threshold, nev_max, preconditioner, max_direct_flops, verbose, raw):
    par = GeneoParams()
    ctx.lib.ddm_geneo_params_default(ctypes.byref(par))
    par.nev, par.tolerance, par.shift, par.maxit, par.extra, par.seed = int(nev), float(tol), float(shift), int(maxit), int(extra), int(seed)
    par.threshold = float(threshold)
    par.nev_max = int(nev_max if nev_max is not None else 2 * nev)
    par.preconditioner = {"auto": 0, "ilu0": 1, "cholesky": 2}[preconditioner]
    if max_direct_flops is not None:          # (None: the library's per-rank time / memory budget, ddm_geneo_params_default)
        par.max_direct_flops = float(max_direct_flops)
    par.verbose = int(bool(verbose))
    par.raw = int(bool(raw))
    return par


def _info(info_c, eig, nconv, ids):
    k = int(info_c.nev)
    return {"iterations": int(info_c.iterations), "converged": bool(info_c.converged), "used_direct": bool(info_c.used_direct), "nev": k,
            "worst_residual": float(info_c.worst_residual), "setup_s": float(info_c.setup_s), "iterate_s": float(info_c.iterate_s),
            "eigenvalues": {s: eig[i, :k].copy() for i, s in enumerate(ids)}, "nconv": {s: int(nconv[i]) for i, s in enumerate(ids)}}


def _finalize(v, pou):
    """detail::finalize_eigenvectors (coarse_spaces.hh:52-61), rows of v"""
    v = v * pou[None, :]
    return v / np.sqrt((v * v).sum(axis=1))[:, None]


def _run_eig(ctx, which, mats, block_ptr, pou, dirichlet, boundary, par, ids, require_convergence):
    """one call of ddm_geneo_basis (which = "geneo": mats = (A, B)) or ddm_msgfem_basis ("msgfem": mats = (A_neu, A_dir))"""
    bp = _np(block_ptr, np.int64)
    n, nsub = int(bp[-1]), len(bp) - 1
    kmax = max(par.nev, par.nev_max if par.threshold > 0 else par.nev)
    basis = np.empty((kmax, n), dtype=np.float64)
    nconv = np.zeros(nsub, dtype=np.int32)
    eig = np.zeros((nsub, kmax), dtype=np.float64)
    info_c = GeneoInfo()
    d0 = CsrMatrix(ctx, mats[0])
    d1 = d0 if mats[1] is mats[0] else CsrMatrix(ctx, mats[1])
    pou = _np(pou, np.float64)
    dm = _np(dirichlet, np.uint8)
    if which == "geneo":
        ctx.check(ctx.lib.ddm_geneo_basis(ctx.h, d0.h, d1.h, nsub, _hp(bp), _hp(pou), _hp(dm), ctypes.byref(par), kmax, _hp(basis), _hp(nconv),
                                          _hp(eig), ctypes.byref(info_c)))
    else:
        bm = _np(boundary, np.uint8)
        ctx.check(ctx.lib.ddm_msgfem_basis(ctx.h, d0.h, d1.h, nsub, _hp(bp), _hp(pou), _hp(dm), _hp(bm), ctypes.byref(par), kmax, _hp(basis),
                                           _hp(nconv), _hp(eig), ctypes.byref(info_c)))
    info = _info(info_c, eig, nconv, ids)
    if require_convergence and not info["converged"]:
        raise RuntimeError(f"{which} eigensolver did not converge in {info['iterations']} block iterations (worst residual {info['worst_residual']:.3e})")
    return basis, nconv, info


def _need(sd, *names):
    for nme in names:
        if getattr(sd, nme) is None:
            raise ValueError(f"this coarse space needs SubdomainData.{nme} (build_structured(..., neumann=True))")


def msgfem_basis(tl, nev=20, tol=1e-5, shift=1e-3, maxit=400, extra=4, seed=0, verbose=False, return_info=False, threshold=-0.5, nev_max=None,
                 preconditioner="auto", max_direct_flops=None, require_convergence=True):
    """MsGFEMCoarseSpace(A_neu, A_dir, pou, dirichlet_mask, subdomain_boundary_mask, ptree) (coarse_spaces.hh:689-697)."""
    rl, ctx = tl.rl, tl.ctx
    for sd in rl.subs:
        _need(sd, "A_neu", "boundary", "pou")
        if len(sd.pou) != sd.n:
            raise ValueError("The matrix and the partition of unity must have the same size")       # :718
    par = _params(ctx, nev, tol, shift, maxit, extra, seed, threshold, nev_max, preconditioner, max_direct_flops, verbose, False)
    basis, nconv, info = _run_eig(ctx, "msgfem", (_block_diag([sd.A_neu for sd in rl.subs]), rl.A_dir), rl.block_ptr, rl.pou, rl.dirichlet_ovlp,
                                  np.concatenate([sd.boundary for sd in rl.subs]), par, [sd.id for sd in rl.subs], require_convergence)
    tl.geneo_info = info
    out = {sd.id: np.ascontiguousarray(basis[:int(nconv[i]), int(rl.block_ptr[i]):int(rl.block_ptr[i + 1])]) for i, sd in enumerate(rl.subs)}
    return (out, info) if return_info else out


def constraint_geneo_basis(tl, **kw):
    """ConstraintGenEOCoarseSpace(A_dir, A, B, pou, subdomain_boundary, ptree) (coarse_spaces.hh:410-488): the constraint callback is
    dropped by solve_gevp in this snapshot (eigensolvers/eigensolvers.hh:27-30), the basis is GenEO's."""
    return geneo_basis(tl, **kw)


def pou_basis(tl):
    """POUCoarseSpace (coarse_spaces.hh:1186-1209): the partition of unity, 2-normalised."""
    return {sd.id: (sd.pou / np.sqrt(float(sd.pou @ sd.pou)))[None, :].copy() for sd in tl.rl.subs}


def _rows(A, idx):
    """column indices of the rows idx of a CSR matrix, with the row position of every entry"""
    cnt = (A.indptr[idx + 1] - A.indptr[idx]).astype(np.int64)
    start = np.repeat(A.indptr[idx].astype(np.int64) - np.concatenate([[0], np.cumsum(cnt)[:-1]]), cnt)
    return np.repeat(np.arange(len(idx)), cnt), A.indices[np.arange(int(cnt.sum())) + start]


def _extend_and_finalize(tl, ring_vecs, rings, interiors, boundaries):
    """ring eigenvectors -> subdomain vectors, energy-minimal extension from `boundaries` into `interiors` on the device (one
    block-diagonal ddm_harmonic for all local subdomains), finalize_eigenvectors (coarse_spaces.hh:612-627 / :1120-1136)."""
    import torch
    rl, ctx = tl.rl, tl.ctx
    k = min(v.shape[0] for v in ring_vecs)
    X = np.zeros((rl.n, k))
    for i, sd in enumerate(rl.subs):
        a = int(rl.block_ptr[i])
        X[a + rings[i]] = ring_vecs[i][:k].T
    dA = CsrMatrix(ctx, rl.A_dir)
    ii = np.concatenate([int(rl.block_ptr[i]) + interiors[i] for i in range(len(rl.subs))])
    bb = np.concatenate([int(rl.block_ptr[i]) + boundaries[i] for i in range(len(rl.subs))])
    H = HarmonicExtension(ctx, dA, ii, bb, rl.block_ptr)
    Xd = torch.as_tensor(X).to(tl.dev)
    H.extend(Xd)
    ctx.sync()
    X = Xd.cpu().numpy()
    H.close()
    return {sd.id: _finalize(np.ascontiguousarray(X[int(rl.block_ptr[i]):int(rl.block_ptr[i + 1])].T), sd.pou) for i, sd in enumerate(rl.subs)}


def geneo_ring_basis(tl, ring_matrices, rings, nev=20, tol=1e-5, shift=1e-3, maxit=400, extra=4, seed=0, verbose=False, return_info=False,
                     preconditioner="auto", max_direct_flops=None, require_convergence=True):
    """GenEORingCoarseSpace(A_dir, A, pou, ring_to_subdomain, ptree) (coarse_spaces.hh:517-633).  ring_matrices[i]: the Neumann
    matrix on the ring's own numbering, rings[i] = ring_to_subdomain, per local subdomain."""
    rl, ctx = tl.rl, tl.ctx
    mod_pou, interiors, boundaries = [], [], []
    for i, sd in enumerate(rl.subs):
        A = sp.csr_matrix(sd.A_dir)
        ring = np.asarray(rings[i], dtype=np.int64)
        in_ring = np.zeros(sd.n, dtype=bool)
        in_ring[ring] = True
        r, c = _rows(A, ring)
        on_irb = np.zeros(sd.n, dtype=bool)                       # ring DoFs with a neighbour outside the ring (:548-557)
        on_irb[ring[np.unique(r[~in_ring[c]])]] = True
        mp = np.where(in_ring & ~on_irb, sd.pou, 0.0)             # :541-546, 555
        mod_pou.append(mp[ring])
        inside = np.zeros(sd.n, dtype=bool)                       # one layer inside the ring (:582-589)
        inside[ring[np.unique(r[on_irb[c]])]] = True
        inside &= ~on_irb
        interiors.append(np.concatenate([np.nonzero(~in_ring)[0], np.nonzero(on_irb)[0]]))   # :592-595
        boundaries.append(np.nonzero(inside)[0])
    bp = np.concatenate([[0], np.cumsum([len(r) for r in rings])]).astype(np.int64)
    Ar = _block_diag(ring_matrices)
    par = _params(ctx, nev, tol, shift, maxit, extra, seed, -0.5, None, preconditioner, max_direct_flops, verbose, True)
    # global Dirichlet DoFs are left out of the eigenproblem as in ddm_geneo_basis (decoupled unit modes that the caller's
    # zero_at_dirichlet would turn into zero vectors, csrc/geneo.hpp)
    dm = np.concatenate([np.asarray(sd.dirichlet_ovlp)[np.asarray(rings[i], dtype=np.int64)] for i, sd in enumerate(rl.subs)]).astype(np.uint8)
    basis, nconv, info = _run_eig(ctx, "geneo", (Ar, Ar), bp, np.concatenate(mod_pou), dm, None, par, [sd.id for sd in rl.subs], require_convergence)
    ring_vecs = [basis[:int(nconv[i]), int(bp[i]):int(bp[i + 1])] for i in range(len(rl.subs))]
    out = _extend_and_finalize(tl, ring_vecs, [np.asarray(r, dtype=np.int64) for r in rings], interiors, boundaries)
    tl.geneo_info = info
    return (out, info) if return_info else out


def msgfem_ring_basis(tl, ring_matrices, rings, overlap, shrink=0, nev=20, tol=1e-5, shift=1e-3, maxit=400, extra=4, seed=0, verbose=False,
                      return_info=False, preconditioner="auto", max_direct_flops=None, require_convergence=True):
    """MsGFEMRingCoarseSpace(A_dir, A, overlap, pou, dirichlet_mask, subdomain_boundary_mask, ring_to_subdomain, ptree)
    (coarse_spaces.hh:931-1149)."""
    from .setup_host import bfs_distance
    rl, ctx = tl.rl, tl.ctx
    width = 2 * overlap - 2 * shrink                                                           # :964
    mod_pou, dms, bms, interiors, boundaries = [], [], [], [], []
    for i, sd in enumerate(rl.subs):
        _need(sd, "boundary")
        ring = np.asarray(rings[i], dtype=np.int64)
        if len(ring) == 0:
            raise ValueError("The ring to subdomain mapping is empty, cannot build MsGFEM ring coarse space")   # :972
        # 2 overlap + 2 Gauss-Seidel sweeps (:950-962) give at least the exact distance up to 2 overlap + 2; only values up to
        # 2 overlap are compared below
        dist = bfs_distance(sp.csr_matrix(sd.A_dir), sd.boundary, 2 * overlap + 2)
        mp = np.where(dist >= shrink + width, 0.0, sd.pou)                                     # :974-976
        mod_pou.append(mp[ring])
        dms.append(np.asarray(sd.dirichlet_ovlp)[ring].astype(np.uint8))
        bms.append((sd.boundary[ring] | (dist[ring] == 2 * overlap)).astype(np.uint8))         # :978-1000
        interiors.append(np.nonzero(dist > shrink + width - 1)[0])                             # :1090-1092
        boundaries.append(np.nonzero(dist == shrink + width - 1)[0])
    bp = np.concatenate([[0], np.cumsum([len(r) for r in rings])]).astype(np.int64)
    Ar = _block_diag(ring_matrices)
    par = _params(ctx, nev, tol, shift, maxit, extra, seed, -0.5, None, preconditioner, max_direct_flops, verbose, True)
    basis, nconv, info = _run_eig(ctx, "msgfem", (Ar, Ar), bp, np.concatenate(mod_pou), np.concatenate(dms), np.concatenate(bms), par,
                                  [sd.id for sd in rl.subs], require_convergence)
    ring_vecs = [basis[:int(nconv[i]), int(bp[i]):int(bp[i + 1])] for i in range(len(rl.subs))]
    out = _extend_and_finalize(tl, ring_vecs, [np.asarray(r, dtype=np.int64) for r in rings], interiors, boundaries)
    tl.geneo_info = info
    return (out, info) if return_info else out


def harmonic_extension_basis(tl, boundary_data):
    """HarmonicExtensionCoarseSpace(A_ovlp, pou, boundary_data, subdomain_boundary_mask) (coarse_spaces.hh:1232-1266);
    boundary_data[i]: (k, number of boundary DoFs of local subdomain i)."""
    rl = tl.rl
    rings, vecs, ints, bnds = [], [], [], []
    for i, sd in enumerate(rl.subs):
        _need(sd, "boundary")
        b = np.nonzero(sd.boundary)[0]
        rings.append(b)
        vecs.append(np.asarray(boundary_data[i], dtype=np.float64).reshape(-1, len(b)))
        ints.append(np.nonzero(~sd.boundary)[0])
        bnds.append(b)
    return _extend_and_finalize(tl, vecs, rings, ints, bnds)


def svd_basis(tl, n_vectors=10, mult_pou=False, tol=1e-8, maxit=400, return_info=False, require_convergence=True):
    """SVDCoarseSpace(A_ovlp, pou, subdomain_boundary_mask, dirichlet_boundary_mask, ptree) (coarse_spaces.hh:1268-1407; keys
    `svd_coarse_space.n` and `.mult_pou`): the leading left singular vectors of T = D A_ii^-1 A_{i,Gamma} (ddm_svd_basis)."""
    rl, ctx = tl.rl, tl.ctx
    for sd in rl.subs:
        _need(sd, "boundary", "pou")
    bp = _np(rl.block_ptr, np.int64)
    nsub = len(rl.subs)
    basis = np.empty((n_vectors, rl.n), dtype=np.float64)
    sv = np.zeros((nsub, n_vectors), dtype=np.float64)
    info_c = GeneoInfo()
    dA = CsrMatrix(ctx, rl.A_dir)
    pou, dm = _np(rl.pou, np.float64), _np(rl.dirichlet_ovlp, np.uint8)
    bm = _np(np.concatenate([sd.boundary for sd in rl.subs]), np.uint8)
    ctx.check(ctx.lib.ddm_svd_basis(ctx.h, dA.h, nsub, _hp(bp), _hp(pou), _hp(dm), _hp(bm), int(n_vectors), int(bool(mult_pou)), float(tol), int(maxit),
                                    _hp(basis), _hp(sv), ctypes.byref(info_c)))
    info = {"iterations": int(info_c.iterations), "converged": bool(info_c.converged), "worst_residual": float(info_c.worst_residual),
            "singular_values": {sd.id: sv[i].copy() for i, sd in enumerate(rl.subs)}}
    if require_convergence and not info["converged"]:
        raise RuntimeError(f"svd coarse space: eigensolver did not converge in {info['iterations']} block iterations (worst residual {info['worst_residual']:.3e})")
    out = {sd.id: np.ascontiguousarray(basis[:, int(bp[i]):int(bp[i + 1])]) for i, sd in enumerate(rl.subs)}
    return (out, info) if return_info else out
