"""Overlapping Dirichlet / Neumann matrices by ASSEMBLY INTERCEPTION, the way the reference's drivers obtain them
(examples/assemblewrapper.hh:182-367 `AssembleWrapper`, examples/pdelab_helper.hh:113-436 `assemble_overlapping_matrices`).

The reference never assembles a Neumann matrix directly.  Every rank assembles its own elements once more with a wrapped local
operator that ALSO records, element by element, what an element adds to pairs of degrees of freedom on the boundary of some
overlapping subdomain although the element itself reaches outside that subdomain ("Neumann corrections"); the corrections travel to
the subdomain's rank as (global row, global column, value) triples and are subtracted from the overlapping Dirichlet matrix:

    A_neu = A_dir - (corrections of the elements that straddle the boundary of the overlapping subdomain)          NeumannRegion::All
    B_neu = A_neu restricted to the overlap region - (own corrections at the inner boundary of the overlap region)  NeumannRegion::Overlap

This module is the host-side mirror of that path for the element providers of dune_ddm_amd.synth (which stand where PDELab's grid
operator stands): vectorised numpy over the elements of a rank, global knowledge of all ranks in one process as the rest of
setup_host.py.  oracle/neumann_oracle.py restates the reference's per-element loops literally; tests/test_neumann_assembly.py checks
this module against it bit for bit and both against the directly assembled region matrices of synth.py (`build_structured(...,
neumann=True)`), which is what the benchmark uses.

A DUNE application keeps using the reference's own AssembleWrapper: nothing in it touches the hot path."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import scipy.sparse as sp

from . import setup_host as sh
from .synth import eliminate_dirichlet

REGIONS = ("overlap", "extended_overlap", "all")      # NeumannRegion (examples/pdelab_helper.hh:17-21)


@dataclass
class Triples:
    """corrections one rank holds for another (rank >= 0: global ids) or for itself (rank -1: local ids), in the order the
    reference's BCRSMatrix iteration produces them (ascending row, ascending column)"""
    rank: int
    row: np.ndarray
    col: np.ndarray
    val: np.ndarray


@dataclass
class OverlappingMatrices:
    A_dir: sp.csr_matrix
    A_neu: sp.csr_matrix
    B_neu: sp.csr_matrix
    dirichlet_mask_ovlp: np.ndarray
    boundary_dst: np.ndarray
    triples_sent: dict       # {destination rank (or -1): Triples}


def boundary_masks(idx, A_dir, nglobal, overlap):
    """per rank: distance to the boundary of the overlapping subdomain (exact up to 4 overlap + 1, pdelab_helper.hh:150-158) and the
    indicator vector the rank sends to its neighbours (1 on the boundary, 2 elsewhere: :160-166)"""
    bmask = sh.subdomain_boundary(idx, A_dir, nglobal)
    dist = [sh.bfs_distance(sp.csr_matrix(A), b, 4 * overlap + 1) for A, b in zip(A_dir, bmask)]
    indicator = [np.where(d == 0, 1, 2).astype(np.uint8) for d in dist]
    return bmask, dist, indicator


def _accumulate(n, rows, cols, vals):
    """sum of the triples as a CSR matrix; duplicates are added in the order they come (the order of the element loop)"""
    key = rows.astype(np.int64) * n + cols.astype(np.int64)
    uk, inv = np.unique(key, return_inverse=True)
    acc = np.zeros(len(uk))
    np.add.at(acc, inv, vals)
    return (uk // n).astype(np.int64), (uk % n).astype(np.int64), acc


def element_corrections(dofs, Ke, on_mask, out_mask, n):
    """AssembleWrapper::jacobian_volume, one destination: elements with a DoF on the boundary (`on_mask`) AND a DoF outside
    (`out_mask`) contribute their entries between boundary DoFs (assemblewrapper.hh:207-236, 239-262).  dofs: (ne, nc) local
    indices, Ke: (ne, nc, nc).  Returns (row, col, val) sorted by (row, col); every (boundary, boundary) pair of the stored pattern
    that no element touches is NOT listed (the reference lists it with value 0: same matrix after subtraction)."""
    onb = on_mask[dofs]
    sel = onb.any(axis=1) & out_mask[dofs].any(axis=1)
    if not sel.any():
        z = np.zeros(0, dtype=np.int64)
        return z, z, np.zeros(0)
    d, K, o = dofs[sel], Ke[sel], onb[sel]
    pair = o[:, :, None] & o[:, None, :]
    e, a, b = np.nonzero(pair)                        # element-major, then row corner, then column corner: the loop order
    return _accumulate(n, d[e, a], d[e, b], K[e, a, b])


def assemble_overlapping_matrices(grid, overlap: int, first_region: str = "all", second_region: str = "overlap"):
    """examples/pdelab_helper.hh:113-436 for all ranks of ``grid`` at once.  ``grid`` provides subdomains() / subdomain(r) (the
    non-overlapping matrices of make_communication), elements(r) -> (dofs, Ke) of the rank's own elements in its local numbering,
    dirichlet_of(glob) and dirichlet_matrix(glob, dmask) (the result of CreateMatrixDataHandle + AddMatrixDataHandle + symmetric
    Dirichlet elimination, checked against the message-passing restatement in tests/test_setup_dist.py).
    Returns a list of OverlappingMatrices, one per rank (matrix_size_eq_subdomain = true)."""
    if first_region not in REGIONS or second_region not in REGIONS:
        raise NotImplementedError("Unknown neumann_region type")                                                   # :428
    if first_region != "all" and first_region != second_region:
        raise NotImplementedError("Two different Neumann regions are only supported if the first is NeumannRegion::All")   # :181
    if second_region == "extended_overlap" and first_region == "all":
        raise NotImplementedError("Unknown neumann_region type")                                                   # :428 (only Overlap as a second region)
    nov = grid.subdomains() if not hasattr(grid, "subdomain") else [grid.subdomain(r) for r in range(grid.nranks)]
    ng = grid.nglobal
    idx = sh.make_overlapping_communication(nov, overlap, ng)
    dmask = [grid.dirichlet_of(i.glob) for i in idx]
    A_dir = [grid.dirichlet_matrix(i.glob, dm) for i, dm in zip(idx, dmask)]
    bmask, dist, indicator = boundary_masks(idx, A_dir, ng, overlap)
    P = len(nov)
    loc = []
    for i in idx:
        l = np.full(ng, -1, dtype=np.int64)
        l[i.glob] = np.arange(len(i.glob))
        loc.append(l)
    width = {"overlap": 2 * overlap, "extended_overlap": 2 * overlap + 1}
    inner = second_region if second_region != "all" else None          # the region that needs the rank's OWN corrections
    # ---- every rank assembles its elements once, recording corrections for every rank whose overlapping subdomain it touches ----
    sent = [dict() for _ in range(P)]
    for p in range(P):
        n_o = idx[p].n_o
        dofs, Ke = grid.elements(p)
        gl = idx[p].glob[:n_o]
        for q in range(P):
            if q == p:
                continue
            lq = loc[q][gl]                            # the copied indicator vector (CopyVectorDataHandleWithRank, :168-180)
            if not (lq >= 0).any():
                continue
            ind = np.where(lq >= 0, indicator[q][np.maximum(lq, 0)], 0)
            r, c, v = element_corrections(dofs, Ke, ind == 1, ind == 0, n_o)
            sent[p][q] = Triples(q, gl[r], gl[c], v)                                  # get_correction_triples: global ids (:440-456)
        if inner is not None:
            w = width[inner]
            r, c, v = element_corrections(dofs, Ke, dist[p][:n_o] == w, dist[p][:n_o] == w + 1, n_o)
            sent[p][-1] = Triples(-1, r, c, v)                                        # own corrections: local ids (:458-470)
    # ---- every rank subtracts what it received (ascending source rank, the std::map order of :224-238) ----
    out = []
    for q in range(P):
        A = sp.csr_matrix(A_dir[q]).copy()
        A.sort_indices()
        n = A.shape[0]

        def subtract(M, rows, cols, vals):
            # M[r][c] -= v entry by entry, in the order given (a pair may occur in the triples of several source ranks)
            pos = _positions(M, rows, cols)
            np.subtract.at(M.data, pos, vals)

        if first_region == "all":
            A_neu = A.copy()
            for p in range(P):
                t = sent[p].get(q)
                if t is not None and len(t.val):
                    subtract(A_neu, loc[q][t.row], loc[q][t.col], t.val)
            A_neu = eliminate_dirichlet(A_neu, dmask[q])
        else:
            A_neu = _restrict(A, dist[q] <= width[first_region])
            for p in range(P):
                t = sent[p].get(q)
                if t is not None and len(t.val):
                    subtract(A_neu, loc[q][t.row], loc[q][t.col], t.val)
            t = sent[q][-1]
            subtract(A_neu, t.row, t.col, t.val)
            A_neu = eliminate_dirichlet(A_neu, dmask[q])
        if second_region == first_region:
            B_neu = A_neu
        else:
            B_neu = _restrict(A_neu, dist[q] <= 2 * overlap)                          # :404-417 (copied from the finished A_neu)
            t = sent[q][-1]
            subtract(B_neu, t.row, t.col, t.val)
            B_neu = eliminate_dirichlet(B_neu, dmask[q])
        out.append(OverlappingMatrices(A_dir[q], A_neu, B_neu, dmask[q], dist[q], sent[q]))
    return out


def _restrict(A: sp.csr_matrix, keep: np.ndarray) -> sp.csr_matrix:
    """the entries of A whose row and column are kept; same size, pattern = the kept entries (pdelab_helper.hh:302-313)"""
    A = sp.csr_matrix(A)
    rows = np.repeat(np.arange(A.shape[0]), np.diff(A.indptr))
    sel = keep[rows] & keep[A.indices]
    M = sp.csr_matrix((A.data[sel], (rows[sel], A.indices[sel])), shape=A.shape)
    M.sort_indices()
    return M


def _positions(M: sp.csr_matrix, rows, cols):
    """positions of the entries (rows[k], cols[k]) in M.data; all of them must be stored entries"""
    rows = np.asarray(rows, dtype=np.int64)
    cols = np.asarray(cols, dtype=np.int64)
    n = M.shape[1]
    key = np.repeat(np.arange(M.shape[0], dtype=np.int64), np.diff(M.indptr)) * n + M.indices      # ascending (sorted indices)
    want = rows * n + cols
    pos = np.searchsorted(key, want)
    if len(want) and ((pos >= len(key)).any() or (key[np.minimum(pos, len(key) - 1)] != want).any()):
        raise ValueError("Neumann correction for an entry the overlapping matrix does not store")
    return pos
