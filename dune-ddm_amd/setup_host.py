"""Host-side domain-decomposition setup (product path, vectorised numpy; no oracle involved).

Produces, from the rank-local non-overlapping data (synth.NovlpSubdomain = what
``make_communication`` + assembly give the reference), the objects the reference builds in its
layer L3 before the hot path starts:

  * the overlapping index sets of ``make_overlapping_communication``
    (dune/ddm/overlap_extension.hh:53-285) with the reference's local numbering (non-overlapping
    indices first, new indices appended per round in arrival order: neighbours by ascending rank,
    shared indices by ascending global id, graph neighbours in row order);
  * the subdomain-boundary mask (IdentifyBoundaryDataHandle, dune/ddm/datahandles.hh:122-192);
  * the PartitionOfUnity vector (dune/ddm/pou.hh:57-141);
  * the index lists of the three DUNE interfaces used in the Krylov loop.

It works with global knowledge of all subdomains (the hot path is what runs distributed), and
replaces the reference's in-place Gauss-Seidel distance sweeps by a breadth-first search: the
sweeps only ever use distances <= 4*overlap, for which both give the exact graph distance.
tests/test_setup_host.py checks all integer maps bit-exactly against the oracle's literal
message-passing restatement.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
import scipy.sparse as sp

INT_INF = np.iinfo(np.int32).max - 1


@dataclass
class OvlpIndexSet:
    rank: int
    n_o: int
    glob: np.ndarray          # int64[n]; first n_o entries = non-overlapping set
    owner: np.ndarray         # uint8[n]
    public: np.ndarray        # uint8[n]
    ext_boundary: np.ndarray  # bool[n]: indices added in the last round (overlap_extension.hh:281-282)
    round_sizes: list = field(default_factory=list)


def _csr_rows(A: sp.csr_matrix, rows: np.ndarray):
    """(row_of_entry (position in ``rows``), col) of the entries of the selected rows, row order."""
    cnt = (A.indptr[rows + 1] - A.indptr[rows]).astype(np.int64)
    tot = int(cnt.sum())
    if tot == 0:
        return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
    start = np.repeat(A.indptr[rows].astype(np.int64) - np.concatenate([[0], np.cumsum(cnt)[:-1]]), cnt)
    pos = np.arange(tot, dtype=np.int64) + start
    return np.repeat(np.arange(len(rows), dtype=np.int64), cnt), A.indices[pos].astype(np.int64)


def bfs_distance(A: sp.csr_matrix, start: np.ndarray, maxdist: int) -> np.ndarray:
    """Graph distance to the start set, exact up to ``maxdist`` (INT_INF beyond)."""
    n = A.shape[0]
    dist = np.full(n, INT_INF, dtype=np.int64)
    frontier = np.nonzero(start)[0]
    dist[frontier] = 0
    d = 0
    while len(frontier) and d < maxdist:
        _, cols = _csr_rows(A, frontier)
        cols = np.unique(cols)
        new = cols[dist[cols] > d + 1]
        dist[new] = d + 1
        frontier = new
        d += 1
    return dist


class _World:
    """loc-of-gid lookup tables for all ranks (global knowledge)."""

    def __init__(self, nglobal, subs):
        self.nglobal = nglobal
        self.P = len(subs)
        self.glob = [np.asarray(s.glob, dtype=np.int64).copy() for s in subs]
        self.loc = []
        self.public = [np.asarray(s.public, dtype=bool).copy() for s in subs]
        for g in self.glob:
            l = np.full(nglobal, -1, dtype=np.int32)
            l[g] = np.arange(len(g), dtype=np.int32)
            self.loc.append(l)

    def pubmask(self, p):
        m = np.zeros(self.nglobal, dtype=bool)
        m[self.glob[p][self.public[p]]] = True
        return m

    def shared(self, pub, p, q):
        """ascending global ids both ranks know and flag public (RemoteIndices::rebuild<false>)"""
        return np.nonzero(pub[p] & pub[q])[0]


def identify_boundary(world: _World, mats, nbrs):
    """datahandles.hh:122-192 -- mats[q] is the matrix whose rows q sends (rows < mats[q].shape[0])."""
    pub = [world.pubmask(p) for p in range(world.P)]
    out = []
    for p in range(world.P):
        mask = np.zeros(len(world.glob[p]), dtype=bool)
        for q in nbrs[p]:
            g = world.shared(pub, p, q)
            lq = world.loc[q][g].astype(np.int64)
            keep = lq < mats[q].shape[0]
            g, lq = g[keep], lq[keep]
            r, c = _csr_rows(mats[q], lq)
            offdiag = c != lq[r]
            unknown = world.loc[p][world.glob[q][c]] < 0
            hit = np.unique(r[offdiag & unknown])
            mask[world.loc[p][g[hit]]] = True
        out.append(mask)
    return out


def _neighbours(world: _World):
    pub = [world.pubmask(p) for p in range(world.P)]
    nb = [[] for _ in range(world.P)]
    for p in range(world.P):
        for q in range(p + 1, world.P):
            if (pub[p] & pub[q]).any():
                nb[p].append(q)
                nb[q].append(p)
    return nb, pub


def make_overlapping_communication(subs, overlap: int, nglobal: int):
    """overlap_extension.hh:53-285 -> list[OvlpIndexSet]."""
    if overlap <= 0:
        raise ValueError(f"make_overlapping_communication: overlap must be positive, got {overlap}")   # :72-75
    world = _World(nglobal, subs)
    mats = [sp.csr_matrix(s.A) for s in subs]
    n_o = [len(s.glob) for s in subs]
    owner = [np.asarray(s.owner, dtype=np.uint8).copy() for s in subs]
    nbrs, _ = _neighbours(world)
    bmask = identify_boundary(world, mats, nbrs)
    for p in range(world.P):   # public |= boundary distance <= overlap + 2  (:105-149, 180)
        dist = bfs_distance(mats[p], bmask[p], overlap + 2)
        world.public[p] = world.public[p] | (dist <= overlap + 2)
    nbrs, pub = _neighbours(world)
    sizes = [[n] for n in n_o]
    for _round in range(overlap):
        additions = []
        for p in range(world.P):
            known = world.loc[p] >= 0
            new_all = []
            for q in nbrs[p]:                                   # ascending rank
                g = world.shared(pub, p, q)                     # ascending global id
                lq = world.loc[q][g].astype(np.int64)
                lq = lq[lq < n_o[q]]                            # "if (i < A.N())"
                r, c = _csr_rows(mats[q], lq)
                cand = world.glob[q][c[c != lq[r]]]
                cand = cand[~known[cand]]
                if len(cand):
                    _, first = np.unique(cand, return_index=True)
                    cand = cand[np.sort(first)]                 # first occurrence, arrival order
                    known[cand] = True
                    new_all.append(cand)
            additions.append(np.concatenate(new_all) if new_all else np.zeros(0, dtype=np.int64))
        for p, new in enumerate(additions):                     # appended as copy / public (:260-262)
            base = len(world.glob[p])
            world.loc[p][new] = base + np.arange(len(new), dtype=np.int32)
            world.glob[p] = np.concatenate([world.glob[p], new])
            world.public[p] = np.concatenate([world.public[p], np.ones(len(new), dtype=bool)])
            owner[p] = np.concatenate([owner[p], np.zeros(len(new), dtype=np.uint8)])
            sizes[p].append(len(world.glob[p]))
        nbrs, pub = _neighbours(world)
    out = []
    for p in range(world.P):
        eb = np.zeros(len(world.glob[p]), dtype=bool)
        eb[sizes[p][overlap - 1]:sizes[p][overlap]] = True
        out.append(OvlpIndexSet(p, n_o[p], world.glob[p], owner[p], world.public[p].astype(np.uint8), eb, sizes[p]))
    return out


def interface_pairs(index_sets, nglobal: int, kind: str):
    """{(src, dst): (idx_src, idx_dst)} ordered by ascending global id; kind = 'all_to_all'
    (addOwnerCopyToOwnerCopy / addOwnerCopyToAll) or 'owner_to_all' (copyOwnerToAll)."""
    world = _World(nglobal, index_sets)
    nbrs, pub = _neighbours(world)
    out = {}
    for src in range(world.P):
        for dst in nbrs[src]:
            g = world.shared(pub, src, dst)
            ls = world.loc[src][g].astype(np.int64)
            ld = world.loc[dst][g].astype(np.int64)
            if kind == "owner_to_all":
                keep = np.asarray(index_sets[src].owner, dtype=bool)[ls]
                ls, ld = ls[keep], ld[keep]
            elif kind != "all_to_all":
                raise ValueError(kind)
            if len(ls):
                out[(src, dst)] = (ls, ld)
    return out


def add_to_all(pairs, vecs):
    """addOwnerCopyToAll on host vectors (setup only): gather first, then add by ascending source."""
    bufs = {k: vecs[k[0]][s].copy() for k, (s, d) in pairs.items()}
    out = [v.copy() for v in vecs]
    for (src, dst) in sorted(pairs, key=lambda k: (k[1], k[0])):
        out[dst][pairs[(src, dst)][1]] += bufs[(src, dst)]
    return out


def subdomain_boundary(index_sets, A_dir, nglobal):
    world = _World(nglobal, index_sets)
    nbrs, _ = _neighbours(world)
    return identify_boundary(world, [sp.csr_matrix(A) for A in A_dir], nbrs)


def partition_of_unity(index_sets, A_dir, pairs_all, nglobal, pou_type="distance", shrink=0, overlap=0):
    """dune/ddm/pou.hh:57-141.  Returns (pou list, boundary masks, boundary distances)."""
    if pou_type == "trivial":
        return [np.asarray(s.owner, dtype=np.float64).copy() for s in index_sets], None, None
    if pou_type not in ("standard", "distance"):
        raise ValueError("Unknown partition of unity type: " + str(pou_type))            # :176
    bmask = subdomain_boundary(index_sets, A_dir, nglobal)
    if pou_type == "standard":
        w = [np.where(b, 0.0, 1.0) for b in bmask]
        s = add_to_all(pairs_all, w)
        return [np.where(b, 0.0, 1.0 / np.where(b, 1.0, sv)) for b, sv in zip(bmask, s)], bmask, None
    if shrink < 0 or shrink >= max(overlap, 1):
        raise ValueError(f"Invalid value for shrink: {shrink} (must be >= 0 and < overlap size {overlap})")   # :184
    w, dists = [], []
    for A, b in zip(A_dir, bmask):
        dist = bfs_distance(sp.csr_matrix(A), b, 4 * overlap + 1)
        wv = np.ones(A.shape[0])
        sel = dist <= 4 * overlap
        wv[sel] = np.where(dist[sel] <= shrink, 0.0, (dist[sel] - shrink).astype(float))
        w.append(wv)
        dists.append(dist)
    s = add_to_all(pairs_all, w)
    pou = [np.where(b, 0.0, wv / np.where(b, 1.0, sv)) for b, wv, sv in zip(bmask, w, s)]
    return pou, bmask, dists
