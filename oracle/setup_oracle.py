"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of dune-ddm's *setup* half of the two-level Schwarz path, as a single-process
simulation of the reference's multi-rank message passing.  Pure Python / numpy loops -- only
for small cases.  Every function cites the reference lines it follows (paths relative to
/root/reference).

Pinned against the reference's own golden data in tests/test_oracle_kat.py:
  * tests/test_galerkin_coarse_matrix.cc:20-48,77-212  (9x9 chain, overlap 6 -> full matrix)
  * tests/test_galerkin_coarse_matrix.cc:50-67,216-283 (overlap 1 + POU -> 4x4 R A R^T)
dune-common / dune-istl semantics (RemoteIndices, Interface, BufferedCommunicator) are not in
the snapshot; they are restated from upstream knowledge as documented in SURVEY.md 8c.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
import scipy.sparse as sp

INT_INF = np.iinfo(np.int32).max - 1  # std::numeric_limits<int>::max() - 1 (pou.hh:100)


@dataclass
class RankIndexSet:
    """Simulated ParallelIndexSet + RemoteIndices of one rank."""
    rank: int
    n_o: int
    glob: list            # local -> global id (grows during the extension)
    owner: list           # attribute == owner
    public: list
    loc: dict = field(default_factory=dict)  # global -> local
    neighbours: set = field(default_factory=set)

    def __post_init__(self):
        self.loc = {int(g): i for i, g in enumerate(self.glob)}

    @property
    def n(self):
        return len(self.glob)


def shared_indices(a: RankIndexSet, b: RankIndexSet):
    """RemoteIndices::rebuild<false>: the indices both ranks know *and* both flag public,
    sorted by global id (so sender and receiver agree on the order without exchanging it,
    SURVEY.md App. A.11).  Returns [(global, loc_a, loc_b)]."""
    if len(a.loc) > len(b.loc):
        return [(g, lb, la) for (g, la, lb) in shared_indices(b, a)]
    out = []
    for g, la in a.loc.items():
        lb = b.loc.get(g)
        if lb is not None and a.public[la] and b.public[lb]:
            out.append((g, la, lb))
    out.sort()
    return out


def _rebuild_neighbours(ranks):
    for r in ranks:
        r.neighbours = set()
    for i, a in enumerate(ranks):
        for b in ranks[i + 1:]:
            if shared_indices(a, b):
                a.neighbours.add(b.rank)
                b.neighbours.add(a.rank)


def identify_boundary(ranks, mats):
    """IdentifyBoundaryDataHandle (dune/ddm/datahandles.hh:122-192): for every index i shared
    with a neighbour q, q sends the global ids of the off-diagonal entries of *its* row i
    (:152-157); the receiver marks i if any id is unknown locally (:168-174)."""
    masks = [np.zeros(r.n, dtype=bool) for r in ranks]
    for p in ranks:
        for qn in sorted(p.neighbours):
            q = ranks[qn]
            Aq = mats[qn]
            for g, lp, lq in shared_indices(p, q):
                if lq >= Aq.shape[0]:
                    continue
                for c in Aq.indices[Aq.indptr[lq]:Aq.indptr[lq + 1]]:
                    if c != lq and int(q.glob[c]) not in p.loc:
                        masks[p.rank][lp] = True
                        break
    return masks


def make_overlapping_communication(subs, overlap: int):
    """make_overlapping_communication (dune/ddm/overlap_extension.hh:53-285).

    ``subs``: list of objects with .rank .glob .owner .public .A (CSR, n_o x n_o).
    Returns (list[RankIndexSet] on the overlapping sets, list[ext_boundary_mask]).
    New indices are appended round by round in arrival order (:260-262): neighbours in
    ascending rank order, shared indices in ascending global id, matrix-graph neighbours in
    row order (datahandles.hh:251-279)."""
    assert overlap > 0  # :72-75
    ranks = [RankIndexSet(s.rank, len(s.glob), [int(g) for g in s.glob], [bool(o) for o in s.owner],
                          [bool(p) for p in s.public]) for s in subs]
    mats = [s.A.tocsr() for s in subs]
    _rebuild_neighbours(ranks)
    # boundary distance by BFS on the local matrix graph, capped at overlap + 2 (:105-140)
    bmask = identify_boundary(ranks, mats)
    for r, A in zip(ranks, mats):
        dist = np.full(r.n, INT_INF, dtype=np.int64)
        queue = [i for i in range(r.n) if bmask[r.rank][i]]
        dist[queue] = 0
        head = 0
        while head < len(queue):
            cur = queue[head]
            head += 1
            if dist[cur] >= overlap + 2:
                continue
            for c in A.indices[A.indptr[cur]:A.indptr[cur + 1]]:
                if dist[c] > dist[cur] + 1:
                    dist[c] = dist[cur] + 1
                    queue.append(int(c))
        # modify_parindexset_public_state (:143-149, 180)
        r.public = [bool(pb or dist[i] <= overlap + 2) for i, pb in enumerate(r.public)]
    _rebuild_neighbours(ranks)
    sizes = [[r.n] for r in ranks]
    for _round in range(overlap):  # :205-276
        snapshot = [list(r.glob) for r in ranks]           # ltg_copy (datahandles.hh:225)
        shared = {(p.rank, qn): shared_indices(p, ranks[qn]) for p in ranks for qn in p.neighbours}
        additions = []
        for p in ranks:
            new, seen = [], set()
            for qn in sorted(p.neighbours):
                Aq = mats[qn]
                for g, lp, lq in shared[(p.rank, qn)]:
                    if lq >= Aq.shape[0]:                   # "if (i < A.N())" (datahandles.hh:246,256)
                        continue
                    for c in Aq.indices[Aq.indptr[lq]:Aq.indptr[lq + 1]]:
                        if c == lq:
                            continue
                        gc = snapshot[qn][c]
                        if gc not in p.loc and gc not in seen:   # gis.count(gi) (:276-279)
                            seen.add(gc)
                            new.append(gc)
            additions.append(new)
        for p, new in zip(ranks, additions):                # ext_indexset.add(..., copy, public) (:260-262)
            for g in new:
                p.loc[g] = len(p.glob)
                p.glob.append(g)
                p.owner.append(False)
                p.public.append(True)
            sizes[p.rank].append(p.n)
        _rebuild_neighbours(ranks)
    ext_boundary = []
    for p in ranks:                                         # :281-282
        m = np.zeros(p.n, dtype=bool)
        m[sizes[p.rank][overlap - 1]:sizes[p.rank][overlap]] = True
        ext_boundary.append(m)
    return ranks, ext_boundary


def interface_pairs(ranks, kind: str):
    """Per ordered pair (src -> dst) the (src_local, dst_local) index lists of the three DUNE
    interfaces used on the hot path: 'owner_to_all' (copyOwnerToAll) keeps the
    pairs whose source attribute is owner; 'all_to_all' (addOwnerCopyToOwnerCopy /
    addOwnerCopyToAll -- only owner and copy attributes exist, overlap_extension.hh:261)."""
    out = {}
    for src in ranks:
        for dn in sorted(src.neighbours):
            dst = ranks[dn]
            sh = shared_indices(src, dst)
            if kind == "owner_to_all":
                sh = [(g, ls, ld) for (g, ls, ld) in sh if src.owner[ls]]
            elif kind != "all_to_all":
                raise ValueError(kind)
            if sh:
                out[(src.rank, dn)] = (np.array([s for _, s, _ in sh], dtype=np.int64),
                                       np.array([d for _, _, d in sh], dtype=np.int64))
    return out


def add_vector(ranks, vecs):
    """AddVectorDataHandle forward on the all-all interface (datahandles.hh:16-79): every holder
    ends with the sum of all holders' pre-exchange values (send buffers are packed first)."""
    pairs = interface_pairs(ranks, "all_to_all")
    bufs = {k: vecs[k[0]][s].copy() for k, (s, d) in pairs.items()}
    out = [v.copy() for v in vecs]
    for (src, dst) in sorted(pairs, key=lambda k: (k[1], k[0])):
        out[dst][pairs[(src, dst)][1]] += bufs[(src, dst)]
    return out


def overlapping_matrix(ranks, subs, dirichlet_novlp=None):
    """CreateMatrixDataHandle + AddMatrixDataHandle (dune/ddm/datahandles.hh:436-591) followed by
    the symmetric Dirichlet elimination of examples/pdelab_helper.hh:33-46,296-304,429:
    row i of the overlapping matrix is the sum of row i of every rank's *additive* matrix,
    restricted to locally known columns.  Returns (A_dir list, dirichlet_mask_ovlp list)."""
    mats = [s.A.tocsr() for s in subs]
    out = []
    for p in ranks:
        A = mats[p.rank].tocoo()
        rows, cols, vals = [A.row.astype(np.int64)], [A.col.astype(np.int64)], [A.data.copy()]
        for qn in sorted(p.neighbours):
            q, Aq = ranks[qn], mats[qn]
            r_, c_, v_ = [], [], []
            for g, lp, lq in shared_indices(p, q):
                if lq >= Aq.shape[0]:
                    continue
                for k in range(Aq.indptr[lq], Aq.indptr[lq + 1]):
                    lc = p.loc.get(int(q.glob[Aq.indices[k]]))
                    if lc is not None:
                        r_.append(lp); c_.append(lc); v_.append(Aq.data[k])
            rows.append(np.array(r_, dtype=np.int64)); cols.append(np.array(c_, dtype=np.int64)); vals.append(np.array(v_))
        M = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(p.n, p.n))
        M.sort_indices()
        out.append(M)
    if dirichlet_novlp is None:
        return out, None
    dm = []
    for p in ranks:
        v = np.zeros(p.n)
        v[:p.n_o] = dirichlet_novlp[p.rank]
        dm.append(v)
    dm = add_vector(ranks, dm)                             # pdelab_helper.hh:296-302
    dm = [(v > 0).astype(np.uint8) for v in dm]
    out = [eliminate_dirichlet(M, m) for M, m in zip(out, dm)]
    return out, dm


def eliminate_dirichlet(M, dmask):
    """examples/pdelab_helper.hh:33-46."""
    M = M.copy().tocsr()
    for i in range(M.shape[0]):
        for k in range(M.indptr[i], M.indptr[i + 1]):
            j = M.indices[k]
            if dmask[i] > 0:
                M.data[k] = 1.0 if j == i else 0.0
            elif dmask[j] > 0:
                M.data[k] = 0.0
    return M


def graph_distance_sweeps(A, start_mask, rounds):
    """In-place Gauss-Seidel minima in index order (pou.hh:98-111)."""
    dist = np.full(A.shape[0], INT_INF, dtype=np.int64)
    dist[start_mask] = 0
    indptr, indices = A.indptr, A.indices
    for _ in range(rounds):
        for i in range(A.shape[0]):
            d = dist[i]
            for c in indices[indptr[i]:indptr[i + 1]]:
                if dist[c] + 1 < d:
                    d = dist[c] + 1
            dist[i] = d
    return dist


def partition_of_unity(ranks, A_dir, pou_type="distance", shrink=0, overlap=0):
    """PartitionOfUnity (dune/ddm/pou.hh:57-141)."""
    if pou_type == "trivial":                               # :132-139
        return [np.array([1.0 if o else 0.0 for o in r.owner]) for r in ranks], None
    bmask = identify_boundary(ranks, A_dir)                 # :65-77
    if pou_type == "standard":                              # :80-94
        w = [np.where(b, 0.0, 1.0) for b in bmask]
        s = add_vector(ranks, w)
        return [np.where(b, 0.0, 1.0 / np.where(b, 1.0, sv)) for b, sv in zip(bmask, s)], bmask
    if pou_type != "distance":
        raise ValueError("Unknown partition of unity type: " + pou_type)   # :176
    if shrink < 0 or shrink >= max(overlap, 1):
        raise ValueError("Invalid value for shrink")        # :184
    w = []
    for r, A, b in zip(ranks, A_dir, bmask):
        dist = graph_distance_sweeps(A, b, 4 * overlap + 1)  # :106-111 (round = 0..4*overlap)
        wv = np.ones(r.n)
        sel = dist <= 4 * overlap                           # :115-120
        wv[sel] = np.where(dist[sel] <= shrink, 0.0, (dist[sel] - shrink).astype(float))
        w.append(wv)
    s = add_vector(ranks, w)                                # :123-124
    pou = [np.where(b, 0.0, wv / np.where(b, 1.0, sv)) for b, wv, sv in zip(bmask, w, s)]  # :127-129
    return pou, bmask


def neumann_region_masks(A_dir, bmask, overlap):
    """examples/pdelab_helper.hh:151-158,181-196: dist = graph distance to the overlapping
    subdomain boundary; 'overlap' region = dist <= 2*overlap."""
    out = []
    for A, b in zip(A_dir, bmask):
        dist = graph_distance_sweeps(A, b, 4 * overlap + 1)
        out.append(dist <= 2 * overlap)
    return out
