"""Distributed domain-decomposition setup: the reference's layer L3 run rank by rank.

``setup_host`` produces the same objects with global knowledge of all subdomains; here every rank holds only what one MPI rank
of the reference holds after ``make_communication`` + assembly (synth.NovlpSubdomain: global ids, owner / public flags, the additive
matrix) and talks to other ranks through ONE primitive, ``Exchange.sparse`` (any-to-any messages of int64 words whose receivers are
not known in advance: counts by an all-gather, payload point to point) -- what DUNE's RemoteIndices::rebuild,
VariableSizeCommunicator and BufferedCommunicator are used for in the reference:

  * ``make_overlapping_communication``   dune/ddm/overlap_extension.hh:53-285 (+ IndexsetExtensionMatrixGraphDataHandle,
                                         dune/ddm/datahandles.hh:209-335; IdentifyBoundaryDataHandle, :122-192)
  * ``overlapping_matrix``               CreateMatrixDataHandle / AddMatrixDataHandle (datahandles.hh:436-591) + the Dirichlet mask
                                         and symmetric elimination of examples/pdelab_helper.hh:33-46, 296-304
  * ``partition_of_unity``               dune/ddm/pou.hh:57-141
  * ``interfaces``                       the index lists of copyOwnerToAll / addOwnerCopyToOwnerCopy

Local numbering, arrival order of new indices (neighbours by ascending rank, shared indices by ascending global id, graph
neighbours in row order) and all floating-point sums (ascending source rank) equal ``setup_host`` / the oracle bit for bit:
tests/test_setup_dist.py runs this on P threads and over gloo processes and compares every array.

Who shares an index with whom is found through a distributed directory (global id g is registered at rank g mod P) instead of
DUNE's ring algorithm in RemoteIndices::rebuild and the rank-map propagation of UpdateRankInfoDataHandle (overlap_extension.hh:
205-257): same result -- two ranks are neighbours iff they share a public index -- in two sparse exchanges per rebuild.
"""
from __future__ import annotations

import threading

import numpy as np
import scipy.sparse as sp

from .setup_host import INT_INF, OvlpIndexSet, bfs_distance


# ---- the exchange primitive -------------------------------------------------------------------------------------------------------
class Exchange:
    rank: int
    size: int

    def sparse(self, send: dict) -> dict:
        """send: {destination rank: int64 array}; returns {source rank: int64 array} of everything addressed to this rank."""
        raise NotImplementedError


class ThreadExchange(Exchange):
    """P ranks as P threads of one process (tests, and one-process multi-subdomain runs)."""

    class Hub:
        def __init__(self, size):
            self.size = size
            self.barrier = threading.Barrier(size)
            self.box = [dict() for _ in range(size)]

    def __init__(self, hub, rank):
        self.hub, self.rank, self.size = hub, rank, hub.size

    def sparse(self, send):
        for dst, buf in send.items():
            self.hub.box[dst][self.rank] = np.ascontiguousarray(buf, dtype=np.int64).copy()
        self.hub.barrier.wait()
        got = self.hub.box[self.rank]
        self.hub.box[self.rank] = dict()
        self.hub.barrier.wait()
        return got


class TorchExchange(Exchange):
    """torch.distributed process group (gloo on the host; the setup phase is host code in the reference as well)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.rank, self.size = dist.get_rank(group), dist.get_world_size(group)

    def sparse(self, send):
        import torch
        dist = self.dist
        counts = torch.zeros(self.size, dtype=torch.int64)
        for dst, buf in send.items():
            counts[dst] = len(buf)
        allc = [torch.zeros(self.size, dtype=torch.int64) for _ in range(self.size)]
        dist.all_gather(allc, counts, group=self.group)
        reqs, bufs, out = [], [], {}
        for src in range(self.size):
            nrecv = int(allc[src][self.rank])
            if src == self.rank:
                if self.rank in send:
                    out[src] = np.ascontiguousarray(send[src], dtype=np.int64).copy()
            elif nrecv:                      # zero-length messages are not delivered (the callers never rely on them)
                t = torch.empty(nrecv, dtype=torch.int64)
                reqs.append(dist.irecv(t, src=src, group=self.group))
                bufs.append((src, t))
        keep = []
        for dst, buf in send.items():
            if dst != self.rank and len(buf):
                t = torch.from_numpy(np.ascontiguousarray(buf, dtype=np.int64).copy())
                keep.append(t)
                reqs.append(dist.isend(t, dst=dst, group=self.group))
        for r in reqs:
            r.wait()
        for src, t in bufs:
            out[src] = t.numpy()
        return out


# ---- helpers ----------------------------------------------------------------------------------------------------------------------
def _f2i(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.int64)


def _i2f(a):
    return np.ascontiguousarray(a, dtype=np.int64).view(np.float64)


class _Lookup:
    """global id -> local index of one rank (sorted arrays + binary search)."""

    def __init__(self, glob):
        glob = np.asarray(glob, dtype=np.int64)
        self.order = np.argsort(glob, kind="stable")
        self.sorted = glob[self.order]

    def find(self, g):
        """local index of every global id in g, -1 if unknown"""
        g = np.asarray(g, dtype=np.int64)
        if len(self.sorted) == 0:
            return np.full(len(g), -1, dtype=np.int64)
        pos = np.minimum(np.searchsorted(self.sorted, g), len(self.sorted) - 1)
        return np.where(self.sorted[pos] == g, self.order[pos], -1).astype(np.int64)


def _pack_rows(A, glob, rows, gids, with_values):
    """[g, count, column global ids ..., (values as int64 words ...)] for every listed row"""
    indptr, indices = A.indptr, A.indices
    cnt = (indptr[rows + 1] - indptr[rows]).astype(np.int64)
    tot = int(cnt.sum())
    start = np.repeat(indptr[rows].astype(np.int64) - np.concatenate([[0], np.cumsum(cnt)[:-1]]), cnt) if len(rows) else np.zeros(0, np.int64)
    pos = np.arange(tot, dtype=np.int64) + start
    cols = glob[indices[pos]] if tot else np.zeros(0, dtype=np.int64)
    parts = [np.asarray(gids, dtype=np.int64), cnt, cols]
    if with_values:
        parts.append(_f2i(A.data[pos]) if tot else np.zeros(0, dtype=np.int64))
    return np.concatenate([[len(rows)], *parts]).astype(np.int64)


def _unpack_rows(buf, with_values):
    m = int(buf[0])
    g, cnt = buf[1:1 + m], buf[1 + m:1 + 2 * m]
    tot = int(cnt.sum())
    cols = buf[1 + 2 * m:1 + 2 * m + tot]
    vals = _i2f(buf[1 + 2 * m + tot:1 + 2 * m + 2 * tot]) if with_values else None
    return g, cnt, cols, vals


class DistSetup:
    """One rank's view.  ``sub``: synth.NovlpSubdomain-like (rank, glob, owner, public, A, dirichlet)."""

    def __init__(self, ex: Exchange, sub):
        self.ex = ex
        self.rank = ex.rank
        self.n_o = len(sub.glob)
        self.glob = np.asarray(sub.glob, dtype=np.int64).copy()
        self.owner = np.asarray(sub.owner, dtype=np.uint8).copy()
        self.public = np.asarray(sub.public, dtype=bool).copy()
        self.A = sp.csr_matrix(sub.A)
        self.A.sort_indices()
        self.dirichlet_novlp = np.asarray(sub.dirichlet, dtype=np.uint8)
        self.lookup = _Lookup(self.glob)
        self.shared = {}          # neighbour rank -> local indices of the shared public indices, ascending global id
        self.round_sizes = [self.n_o]
        self._rebuild_neighbours()

    # RemoteIndices::rebuild<false>: q is a neighbour iff some global id is public on both ranks
    def _rebuild_neighbours(self):
        P = self.ex.size
        pub = np.sort(self.glob[self.public])
        home = pub % P
        got = self.ex.sparse({int(d): pub[home == d] for d in np.unique(home)})
        # directory: (gid, holder) pairs registered here -> every holder learns the other holders of its gids
        if got:
            gid = np.concatenate([got[s] for s in sorted(got)])
            holder = np.concatenate([np.full(len(got[s]), s, dtype=np.int64) for s in sorted(got)])
            order = np.lexsort((holder, gid))
            gid, holder = gid[order], holder[order]
            uniq, first, cnt = np.unique(gid, return_index=True, return_counts=True)
            grp = np.repeat(np.arange(len(uniq)), cnt)
        else:
            gid = holder = grp = cnt = first = np.zeros(0, dtype=np.int64)
        reply = {}
        if len(gid):
            # all ordered pairs (holder a, holder b != a) within a group: element e is paired with every element of its group
            c_e = cnt[grp]
            e = np.repeat(np.arange(len(gid)), c_e)
            within = np.arange(int(c_e.sum())) - np.repeat(np.cumsum(c_e) - c_e, c_e)
            partner = first[grp[e]] + within
            ok = partner != e
            e, partner = e[ok], partner[ok]
            he = holder[e]
            for s_ in np.unique(he):
                m = he == s_
                reply[int(s_)] = np.concatenate([gid[e[m]], holder[partner[m]]])
        ans = self.ex.sparse(reply)
        g_all, r_all = [], []
        for s in sorted(ans):
            half = len(ans[s]) // 2
            g_all.append(ans[s][:half])
            r_all.append(ans[s][half:])
        self.shared = {}
        if g_all:
            g_all, r_all = np.concatenate(g_all), np.concatenate(r_all)
            for q in np.unique(r_all):
                g = np.sort(g_all[r_all == q])
                self.shared[int(q)] = self.lookup.find(g)
        self.neighbours = sorted(self.shared)

    def _send_rows(self, mat, nrows, with_values):
        """every neighbour gets, for each shared index whose row this rank holds (local index < nrows: "if (i < A.N())",
        datahandles.hh:246), the row's column global ids (and values)"""
        send = {}
        for q in self.neighbours:
            l = self.shared[q]
            l = l[l < nrows]
            send[q] = _pack_rows(mat, self.glob, l, self.glob[l], with_values)
        return self.ex.sparse(send)

    def identify_boundary(self, mat):
        """IdentifyBoundaryDataHandle (datahandles.hh:122-192): a shared index is on the subdomain boundary if a neighbour's row of
        it has an off-diagonal column this rank does not know"""
        mat = sp.csr_matrix(mat)
        got = self._send_rows(mat, mat.shape[0], False)
        mask = np.zeros(len(self.glob), dtype=bool)
        for q in sorted(got):
            g, cnt, cols, _ = _unpack_rows(got[q], False)
            row = np.repeat(np.arange(len(g)), cnt)
            unknown = (self.lookup.find(cols) < 0) & (cols != g[row])
            mask[self.lookup.find(g[np.unique(row[unknown])])] = True
        return mask

    def make_overlapping_communication(self, overlap: int):
        """overlap_extension.hh:53-285 -> OvlpIndexSet of this rank"""
        if overlap <= 0:
            raise ValueError(f"make_overlapping_communication: overlap must be positive, got {overlap}")   # :72-75
        bmask = self.identify_boundary(self.A)
        dist = bfs_distance(self.A, bmask, overlap + 2)
        self.public = self.public | (dist <= overlap + 2)                                                 # :105-149, 180
        self._rebuild_neighbours()
        for _round in range(overlap):                                                                     # :205-276
            got = self._send_rows(self.A, self.n_o, False)
            new = []
            for q in sorted(got):                                                                         # ascending rank
                g, cnt, cols, _ = _unpack_rows(got[q], False)
                row = np.repeat(np.arange(len(g)), cnt)
                cand = cols[(cols != g[row]) & (self.lookup.find(cols) < 0)]
                if len(cand):
                    _, first = np.unique(cand, return_index=True)
                    cand = cand[np.sort(first)]                                                           # first occurrence, arrival order
                    if new:
                        cand = cand[~np.isin(cand, np.concatenate(new))]
                    new.append(cand)
            new = np.concatenate(new) if new else np.zeros(0, dtype=np.int64)
            self.glob = np.concatenate([self.glob, new])                                                  # copy / public (:260-262)
            self.owner = np.concatenate([self.owner, np.zeros(len(new), dtype=np.uint8)])
            self.public = np.concatenate([self.public, np.ones(len(new), dtype=bool)])
            self.lookup = _Lookup(self.glob)
            self.round_sizes.append(len(self.glob))
            self._rebuild_neighbours()
        eb = np.zeros(len(self.glob), dtype=bool)
        eb[self.round_sizes[overlap - 1]:self.round_sizes[overlap]] = True                                # :281-282
        self.overlap = overlap
        return OvlpIndexSet(self.rank, self.n_o, self.glob, self.owner, self.public.astype(np.uint8), eb, list(self.round_sizes))

    def add_to_all(self, v):
        """addOwnerCopyToOwnerCopy / addOwnerCopyToAll on a host vector: buffers packed first, added by ascending source rank"""
        got = self.ex.sparse({q: _f2i(v[self.shared[q]]) for q in self.neighbours})
        out = np.array(v, dtype=np.float64, copy=True)
        for q in sorted(got):
            np.add.at(out, self.shared[q], _i2f(got[q]))
        return out

    def overlapping_matrix(self):
        """CreateMatrixDataHandle + AddMatrixDataHandle (datahandles.hh:436-591): row i = sum over all ranks of row i of their additive
        matrices, restricted to the columns known here; then the Dirichlet mask (pdelab_helper.hh:296-302) and the symmetric
        elimination (:33-46).  Returns (A_dir, dirichlet mask on the overlapping set)."""
        n = len(self.glob)
        got = self._send_rows(self.A, self.n_o, True)
        A = self.A.tocoo()
        rows, cols, vals = [A.row.astype(np.int64)], [A.col.astype(np.int64)], [A.data.copy()]
        for q in sorted(got):
            g, cnt, cg, v = _unpack_rows(got[q], True)
            lr = np.repeat(self.lookup.find(g), cnt)
            lc = self.lookup.find(cg)
            keep = lc >= 0
            rows.append(lr[keep])
            cols.append(lc[keep])
            vals.append(v[keep])
        # duplicates are summed in the order own, then neighbours by ascending rank
        r, c, v = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
        order = np.lexsort((np.arange(len(r)), c, r))
        r, c, v = r[order], c[order], v[order]
        key_new = np.ones(len(r), dtype=bool)
        key_new[1:] = (r[1:] != r[:-1]) | (c[1:] != c[:-1])
        idx = np.cumsum(key_new) - 1
        data = np.zeros(int(idx[-1]) + 1 if len(idx) else 0)
        np.add.at(data, idx, v)
        M = sp.csr_matrix((data, (r[key_new], c[key_new])), shape=(n, n))
        M.sort_indices()
        dm = np.zeros(n)
        dm[:self.n_o] = self.dirichlet_novlp
        dm = (self.add_to_all(dm) > 0).astype(np.uint8)
        rr = np.repeat(np.arange(n), np.diff(M.indptr))
        isd_r, isd_c = dm[rr] > 0, dm[M.indices] > 0
        M.data = np.where(isd_r, np.where(M.indices == rr, 1.0, 0.0), np.where(isd_c, 0.0, M.data))
        return M, dm

    def partition_of_unity(self, A_dir, pou_type="distance", shrink=0):
        """dune/ddm/pou.hh:57-141 -> (pou, boundary mask, boundary distance)"""
        overlap = self.overlap
        if pou_type == "trivial":
            return self.owner.astype(np.float64), None, None
        if pou_type not in ("standard", "distance"):
            raise ValueError("Unknown partition of unity type: " + str(pou_type))                          # :176
        b = self.identify_boundary(A_dir)
        if pou_type == "standard":
            s = self.add_to_all(np.where(b, 0.0, 1.0))
            return np.where(b, 0.0, 1.0 / np.where(b, 1.0, s)), b, None
        if shrink < 0 or shrink >= max(overlap, 1):
            raise ValueError(f"Invalid value for shrink: {shrink} (must be >= 0 and < overlap size {overlap})")   # :184
        dist = bfs_distance(sp.csr_matrix(A_dir), b, 4 * overlap + 1)
        w = np.ones(len(self.glob))
        sel = dist <= 4 * overlap
        w[sel] = np.where(dist[sel] <= shrink, 0.0, (dist[sel] - shrink).astype(float))
        s = self.add_to_all(w)
        return np.where(b, 0.0, w / np.where(b, 1.0, s)), b, dist

    def interfaces(self):
        """{"all_to_all": {q: local idx}, "owner_send": {q: local idx this rank owns}, "owner_recv": {q: local idx q owns}} -- the
        lists of addOwnerCopyToOwnerCopy / copyOwnerToAll, each ordered by ascending global id on both sides"""
        got = self.ex.sparse({q: self.owner[self.shared[q]].astype(np.int64) for q in self.neighbours})
        return {"all_to_all": {q: self.shared[q].copy() for q in self.neighbours},
                "owner_send": {q: self.shared[q][self.owner[self.shared[q]] > 0] for q in self.neighbours},
                "owner_recv": {q: self.shared[q][got[q] > 0] for q in self.neighbours}}
