"""Decomposition containers and the flattening of a set of subdomains into one rank-local problem.

``Decomposition``     what the DUNE side holds before the Krylov loop starts: per subdomain the
                      additive non-overlapping matrix, the overlapping Dirichlet / Neumann matrices,
                      the partition of unity, masks, plus the index lists of the three interfaces.
``RankLocal``         the concatenation of the subdomains assigned to one GPU (one per rank in the
                      reference, examples/poisson.cc:128-131; 8/N per GPU here so that the same
                      8-subdomain problem runs on N = 1, 2, 4, 8 GPUs) together with the halo plans
                      in the layout ``ddm_halo_create`` expects.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
import scipy.sparse as sp

from . import setup_host as sh


@dataclass
class SubdomainData:
    id: int
    n_o: int
    n: int
    glob: np.ndarray
    A: sp.csr_matrix            # additive, non-overlapping (n_o x n_o)
    owner_novlp: np.ndarray     # uint8[n_o]
    b: np.ndarray               # consistent rhs, f64[n_o]
    A_dir: sp.csr_matrix        # overlapping Dirichlet matrix (n x n)
    owner_ovlp: np.ndarray      # uint8[n]
    dirichlet_ovlp: np.ndarray  # uint8[n]
    pou: np.ndarray | None      # f64[n]
    A_neu: sp.csr_matrix | None = None
    B_neu: sp.csr_matrix | None = None
    boundary: np.ndarray | None = None     # bool[n]: subdomain boundary mask (IdentifyBoundaryDataHandle)
    boundary_dist: np.ndarray | None = None  # graph distance to the subdomain boundary (exact up to 4 * overlap)


@dataclass
class Decomposition:
    subs: list
    novlp_all: dict             # (src,dst) -> (idx_src, idx_dst): addOwnerCopyToOwnerCopy on novlp_comm (C1)
    ovlp_owner: dict            # copyOwnerToAll on ovlp_comm (C2)
    ovlp_all: dict              # addOwnerCopyToOwnerCopy / addOwnerCopyToAll on ovlp_comm (C3, C4)
    overlap: int
    nglobal: int
    meta: dict = field(default_factory=dict)

    @property
    def nsub(self):
        return len(self.subs)


def restrict_decomposition(dec: "Decomposition", keep) -> "Decomposition":
    """The decomposition reduced to the subdomains `keep` (renumbered 0 .. len(keep) - 1): their matrices, masks and partitions of
    unity as they are, the exchange pairs among them; pairs to dropped subdomains are removed.  Every per-subdomain object (local
    solves, GenEO pencils, restriction / prolongation) is exactly what the rank owning `keep` works on in a run over several GPUs;
    only the exchanges with the other ranks are cut.  Used to time the rank-local workload of N GPUs on one (bench.py
    --emulate-rank-of N): NOT a solvable restriction of the global problem."""
    from dataclasses import replace
    keep = list(keep)
    new_id = {s: i for i, s in enumerate(keep)}
    subs = [replace(dec.subs[s], id=new_id[s]) for s in keep]

    def pairs(d):
        return {(new_id[a], new_id[b]): v for (a, b), v in d.items() if a in new_id and b in new_id}

    return Decomposition(subs, pairs(dec.novlp_all), pairs(dec.ovlp_owner), pairs(dec.ovlp_all), dec.overlap, dec.nglobal,
                         {"restricted_from": dec.nsub, "kept": keep})


def _pmap(fn, items):
    """[fn(x) for x in items] on host threads: the per-subdomain matrix generation is large numpy array arithmetic, which releases
    the GIL (216^3: 34 s of the run time of bench.py were this loop, sequential).  DDM_HOST_THREADS = 1 switches it off."""
    import os
    items = list(items)
    nt = min(len(items), int(os.environ.get("DDM_HOST_THREADS", min(8, os.cpu_count() or 1))))
    if nt <= 1:
        return [fn(x) for x in items]
    from concurrent.futures import ThreadPoolExecutor
    from . import synth
    synth.INNER_THREADS = max(1, (os.cpu_count() or 1) // nt)     # the native generator's own threads: share the cores with the pool
    try:
        with ThreadPoolExecutor(nt) as ex:
            return list(ex.map(fn, items))
    finally:
        synth.INNER_THREADS = None


def build_structured(grid, overlap=2, pou_type="distance", shrink=0, neumann=False, second_region="overlap") -> Decomposition:
    """Runs the L3 setup of the reference (make_communication ->
    make_overlapping_communication -> assemble_overlapping_matrices -> PartitionOfUnity) on one of the
    synth problems (StructuredPoisson, StructuredDG2D, StructuredElasticity).  neumann=True also assembles
    A_neu (NeumannRegion::All) and B_neu: on NeumannRegion::Overlap as examples/poisson.cc:206 and
    examples/pdelab_schwarz.hh:62 request for GenEO, or, with second_region="all", the same matrix as A_neu
    (examples/linearelasticity.hh:222)."""
    nov = _pmap(grid.subdomain, range(grid.nranks)) if hasattr(grid, "subdomain") else grid.subdomains()
    ng = grid.nglobal
    if grid.nranks == 1:
        s = nov[0]
        n = len(s.glob)
        sd = SubdomainData(0, n, n, s.glob, s.A, s.owner, s.b, s.A, s.owner.copy(), s.dirichlet,
                           np.ones(n) if pou_type else None)
        return Decomposition([sd], {}, {}, {}, overlap, ng, {"pou_type": pou_type, "shrink": shrink})
    idx = sh.make_overlapping_communication(nov, overlap, ng)
    novlp_all = sh.interface_pairs(nov, ng, "all_to_all")
    ovlp_all = sh.interface_pairs(idx, ng, "all_to_all")
    ovlp_owner = sh.interface_pairs(idx, ng, "owner_to_all")
    dmask = [grid.dirichlet_of(i.glob) for i in idx]
    A_dir = _pmap(lambda t: grid.dirichlet_matrix(t[0].glob, t[1]), list(zip(idx, dmask)))
    pou, bmask, dist = sh.partition_of_unity(idx, A_dir, ovlp_all, ng, pou_type, shrink, overlap)
    if bmask is None and neumann:
        bmask = sh.subdomain_boundary(idx, A_dir, ng)
    def one(r):
        s, i = nov[r], idx[r]
        sd = SubdomainData(r, i.n_o, len(i.glob), i.glob, s.A, s.owner, s.b, A_dir[r], i.owner, dmask[r], pou[r])
        if bmask is not None:
            sd.boundary = np.asarray(bmask[r], dtype=bool)
        if neumann:
            d = dist[r] if dist is not None else sh.bfs_distance(A_dir[r], bmask[r], 4 * overlap + 1)
            sd.boundary_dist = d
            sd.A_neu = grid.neumann_matrix(i.glob, None, dmask[r])
            sd.B_neu = sd.A_neu if second_region == "all" else grid.neumann_matrix(i.glob, d <= 2 * overlap, dmask[r])
        return sd

    subs = _pmap(one, range(len(nov)))
    return Decomposition(subs, novlp_all, ovlp_owner, ovlp_all, overlap, ng,
                         {"pou_type": pou_type, "shrink": shrink, "ext_boundary": [i.ext_boundary for i in idx],
                          "boundary": bmask})


def _block_diag(mats):
    """Block-diagonal CSR matrix (int64 row pointers, int32 columns, float64 values) of the given square matrices.  The three arrays
    are allocated once and every block is written into its slice by a host thread (numpy's slice copies and in-place adds release
    the GIL): at 216^3 the list-of-temporaries + concatenate version of this was 1.4 s per call, three calls per run."""
    mats = [sp.csr_matrix(M) for M in mats]
    nrows = np.array([0] + [M.shape[0] for M in mats], dtype=np.int64).cumsum()
    nnz = np.array([0] + [M.nnz for M in mats], dtype=np.int64).cumsum()
    n, z = int(nrows[-1]), int(nnz[-1])
    rp = np.empty(n + 1, dtype=np.int64)
    ci = np.empty(z, dtype=np.int32)
    va = np.empty(z, dtype=np.float64)
    rp[0] = 0

    def put(k):
        M = mats[k]
        r0, r1, z0, z1 = int(nrows[k]), int(nrows[k + 1]), int(nnz[k]), int(nnz[k + 1])
        rp[r0 + 1:r1 + 1] = M.indptr[1:]
        rp[r0 + 1:r1 + 1] += z0
        ci[z0:z1] = M.indices
        ci[z0:z1] += np.int32(r0)
        va[z0:z1] = M.data

    _pmap(put, range(len(mats)))
    return sp.csr_matrix((va, ci, rp), shape=(n, n))


def halo_plan(pairs: dict, local_subs, sub2rank, offsets, rank, nranks):
    """Flattens one interface into the arrays of ddm_halo_create for ``rank``.

    Send segment to rank r: the pairs (src local, dst on r) sorted by (src, dst), each in its
    stored (ascending global id) order.  Receive side mirrors it.  Contributions to one
    destination entry are listed by ascending source subdomain -- the order in which DUNE's
    BufferedCommunicator scatters the neighbours' messages."""
    local = set(local_subs)
    send_idx, send_counts = [], np.zeros(nranks, dtype=np.int64)
    for r in range(nranks):
        for (s, d) in sorted(k for k in pairs if k[0] in local and sub2rank[k[1]] == r):
            send_idx.append(offsets[s] + pairs[(s, d)][0])
            send_counts[r] += len(pairs[(s, d)][0])
    recv_counts = np.zeros(nranks, dtype=np.int64)
    dst_l, src_l, pos_l = [], [], []
    pos = 0
    for r in range(nranks):
        for (s, d) in sorted(k for k in pairs if k[1] in local and sub2rank[k[0]] == r):
            m = len(pairs[(s, d)][1])
            dst_l.append(offsets[d] + pairs[(s, d)][1])
            src_l.append(np.full(m, s, dtype=np.int64))
            pos_l.append(pos + np.arange(m, dtype=np.int64))
            pos += m
            recv_counts[r] += m
    if dst_l:
        dst = np.concatenate(dst_l)
        src = np.concatenate(src_l)
        posa = np.concatenate(pos_l)
        order = np.lexsort((src, dst))
        dst, posa = dst[order], posa[order]
        dst_idx, start = np.unique(dst, return_index=True)
        dst_ptr = np.concatenate([start, [len(dst)]]).astype(np.int64)
    else:
        dst_idx, dst_ptr, posa = np.zeros(0, np.int64), np.zeros(1, np.int64), np.zeros(0, np.int64)
    return {"send_idx": np.concatenate(send_idx).astype(np.int64) if send_idx else np.zeros(0, np.int64),
            "send_counts": send_counts, "recv_counts": recv_counts, "dst_idx": dst_idx.astype(np.int64), "dst_ptr": dst_ptr,
            "src_pos": posa.astype(np.int64)}


class RankLocal:
    """Concatenated data of the subdomains owned by one rank + exchange plans."""

    def __init__(self, dec: Decomposition, rank=0, nranks=1, sub2rank=None):
        P = dec.nsub
        if sub2rank is None:
            assert P % nranks == 0, "number of subdomains must be a multiple of the number of ranks"
            per = P // nranks
            sub2rank = [s // per for s in range(P)]
        self.sub2rank = list(sub2rank)
        self.rank, self.nranks = rank, nranks
        self.local = [s for s in range(P) if self.sub2rank[s] == rank]
        subs = [dec.subs[s] for s in self.local]
        self.subs = subs
        self.off_o, self.off = {}, {}
        o1 = o2 = 0
        for sd in subs:
            self.off_o[sd.id], self.off[sd.id] = o1, o2
            o1 += sd.n_o
            o2 += sd.n
        self.n_o, self.n = o1, o2
        self.block_ptr_o = np.array([self.off_o[sd.id] for sd in subs] + [o1], dtype=np.int64)
        self.block_ptr = np.array([self.off[sd.id] for sd in subs] + [o2], dtype=np.int64)
        self.A = _block_diag([sd.A for sd in subs])
        self.A_dir = _block_diag([sd.A_dir for sd in subs])
        self.owner_novlp = np.concatenate([sd.owner_novlp for sd in subs]).astype(np.uint8)
        self.b = np.concatenate([sd.b for sd in subs]).astype(np.float64)
        self.pou = None if subs[0].pou is None else np.concatenate([sd.pou for sd in subs]).astype(np.float64)
        self.dirichlet_ovlp = np.concatenate([sd.dirichlet_ovlp for sd in subs]).astype(np.uint8)
        em = np.full(self.n, -1, dtype=np.int32)
        for sd in subs:
            em[self.off[sd.id]:self.off[sd.id] + sd.n_o] = self.off_o[sd.id] + np.arange(sd.n_o, dtype=np.int32)
        self.ext_map = em
        # offsets of *all* subdomains are needed only for the local ones; remote entries never index
        offs_o = {s: self.off_o.get(s, 0) for s in range(P)}
        offs = {s: self.off.get(s, 0) for s in range(P)}
        self.plan_novlp_add = halo_plan(dec.novlp_all, self.local, self.sub2rank, offs_o, rank, nranks)
        self.plan_ovlp_copy = halo_plan(dec.ovlp_owner, self.local, self.sub2rank, offs, rank, nranks)
        self.plan_ovlp_add = halo_plan(dec.ovlp_all, self.local, self.sub2rank, offs, rank, nranks)

    # ---- helpers to move between per-subdomain lists and rank-local vectors
    def cat_novlp(self, vecs_by_sub):
        return np.concatenate([np.asarray(vecs_by_sub[s], dtype=np.float64) for s in self.local])

    def cat_ovlp(self, vecs_by_sub):
        return np.concatenate([np.asarray(vecs_by_sub[s], dtype=np.float64) for s in self.local])

    def split_novlp(self, v):
        return {sd.id: v[self.off_o[sd.id]:self.off_o[sd.id] + sd.n_o] for sd in self.subs}

    def split_ovlp(self, v):
        return {sd.id: v[self.off[sd.id]:self.off[sd.id] + sd.n] for sd in self.subs}

    def neumann_matrices(self):
        return _block_diag([sd.A_neu for sd in self.subs]), _block_diag([sd.B_neu for sd in self.subs])


def build_distributed(ex, sub, nranks, overlap=2, pou_type="distance", shrink=0, nglobal=None) -> Decomposition:
    """The same L3 setup run by ONE rank on its own non-overlapping data ``sub`` (synth.NovlpSubdomain) with neighbour exchanges only
    (setup_dist.DistSetup over ``ex``: torch.distributed / threads) -- no global knowledge of the other subdomains.  Returns a
    Decomposition whose ``subs`` list holds this rank's entry only (None elsewhere) and whose interface dictionaries hold the local
    halves of the index lists, which is all ``RankLocal(dec, rank, nranks)`` reads when every rank owns one subdomain.
    (The Neumann matrices of the GenEO-type coarse spaces come from the application's assembler, as in the reference:
    examples/pdelab_helper.hh:113-436.)"""
    from . import setup_dist as sdist
    ds = sdist.DistSetup(ex, sub)
    r = ex.rank
    ifc0 = ds.interfaces()
    novlp_all = {}
    for q, l in ifc0["all_to_all"].items():
        novlp_all[(r, q)] = (l, None)
        novlp_all[(q, r)] = (None, l)
    idx = ds.make_overlapping_communication(overlap)
    A_dir, dm = ds.overlapping_matrix()
    pou, bmask, dist = ds.partition_of_unity(A_dir, pou_type, shrink)
    ifc = ds.interfaces()
    ovlp_all, ovlp_owner = {}, {}
    for q, l in ifc["all_to_all"].items():
        ovlp_all[(r, q)] = (l, None)
        ovlp_all[(q, r)] = (None, l)
        if len(ifc["owner_send"][q]):
            ovlp_owner[(r, q)] = (ifc["owner_send"][q], None)
        if len(ifc["owner_recv"][q]):
            ovlp_owner[(q, r)] = (None, ifc["owner_recv"][q])
    sd = SubdomainData(r, idx.n_o, len(idx.glob), idx.glob, sub.A, sub.owner, sub.b, A_dir, idx.owner, dm, pou)
    if bmask is not None:
        sd.boundary = np.asarray(bmask, dtype=bool)
    sd.boundary_dist = dist
    subs = [None] * nranks
    subs[r] = sd
    # the Galerkin assembly exchanges basis vectors one neighbour slot at a time (galerkin_preconditioner.hh:298-309): the sender
    # must know its position in the RECEIVER's sorted neighbour list
    got = ex.sparse({q: np.asarray(ds.neighbours, dtype=np.int64) for q in ds.neighbours})
    slots = {q: {r: int(np.nonzero(got[q] == r)[0][0])} for q in ds.neighbours}
    slots[r] = {a: i for i, a in enumerate(ds.neighbours)}
    return Decomposition(subs, novlp_all, ovlp_owner, ovlp_all, overlap, nglobal if nglobal is not None else -1,
                         {"pou_type": pou_type, "shrink": shrink, "ext_boundary": {r: idx.ext_boundary}, "boundary": {r: bmask}, "distributed": True,
                          "_nbr_slots": slots})
