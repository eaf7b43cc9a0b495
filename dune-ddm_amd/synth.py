"""Synthetic "DUNE-assembled" inputs for the two-level Schwarz + GenEO path.

The reference obtains its inputs from dune-pdelab (grid, FE assembly, constraints) -- all of
which is out of scope here.  This module produces the same
*objects* for structured Q1 (bi-/tri-linear) diffusion problems:

  * the rank-local **non-overlapping** data the reference gets from
    ``make_communication(gfs)`` (dune/ddm/pdelab_helper.hh:15-94) and ``problem.getA()``:
    global ids (rank-contiguous numbering of owned DoFs, :51-66), owner/copy attribute
    (owner = lowest rank holding the DoF), public flag, neighbour set, the *additive* local
    matrix and a consistent right-hand side;
  * the element-wise Neumann matrices ``A_neu`` / ``B_neu`` the reference obtains by
    intercepting the element assembly (examples/pdelab_helper.hh:113-436): the sum of element
    matrices over all elements that lie completely inside the respective node region,
    with global Dirichlet DoFs eliminated symmetrically (:33-46).

All matrix entries are small integers (element matrix scaled by 12 in 3-D / 6 in 2-D, unit
mesh width, integer coefficients), so every assembly order gives bit-identical matrices.
This is input generation only; the DD algorithms proper live in ``setup_host`` (product) and
``oracle/`` (checker).
"""
from __future__ import annotations

import itertools
from dataclasses import dataclass, field

import numpy as np
import scipy.sparse as sp


def q1_element_matrix(dim: int) -> np.ndarray:
    """Integer-scaled Q1 Laplace element matrix on the unit cube (x12 in 3-D, x6 in 2-D)."""
    S = np.array([[1.0, -1.0], [-1.0, 1.0]])
    M = np.array([[2.0, 1.0], [1.0, 2.0]]) / 6.0
    nc = 1 << dim
    K = np.zeros((nc, nc))
    for a in range(nc):
        for b in range(nc):
            tot = 0.0
            for ax in range(dim):
                t = 1.0
                for d in range(dim):
                    ia, ib = (a >> d) & 1, (b >> d) & 1
                    t *= S[ia, ib] if d == ax else M[ia, ib]
                tot += t
            K[a, b] = tot
    K *= 12.0 if dim == 3 else 6.0
    Ki = np.rint(K)
    assert np.abs(K - Ki).max() < 1e-12
    return Ki


def _split(nel: int, parts: int):
    base, rem = divmod(nel, parts)
    sizes = [base + (1 if p < rem else 0) for p in range(parts)]
    lo = np.concatenate([[0], np.cumsum(sizes)])
    return [(int(lo[p]), int(lo[p + 1])) for p in range(parts)]  # element ranges [lo,hi)


def assemble_stencil(kappa_box: np.ndarray, dim: int) -> np.ndarray:
    """Sum of kappa_e * K_e as a 3^dim-point stencil array over the node box of an element box.

    kappa_box has shape e[::-1] (C order, x fastest); returns S[3^dim, *nodes_shape] with
    S[o][node] = entry (node, node + offset_o), offsets enumerated x fastest in {-1,0,1}^dim.
    """
    K = q1_element_matrix(dim)
    eshape = kappa_box.shape
    nshape = tuple(s + 1 for s in eshape)
    S = np.zeros((3 ** dim,) + nshape)
    nc = 1 << dim
    for a in range(nc):
        sl = tuple(slice((a >> (dim - 1 - ax)) & 1, ((a >> (dim - 1 - ax)) & 1) + eshape[ax]) for ax in range(dim))
        for b in range(nc):
            o = 0
            for d in range(dim):
                o += (((b >> d) & 1) - ((a >> d) & 1) + 1) * 3 ** d
            if K[a, b] != 0.0:
                S[(o,) + sl] += kappa_box * K[a, b]
    return S


def stencil_to_csr(S: np.ndarray, dim: int, inset: np.ndarray | None = None) -> sp.csr_matrix:
    """Clipped 3^dim-point stencil -> CSR over the lexicographic node box (explicit zeros kept).

    ``inset`` (bool over the node box) restricts rows and columns to a node subset (pattern
    entries touching nodes outside the subset are dropped; numbering stays box-lexicographic).
    """
    nshape = S.shape[1:]
    n = int(np.prod(nshape))
    strides = [1]
    for d in range(1, dim):
        strides.append(strides[-1] * nshape[dim - d])  # x fastest: axis index dim-1 is x
    coords = np.indices(nshape).reshape(dim, n)  # coords[ax] for ax = z,y,x order
    idx = np.arange(n, dtype=np.int64)
    cols = np.empty((n, 3 ** dim), dtype=np.int64)
    valid = np.empty((n, 3 ** dim), dtype=bool)
    for o in range(3 ** dim):
        off = [(o // 3 ** d) % 3 - 1 for d in range(dim)]  # d=0 is x
        lin = sum(off[d] * strides[d] for d in range(dim))
        ok = np.ones(n, dtype=bool)
        for d in range(dim):
            c = coords[dim - 1 - d] + off[d]
            ok &= (c >= 0) & (c < nshape[dim - 1 - d])
        cols[:, o] = idx + lin
        valid[:, o] = ok
    if inset is not None:
        flat = inset.reshape(n)
        valid &= flat[:, None]
        nb = np.where(valid, cols, 0)
        valid &= flat[nb]
    vals = S.reshape(3 ** dim, n).T
    indptr = np.concatenate([[0], np.cumsum(valid.sum(axis=1))]).astype(np.int64)
    M = sp.csr_matrix((vals[valid], cols[valid].astype(np.int32), indptr), shape=(n, n))
    if min(nshape) < 3:
        M.sort_indices()
    return M


def eliminate_dirichlet(M: sp.csr_matrix, dmask: np.ndarray, diag_value=1.0) -> sp.csr_matrix:
    """Symmetric Dirichlet elimination on the stored pattern (examples/pdelab_helper.hh:33-46)."""
    M = M.copy()
    rows = np.repeat(np.arange(M.shape[0]), np.diff(M.indptr))
    drow = dmask[rows] > 0
    dcol = dmask[M.indices] > 0
    isdiag = rows == M.indices
    dv = np.broadcast_to(np.asarray(diag_value, dtype=float), (M.shape[0],))
    M.data = np.where(drow, np.where(isdiag, dv[rows], 0.0), np.where(dcol, 0.0, M.data))
    return M


@dataclass
class NovlpSubdomain:
    """What one MPI rank holds after make_communication + assembly (non-overlapping)."""
    rank: int
    glob: np.ndarray        # int64[n_o] global id per local index
    owner: np.ndarray       # uint8[n_o] 1 = owner, 0 = copy
    public: np.ndarray      # uint8[n_o] shared with another rank
    A: sp.csr_matrix        # additive local matrix (n_o x n_o)
    b: np.ndarray           # consistent right-hand side
    dirichlet: np.ndarray   # uint8[n_o]
    neighbours: list = field(default_factory=list)
    node_lo: tuple = ()     # node box (inclusive lo, inclusive hi), x,y,z order
    node_hi: tuple = ()


class StructuredPoisson:
    """Q1 diffusion -div(kappa grad u) = 1, u = 0 on the boundary, on N[0] x N[1] (x N[2]) nodes,
    elements partitioned into P[0] x P[1] (x P[2]) boxes (one per rank)."""

    def __init__(self, N, P, kappa: np.ndarray | None = None):
        self.dim = len(N)
        assert self.dim in (2, 3) and len(P) == self.dim
        self.N = tuple(int(x) for x in N)
        self.P = tuple(int(x) for x in P)
        self.nel = tuple(n - 1 for n in self.N)
        eshape = self.nel[::-1]
        self.kappa = np.ones(eshape) if kappa is None else np.asarray(kappa, dtype=float)
        assert self.kappa.shape == eshape
        self.nranks = int(np.prod(self.P))
        self.splits = [_split(self.nel[d], self.P[d]) for d in range(self.dim)]
        self._number_globally()

    # -- rank <-> part coordinates (x fastest)
    def part_of_rank(self, r):
        out = []
        for d in range(self.dim):
            out.append(r % self.P[d])
            r //= self.P[d]
        return tuple(out)

    def _node_box(self, r):
        pc = self.part_of_rank(r)
        lo = tuple(self.splits[d][pc[d]][0] for d in range(self.dim))
        hi = tuple(self.splits[d][pc[d]][1] for d in range(self.dim))  # inclusive node hi = element hi
        return lo, hi

    def _box_slices(self, lo, hi):
        return tuple(slice(lo[d], hi[d] + 1) for d in reversed(range(self.dim)))

    def _holders_axis(self, d):
        """per node coordinate on axis d: (first part holding it, number of parts holding it)"""
        first = np.zeros(self.N[d], dtype=np.int64)
        cnt = np.zeros(self.N[d], dtype=np.int64)
        for p, (lo, hi) in reversed(list(enumerate(self.splits[d]))):
            first[lo:hi + 1] = p
            cnt[lo:hi + 1] += 1
        return first, cnt

    def _number_globally(self):
        dim = self.dim
        nshape = self.N[::-1]
        first, cnt = zip(*[self._holders_axis(d) for d in range(dim)])
        grids = np.meshgrid(*[np.arange(self.N[d]) for d in reversed(range(dim))], indexing="ij")
        owner = np.zeros(nshape, dtype=np.int64)
        holders = np.ones(nshape, dtype=np.int64)
        mult = 1
        for d in range(dim):
            c = grids[dim - 1 - d]
            owner += first[d][c] * mult
            holders *= cnt[d][c]
            mult *= self.P[d]
        self.owner_rank = owner
        self.holders = holders
        gid = np.full(nshape, -1, dtype=np.int64)
        start = 0
        for r in range(self.nranks):
            lo, hi = self._node_box(r)
            sl = self._box_slices(lo, hi)
            own = self.owner_rank[sl] == r
            k = int(own.sum())
            sub = gid[sl]
            sub[own] = start + np.arange(k)
            start += k
        assert (gid >= 0).all() and start == gid.size
        self.gid_of_node = gid
        node_of_gid = np.empty(gid.size, dtype=np.int64)
        node_of_gid[gid.reshape(-1)] = np.arange(gid.size)
        self.node_of_gid = node_of_gid
        bnd = np.zeros(nshape, dtype=bool)
        for ax in range(dim):
            idx = [slice(None)] * dim
            idx[ax] = 0
            bnd[tuple(idx)] = True
            idx[ax] = -1
            bnd[tuple(idx)] = True
        self.dirichlet_node = bnd
        # consistent load vector: number of adjacent elements (f = 1, scaled), 0 on Dirichlet nodes
        load = np.zeros(nshape)
        ones = np.ones(self.nel[::-1])
        for a in range(1 << dim):
            sl = tuple(slice((a >> (dim - 1 - ax)) & 1, ((a >> (dim - 1 - ax)) & 1) + self.nel[::-1][ax]) for ax in range(dim))
            load[sl] += ones
        load[bnd] = 0.0
        self.load = load

    @property
    def nglobal(self):
        return int(np.prod(self.N))

    def subdomain(self, r) -> NovlpSubdomain:
        dim = self.dim
        lo, hi = self._node_box(r)
        nsl = self._box_slices(lo, hi)
        esl = tuple(slice(lo[d], hi[d]) for d in reversed(range(dim)))
        S = assemble_stencil(self.kappa[esl], dim)
        A = stencil_to_csr(S, dim)
        dmask = self.dirichlet_node[nsl].reshape(-1).astype(np.uint8)
        holders = self.holders[nsl].reshape(-1)
        A = eliminate_dirichlet(A, dmask, 1.0 / holders)
        glob = self.gid_of_node[nsl].reshape(-1).copy()
        owner = (self.owner_rank[nsl].reshape(-1) == r).astype(np.uint8)
        public = (holders > 1).astype(np.uint8)
        pc = self.part_of_rank(r)
        nb = set()
        for off in itertools.product((-1, 0, 1), repeat=dim):
            q = tuple(pc[d] + off[d] for d in range(dim))
            if any(off) and all(0 <= q[d] < self.P[d] for d in range(dim)):
                nb.add(sum(q[d] * int(np.prod(self.P[:d])) for d in range(dim)))
        return NovlpSubdomain(rank=r, glob=glob, owner=owner, public=public, A=A,
                              b=self.load[nsl].reshape(-1).copy(), dirichlet=dmask,
                              neighbours=sorted(nb), node_lo=lo, node_hi=hi)

    def subdomains(self):
        return [self.subdomain(r) for r in range(self.nranks)]

    # ---- element-wise Neumann matrices on an overlapping node set -------------------------
    def node_coords(self, glob: np.ndarray) -> np.ndarray:
        """(len, dim) integer node coordinates (x,y,z order) of global ids."""
        node = self.node_of_gid[glob]
        out = np.empty((len(glob), self.dim), dtype=np.int64)
        for d in range(self.dim):
            out[:, d] = node % self.N[d]
            node = node // self.N[d]
        return out

    def region_matrix(self, glob: np.ndarray, region: np.ndarray | None, dirichlet_ovlp: np.ndarray,
                      neumann: bool = True) -> sp.csr_matrix:
        """Sum of element matrices over the elements all of whose nodes are in ``glob`` (and in
        ``region`` if given) -- Neumann condition on the region boundary -- in the local
        numbering of ``glob``; with neumann=False all global elements touching the nodes
        contribute (principal submatrix of the global matrix = the Dirichlet matrix).
        Pattern: pairs of region nodes sharing an element of the *global* mesh
        (examples/pdelab_helper.hh:304, 404-417).  Dirichlet DoFs eliminated symmetrically.
        """
        dim = self.dim
        n = len(glob)
        xyz = self.node_coords(glob)
        lo = xyz.min(axis=0)
        hi = xyz.max(axis=0)
        bshape = tuple(int(hi[d] - lo[d] + 1) for d in reversed(range(dim)))
        box_index = np.zeros(n, dtype=np.int64)
        mult = 1
        for d in range(dim):
            box_index += (xyz[:, d] - lo[d]) * mult
            mult *= int(hi[d] - lo[d] + 1)
        nbox = int(np.prod(bshape))
        inset = np.zeros(nbox, dtype=bool)
        sel = np.ones(n, dtype=bool) if region is None else region.astype(bool)
        inset[box_index[sel]] = True
        inset = inset.reshape(bshape)
        if neumann:
            ke = self.kappa[tuple(slice(lo[d], hi[d]) for d in reversed(range(dim)))].copy()
            allin = np.ones(ke.shape, dtype=bool)
            for a in range(1 << dim):
                sl = tuple(slice((a >> (dim - 1 - ax)) & 1, ((a >> (dim - 1 - ax)) & 1) + ke.shape[ax]) for ax in range(dim))
                allin &= inset[sl]
            ke[~allin] = 0.0
            S = assemble_stencil(ke, dim)
        else:
            # one extra element layer around the node box, clipped to the mesh
            elo = [max(lo[d] - 1, 0) for d in range(dim)]
            ehi = [min(hi[d] + 1, self.N[d] - 1) for d in range(dim)]
            ke = self.kappa[tuple(slice(elo[d], ehi[d]) for d in reversed(range(dim)))]
            Sbig = assemble_stencil(ke, dim)
            crop = tuple(slice(lo[d] - elo[d], lo[d] - elo[d] + (hi[d] - lo[d] + 1)) for d in reversed(range(dim)))
            S = np.ascontiguousarray(Sbig[(slice(None),) + crop])
        Mbox = stencil_to_csr(S, dim, inset=inset).tocoo()
        loc_of_box = np.full(nbox, -1, dtype=np.int64)
        loc_of_box[box_index] = np.arange(n)
        M = sp.csr_matrix((Mbox.data, (loc_of_box[Mbox.row], loc_of_box[Mbox.col])), shape=(n, n))
        M.sort_indices()
        return eliminate_dirichlet(M, dirichlet_ovlp)

    def dirichlet_of(self, glob: np.ndarray) -> np.ndarray:
        return self.dirichlet_node.reshape(-1)[self.node_of_gid[glob]].astype(np.uint8)


def islands_kappa(nel, contrast=1e6, period=8, width=2):
    """High-contrast coefficient: channels/islands of value ``contrast`` in a background of 1
    (cf. the IslandsModelProblem hard-coded in examples/poisson.hh:234).  Integer-valued."""
    dim = len(nel)
    grids = np.meshgrid(*[np.arange(nel[d]) for d in reversed(range(dim))], indexing="ij")
    hit = np.ones(nel[::-1], dtype=bool)
    for ax in range(1, dim):
        hit &= (grids[ax - 1] % period) < width
    k = np.ones(nel[::-1])
    k[hit] = float(contrast)
    return k
