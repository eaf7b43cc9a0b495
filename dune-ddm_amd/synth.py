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


INNER_THREADS = None     # set by problem._pmap while it runs one generator call per host thread: threads each call may start itself


def _native():
    """libddm_hip.so's host-side generator (csrc/synth_host.hpp) unless DDM_SYNTH_NATIVE=0 asks for the numpy passes below (kept as
    its checker: tests/test_synth_native.py compares the two bit for bit)."""
    import os
    if os.environ.get("DDM_SYNTH_NATIVE", "1") == "0":
        return None
    from . import load_library
    return load_library()


def q1_matrix_native(lib, dim, bshape, ke, eoff, inset, loc_of_box, n, box_index, dmask, diag) -> sp.csr_matrix:
    """ddm_synth_q1_matrix (include/ddm_hip.h): Q1 matrix of the node box ``bshape`` (x first) from the element coefficients ``ke``
    (array in C order, z/y/x axes; element (0,..,0) of it is the element ``eoff`` places before box node 0), rows = the ``n`` local
    indices, pattern restricted to ``inset``, Dirichlet-eliminated."""
    import ctypes
    import os
    i64, u8, f64 = np.int64, np.uint8, np.float64
    c = np.ascontiguousarray
    bs = c(bshape, dtype=i64)
    ke = c(ke, dtype=f64)
    es = c(ke.shape[::-1], dtype=i64)
    eo = c(eoff, dtype=i64)
    K = c(q1_element_matrix(dim), dtype=f64)
    inset = None if inset is None else c(inset.reshape(-1), dtype=u8)
    loc = None if loc_of_box is None else c(loc_of_box, dtype=i64)
    bi = None if box_index is None else c(box_index, dtype=i64)
    dm = c(dmask, dtype=u8)
    dg = None if diag is None else c(np.broadcast_to(np.asarray(diag, dtype=f64), (n,)))
    P = ctypes.c_void_p

    def ptr(a):
        return None if a is None else a.ctypes.data_as(P)

    nt = INNER_THREADS or int(os.environ.get("DDM_HOST_THREADS", min(8, os.cpu_count() or 1)))
    indptr = np.empty(n + 1, dtype=i64)
    args = [dim, ptr(bs), ptr(ke), ptr(es), ptr(eo), ptr(K), ptr(inset), ptr(loc), n, ptr(bi), ptr(dm), ptr(dg)]
    if lib.ddm_synth_q1_matrix(*args, ptr(indptr), None, None, nt) != 0:
        raise RuntimeError("ddm_synth_q1_matrix: " + lib.ddm_last_error(None).decode())
    indices = np.empty(int(indptr[-1]), dtype=np.int32)
    data = np.empty(int(indptr[-1]), dtype=f64)
    if lib.ddm_synth_q1_matrix(*args, ptr(indptr), ptr(indices), ptr(data), nt) != 0:
        raise RuntimeError("ddm_synth_q1_matrix: " + lib.ddm_last_error(None).decode())
    M = sp.csr_matrix((data, indices, indptr), shape=(n, n))
    M.has_sorted_indices = True
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
        dmask = self.dirichlet_node[nsl].reshape(-1).astype(np.uint8)
        holders = self.holders[nsl].reshape(-1)
        lib = _native()
        if lib is not None:
            bshape = [hi[d] - lo[d] + 1 for d in range(dim)]
            A = q1_matrix_native(lib, dim, bshape, self.kappa[esl], [0] * dim, None, None, len(dmask), None, dmask, 1.0 / holders)
        else:
            S = assemble_stencil(self.kappa[esl], dim)
            A = stencil_to_csr(S, dim)
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

    def elements(self, r):
        """The rank's own elements as the grid operator visits them (x fastest): (dofs, Ke) with dofs[e, a] the local index (numbering
        of subdomain(r)) of corner a (bit d of a = offset in direction d) and Ke[e] = kappa_e K, the raw element matrix -- what
        `mat.container() - M_before` is in examples/assemblewrapper.hh:229.  Used by neumann_assembly.py."""
        dim = self.dim
        lo, hi = self._node_box(r)
        nn = [hi[d] - lo[d] + 1 for d in range(dim)]                     # nodes per direction, x first
        ne = [n - 1 for n in nn]
        esl = tuple(slice(lo[d], hi[d]) for d in reversed(range(dim)))
        kap = self.kappa[esl].reshape(-1)                                # C order: x fastest
        eidx = np.arange(int(np.prod(ne)), dtype=np.int64)
        base = np.zeros_like(eidx)
        stride, rem = 1, eidx
        for d in range(dim):
            base += (rem % ne[d]) * stride
            rem = rem // ne[d]
            stride *= nn[d]
        nc = 1 << dim
        off = np.zeros(nc, dtype=np.int64)
        for a in range(nc):
            stride = 1
            for d in range(dim):
                off[a] += ((a >> d) & 1) * stride
                stride *= nn[d]
        dofs = base[:, None] + off[None, :]
        Ke = kap[:, None, None] * q1_element_matrix(dim)[None, :, :]
        return dofs, Ke

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
        lib = _native()
        if neumann:
            ke = self.kappa[tuple(slice(lo[d], hi[d]) for d in reversed(range(dim)))].copy()
            allin = np.ones(ke.shape, dtype=bool)
            for a in range(1 << dim):
                sl = tuple(slice((a >> (dim - 1 - ax)) & 1, ((a >> (dim - 1 - ax)) & 1) + ke.shape[ax]) for ax in range(dim))
                allin &= inset[sl]
            ke[~allin] = 0.0
            if lib is not None:
                loc_of_box = np.full(nbox, -1, dtype=np.int64)
                loc_of_box[box_index] = np.arange(n)
                return q1_matrix_native(lib, dim, bshape[::-1], ke, [0] * dim, inset, loc_of_box, n, box_index, dirichlet_ovlp, None)
            S = assemble_stencil(ke, dim)
        else:
            # one extra element layer around the node box, clipped to the mesh
            elo = [max(lo[d] - 1, 0) for d in range(dim)]
            ehi = [min(hi[d] + 1, self.N[d] - 1) for d in range(dim)]
            ke = self.kappa[tuple(slice(elo[d], ehi[d]) for d in reversed(range(dim)))]
            if lib is not None:
                loc_of_box = np.full(nbox, -1, dtype=np.int64)
                loc_of_box[box_index] = np.arange(n)
                return q1_matrix_native(lib, dim, bshape[::-1], ke, [int(lo[d] - elo[d]) for d in range(dim)], inset, loc_of_box, n, box_index,
                                        dirichlet_ovlp, None)
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

    # the two calls problem.build_structured makes (shared with ElementProblem / StructuredDG2D)
    def dirichlet_matrix(self, glob, dirichlet_ovlp):
        return self.region_matrix(glob, None, dirichlet_ovlp, neumann=False)

    def neumann_matrix(self, glob, region, dirichlet_ovlp):
        return self.region_matrix(glob, region, dirichlet_ovlp, neumann=True)


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


# =================================================================================================
# BASELINE.json configs[4]: P1 vector elasticity on a simplex box  (examples/linearelasticity.cc)
# =================================================================================================
def _coo_to_csr(rows, cols, vals, shape):
    """COO -> CSR with duplicates summed, sorted columns, explicit zeros kept (pattern entries)."""
    M = sp.coo_matrix((vals, (rows, cols)), shape=shape).tocsr()
    M.sum_duplicates()
    M.sort_indices()
    return M


class ElementProblem:
    """Conforming vector-valued FE problem given by explicit elements, partitioned element-wise
    (= a non-overlapping DUNE grid partition; OverlappingEntitySet without ghosts, CG assembly is
    additive: examples/problem_traits.hh:150).  Produces the same objects as StructuredPoisson:

      * ``subdomains()``: what make_communication (dune/ddm/pdelab_helper.hh:15-94) + assembly give
        -- owner = lowest rank holding the DoF, rank-contiguous global ids of the owned DoFs in
        local index order (:51-66), additive matrix, consistent residual;
      * ``dirichlet_matrix`` / ``neumann_matrix``: the overlapping matrices of
        assemble_overlapping_matrices (examples/pdelab_helper.hh:113-436): A_dir = principal
        submatrix of the global matrix; the Neumann matrix of a DoF region = sum of the element
        matrices of the elements all of whose DoFs lie in the region, on the pattern of A_dir
        restricted to the region; Dirichlet DoFs eliminated symmetrically (:33-46).

    Local DoF order on a rank: component-major ("lexicographic" ordering of the vector-valued
    grid function space, examples/linearelasticity.hh:152-155), nodes by ascending global node id.
    elem_mats[e] is (nv*ncomp) x (nv*ncomp) in element DoF order (vertex a, component c) -> a*ncomp + c.
    """

    def __init__(self, nnodes, ncomp, elem_nodes, elem_mats, elem_rank, nranks, dirichlet, load):
        self.nnodes, self.ncomp = int(nnodes), int(ncomp)
        self.elem_nodes = np.asarray(elem_nodes, dtype=np.int64)
        self.elem_mats = np.asarray(elem_mats, dtype=np.float64)
        self.elem_rank = np.asarray(elem_rank, dtype=np.int64)
        self.nranks = int(nranks)
        self.dirichlet = np.asarray(dirichlet, dtype=bool).reshape(self.nnodes, self.ncomp)
        self.load = np.where(self.dirichlet, 0.0, np.asarray(load, dtype=np.float64).reshape(self.nnodes, self.ncomp))
        ne, nv = self.elem_nodes.shape
        assert self.elem_mats.shape == (ne, nv * ncomp, nv * ncomp)
        # holders / owner per node
        held = np.zeros((self.nranks, self.nnodes), dtype=bool)
        for r in range(self.nranks):
            held[r, np.unique(self.elem_nodes[self.elem_rank == r])] = True
        self._held = held
        self.node_holders = held.sum(axis=0)
        assert (self.node_holders > 0).all(), "every node must belong to an element"
        self.node_owner = np.argmax(held, axis=0)                      # lowest rank holding the node
        # rank-contiguous global ids of owned DoFs, local (component-major) order
        gid = np.full((self.nnodes, self.ncomp), -1, dtype=np.int64)
        start = 0
        self._local_nodes = []
        for r in range(self.nranks):
            ln = np.nonzero(held[r])[0]
            self._local_nodes.append(ln)
            own = ln[self.node_owner[ln] == r]
            for c in range(self.ncomp):
                gid[own, c] = start + np.arange(len(own))
                start += len(own)
        assert start == self.nnodes * self.ncomp and (gid >= 0).all()
        self.gid = gid
        self.node_of_gid = np.empty(start, dtype=np.int64)
        self.comp_of_gid = np.empty(start, dtype=np.int64)
        self.node_of_gid[gid.reshape(-1)] = np.repeat(np.arange(self.nnodes), self.ncomp)
        self.comp_of_gid[gid.reshape(-1)] = np.tile(np.arange(self.ncomp), self.nnodes)
        # element DoF -> global id, (ne, nv*ncomp)
        self.elem_gid = gid[self.elem_nodes].reshape(ne, nv * ncomp)

    @property
    def nglobal(self):
        return self.nnodes * self.ncomp

    def subdomain(self, r) -> NovlpSubdomain:
        ln = self._local_nodes[r]
        nl = len(ln)
        lnode = np.full(self.nnodes, -1, dtype=np.int64)
        lnode[ln] = np.arange(nl)
        sel = self.elem_rank == r
        en = self.elem_nodes[sel]
        nv = en.shape[1]
        nd = nv * self.ncomp
        ldof = (lnode[en][:, :, None] + nl * np.arange(self.ncomp)[None, None, :]).reshape(len(en), nd)
        rows = np.repeat(ldof, nd, axis=1).reshape(-1)
        cols = np.tile(ldof, (1, nd)).reshape(-1)
        A = _coo_to_csr(rows, cols, self.elem_mats[sel].reshape(-1), (nl * self.ncomp,) * 2)
        dmask = self.dirichlet[ln].T.reshape(-1).astype(np.uint8)
        holders = np.tile(self.node_holders[ln], self.ncomp)
        A = eliminate_dirichlet(A, dmask, 1.0 / holders)
        glob = self.gid[ln].T.reshape(-1).copy()
        owner = np.tile(self.node_owner[ln] == r, self.ncomp).astype(np.uint8)
        public = (holders > 1).astype(np.uint8)
        nb = np.nonzero(self._held[:, ln[self.node_holders[ln] > 1]].any(axis=1))[0]
        return NovlpSubdomain(rank=r, glob=glob, owner=owner, public=public, A=A, b=self.load[ln].T.reshape(-1).copy(),
                              dirichlet=dmask, neighbours=sorted(int(q) for q in nb if q != r))

    def subdomains(self):
        return [self.subdomain(r) for r in range(self.nranks)]

    def dirichlet_of(self, glob):
        return self.dirichlet[self.node_of_gid[glob], self.comp_of_gid[glob]].astype(np.uint8)

    def _region(self, glob, region, with_values_outside_region):
        n = len(glob)
        loc = np.full(self.nglobal, -1, dtype=np.int64)
        loc[glob] = np.arange(n)
        inreg = np.zeros(self.nglobal, dtype=bool)
        inreg[glob if region is None else np.asarray(glob)[np.asarray(region, dtype=bool)]] = True
        eg = self.elem_gid
        touch = (loc[eg] >= 0).any(axis=1)
        eg = eg[touch]
        mats = self.elem_mats[touch]
        nd = eg.shape[1]
        full = inreg[eg].all(axis=1)                       # elements completely inside the region
        if not with_values_outside_region:
            mats = mats * full[:, None, None]
        rows = np.repeat(eg, nd, axis=1).reshape(-1)
        cols = np.tile(eg, (1, nd)).reshape(-1)
        keep = inreg[rows] & inreg[cols]
        return _coo_to_csr(loc[rows[keep]], loc[cols[keep]], mats.reshape(-1)[keep], (n, n))

    def dirichlet_matrix(self, glob, dirichlet_ovlp):
        """principal submatrix of the global matrix on the overlapping DoF set (CreateMatrixDataHandle +
        AddMatrixDataHandle, dune/ddm/datahandles.hh:436-591), Dirichlet DoFs eliminated"""
        return eliminate_dirichlet(self._region(glob, None, True), dirichlet_ovlp)

    def neumann_matrix(self, glob, region, dirichlet_ovlp):
        return eliminate_dirichlet(self._region(glob, region, False), dirichlet_ovlp)


def _steel_rubber(x, y, z):
    """lambda, mu of examples/coefficient.lua:1-72 (rubber block with 2 x 4 steel bars along x)."""
    steel = np.zeros(x.shape, dtype=bool)
    for by in (0.25, 0.75):
        for bz in (0.3, 0.6, 0.9, 1.2):
            steel |= (np.sqrt((y - by) ** 2 + (z - bz) ** 2) <= 0.04)
    steel &= (x >= 0.0) & (x <= 3.0)
    E = np.where(steel, 2e11, 2e7)
    nu = np.where(steel, 0.3, 0.45)
    return E * nu / (1.0 + nu) / (1.0 - 2.0 * nu), E / 2.0 / (1.0 + nu)


class StructuredElasticity(ElementProblem):
    """BASELINE.json configs[4]: P1 linear elasticity on the box [0,10] x [0,1] x [0,1.5] cut into
    cells x 6 Kuhn simplices (examples/linearelasticity.cc:39-41: 80 x 8 x 12 cells, 28 431 DoF;
    ``refine`` halves the mesh width like grid.globalRefine), clamped at x = 0, gravity load, Lame
    parameters of examples/coefficient.lua evaluated at the element centroid (``coefficient="lua"``) or the
    constants of LinearElasticityParameters (linearelasticity.hh:14-44, ``"const"``).  Elements are
    partitioned into ``parts`` slabs along x (the reference uses ParMETIS; any element partition will do)."""

    def __init__(self, cells=(80, 8, 12), parts=8, refine=0, size=(10.0, 1.0, 1.5), coefficient="lua"):
        cells = tuple(int(c) << int(refine) for c in cells)
        self.cells, self.parts = cells, int(parts)
        nx, ny, nz = cells
        N = (nx + 1, ny + 1, nz + 1)
        h = np.array([size[d] / cells[d] for d in range(3)])
        kk, jj, ii = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")   # cell coordinates, x fastest
        cell0 = (ii + N[0] * (jj + N[1] * kk)).reshape(-1)
        stride = np.array([1, N[0], N[0] * N[1]])
        import itertools as it
        tets, types = [], []
        for tpe, perm in enumerate(it.permutations(range(3))):         # Kuhn: paths (0,0,0) -> (1,1,1)
            off = [0]
            for ax in perm:
                off.append(off[-1] + stride[ax])
            tets.append(cell0[:, None] + np.array(off)[None, :])
            types.append(np.full(len(cell0), tpe))
        elem_nodes = np.concatenate(tets)
        etype = np.concatenate(types)
        ecell_x = np.tile(ii.reshape(-1), 6)
        node = np.arange(int(np.prod(N)))
        coords = np.stack([(node % N[0]) * h[0], ((node // N[0]) % N[1]) * h[1], (node // (N[0] * N[1])) * h[2]], axis=1)
        X = coords[elem_nodes]                                          # (ne, 4, 3)
        cen = X.mean(axis=1)
        if coefficient == "lua":
            lam, mu = _steel_rubber(cen[:, 0], cen[:, 1], cen[:, 2])
            fz = -20000.0
        else:
            lam, mu = np.full(len(cen), 100.0), np.full(len(cen), 10000.0)
            fz = -10.0
        # one reference matrix pair per Kuhn type (uniform mesh): K = lam * Kl + mu * Km
        Kl, Km, vol = [], [], None
        for tpe in range(6):
            x = X[np.argmax(etype == tpe)]
            T = np.concatenate([np.ones((4, 1)), x], axis=1)
            g = np.linalg.inv(T)[1:, :].T                               # (4 vertices, 3): gradients of the barycentric basis
            vol = abs(np.linalg.det(T)) / 6.0
            kl = np.einsum("ai,bj->aibj", g, g)                          # lam div u div v
            km = np.einsum("aj,bi->aibj", g, g) + np.einsum("ak,bk,ij->aibj", g, g, np.eye(3))   # 2 mu eps(u):eps(v)
            Kl.append(vol * kl.reshape(12, 12))
            Km.append(vol * km.reshape(12, 12))
        Kl, Km = np.array(Kl), np.array(Km)
        elem_mats = lam[:, None, None] * Kl[etype] + mu[:, None, None] * Km[etype]
        elem_rank = np.minimum(ecell_x * self.parts // nx, self.parts - 1)
        dirichlet = np.repeat((coords[:, 0] < 1e-9)[:, None], 3, axis=1)
        load = np.zeros((len(node), 3))
        np.add.at(load[:, 2], elem_nodes.reshape(-1), -fz * vol / 4.0)   # residual of x = 0: -int f.v
        super().__init__(len(node), 3, elem_nodes, elem_mats, elem_rank, self.parts, dirichlet, load)
        self.coords = coords


# =================================================================================================
# BASELINE.json configs[3]: Q1-DG convection-diffusion, high-contrast coefficient  (examples/pdelab_example.cc)
# =================================================================================================
def checkerboard_kappa(ncells, blocks=8, a1=1e-6, a2=1.0):
    """alpha(x, y) of examples/convection_diffusion_coefficient.lua:1-17 on an nx x ny cell grid (shape (ny, nx))."""
    nx, ny = ncells
    ix = (np.arange(nx) * blocks) // nx
    iy = (np.arange(ny) * blocks) // ny
    return np.where((ix[None, :] % 2) == (iy[:, None] % 2), a2, a1)


class StructuredDG2D:
    """Q1-DG (4 DoF per cell) symmetric-interior-penalty diffusion + upwind convection on nx x ny square cells of the
    unit square -- a restatement of dune-pdelab's ConvectionDiffusionDG local operator (weighted averages "weightsOn",
    penalty alpha * k(k+d-1) / h_F * harmonic average; dune-pdelab is absent from the snapshot, so the matrices are
    "parity unpinned" inputs) for the problem of examples/convection_diffusion_coefficient.lua: checkerboard
    diffusion 1e-6 / 1, b = (1/3, 1), f = 0, Dirichlet g = [x < 1e-6] on x = 0 and y = 0, outflow elsewhere.

    Decomposition as the reference's DG set-up (examples/problem_traits.hh:80-83, AllEntitySet): a rank holds its
    interior cells plus one layer of face-neighbour ghost cells; ghost DoFs are copies owned by the neighbour; the
    locally assembled matrix is consistent and ``make_additive`` (dune/ddm/pdelab_helper.hh:108-149) zeroes the
    copy rows.  The system matrix is non-symmetric; the Neumann matrices of the GenEO eigenproblem come from the
    SYMMETRIC part (no convection: examples/generic_ddm_problem.hh:253-266, LuaConvectionDiffusionProblem<.., true>):
    A_dir of the symmetric problem minus the self-coupling blocks of the faces between a cell of the subdomain and a
    cell outside (examples/assemblewrapper.hh:268-352)."""

    def __init__(self, ncells, P, kappa=None, b=(1.0 / 3.0, 1.0), alpha=1.0):
        self.nx, self.ny = int(ncells[0]), int(ncells[1])
        self.P = (int(P[0]), int(P[1]))
        self.nranks = self.P[0] * self.P[1]
        self.kappa = checkerboard_kappa((self.nx, self.ny)) if kappa is None else np.asarray(kappa, dtype=float)
        assert self.kappa.shape == (self.ny, self.nx)
        self.b = (float(b[0]), float(b[1]))
        self.alpha = float(alpha)
        self.h = 1.0 / self.nx
        assert self.nx == self.ny, "square cells on the unit square"
        nc = self.nx * self.ny
        cx, cy = np.arange(nc) % self.nx, np.arange(nc) // self.nx
        sx, sy = _split(self.nx, self.P[0]), _split(self.ny, self.P[1])
        px = np.searchsorted([s[1] for s in sx], cx, side="right")
        py = np.searchsorted([s[1] for s in sy], cy, side="right")
        self.cell_rank = px + self.P[0] * py
        # rank-contiguous global ids: interior cells of rank 0 (lexicographic), then rank 1, ...
        order = np.argsort(self.cell_rank, kind="stable")
        self.cell_gid0 = np.empty(nc, dtype=np.int64)
        self.cell_gid0[order] = 4 * np.arange(nc)
        self.cell_of_gid = np.repeat(order, 4)
        self._assemble()

    @property
    def nglobal(self):
        return 4 * self.nx * self.ny

    # ---- element / face matrices -------------------------------------------------------------------
    def _assemble(self):
        nx, ny, h, al = self.nx, self.ny, self.h, self.alpha
        b1, b2 = self.b
        nc = nx * ny
        kap = self.kappa.reshape(-1)
        M = np.array([[2.0, 1.0], [1.0, 2.0]]) / 6.0
        S = np.array([[1.0, -1.0], [-1.0, 1.0]])
        Gd = np.array([[-0.5, -0.5], [0.5, 0.5]])                      # int l_i' l_j
        vs, vn = np.array([0.0, 1.0]), np.array([1.0, 0.0])            # traces of the lower (s) / upper (n) cell on the shared face
        der = np.array([-1.0, 1.0]) / h                                 # derivative along the face normal (+axis)

        def blk(axis, N2):                                              # 2x2 in the normal index (x) h*M in the tangential index
            return np.kron(h * M, N2) if axis == 0 else np.kron(N2, h * M)

        o = np.outer
        rows, cols, vals_f, vals_s = [], [], [], []

        def add(rc, cc, B_full, B_sym):
            r = (4 * rc[:, None, None] + np.arange(4)[None, :, None]) + np.zeros((1, 1, 4), dtype=np.int64)
            c = (4 * cc[:, None, None] + np.arange(4)[None, None, :]) + np.zeros((1, 4, 1), dtype=np.int64)
            rows.append(r.reshape(-1))
            cols.append(c.reshape(-1))
            vals_f.append(np.broadcast_to(B_full, (len(rc), 4, 4)).reshape(-1))
            vals_s.append(np.broadcast_to(B_sym, (len(rc), 4, 4)).reshape(-1))

        allc = np.arange(nc)
        Vd = np.kron(M, S) + np.kron(S, M)
        Vc = -h * (b1 * np.kron(M, Gd) + b2 * np.kron(Gd, M))
        add(allc, allc, kap[:, None, None] * Vd[None] + Vc[None], kap[:, None, None] * Vd[None])
        self.self_face = np.zeros((4, nc, 4, 4))                        # symmetric problem: face self-coupling per direction (-x, +x, -y, +y)
        cxa, cya = allc % nx, allc // nx
        for axis, bn in ((0, b1), (1, b2)):
            ok = (cxa < nx - 1) if axis == 0 else (cya < ny - 1)
            s = allc[ok]
            n = s + (1 if axis == 0 else nx)
            ks, kn = kap[s], kap[n]
            ws, wn = kn / (ks + kn), ks / (ks + kn)                     # weights of the weighted average ("weightsOn")
            gam = al * 2.0 * (2.0 * ks * kn / (ks + kn)) / h            # alpha * k(k+d-1) / h_F * harmonic average, k = 1, d = 2
            cs, cn = (ws * ks)[:, None, None], (wn * kn)[:, None, None]
            g3 = gam[:, None, None]
            Bss = -cs * blk(axis, o(vs, der) + o(der, vs))[None] + g3 * blk(axis, o(vs, vs))[None]
            Bsn = -cn * blk(axis, o(vs, der))[None] + cs * blk(axis, o(der, vn))[None] - g3 * blk(axis, o(vs, vn))[None]
            Bns = cs * blk(axis, o(vn, der))[None] - cn * blk(axis, o(der, vs))[None] - g3 * blk(axis, o(vn, vs))[None]
            Bnn = cn * blk(axis, o(vn, der) + o(der, vn))[None] + g3 * blk(axis, o(vn, vn))[None]
            up = max(bn, 0.0), min(bn, 0.0)                             # upwind flux (b.n) u_up [v]
            add(s, s, Bss + up[0] * blk(axis, o(vs, vs))[None], Bss)
            add(s, n, Bsn + up[1] * blk(axis, o(vs, vn))[None], Bsn)
            add(n, s, Bns - up[0] * blk(axis, o(vn, vs))[None], Bns)
            add(n, n, Bnn - up[1] * blk(axis, o(vn, vn))[None], Bnn)
            self.self_face[2 * axis + 1, s] = Bss
            self.self_face[2 * axis, n] = Bnn
        # boundary faces: Dirichlet on x = 0 and y = 0 (g = 1 on x = 0), outflow on x = 1 and y = 1
        x0 = np.zeros(4 * nc)
        rhs = np.zeros(4 * nc)
        for axis, bn_axis in ((0, b1), (1, b2)):
            lo = allc[(cxa == 0) if axis == 0 else (cya == 0)]
            hi = allc[(cxa == nx - 1) if axis == 0 else (cya == ny - 1)]
            k = kap[lo][:, None, None]
            gam = (al * 2.0 * kap[lo] / h)[:, None, None]
            # outward normal -axis: outward derivative = -der, own trace = vn (index 0)
            Bd = -k * blk(axis, o(vn, -der) + o(-der, vn))[None] + gam * blk(axis, o(vn, vn))[None]
            bdotn = -bn_axis
            add(lo, lo, Bd + max(bdotn, 0.0) * blk(axis, o(vn, vn))[None], Bd)
            g = 1.0 if axis == 0 else 0.0
            if g != 0.0:
                dofs = (4 * lo[:, None] + np.arange(4)[None, :])
                gvec = np.kron(np.ones(2), vn) if axis == 0 else np.kron(vn, np.ones(2))     # interpolation of g: corner DoFs on the face
                x0[dofs] = np.maximum(x0[dofs], gvec[None, :])
                # l(v) = theta g k dn v + gamma g v - min(b.n, 0) g v, g constant along the face: apply the blocks to the trace vector
                lv = (-k * blk(axis, o(-der, vn))[None] + gam * blk(axis, o(vn, vn))[None] - min(bdotn, 0.0) * blk(axis, o(vn, vn))[None]) @ gvec
                rhs[dofs] += g * lv
            add(hi, hi, bn_axis * blk(axis, o(vs, vs))[None] * np.ones((len(hi), 1, 1)), np.zeros((len(hi), 4, 4)))
        rows, cols = np.concatenate(rows), np.concatenate(cols)
        # natural numbering (4 * cell + a) -> rank-contiguous global ids
        nat2gid = (self.cell_gid0[:, None] + np.arange(4)[None, :]).reshape(-1)
        n = 4 * nc
        self.G = _coo_to_csr(nat2gid[rows], nat2gid[cols], np.concatenate(vals_f), (n, n))
        self.Gsym = _coo_to_csr(nat2gid[rows], nat2gid[cols], np.concatenate(vals_s), (n, n))
        xg = np.empty(n)
        xg[nat2gid] = x0
        rg = np.empty(n)
        rg[nat2gid] = rhs
        self.x0 = xg
        self.rhs = self.G @ xg - rg                                     # residual of the Dirichlet extension (generic_ddm_problem.hh:224)

    # ---- rank-local non-overlapping data ---------------------------------------------------------------
    def _neighbour_cells(self, cells):
        nx, ny = self.nx, self.ny
        cx, cy = cells % nx, cells // nx
        out = []
        for d, (dx, dy) in enumerate(((-1, 0), (1, 0), (0, -1), (0, 1))):
            ok = (cx + dx >= 0) & (cx + dx < nx) & (cy + dy >= 0) & (cy + dy < ny)
            out.append(np.where(ok, cells + dx + nx * dy, -1))
        return np.stack(out)                                            # (4, len(cells)), -1 = domain boundary

    def subdomain(self, r) -> NovlpSubdomain:
        interior = np.nonzero(self.cell_rank == r)[0]
        nb = self._neighbour_cells(interior)
        ghosts = np.setdiff1d(np.unique(nb[nb >= 0]), interior)
        cells = np.sort(np.concatenate([interior, ghosts]))            # local cell order: lexicographic
        is_int = self.cell_rank[cells] == r
        glob = (self.cell_gid0[cells][:, None] + np.arange(4)[None, :]).reshape(-1)
        A = self.G[glob][:, glob].tocsr()
        A.sort_indices()
        owner = np.repeat(is_int, 4).astype(np.uint8)
        rows = np.repeat(np.arange(A.shape[0]), np.diff(A.indptr))
        A.data = np.where(owner[rows] > 0, A.data, 0.0)                 # make_additive: copy rows are zeroed
        # shared DoFs: ghost cells, and interior cells that are ghosts of another rank
        nbc = self._neighbour_cells(cells)
        other = np.zeros(len(cells), dtype=bool)
        for d in range(4):
            other |= (nbc[d] >= 0) & (self.cell_rank[np.maximum(nbc[d], 0)] != r)
        public = np.repeat(~is_int | other, 4).astype(np.uint8)
        nbr = np.unique(self.cell_rank[ghosts])
        return NovlpSubdomain(rank=r, glob=glob, owner=owner, public=public, A=A, b=self.rhs[glob].copy(),
                              dirichlet=np.zeros(len(glob), dtype=np.uint8), neighbours=[int(q) for q in nbr])

    def subdomains(self):
        return [self.subdomain(r) for r in range(self.nranks)]

    def dof_coords(self, glob):
        """(len, 2) coordinates of the Lagrange nodes of the DoFs (cell corners): what Dune::PDELab::interpolate evaluates the
        template functions 1, x, y, xy at (dune/ddm/twolevel_schwarz.hh:68-75)"""
        glob = np.asarray(glob, dtype=np.int64)
        cells = self.cell_of_gid[glob]
        a = (glob - self.cell_gid0[cells]).astype(np.int64)
        return np.stack([((cells % self.nx) + (a & 1)) * self.h, ((cells // self.nx) + (a >> 1)) * self.h], axis=1)

    # ---- overlapping matrices ------------------------------------------------------------------------
    def dirichlet_of(self, glob):
        return np.zeros(len(glob), dtype=np.uint8)                      # NoConstraints: Dirichlet data enter weakly

    def dirichlet_matrix(self, glob, dirichlet_ovlp=None):
        A = self.G[glob][:, glob].tocsr()
        A.sort_indices()
        return A

    def neumann_matrix(self, glob, region, dirichlet_ovlp=None):
        glob = np.asarray(glob, dtype=np.int64)
        n = len(glob)
        A = self.Gsym[glob][:, glob].tocsr()
        A.sort_indices()
        cells = self.cell_of_gid[glob]
        assert n % 4 == 0 and (cells.reshape(-1, 4) == cells[::4, None]).all(), "DG subdomains consist of whole cells"
        cells = cells[::4]
        inset = np.zeros(self.nx * self.ny, dtype=bool)
        inset[cells] = True
        nbc = self._neighbour_cells(cells)
        corr = np.zeros((len(cells), 4, 4))
        for d in range(4):
            cut = (nbc[d] >= 0) & ~inset[np.maximum(nbc[d], 0)]         # face towards a cell outside the subdomain
            corr[cut] += self.self_face[d, cells[cut]]
        base = 4 * np.arange(len(cells))
        r = (base[:, None, None] + np.arange(4)[None, :, None] + np.zeros((1, 1, 4), dtype=np.int64)).reshape(-1)
        c = (base[:, None, None] + np.arange(4)[None, None, :] + np.zeros((1, 4, 1), dtype=np.int64)).reshape(-1)
        A = (A - sp.csr_matrix((corr.reshape(-1), (r, c)), shape=(n, n))).tocsr()
        if region is not None:                                          # second Neumann matrix: entries inside the region only
            reg = np.asarray(region, dtype=bool)
            A = A.tocoo()
            keep = reg[A.row] & reg[A.col]
            A = sp.csr_matrix((A.data[keep], (A.row[keep], A.col[keep])), shape=(n, n))
        A.sort_indices()
        return A
