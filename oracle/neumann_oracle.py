"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of how the reference's drivers obtain the overlapping Neumann matrices: assembly through a wrapped local operator that
records "Neumann corrections" element by element, exchange of the corrections as (global row, global column, value) triples, and their
subtraction from the overlapping Dirichlet matrix.  Plain Python loops over elements and element corners -- only for small cases.

  AssembleWrapper::set_masks / jacobian_volume / get_correction_triples     examples/assemblewrapper.hh:182-262, 385-470
  assemble_overlapping_matrices                                            examples/pdelab_helper.hh:113-436

The element matrices come from the caller (in the reference: PDELab's local operator, `mat.container() - M_before`).  Q1 / P1 volume
terms only (jacobian_volume); the skeleton variant for DG (assemblewrapper.hh:265-367) is not restated.

Pinned by construction rather than by a fixture (the reference holds no golden data for this path: "parity unpinned" against the
reference itself): tests/test_neumann_assembly.py checks that the matrices this procedure produces are the element sums over the
elements inside the region, i.e. the definition of the Neumann matrix the procedure exists to compute."""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from . import setup_oracle as so


class AssembleWrapper:
    """examples/assemblewrapper.hh:27-490 around an element provider: `elements` = list of (dofs, Ke), dofs local indices."""

    def __init__(self, elements):
        self.elements = elements
        self.masks = None

    def set_masks(self, A, on_boundary_mask_for_rank, inside_boundary_mask_for_rank, on_boundary_mask, outside_boundary_mask):
        """:385-428 -- correction matrices on the pattern of A restricted to (mask, mask) pairs, zero-initialised"""
        self.masks = (on_boundary_mask_for_rank, inside_boundary_mask_for_rank, on_boundary_mask, outside_boundary_mask)
        A = sp.csr_matrix(A)
        self.corr = {}
        for rank, mask in list(on_boundary_mask_for_rank.items()) + [(-1, on_boundary_mask)]:
            An = {}
            for i in range(A.shape[0]):
                if mask[i]:
                    for k in range(A.indptr[i], A.indptr[i + 1]):
                        if mask[A.indices[k]]:
                            An[(i, int(A.indices[k]))] = 0.0
            self.corr[rank] = An

    def jacobian(self):
        """the grid operator's element loop calling jacobian_volume (:182-263)"""
        on_for, in_for, on_own, out_own = self.masks
        for dofs, Ke in self.elements:
            gi = [int(v) for v in dofs]
            for rank, mask in on_for.items():                       # corrections for other ranks (:207-236)
                inside = in_for[rank]
                at_boundary = any(mask[g] for g in gi)
                outside = any((not mask[g]) and (not inside[g]) for g in gi)
                if at_boundary and outside:
                    An = self.corr[rank]
                    for a, ga in enumerate(gi):
                        if not mask[ga]:
                            continue
                        for b, gb in enumerate(gi):
                            if not mask[gb]:
                                continue
                            An[(ga, gb)] = An[(ga, gb)] + Ke[a][b]
            at_boundary = any(on_own[g] for g in gi)                # corrections for ourselves (:239-262)
            outside = any(out_own[g] for g in gi)
            if at_boundary and outside:
                An = self.corr[-1]
                for a, ga in enumerate(gi):
                    if not on_own[ga]:
                        continue
                    for b, gb in enumerate(gi):
                        if not on_own[gb]:
                            continue
                        An[(ga, gb)] = An[(ga, gb)] + Ke[a][b]

    def get_correction_triples(self, glob):
        """:440-470 -- rank >= 0: global ids; rank -1: local ids; rows ascending, columns ascending"""
        out = {}
        for rank, An in self.corr.items():
            keys = sorted(An)
            if rank >= 0:
                out[rank] = [(int(glob[i]), int(glob[j]), An[(i, j)]) for i, j in keys]
            else:
                out[rank] = [(i, j, An[(i, j)]) for i, j in keys]
        return out


def assemble_overlapping_matrices(subs, elements, dirichlet_novlp, overlap, first_region="all", second_region="overlap"):
    """examples/pdelab_helper.hh:113-436 for all ranks.  subs: the non-overlapping subdomains (A additive, glob, owner, public),
    elements[p]: list of (dofs, Ke) of rank p.  Returns per rank (A_dir, A_neu, B_neu, dirichlet_mask_ovlp, boundary_dst, triples)."""
    if first_region != "all" and first_region != second_region:
        raise NotImplementedError("Two different Neumann regions are only supported if the first is NeumannRegion::All")   # :181
    ranks, _ = so.make_overlapping_communication(subs, overlap)
    A_raw, _ = so.overlapping_matrix(ranks, subs)                          # CreateMatrixDataHandle + AddMatrixDataHandle (:134-137, 288-289)
    bmask = so.identify_boundary(ranks, A_raw)                             # :140-141
    dst = [so.graph_distance_sweeps(A, b, 4 * overlap + 1) for A, b in zip(A_raw, bmask)]     # :150-158
    indicator = [np.where(d == 0, 1, 2) for d in dst]                      # :160-163
    P = len(ranks)
    width = {"overlap": 2 * overlap, "extended_overlap": 2 * overlap + 1}
    triples = []
    for p in ranks:
        n_o = p.n_o
        on_for, in_for = {}, {}
        for q in ranks:                                                    # CopyVectorDataHandleWithRank (:165-180)
            if q.rank == p.rank:
                continue
            copied = np.zeros(n_o, dtype=np.int64)
            seen = False
            for i in range(n_o):
                lq = q.loc.get(int(p.glob[i]))
                if lq is not None:
                    copied[i] = indicator[q.rank][lq]
                    seen = True
            if seen:
                on_for[q.rank] = copied == 1
                in_for[q.rank] = copied == 2
        on_own = np.zeros(n_o, dtype=bool)
        out_own = np.zeros(n_o, dtype=bool)
        inner = None
        if "overlap" in (first_region, second_region):                     # :183-188
            inner = 2 * overlap
        elif "extended_overlap" in (first_region, second_region):          # :189-194
            inner = 2 * overlap + 1
        if inner is not None:
            on_own = dst[p.rank][:n_o] == inner
            out_own = dst[p.rank][:n_o] == inner + 1
        w = AssembleWrapper(elements[p.rank])
        w.set_masks(subs[p.rank].A, on_for, in_for, on_own, out_own)       # :199-200
        w.jacobian()                                                       # :201
        triples.append(w.get_correction_triples(p.glob))                   # :210
    # dirichlet mask on the overlapping subdomain (:291-302)
    dm = []
    for p in ranks:
        v = np.zeros(p.n)
        v[:p.n_o] = dirichlet_novlp[p.rank]
        dm.append(v)
    dm = [(v > 0).astype(np.uint8) for v in so.add_vector(ranks, dm)]
    out = []
    for q in ranks:
        A_dir = A_raw[q.rank].copy().tolil()
        remote = {p.rank: triples[p.rank][q.rank] for p in ranks if q.rank in triples[p.rank]}     # MPI exchange (:212-262)
        own = triples[q.rank][-1]

        def restricted(M, w_):
            R = sp.lil_matrix(M.shape)
            Mc = sp.csr_matrix(M)
            for i in range(M.shape[0]):
                if dst[q.rank][i] > w_:
                    continue
                for k in range(Mc.indptr[i], Mc.indptr[i + 1]):
                    j = Mc.indices[k]
                    if dst[q.rank][j] > w_:
                        continue
                    R[i, j] = Mc.data[k]
            return R, {(i, int(j)) for i in range(M.shape[0]) if dst[q.rank][i] <= w_
                       for j in Mc.indices[Mc.indptr[i]:Mc.indptr[i + 1]] if dst[q.rank][j] <= w_}

        def finish(L, pattern):
            """CSR with exactly `pattern` stored (explicit zeros included), Dirichlet rows / columns eliminated"""
            keys = sorted(pattern)
            M = sp.csr_matrix((np.array([L[i, j] for i, j in keys]), (np.array([k[0] for k in keys]), np.array([k[1] for k in keys]))), shape=L.shape)
            M.sort_indices()
            return so.eliminate_dirichlet(M, dm[q.rank])

        Ac = sp.csr_matrix(A_raw[q.rank])
        full = {(i, int(j)) for i in range(Ac.shape[0]) for j in Ac.indices[Ac.indptr[i]:Ac.indptr[i + 1]]}
        if first_region == "all":                                          # :308-330
            A_neu = A_dir.copy()
            for rank in sorted(remote):
                for (grow, gcol, val) in remote[rank]:
                    if grow in q.loc and gcol in q.loc:
                        A_neu[q.loc[grow], q.loc[gcol]] = A_neu[q.loc[grow], q.loc[gcol]] - val
            A_neu_c = finish(A_neu, full)
            pat_neu = full
        else:                                                              # :331-362
            A_neu, pat_neu = restricted(A_dir, width[first_region])
            for rank in sorted(remote):
                for (grow, gcol, val) in remote[rank]:
                    if grow in q.loc and gcol in q.loc:
                        A_neu[q.loc[grow], q.loc[gcol]] = A_neu[q.loc[grow], q.loc[gcol]] - val
            for (i, j, val) in own:
                A_neu[i, j] = A_neu[i, j] - val
            A_neu_c = finish(A_neu, pat_neu)
        if second_region == first_region:                                  # :400
            B_neu_c = A_neu_c
        else:                                                              # :401-423: copied from the finished A_neu
            B_neu, pat_b = restricted(A_neu_c, 2 * overlap)
            for (i, j, val) in own:
                B_neu[i, j] = B_neu[i, j] - val
            B_neu_c = finish(B_neu, pat_b)
        out.append((so.eliminate_dirichlet(sp.csr_matrix(A_raw[q.rank]), dm[q.rank]), A_neu_c, B_neu_c, dm[q.rank], dst[q.rank], triples[q.rank]))
    return out
