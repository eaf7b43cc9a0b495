"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of the remaining coarse-space builders, numpy / scipy, small
problems only.  Citations are relative to /root/reference/dune/ddm/coarsespaces.

  * EnergyMinimalExtension            energy_minimal_extension.hh:36-229
  * MsGFEMCoarseSpace                 coarse_spaces.hh:663-831   (saddle-point pencil, literally)
  * GenEORingCoarseSpace              coarse_spaces.hh:502-648
  * MsGFEMRingCoarseSpace             coarse_spaces.hh:913-1163
  * HarmonicExtensionCoarseSpace      coarse_spaces.hh:1232-1266
  * SVDCoarseSpace                    coarse_spaces.hh:1268-1407
  * ConstraintGenEOCoarseSpace        coarse_spaces.hh:394-490: in this snapshot solve_gevp ignores the constraint callback
    (eigensolvers/eigensolvers.hh:27-30, "(void)callback"), so its basis is GenEO's -> geneo_oracle.geneo_basis.

The eigenproblems go through geneo_oracle.spectra_gevp (the shift-invert IRLM restatement; UMFPACK -> scipy splu).
Parity status: the reference stores no basis vectors / eigenvalues of these spaces => "parity unpinned" by
reference fixtures; tests/test_oracle_coarse.py pins this file by properties (a-harmonicity, an independent dense
formulation of the constrained eigenproblem on the boundary unknowns, SVD against numpy).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spl

from . import geneo_oracle as go

INTERIOR, BOUNDARY, DIRICHLET = 0, 1, 2


class EnergyMinimalExtension:
    """energy_minimal_extension.hh:36-229: u_i = -A_ii^-1 (A [0; u_b])_i."""

    def __init__(self, A, interior_indices, boundary_indices):
        self.A = sp.csr_matrix(A)
        self.interior = np.asarray(interior_indices, dtype=np.int64)
        self.boundary = np.asarray(boundary_indices, dtype=np.int64)
        Aii = self.A[self.interior][:, self.interior].tocsc()      # :46-69
        self.lu = spl.splu(Aii)                                    # :76-86 (UMFPACK, no iterative refinement)

    def extend(self, boundary_values):                             # :104-131
        v_full = np.zeros(self.A.shape[0])
        v_full[self.boundary] = boundary_values
        rhs = (self.A @ v_full)[self.interior]
        return -self.lu.solve(rhs)


def _partition_dofs(n, dirichlet_mask, boundary_mask):
    """coarse_spaces.hh:722-751: DOF classes and the reordering interior | boundary | Dirichlet."""
    part = np.where(np.asarray(dirichlet_mask) > 0, DIRICHLET, np.where(np.asarray(boundary_mask) != 0, BOUNDARY, INTERIOR))
    ni, nb = int((part == INTERIOR).sum()), int((part == BOUNDARY).sum())
    reorder = np.empty(n, dtype=np.int64)
    reorder[part == INTERIOR] = np.arange(ni)
    reorder[part == BOUNDARY] = ni + np.arange(nb)
    reorder[part == DIRICHLET] = ni + nb + np.arange(n - ni - nb)
    return part, reorder, ni, nb


def _msgfem_pencil(A_neu, A_con, pou_rows, part, reorder, ni, nb, rhs_interior_only):
    """The saddle-point pencil of coarse_spaces.hh:753-812 (and :1003-1068 for the ring variant).
    A_con supplies the a-harmonic constraint rows, A_neu the (1,1) block and the right-hand side."""
    n_big = ni + nb + ni
    A_neu = sp.coo_matrix(A_neu)
    A_con = sp.coo_matrix(A_con)
    r, c, v = A_con.row, A_con.col, A_con.data
    k = (part[r] == INTERIOR) & (part[c] != DIRICHLET)             # constraint block, both triangles (:763-777)
    rows = [reorder[c[k]], ni + nb + reorder[r[k]]]
    cols = [ni + nb + reorder[r[k]], reorder[c[k]]]
    vals = [v[k], v[k]]
    r, c, v = A_neu.row, A_neu.col, A_neu.data
    k = (part[r] != DIRICHLET) & (part[c] != DIRICHLET)            # (1,1) block (:780-792)
    rows.append(reorder[r[k]])
    cols.append(reorder[c[k]])
    vals.append(v[k])
    A_lhs = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n_big, n_big))
    if rhs_interior_only:
        k = (part[r] == INTERIOR) & (part[c] == INTERIOR)          # :801-811
    B = sp.csr_matrix((pou_rows[r[k]] * pou_rows[c[k]] * v[k], (reorder[r[k]], reorder[c[k]])), shape=(n_big, n_big))
    return A_lhs, B


def msgfem_eigenvectors(A_neu, A_dir, pou, dirichlet_mask, boundary_mask, eig_ptree=None):
    """MsGFEMCoarseSpace::setup_msgfem_impl up to the extraction (coarse_spaces.hh:712-826): (vectors as columns, eigenvalues)."""
    n = A_dir.shape[0]
    if A_neu.shape[0] != n:
        raise ValueError("The two matrices must have the same size")                                   # :714
    if len(dirichlet_mask) != n:
        raise ValueError("The matrix and the Dirichlet mask must have the same size")                 # :716
    if len(pou) != n:
        raise ValueError("The matrix and the partition of unity must have the same size")             # :718
    part, reorder, ni, nb = _partition_dofs(n, dirichlet_mask, boundary_mask)
    A_lhs, B = _msgfem_pencil(A_neu, A_dir, np.asarray(pou, dtype=float), part, reorder, ni, nb, rhs_interior_only=True)
    lam, X, _ = go.spectra_gevp(A_lhs, B, go.EigensolverParams(eig_ptree))                              # :815
    V = np.zeros((n, X.shape[1]))
    free = part != DIRICHLET
    V[free] = X[reorder[free]]                                                                          # :818-823
    return V, lam


def msgfem_basis(A_neu, A_dir, pou, dirichlet_mask, boundary_mask, eig_ptree=None):
    V, lam = msgfem_eigenvectors(A_neu, A_dir, pou, dirichlet_mask, boundary_mask, eig_ptree)
    return go.finalize_eigenvectors([V[:, j].copy() for j in range(V.shape[1])], np.asarray(pou, dtype=float)), lam   # :825-826


def _row_neighbours(A, i):
    return A.indices[A.indptr[i]:A.indptr[i + 1]]


def geneo_ring_basis(A_dir, A_ring, pou, ring_to_subdomain, eig_ptree=None):
    """GenEORingCoarseSpace (coarse_spaces.hh:517-633).  A_ring lives on the ring's own numbering."""
    A_dir = sp.csr_matrix(A_dir)
    n = A_dir.shape[0]
    ring = np.asarray(ring_to_subdomain, dtype=np.int64)
    in_ring = np.zeros(n, dtype=bool)
    in_ring[ring] = True
    pou = np.asarray(pou, dtype=float)
    mod_pou = pou.copy()                                                                                # :541
    interior_to_subdomain, inner_ring_boundary = [], []
    for i in range(n):                                                                                  # :543-560
        if not in_ring[i]:
            interior_to_subdomain.append(i)
            mod_pou[i] = 0
        elif not in_ring[_row_neighbours(A_dir, i)].all():
            inner_ring_boundary.append(i)
            mod_pou[i] = 0
    on_irb = np.zeros(n, dtype=bool)
    on_irb[inner_ring_boundary] = True
    C = go.scale_matrix_with_pou(A_ring, mod_pou[ring])                                                 # :567-568 (ring indices -> subdomain pou)
    lam, X, _ = go.spectra_gevp(A_ring, C, go.EigensolverParams(eig_ptree))                             # :571
    inside = []                                                                                         # :582-589 (one entry per such neighbour: duplicates as in the reference)
    for i in ring:
        for j in _row_neighbours(A_dir, i):
            if on_irb[j] and not on_irb[i]:
                inside.append(int(i))
    ext_interior = np.array(interior_to_subdomain + inner_ring_boundary, dtype=np.int64)                # :592-595
    ext = EnergyMinimalExtension(A_dir, ext_interior, np.array(inside, dtype=np.int64))                 # :598
    sub_to_ring = {int(s): k for k, s in enumerate(ring)}
    inside_to_ring = np.array([sub_to_ring[i] for i in inside], dtype=np.int64)
    vecs = []
    for k in range(X.shape[1]):                                                                         # :612-624
        v = np.zeros(n)
        v[ring] = X[:, k]
        v[ext_interior] = ext.extend(X[inside_to_ring, k])
        vecs.append(v)
    return go.finalize_eigenvectors(vecs, pou), lam                                                      # :627


def boundary_distance(A_dir, boundary_mask, rounds):
    """coarse_spaces.hh:950-962: in-place relaxation sweeps in row order (Gauss-Seidel), `rounds` times."""
    A_dir = sp.csr_matrix(A_dir)
    n = A_dir.shape[0]
    dist = np.full(n, np.iinfo(np.int32).max - 1, dtype=np.int64)
    dist[np.asarray(boundary_mask) > 0] = 0
    for _ in range(rounds):
        for i in range(n):
            nb = _row_neighbours(A_dir, i)
            if len(nb):
                dist[i] = min(dist[i], int(dist[nb].min()) + 1)
    return dist


def msgfem_ring_basis(A_dir, A_ring, overlap, pou, shrink, dirichlet_mask, boundary_mask, ring_to_subdomain, eig_ptree=None):
    """MsGFEMRingCoarseSpace (coarse_spaces.hh:931-1149)."""
    A_dir = sp.csr_matrix(A_dir)
    n = A_dir.shape[0]
    ring = np.asarray(ring_to_subdomain, dtype=np.int64)
    if len(ring) == 0:
        raise ValueError("The ring to subdomain mapping is empty, cannot build MsGFEM ring coarse space")   # :972
    pou = np.asarray(pou, dtype=float)
    dist = boundary_distance(A_dir, boundary_mask, 2 * overlap + 2)
    ring_width = 2 * overlap - 2 * shrink                                                               # :964
    mod_pou = pou.copy()
    mod_pou[dist >= shrink + ring_width] = 0                                                            # :974-976
    inside_ring_boundary = dist[ring] == 2 * overlap                                                    # :978-980
    dmask = np.asarray(dirichlet_mask)[ring]
    bmask = (np.asarray(boundary_mask)[ring] != 0) | inside_ring_boundary                              # :992-1000
    part, reorder, ni, nb = _partition_dofs(len(ring), dmask, bmask)
    A_lhs, B = _msgfem_pencil(A_ring, A_ring, mod_pou[ring], part, reorder, ni, nb, rhs_interior_only=False)   # :1018-1068
    lam, X, _ = go.spectra_gevp(A_lhs, B, go.EigensolverParams(eig_ptree))                              # :1071
    Vr = np.zeros((len(ring), X.shape[1]))
    free = part != DIRICHLET
    Vr[free] = X[reorder[free]]                                                                         # :1078-1081
    ext_interior = np.nonzero(dist > shrink + ring_width - 1)[0]                                        # :1090-1092
    ext_boundary = np.nonzero(dist == shrink + ring_width - 1)[0]
    ext = EnergyMinimalExtension(A_dir, ext_interior, ext_boundary)
    sub_to_ring = {int(s): k for k, s in enumerate(ring)}
    b_to_ring = np.array([sub_to_ring[int(i)] for i in ext_boundary], dtype=np.int64)
    vecs = []
    for k in range(Vr.shape[1]):                                                                        # :1120-1133
        v = np.zeros(n)
        v[ring] = Vr[:, k]
        v[ext_interior] = ext.extend(Vr[b_to_ring, k])
        vecs.append(v)
    return go.finalize_eigenvectors(vecs, pou), lam


def harmonic_extension_basis(A_ovlp, pou, boundary_data, boundary_mask):
    """HarmonicExtensionCoarseSpace (coarse_spaces.hh:1232-1266); boundary_data: list of vectors on the boundary DoFs."""
    bmask = np.asarray(boundary_mask) != 0
    b_idx, i_idx = np.nonzero(bmask)[0], np.nonzero(~bmask)[0]
    ext = EnergyMinimalExtension(A_ovlp, i_idx, b_idx)
    vecs = []
    for g in boundary_data:
        v = np.zeros(A_ovlp.shape[0])
        v[b_idx] = g
        v[i_idx] = ext.extend(np.asarray(g, dtype=float))
        vecs.append(v)
    return go.finalize_eigenvectors(vecs, np.asarray(pou, dtype=float))


def svd_basis(A_ovlp, pou, boundary_mask, dirichlet_mask, n_vectors=10, mult_pou=False):
    """SVDCoarseSpace (coarse_spaces.hh:1268-1407): left singular vectors of T = D A_ii^-1 A_{i,Gamma}."""
    A = sp.csr_matrix(A_ovlp)
    n = A.shape[0]
    part, _, _, _ = _partition_dofs(n, dirichlet_mask, boundary_mask)
    i_idx, b_idx = np.nonzero(part == INTERIOR)[0], np.nonzero(part == BOUNDARY)[0]
    lu = spl.splu(A[i_idx][:, i_idx].tocsc())
    T = lu.solve(A[i_idx][:, b_idx].toarray())                                                          # :1345-1362
    T = np.asarray(pou, dtype=float)[i_idx, None] * T                                                   # :1364-1371
    U, s, _ = np.linalg.svd(T, full_matrices=False)                                                     # :1379 (Eigen bdcSvd, thin U)
    vecs = []
    for k in range(n_vectors):                                                                          # :1393-1401
        v = np.zeros(n)
        v[i_idx] = U[:, k]
        vecs.append(v)
    if mult_pou:
        vecs = go.finalize_eigenvectors(vecs, np.asarray(pou, dtype=float))
    return vecs, s
