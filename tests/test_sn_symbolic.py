"""Host half of the device supernodal Cholesky (csrc/sn_chol_host.hpp: nested-dissection supernodes of at most 128 columns, one
shared row list per supernode, supernodal elimination tree with levels), checked on the CPU: the numeric algorithm the device
kernels implement (csrc/sn_chol.hpp: diagonal block Cholesky, panel solve, scatter of R R^T into the ancestors, level by level) is
replayed in numpy ON THE STRUCTURE the analysis produced -- if the structure missed a fill entry the scatter would fail to find its
row, and L L^T would not reproduce the permuted matrix."""
import numpy as np
import pytest
import scipy.sparse as sp


def _replay(A, sym):
    """dense replay of the right-looking supernodal factorisation; returns (L dense in permuted order, perm)"""
    perm, first, rptr, rows, level = sym["perm"], sym["first"], sym["rptr"], sym["rows"], sym["level"]
    n = len(perm)
    Ap = sp.csr_matrix(A)[perm][:, perm].toarray()
    nsn = len(first) - 1
    sn_of = np.zeros(n, dtype=np.int64)
    panels, idx = [], []
    for s in range(nsn):
        cols = np.arange(first[s], first[s + 1])
        sn_of[cols] = s
        R = rows[rptr[s]:rptr[s + 1]].astype(np.int64)
        assert (np.diff(R) > 0).all() and (len(R) == 0 or R[0] >= first[s + 1])
        ind = np.concatenate([cols, R])
        idx.append(ind)
        P = np.tril(Ap[np.ix_(ind, cols)].copy(), 0) if len(R) == 0 else Ap[np.ix_(ind, cols)].copy()
        P[:len(cols)] = np.tril(P[:len(cols)])
        # every nonzero of the lower triangle of these columns must sit inside the panel's row list
        colnz = np.nonzero(np.abs(np.tril(Ap)[:, cols]).sum(axis=1))[0]
        assert np.isin(colnz, ind).all()
        panels.append(P)
    for lev in range(int(level.max()) + 1):
        for s in np.nonzero(level == lev)[0]:
            nc = first[s + 1] - first[s]
            P = panels[s]
            D = P[:nc]
            D = np.tril(D) + np.tril(D, -1).T
            Lss = np.linalg.cholesky(D)
            W = np.linalg.inv(Lss)
            P[:nc] = Lss
            if P.shape[0] > nc:
                P[nc:] = P[nc:] @ W.T
                R = idx[s][nc:]
                U = P[nc:] @ P[nc:].T
                for a, ra in enumerate(R):          # scatter the lower triangle into the owners of the columns
                    t = sn_of[ra]
                    pos = {int(g): k for k, g in enumerate(idx[t])}
                    ca = ra - first[t]
                    for b in range(a, len(R)):
                        assert int(R[b]) in pos, ("missing fill", s, t, int(R[b]))
                        panels[t][pos[int(R[b])], ca] -= U[b, a]
    L = np.zeros((n, n))
    for s in range(nsn):
        cols = np.arange(first[s], first[s + 1])
        L[np.ix_(idx[s], cols)] = panels[s]
    return L, Ap


@pytest.mark.parametrize("shape,parts", [((9, 8, 7), (1, 1, 1)), ((14, 13, 6), (2, 1, 1)), ((24, 23), (1, 1))])
def test_supernodal_structure_supports_the_factorisation(ddm, shape, parts):
    from dune_ddm_amd import synth
    from dune_ddm_amd.problem import build_structured
    dec = build_structured(synth.StructuredPoisson(shape, parts), overlap=1, pou_type="distance")
    for sd in dec.subs:
        A = sp.csr_matrix(sd.A_dir)
        sym = ddm.sn_symbolic_host(A)[0]
        n = A.shape[0]
        assert sorted(sym["perm"]) == list(range(n))
        assert sym["first"][0] == 0 and sym["first"][-1] == n and (np.diff(sym["first"]) >= 1).all() and (np.diff(sym["first"]) <= 128).all()
        nsn = len(sym["first"]) - 1
        # tree: the parent owns the first row below; levels strictly increase towards the root
        sn_of = np.repeat(np.arange(nsn), np.diff(sym["first"]))
        for s in range(nsn):
            R = sym["rows"][sym["rptr"][s]:sym["rptr"][s + 1]]
            assert sym["parent"][s] == (sn_of[R[0]] if len(R) else -1)
            if len(R):
                assert sym["level"][sym["parent"][s]] > sym["level"][s]
        L, Ap = _replay(A, sym)
        assert np.abs(L @ L.T - Ap).max() <= 1e-12 * np.abs(Ap).max()
        assert sym["entries"] == sum((sym["first"][s + 1] - sym["first"][s]) * (sym["first"][s + 1] - sym["first"][s] + sym["rptr"][s + 1] - sym["rptr"][s]) for s in range(nsn))


def test_wide_separators_are_cut_into_chains(ddm):
    """a 2-D 60 x 60 grid has separators of ~60 vertices; a 3-D 16^3 grid of 256: the latter must come out as chains of <= 128 columns"""
    from dune_ddm_amd import synth
    from dune_ddm_amd.problem import build_structured
    dec = build_structured(synth.StructuredPoisson((16, 16, 16), (1, 1, 1)), overlap=1, pou_type="distance")
    A = sp.csr_matrix(dec.subs[0].A_dir)
    sym = ddm.sn_symbolic_host(A)[0]
    w = np.diff(sym["first"])
    assert w.max() == 128 and sym["levels"] >= 6 and sym["flops"] > 0
    L, Ap = _replay(A, sym)
    assert np.abs(L @ L.T - Ap).max() <= 1e-12 * np.abs(Ap).max()
