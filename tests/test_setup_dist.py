"""the distributed overlap extension / overlapping-matrix build / partition of unity (dune_ddm_amd.setup_dist,
every rank holding only its own non-overlapping data) against the global-knowledge product path (setup_host) and the oracle's
literal message-passing restatement (oracle/setup_oracle.py, pinned by the reference's 9 x 9 KAT in tests/test_oracle_kat.py):
integer maps bit-exact, matrices to summation order (2e-15 of the largest entry), weights 1e-15."""
import os
import subprocess
import sys
import threading

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_threads(ddm, subs, overlap, pou_type="distance", shrink=0):
    from dune_ddm_amd import setup_dist as sd
    hub = sd.ThreadExchange.Hub(len(subs))
    out, err = [None] * len(subs), []

    def work(r):
        try:
            ds = sd.DistSetup(sd.ThreadExchange(hub, r), subs[r])
            idx = ds.make_overlapping_communication(overlap)
            A_dir, dm = ds.overlapping_matrix()
            pou, bmask, dist = ds.partition_of_unity(A_dir, pou_type, shrink)
            out[r] = dict(idx=idx, A_dir=A_dir, dm=dm, pou=pou, bmask=bmask, ifc=ds.interfaces(), nbrs=list(ds.neighbours))
        except BaseException as e:      # a failing rank must not leave the others in the barrier
            err.append((r, e))
            hub.barrier.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(len(subs))]
    [t.start() for t in th]
    [t.join() for t in th]
    if err:
        raise err[0][1]
    return out


def check_against_host(ddm, grid, overlap, pou_type="distance", shrink=0):
    from dune_ddm_amd import setup_host as sh
    from oracle import setup_oracle as so
    nov = grid.subdomains()
    ng = grid.nglobal
    got = run_threads(ddm, nov, overlap, pou_type, shrink)
    idx = sh.make_overlapping_communication(nov, overlap, ng)
    pairs_all = sh.interface_pairs(idx, ng, "all_to_all")
    pairs_own = sh.interface_pairs(idx, ng, "owner_to_all")
    dmask = [grid.dirichlet_of(i.glob) for i in idx]
    A_dir = [grid.dirichlet_matrix(i.glob, dm) for i, dm in zip(idx, dmask)]
    pou, bmask, _ = sh.partition_of_unity(idx, A_dir, pairs_all, ng, pou_type, shrink, overlap)
    oranks, oext = so.make_overlapping_communication(nov, overlap)
    oA, odm = so.overlapping_matrix(oranks, nov, [s.dirichlet for s in nov])
    for r, g in enumerate(got):
        i = g["idx"]
        assert np.array_equal(i.glob, idx[r].glob) and np.array_equal(i.owner, idx[r].owner) and np.array_equal(i.public, idx[r].public)
        assert np.array_equal(i.ext_boundary, idx[r].ext_boundary) and i.round_sizes == idx[r].round_sizes and i.n_o == idx[r].n_o
        assert np.array_equal(i.glob, np.array(oranks[r].glob)) and np.array_equal(i.ext_boundary, oext[r])
        assert np.array_equal(g["dm"], dmask[r]) and np.array_equal(g["dm"], odm[r])
        for ref in (sp.csr_matrix(A_dir[r]), sp.csr_matrix(oA[r])):
            M = g["A_dir"]
            ref.sort_indices()
            D = (M - ref).tocsr()
            assert M.shape == ref.shape and (abs(D).max() if D.nnz else 0.0) <= 2e-15 * abs(ref).max()
        if bmask is not None:
            assert np.array_equal(g["bmask"], bmask[r])
        assert np.abs(g["pou"] - pou[r]).max() <= 1e-15
        # interfaces: same neighbours, same index lists in the same order
        nb = sorted(q for (p, q) in pairs_all if p == r)
        assert g["nbrs"] == nb
        for q in nb:
            assert np.array_equal(g["ifc"]["all_to_all"][q], pairs_all[(r, q)][0])
            assert np.array_equal(g["ifc"]["all_to_all"][q], pairs_all[(q, r)][1])
            own_s = pairs_own.get((r, q), (np.zeros(0, np.int64),) * 2)[0]
            own_r = pairs_own.get((q, r), (np.zeros(0, np.int64),) * 2)[1]
            assert np.array_equal(g["ifc"]["owner_send"][q], own_s) and np.array_equal(g["ifc"]["owner_recv"][q], own_r)
    return got


@pytest.mark.parametrize("N,P,overlap", [((9, 8, 7), (2, 2, 2), 1), ((11, 10, 9), (2, 2, 2), 2), ((14, 13), (3, 2), 3), ((10, 9, 8), (3, 1, 2), 2)])
def test_distributed_setup_equals_global_knowledge_setup(ddm, N, P, overlap):
    from dune_ddm_amd import synth
    check_against_host(ddm, synth.StructuredPoisson(N, P), overlap)


def test_distributed_setup_pou_variants_and_errors(ddm):
    from dune_ddm_amd import setup_dist as sd
    from dune_ddm_amd import synth
    grid = synth.StructuredPoisson((11, 10, 9), (2, 2, 2))
    check_against_host(ddm, grid, 2, "standard")
    check_against_host(ddm, grid, 2, "distance", shrink=1)
    one = synth.StructuredPoisson((6, 5), (1, 1)).subdomains()
    ds = sd.DistSetup(sd.ThreadExchange(sd.ThreadExchange.Hub(1), 0), one[0])
    with pytest.raises(ValueError):
        ds.make_overlapping_communication(0)          # overlap_extension.hh:72-75


def test_distributed_setup_elasticity(ddm):
    """unstructured-style input (P1 elasticity on simplices, 3 DoFs per node, slab partition)"""
    from dune_ddm_amd import synth
    check_against_host(ddm, synth.StructuredElasticity(cells=(8, 2, 3), parts=4), 1)


@pytest.mark.parametrize("nproc", [2, 4])
def test_distributed_setup_over_gloo(nproc):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(29571 + nproc), os.path.join(ROOT, "tests", "mp_worker.py"), "distsetup"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and f"DISTSETUP_OK {nproc}" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


def test_rank_local_inputs_from_distributed_setup_are_identical(ddm):
    """problem.build_distributed (one rank, neighbour exchanges only) hands RankLocal -- the flattening every solver object is
    built from -- the same arrays as the global-knowledge build_structured: matrices, partition of unity, ext_map and the three
    halo plans."""
    from dune_ddm_amd import setup_dist as sd
    from dune_ddm_amd import synth
    from dune_ddm_amd.problem import RankLocal, build_distributed, build_structured
    grid = synth.StructuredPoisson((11, 10, 9), (2, 2, 2))
    nov = grid.subdomains()
    P = len(nov)
    full = build_structured(grid, overlap=2, pou_type="distance")
    hub = sd.ThreadExchange.Hub(P)
    parts, err = [None] * P, []

    def work(r):
        try:
            parts[r] = build_distributed(sd.ThreadExchange(hub, r), nov[r], P, overlap=2, pou_type="distance", nglobal=grid.nglobal)
        except BaseException as e:
            err.append(e)
            hub.barrier.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(P)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not err, err
    for r in range(P):
        a, b = RankLocal(parts[r], r, P), RankLocal(full, r, P)
        assert a.n == b.n and a.n_o == b.n_o and np.array_equal(a.ext_map, b.ext_map) and np.array_equal(a.block_ptr, b.block_ptr)
        assert np.array_equal(a.owner_novlp, b.owner_novlp) and np.array_equal(a.dirichlet_ovlp, b.dirichlet_ovlp) and np.array_equal(a.b, b.b)
        assert np.abs(a.pou - b.pou).max() <= 1e-15
        D = (a.A_dir - b.A_dir).tocsr()
        assert (abs(D).max() if D.nnz else 0.0) <= 2e-15 * abs(b.A_dir).max() and (a.A != b.A).nnz == 0
        for name in ("plan_novlp_add", "plan_ovlp_copy", "plan_ovlp_add"):
            pa, pb = getattr(a, name), getattr(b, name)
            assert pa.keys() == pb.keys()
            for k in pa:
                assert np.array_equal(pa[k], pb[k]), (r, name, k)
