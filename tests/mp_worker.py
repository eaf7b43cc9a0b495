"""Worker for the world_size>1 tests (launched by torch.distributed.run, backend gloo).

mode "plans": CPU only.  Every rank flattens its share of the subdomains into halo plans
              (problem.halo_plan), executes pack / exchange / unpack in numpy with exactly the
              semantics of ddm_halo_exchange, and rank 0 compares the result of the three DUNE
              interfaces (copy / add) bit for bit with the oracle's single-process communication.
mode "solve": needs a GPU (all ranks share cuda:0, exchange staged through gloo).  Full two-level
              CG solve through the C ABI; the residual history must match the oracle.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.import_package()
from dune_ddm_amd import synth  # noqa: E402
from dune_ddm_amd.problem import RankLocal, build_structured  # noqa: E402


def numpy_halo_exchange(plan, mode, v, rank, world):
    """numpy mirror of ddm_halo_exchange (include/ddm_hip.h) with a gloo all-to-all."""
    sendbuf = v[plan["send_idx"]].copy()
    sc, rc = [int(c) for c in plan["send_counts"]], [int(c) for c in plan["recv_counts"]]
    recv = np.zeros(sum(rc))
    so = np.concatenate([[0], np.cumsum(sc)])
    ro = np.concatenate([[0], np.cumsum(rc)])
    ops, keep = [], []
    for r in range(world):
        if r == rank:
            recv[ro[r]:ro[r + 1]] = sendbuf[so[r]:so[r + 1]]
            continue
        if sc[r]:
            t = torch.from_numpy(sendbuf[so[r]:so[r + 1]].copy())
            keep.append(t)
            ops.append(dist.P2POp(dist.isend, t, r))
        if rc[r]:
            t = torch.zeros(rc[r], dtype=torch.float64)
            keep.append((r, t))
            ops.append(dist.P2POp(dist.irecv, t, r))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    for item in keep:
        if isinstance(item, tuple):
            r, t = item
            recv[ro[r]:ro[r + 1]] = t.numpy()
    out = v.copy()
    for t in range(len(plan["dst_idx"])):
        i = plan["dst_idx"][t]
        s = out[i] if mode == "add" else 0.0
        for k in range(plan["dst_ptr"][t], plan["dst_ptr"][t + 1]):
            s = s + recv[plan["src_pos"][k]] if mode == "add" else recv[plan["src_pos"][k]]
        out[i] = s
    return out


def main():
    mode = sys.argv[1]
    if mode == "solve_nccl":
        # one process per GPU, RCCL: the in-library exchange (ddm_ctx_set_rccl) -- or, with DDM_EXCHANGE=callback, the callbacks
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        rank, world = dist.get_rank(), dist.get_world_size()
        from dune_ddm_amd.solver import TorchComm, TwoLevelSchwarz
        grid = synth.StructuredPoisson((15, 14, 13), (2, 2, 2))
        dec = build_structured(grid, overlap=2, pou_type="distance")
        tl = TwoLevelSchwarz(dec, rank, world, local, TorchComm(), schwarz_type="standard", mode="additive", coarse="pou")
        res, hist, x = tl.solve(reduction=1e-10, maxit=300)
        tl.prec.check_status()
        if rank == 0:
            from tests.oracle_bridge import oracle_solve
            it, conv, hist_o, xo = oracle_solve(dec, reduction=1e-10, maxit=300, coarse="pou", schwarz_type="standard", mode="additive")
            ho = np.array(hist_o)
            assert res.iterations == it and res.converged and conv, (res.iterations, it)
            assert (np.abs(hist - ho) <= 1e-8 * ho + 1e-14 * ho[0]).all()
            print("NCCL_SOLVE_OK", world, it, tl.exchange, tl.schwarz.engine())
        dist.barrier()
        dist.destroy_process_group()
        return
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    if mode == "distsetup":
        # every process holds ONE subdomain's non-overlapping data and runs the distributed setup over gloo; the
        # result is compared (on every rank, for its own subdomain) with the global-knowledge product path
        from dune_ddm_amd import setup_dist as sdist
        from dune_ddm_amd import setup_host as sh
        import scipy.sparse as sp
        P = {2: (2, 1, 1), 4: (2, 2, 1)}[world]
        grid = synth.StructuredPoisson((11, 10, 9), P)
        nov = grid.subdomains()
        overlap = 2
        ds = sdist.DistSetup(sdist.TorchExchange(), nov[rank])
        idx = ds.make_overlapping_communication(overlap)
        A_dir, dm = ds.overlapping_matrix()
        pou, bmask, _ = ds.partition_of_unity(A_dir, "distance", 0)
        ifc = ds.interfaces()
        ref = sh.make_overlapping_communication(nov, overlap, grid.nglobal)
        pairs = sh.interface_pairs(ref, grid.nglobal, "all_to_all")
        dmask = [grid.dirichlet_of(i.glob) for i in ref]
        A_ref = [grid.dirichlet_matrix(i.glob, d) for i, d in zip(ref, dmask)]
        pou_ref, bm_ref, _ = sh.partition_of_unity(ref, A_ref, pairs, grid.nglobal, "distance", 0, overlap)
        ok = np.array_equal(idx.glob, ref[rank].glob) and np.array_equal(idx.owner, ref[rank].owner) and np.array_equal(idx.ext_boundary, ref[rank].ext_boundary)
        ok = ok and np.array_equal(dm, dmask[rank]) and np.array_equal(bmask, bm_ref[rank]) and np.abs(pou - pou_ref[rank]).max() <= 1e-15
        D = (A_dir - sp.csr_matrix(A_ref[rank])).tocsr()
        ok = ok and (abs(D).max() if D.nnz else 0.0) <= 4e-16 * abs(A_ref[rank]).max()
        for q in ds.neighbours:
            ok = ok and np.array_equal(ifc["all_to_all"][q], pairs[(rank, q)][0])
        flag = torch.tensor([1 if ok else 0])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if rank == 0:
            print(("DISTSETUP_OK" if int(flag) else "DISTSETUP_FAIL") + f" {world}", flush=True)
        dist.destroy_process_group()
        return
    if mode == "plans":
        grid = synth.StructuredPoisson((11, 10, 9), (2, 2, 2))
        dec = build_structured(grid, overlap=2, pou_type="distance")
        rl = RankLocal(dec, rank, world)
        rng = np.random.default_rng(5)
        vo = [rng.standard_normal(sd.n) for sd in dec.subs]          # same on every rank
        vn = [rng.standard_normal(sd.n_o) for sd in dec.subs]
        res = {}
        for name, plan, m, vecs, cat in (("ovlp_copy", rl.plan_ovlp_copy, "copy", vo, rl.cat_ovlp),
                                         ("ovlp_add", rl.plan_ovlp_add, "add", vo, rl.cat_ovlp),
                                         ("novlp_add", rl.plan_novlp_add, "add", vn, rl.cat_novlp)):
            res[name] = numpy_halo_exchange(plan, m, cat(vecs), rank, world)
        gathered = [None] * world
        dist.all_gather_object(gathered, (rl.local, res))
        if rank == 0:
            from oracle import apply_oracle as ao
            ocomm = ao.Comm(dec.nsub, dec.ovlp_owner, dec.ovlp_all, [sd.owner_ovlp for sd in dec.subs])
            ncomm = ao.Comm(dec.nsub, {}, dec.novlp_all, [sd.owner_novlp for sd in dec.subs])
            ref = {}
            a = [v.copy() for v in vo]; ocomm.copyOwnerToAll(a); ref["ovlp_copy"] = a
            a = [v.copy() for v in vo]; ocomm.addOwnerCopyToAll(a); ref["ovlp_add"] = a
            a = [v.copy() for v in vn]; ncomm.addOwnerCopyToOwnerCopy(a); ref["novlp_add"] = a
            for local, r_ in gathered:
                for name in ref:
                    want = np.concatenate([ref[name][s] for s in local])
                    assert (r_[name] == want).all(), (name, local)
            print("PLANS_OK", world)
    elif mode == "solve_dist":
        # SURVEY 8f-1 end to end: each rank sets up ITS subdomain with neighbour exchanges only (problem.build_distributed over gloo),
        # then the two-level CG solve through the C ABI; rank 0 checks the history against the oracle on the global-knowledge setup
        from dune_ddm_amd import setup_dist as sdist
        from dune_ddm_amd.problem import build_distributed
        from dune_ddm_amd.solver import TorchComm, TwoLevelSchwarz
        grid = synth.StructuredPoisson((15, 14, 13), {2: (2, 1, 1), 4: (2, 2, 1)}[world])
        nov = grid.subdomains()
        dec = build_distributed(sdist.TorchExchange(), nov[rank], world, overlap=2, pou_type="distance", nglobal=grid.nglobal)
        comm = TorchComm()
        tl = TwoLevelSchwarz(dec, rank, world, 0, comm, schwarz_type="standard", mode="additive", coarse="pou")
        res, hist, x = tl.solve(reduction=1e-10, maxit=300)
        if rank == 0:
            from tests.oracle_bridge import oracle_solve
            full = build_structured(grid, overlap=2, pou_type="distance")
            it, conv, hist_o, xo = oracle_solve(full, reduction=1e-10, maxit=300, coarse="pou", schwarz_type="standard", mode="additive")
            ho = np.array(hist_o)
            assert res.iterations == it and res.converged and conv, (res.iterations, it)
            assert (np.abs(hist - ho) <= 1e-8 * ho + 1e-14 * ho[0]).all()
            assert np.max(np.abs(x.cpu().numpy() - xo[0])) <= 1e-8 * np.max(np.abs(xo[0]))
            print("SOLVE_DIST_OK", world, it)
        dist.barrier()
    elif mode == "solve":
        from dune_ddm_amd.solver import TorchComm, TwoLevelSchwarz
        grid = synth.StructuredPoisson((15, 14, 13), (2, 2, 2))
        dec = build_structured(grid, overlap=2, pou_type="distance")
        comm = TorchComm()
        tl = TwoLevelSchwarz(dec, rank, world, 0, comm, schwarz_type="standard", mode="additive", coarse="pou")
        res, hist, x = tl.solve(reduction=1e-10, maxit=300)
        if rank == 0:
            from tests.oracle_bridge import oracle_solve
            it, conv, hist_o, xo = oracle_solve(dec, reduction=1e-10, maxit=300, coarse="pou", schwarz_type="standard", mode="additive")
            ho = np.array(hist_o)
            assert res.iterations == it and res.converged and conv, (res.iterations, it)
            assert (np.abs(hist - ho) <= 1e-8 * ho + 1e-14 * ho[0]).all()
            want = np.concatenate([xo[s] for s in tl.rl.local])
            assert np.max(np.abs(x.cpu().numpy() - want)) <= 1e-8 * np.max(np.abs(want))
        # chunked iterations (ddm_cg_steps(k), what bench.py times): the defect norm of an iteration rides on the coarse-defect
        # all-reduce of the next one (K + 1 doubles) -- same iterates and the same final defect as one all-reduce per norm
        from dune_ddm_amd import CgIteration
        defects = []
        counts = []
        for chunks in ([6], [1] * 6):
            xx, bb = tl.zeros(tl.rl.n_o), tl.to_device(tl.rl.b)
            cg = CgIteration(tl.ctx, tl.op, tl.prec, xx, bb)
            c0 = tl.ctx.comm_counts()
            for k in chunks:
                cg.steps(k)
            defects.append(cg.defect())
            c1 = tl.ctx.comm_counts()
            counts.append([(c1[i] - c0[i]) for i in range(3)])
            cg.end()
        assert defects[0] == defects[1], defects
        assert counts[0][0] == counts[1][0] - 5 and counts[0][1] == counts[1][1] and counts[0][2] == counts[1][2], counts   # 5 launches saved, same doubles, same halos
        if rank == 0:
            print("PIGGYBACK_OK", counts[0], counts[1])
            print("SOLVE_OK", world, it)
        dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
