"""High-level driver mirroring the call order of examples/poisson.cc:198-321 on one rank:

    A_dir, pou, interfaces  ->  SchwarzPreconditioner  ->  coarse basis (POU / GenEO / given)
    -> zero_at_dirichlet -> GalerkinPreconditioner (R A R^T assembled on the device)
    -> NonOverlappingOperator -> CombinedPreconditioner -> CG.

torch is used for device memory, the stream and torch.distributed (RCCL over xGMI); all compute
goes through the C ABI of libddm_hip.so.
"""
from __future__ import annotations

import numpy as np

from . import (CombinedPreconditioner, Context, torch_context, gmres_solve, bicgstab_solve, CsrMatrix, GalerkinPreconditioner, Halo, NonOverlappingOperator,
               SchwarzPreconditioner, cg_solve, galerkin_products)
from .problem import Decomposition, RankLocal


class _DevView:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


class TorchComm:
    """Inter-rank exchange over torch.distributed.  backend 'nccl' (= RCCL over xGMI) moves device
    buffers directly; with 'gloo' (CPU rehearsal of the N>1 path) buffers are staged through the host."""

    def __init__(self):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.nranks = dist.get_rank(), dist.get_world_size()
        self.backend = dist.get_backend()
        self.halos = {}
        self._views = {}

    def view(self, ptr, n):
        key = (int(ptr), int(n))
        if key not in self._views:
            self._views[key] = self.torch.as_tensor(_DevView(ptr, n), device="cuda")
        return self._views[key]

    def register(self, halo: Halo):
        self.halos[halo.tag] = halo

    def alltoall(self, tag, sptr, rptr):
        h = self.halos[tag]
        ns, nr = sum(h.send_counts), sum(h.recv_counts)
        send = self.view(sptr, max(ns, 1))
        recv = self.view(rptr, max(nr, 1))
        staged = self.backend != "nccl"
        s_host = send.cpu() if staged else send
        r_host = self.torch.empty(max(nr, 1), dtype=self.torch.float64) if staged else recv
        ops = []
        so = ro = 0
        for r in range(self.nranks):
            sc, rc = h.send_counts[r], h.recv_counts[r]
            if r == self.rank:
                if sc:
                    r_host[ro:ro + rc].copy_(s_host[so:so + sc])
            else:
                if sc:
                    ops.append(self.dist.P2POp(self.dist.isend, s_host[so:so + sc], r))
                if rc:
                    ops.append(self.dist.P2POp(self.dist.irecv, r_host[ro:ro + rc], r))
            so += sc
            ro += rc
        if ops:
            for w in self.dist.batch_isend_irecv(ops):
                w.wait()
        if staged:
            recv[:max(nr, 1)].copy_(r_host)
        return 0

    def allreduce(self, ptr, n):
        t = self.view(ptr, n)
        if self.backend == "nccl":
            self.dist.all_reduce(t)
        else:
            h = t.cpu()
            self.dist.all_reduce(h)
            t.copy_(h)
        return 0


def pou_basis(rl: RankLocal, template_vecs=None):
    """POUCoarseSpace (coarse_spaces.hh:1175-1231): 1 * pou / ||pou||_2 per subdomain -- or, with
    ``template_vecs`` = {sub id: (t, n_s)}, the POU-scaled template vectors of the second constructor
    (:1226-1230, used by TwoLevelSchwarzSolver with 1, x, y, xy, twolevel_schwarz.hh:68-107) -- then
    zero_at_dirichlet (examples/poisson.cc:235-238,282).  Returns {sub id: (k, n_s) array}."""
    out = {}
    for sd in rl.subs:
        T = np.ones((1, sd.n)) if template_vecs is None else np.asarray(template_vecs[sd.id], dtype=np.float64)
        vecs = []
        for t in T:
            v = t * sd.pou                                          # finalize_eigenvectors (:52-61)
            v = v * (1.0 / np.sqrt(float(np.dot(v, v))))
            v[sd.dirichlet_ovlp > 0] = 0.0
            vecs.append(v)
        out[sd.id] = np.array(vecs)
    return out


class TwoLevelSchwarz:
    def __init__(self, dec: Decomposition, rank=0, nranks=1, device=0, comm: TorchComm | None = None,
                 schwarz_type="standard", mode="additive", coarse="pou", use_pou_in_schwarz=True, subdomain_solver="ilu0"):
        import torch
        self.torch = torch
        torch.cuda.set_device(device)
        self.dev = torch.device("cuda", device)
        self.dec = dec
        self.comm = comm
        self.ctx = torch_context(device)
        import os
        self.exchange = "none"
        if nranks > 1 or os.environ.get("DDM_RCCL_SELFTEST") == "1":
            # inter-rank exchange: inside the library over RCCL / xGMI when the job runs on the nccl backend (one process per
            # GPU), else the callbacks over torch.distributed (gloo rehearsal: staged through the host; DDM_EXCHANGE=callback
            # forces them on nccl too).  torch.distributed only distributes the 128-byte communicator id.
            in_library = (comm is not None and comm.backend == "nccl" and os.environ.get("DDM_EXCHANGE", "rccl") != "callback") or nranks == 1
            if in_library:
                uid = [self.ctx.rccl_unique_id() if rank == 0 else None]
                if nranks > 1:
                    comm.dist.broadcast_object_list(uid, src=0)
                self.ctx.set_rccl(rank, nranks, uid[0], self_test=(nranks == 1))
                self.exchange = "rccl"
            else:
                assert comm is not None
                self.ctx.set_comm(rank, nranks, comm.alltoall, comm.allreduce)
                self.exchange = "callback"
        import time
        self.setup_times = {}                                          # seconds per phase of the device setup (diagnostics: bench.py logs them)
        _t = [time.perf_counter()]

        def lap(name):
            now = time.perf_counter()
            self.setup_times[name] = self.setup_times.get(name, 0.0) + now - _t[0]
            _t[0] = now
        self._lap = lap
        rl = self.rl = RankLocal(dec, rank, nranks)
        lap("rank-local numbering")
        ctx = self.ctx
        if comm is not None and comm.backend != "nccl":
            # gloo rehearsal: several ranks share one GPU.  The single-launch triangular solves need all
            # their workgroups co-resident (one process per GPU); fall back to one launch per level.
            os.environ["DDM_TRSV_MODE"] = "levels"
        self.A = CsrMatrix(ctx, rl.A)
        self.A_dir = CsrMatrix(ctx, rl.A_dir)
        lap("matrix upload (A, A_dir)")
        self.h_novlp = Halo(ctx, 1, Halo.ADD, rl.plan_novlp_add)
        self.h_copy = Halo(ctx, 2, Halo.COPY, rl.plan_ovlp_copy)
        self.h_add = Halo(ctx, 3, Halo.ADD, rl.plan_ovlp_add)
        if comm is not None:
            for h in (self.h_novlp, self.h_copy, self.h_add):
                comm.register(h)
        self.op = NonOverlappingOperator(ctx, self.A, self.h_novlp, rl.owner_novlp)
        self.schwarz = SchwarzPreconditioner(ctx, self.A_dir, rl.block_ptr, rl.n_o, rl.ext_map,
                                             rl.pou if use_pou_in_schwarz else None, schwarz_type, self.h_copy, self.h_add,
                                             subdomain_solver=subdomain_solver)   # [schwarz.subdomain_solver] type (schwarz.hh:85-92)
        lap("halo plans, operator, local solver (factorisation, schedules)")
        self.galerkin = None
        self.a0 = None
        if coarse is not None and coarse != "none":
            basis = pou_basis(rl) if isinstance(coarse, str) and coarse == "pou" else coarse
            self.set_coarse_basis(basis)
        self.prec = CombinedPreconditioner(ctx, mode, self.op, self.schwarz, self.galerkin)

    def schwarz_levels(self):
        return self.schwarz.num_levels()

    def rebuild_combined(self, mode="additive"):
        self.prec = CombinedPreconditioner(self.ctx, mode, self.op, self.schwarz, self.galerkin)

    def host_basis(self):
        """coarse basis as {sub id: [vectors]} for every subdomain (single-rank runs; checker input)"""
        return {s: [v.copy() for v in self.basis_by_sub[s]] for s in self.rl.local}

    # -- device vectors
    def zeros(self, n):
        return self.torch.zeros(int(n), dtype=self.torch.float64, device=self.dev)

    def to_device(self, a):
        return self.torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64)).to(self.dev)

    # -- coarse level -------------------------------------------------------------------------
    def set_coarse_basis(self, basis_by_sub):
        """basis_by_sub: {local sub id: (k_s, n_s) array}; builds R A R^T and the Galerkin level."""
        rl, dec, torch = self.rl, self.dec, self.torch
        P = dec.nsub
        self.basis_by_sub = {s: np.asarray(basis_by_sub[s], dtype=np.float64) for s in rl.local}
        k_local = {s: int(basis_by_sub[s].shape[0]) for s in rl.local}
        k_all = self._allgather_small(k_local, P)              # MPI_Allgather of num_t (galerkin_preconditioner.hh:248)
        kmax = max(k_all)
        offset = np.concatenate([[0], np.cumsum(k_all)[:-1]]).astype(np.int64)   # offset_per_rank (:256)
        K = int(sum(k_all))
        basis = np.zeros((kmax, rl.n))
        coarse_index = np.full((len(rl.local), kmax), -1, dtype=np.int64)
        for li, s in enumerate(rl.local):
            basis[:k_all[s], rl.off[s]:rl.off[s] + dec.subs[s].n] = basis_by_sub[s]
            coarse_index[li, :k_all[s]] = offset[s] + np.arange(k_all[s])
        self.k_all, self.coarse_offset, self.K = k_all, offset, K
        import time
        t0 = time.perf_counter()
        A0 = self._build_coarse_matrix(basis, k_all, offset, K)
        t1 = time.perf_counter()
        self.a0 = A0
        a0inv = np.linalg.inv(A0)
        self.galerkin = GalerkinPreconditioner(self.ctx, rl.n, rl.n_o, rl.ext_map, rl.block_ptr, basis, coarse_index.reshape(-1),
                                               a0inv, self.h_copy, self.h_add)
        self.setup_times["coarse matrix R A R^T"] = t1 - t0
        self.setup_times["coarse inverse, basis upload"] = time.perf_counter() - t1

    def _allgather_small(self, local: dict, P):
        if self.comm is None:
            return [local[s] for s in range(P)]
        objs = [None] * self.comm.nranks
        self.comm.dist.all_gather_object(objs, local)
        merged = {}
        for o in objs:
            merged.update(o)
        return [merged[s] for s in range(P)]

    def _build_coarse_matrix(self, basis, k_all, offset, K):
        """GalerkinPreconditioner::build_solver (galerkin_preconditioner.hh:219-349) on the device:
        local x local blocks from Y = A_dir R^T, local x neighbour blocks from the neighbours' vectors
        restricted to the shared indices (CopyGatherScatterWithRank, :66-103), one neighbour slot
        at a time for all local subdomains at once (A_dir is block diagonal)."""
        rl, dec, torch, ctx = self.rl, self.dec, self.torch, self.ctx
        kmax = basis.shape[0]
        R = self.to_device(basis)                                   # (kmax, n)
        A0 = np.zeros((K, K))
        for s in rl.local:                                           # :292-295
            blk = galerkin_products(ctx, self.A_dir, R, R, rl.off[s], rl.off[s] + dec.subs[s].n)   # [i, j] = <r_i, A r_j>
            ks = k_all[s]
            A0[offset[s]:offset[s] + ks, offset[s]:offset[s] + ks] = blk[:ks, :ks]
        nbrs = {s: sorted({a for (a, b) in dec.ovlp_all if b == s}) for s in rl.local}
        nslots = max([len(v) for v in nbrs.values()] + [0])
        if self.comm is not None:
            nslots = max(self._allgather_small({s: len(nbrs[s]) for s in rl.local}, dec.nsub) + [0])
        from .problem import halo_plan
        offs = {s: rl.off.get(s, 0) for s in range(dec.nsub)}
        for t in range(nslots):                                      # :298-309, 321-327
            pairs_t = {(a, b): v for (a, b), v in dec.ovlp_all.items()
                       if self._slot_of(dec, a, b) == t}
            plan = halo_plan(pairs_t, rl.local, rl.sub2rank, offs, rl.rank, rl.nranks)
            halo = Halo(ctx, 100 + t, Halo.COPY, plan)
            if self.comm is not None:
                self.comm.register(halo)
            V = torch.zeros_like(R)         # the neighbours' vectors restricted to the shared indices, zero elsewhere (:96-101)
            for j in range(kmax):
                halo.exchange_to(R[j], V[j])
            for s in rl.local:
                if t >= len(nbrs[s]):
                    continue
                src = nbrs[s][t]
                blk = galerkin_products(ctx, self.A_dir, R, V, rl.off[s], rl.off[s] + dec.subs[s].n)
                A0[offset[s]:offset[s] + k_all[s], offset[src]:offset[src] + k_all[src]] = blk[:k_all[s], :k_all[src]]
            self.ctx.sync()
        if self.comm is not None:                                    # gatherMatrixFromRowsFlat (helpers.hh:204-339), replicated
            t_ = torch.as_tensor(A0)
            if self.comm.backend == "nccl":
                t_ = t_.to(self.dev)
            self.comm.dist.all_reduce(t_)
            A0 = t_.cpu().numpy()
        return A0

    @staticmethod
    def _slot_of(dec, src, dst):
        cache = dec.meta.setdefault("_nbr_slots", {})
        if dst not in cache:
            cache[dst] = {a: i for i, a in enumerate(sorted({a for (a, b) in dec.ovlp_all if b == dst}))}
        return cache[dst][src]

    # -- solve -------------------------------------------------------------------------------
    def solve(self, reduction=1e-10, maxit=1000, fixed_iterations=0, history=True, x0=None, b=None, solver="cgsolver", restart=100):
        """v = 0; solver->apply(v, b, res)  (examples/poisson.cc:318-319).  solver: "cgsolver" or
        "restartedgmressolver" (the [solver] type keys of examples/poisson.ini).  Returns (res, hist, x)."""
        x = self.zeros(self.rl.n_o) if x0 is None else self.to_device(x0)
        bd = self.to_device(self.rl.b if b is None else b)
        if solver == "restartedgmressolver":
            res, hist = gmres_solve(self.ctx, self.op, self.prec, x, bd, reduction, maxit, restart, history)
            return res, hist, x
        if solver == "bicgstabsolver":
            res, hist = bicgstab_solve(self.ctx, self.op, self.prec, x, bd, reduction, maxit, history)
            return res, hist, x
        if solver != "cgsolver":
            raise NotImplementedError("solver type '" + str(solver) + "' (cgsolver, restartedgmressolver and bicgstabsolver are available on the device)")
        res, hist = cg_solve(self.ctx, self.op, self.prec, x, bd, reduction, maxit, fixed_iterations, history)
        return res, hist, x


class TwoLevelSchwarzSolver:
    """Host mirror of the PDELab linear-solver backend ``TwoLevelSchwarzSolver`` (dune/ddm/twolevel_schwarz.hh:27-174), the class
    examples/convectiondiffusiondg.cc:75-78 and nonlinearpoisson.cc:151-154 hand to StationaryLinearProblemSolver / Newton:

      ctor   : non-overlapping communication + the four template vectors 1, x, y, xy with constrained DoFs zeroed (:58-81)
      apply  : (first call) overlap extension by ``overlap`` layers, overlapping matrix, PartitionOfUnity from the ``pou`` sub-tree,
               template vectors extended by copyOwnerToAll (:93-128); (every call) POUCoarseSpace(template vectors, pou) ->
               SchwarzPreconditioner (sub-tree "fine") + GalerkinPreconditioner ("coarse") in a CombinedPreconditioner whose mode
               key sits in the sub-tree itself (:131-142), solver from the "solver" sub-tree or restarted GMRES(30), maxit 1000
               (:146-160), right-hand side made consistent (:163-164), solve (:167-168).

    ``problem`` is one of the synth problems (it plays the role of the grid function space + assembled matrix); ``ptree`` the
    ``twolevelschwarz`` sub-tree as a nested dict (the keys of examples/convectiondiffusiondg.ini)."""

    def __init__(self, problem, ptree=None, coords=None, device=0):
        from .problem import build_structured
        self.ptree = dict(ptree or {})
        self.problem = problem
        overlap = int(self.ptree.get("overlap", 1))                                                    # :95
        pou = dict(self.ptree.get("pou", {}))
        self.dec = build_structured(problem, overlap=overlap, pou_type=pou.get("type", "distance"), shrink=int(pou.get("shrink", 0)))
        fine = dict(self.ptree.get("fine", {}))
        solver_type = dict(fine.get("subdomain_solver", {})).get("type")
        if solver_type is None:
            raise ValueError("You must specify the solver in the subtree fine.subdomain_solver using the key 'type'")   # schwarz.hh:89-91
        self.tl = TwoLevelSchwarz(self.dec, device=device, coarse="none", schwarz_type=fine.get("type", "restricted"),
                                  mode=self.ptree.get("mode", "additive"), subdomain_solver=solver_type)
        # template vectors 1, x, y, xy on the overlapping index sets (interpolation at the DoF positions, constrained DoFs zeroed)
        coords = coords if coords is not None else problem.dof_coords
        basis = {}
        for sd in self.tl.rl.subs:
            X = coords(sd.glob)
            T = np.stack([np.ones(sd.n), X[:, 0], X[:, 1], X[:, 0] * X[:, 1]])
            T[:, sd.dirichlet_ovlp > 0] = 0.0                                                          # set_constrained_dofs(cc, 0., v) (:75)
            V = T * sd.pou[None, :]                                                                    # POUCoarseSpace (coarse_spaces.hh:1226-1230)
            basis[sd.id] = V / np.linalg.norm(V, axis=1)[:, None]
        self.template_basis = basis
        self.tl.set_coarse_basis(basis)
        self.tl.rebuild_combined(self.ptree.get("mode", "additive"))
        self.result = None

    def apply(self, reduction, b=None):
        """solve A z = r to the given reduction (twolevel_schwarz.hh:84-169); returns (SolveResult, history, z)"""
        sol = dict(self.ptree.get("solver", {"type": "restartedgmressolver", "restart": 30, "maxit": 1000}))   # :146-153
        res, hist, z = self.tl.solve(reduction=reduction, maxit=int(sol.get("maxit", 1000)), solver=sol.get("type", "restartedgmressolver"),
                                     restart=int(sol.get("restart", 30)), b=b)
        self.tl.prec.check_status()
        self.result = res                                                                              # LinearResultStorage (:170-174)
        return res, hist, z
