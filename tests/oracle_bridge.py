"""Test-side glue: runs the CPU oracle (oracle/) on the same Decomposition the HIP path consumes.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this."""
from __future__ import annotations

import time

import numpy as np

from oracle import apply_oracle as ao


def oracle_objects(dec, schwarz_type="standard", mode="additive", coarse="pou", local_solver="ilu0", use_pou=True):
    P = dec.nsub
    ncomm = ao.Comm(P, {}, dec.novlp_all, [sd.owner_novlp for sd in dec.subs])
    ocomm = ao.Comm(P, dec.ovlp_owner, dec.ovlp_all, [sd.owner_ovlp for sd in dec.subs])
    op = ao.NonOverlappingOperator([ao.Csr(sd.A) for sd in dec.subs], ncomm)
    sp_ = ao.NonOverlappingScalarProduct(ncomm)
    Ad = [ao.Csr(sd.A_dir) for sd in dec.subs]
    pou = [sd.pou for sd in dec.subs] if use_pou else None
    factory = ao.Ilu0 if local_solver == "ilu0" else ao.DirectSolver
    sch = ao.SchwarzPreconditioner(Ad, ocomm, pou, schwarz_type, factory)
    prec = ao.CombinedPreconditioner(mode)
    prec.set_op(op)
    prec.add(sch)
    gal = None
    if coarse is not None and coarse != "none":
        if isinstance(coarse, str) and coarse == "pou":
            basis = ao.pou_coarse_space([sd.pou for sd in dec.subs])
        else:
            basis = [[np.array(v, dtype=float) for v in coarse[s]] for s in range(P)]
        for r, sd in enumerate(dec.subs):                      # zero_at_dirichlet (examples/poisson.cc:235-238)
            for v in basis[r]:
                v[sd.dirichlet_ovlp > 0] = 0.0
        gal = ao.GalerkinPreconditioner(Ad, basis, ocomm)
        prec.add(gal)
    return op, sp_, prec, sch, gal


def oracle_solve(dec, reduction=1e-10, maxit=1000, solver="cgsolver", restart=100, **kw):
    op, sp_, prec, sch, gal = oracle_objects(dec, **kw)
    x = [np.zeros(sd.n_o) for sd in dec.subs]
    b = [sd.b.copy() for sd in dec.subs]
    if solver == "bicgstabsolver":
        it, conv, hist = ao.bicgstab_solve(op, sp_, prec, x, b, reduction, maxit)
    elif solver == "restartedgmressolver":
        it, conv, hist = ao.gmres_solve(op, sp_, prec, x, b, reduction, maxit, restart)
    else:
        it, conv, hist = ao.cg_solve(op, sp_, prec, x, b, reduction, maxit)
    return it, conv, hist, x


def oracle_time_iterations(dec, iters, **kw):
    """Times ``iters`` CG iterations of the oracle (setup excluded) -- bench.py's cpu_baseline leg."""
    op, sp_, prec, sch, gal = oracle_objects(dec, **kw)
    x = [np.zeros(sd.n_o) for sd in dec.subs]
    b = [sd.b.copy() for sd in dec.subs]
    t0 = time.perf_counter()
    it, conv, hist = ao.cg_solve(op, sp_, prec, x, b, 0.0, iters)
    oracle_time_iterations.last_history = np.asarray(hist, dtype=float)   # ||r_0|| .. ||r_iters|| (full-size parity check in bench.py)
    return time.perf_counter() - t0, it
