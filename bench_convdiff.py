#!/usr/bin/env python3
"""bench_convdiff.py -- the second workload `north_star` names: preconditioned restarted-GMRES iterations/s of the two-level
Schwarz + GenEO path on the convection-diffusion DG system of BASELINE.json configs[3] (examples/convectiondiffusiondg.cc).

Workload: Q1-DG (4 DoF per cell) SIPG diffusion + upwind convection on 512 x 512 cells = 1 048 576 DoF, checkerboard coefficient
1e-6 / 1, b = (1/3, 1), 4 x 2 = 8 overlapping subdomains (overlap 2), local solver per `--local-solver` (`umfpack` = the shipped
.ini's sparse direct solve: host L U, device triangular solves; or `ilu0`), GenEO on the symmetric part (nev 16), additive, GMRES(100).
A "step" is one GMRES iteration (operator apply with halo sum, preconditioner apply, modified Gram-Schmidt against the current
Krylov basis).  Same JSON contract as bench.py (which stays the driver's headline bench: the Poisson metric of BASELINE.json);
the 8 subdomains are distributed 8 / N per GPU for N = 1, 2, 4, 8.

  python bench_convdiff.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0


def log(rank, *a):
    if rank == 0:
        print("[bench_convdiff]", *a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--problem", default="dg", choices=["dg", "elasticity"], help="dg = BASELINE configs[3] (default), elasticity = configs[4]")
    ap.add_argument("--cells", type=int, default=512)
    ap.add_argument("--refine", type=int, default=1, help="elasticity: refinement levels of the 80 x 8 x 12 bar (1 -> 205 275 DoF)")
    ap.add_argument("--nev", type=int, default=16)
    ap.add_argument("--restart", type=int, default=100)
    ap.add_argument("--local-solver", default="umfpack", choices=["umfpack", "ilu0"])
    ap.add_argument("--cpu-iters", type=int, default=20, help="GMRES iterations of the CPU oracle timed for cpu_baseline (0 = skip)")
    ap.add_argument("--no-solve", action="store_true", help="skip the full solve (profiling runs)")
    ap.add_argument("--profile-counts", action="store_true", help="count every local solve of the run (event timers on from the start): the divisor of the per-kernel PMC totals")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import bench as _b
        sys.exit(_b.launch_workers(args.gpus, __file__))      # this process becomes the launcher (see bench.py)

    import numpy as np
    import torch
    import __graft_entry__ as ge
    import bench as _b
    pkg = ge.import_package()
    rank, world, local_rank, comm, backend, shared, ndev, dist = _b.init_distributed(args.gpus)
    from dune_ddm_amd import gmres_solve, synth
    from dune_ddm_amd.geneo import geneo_basis
    from dune_ddm_amd.problem import build_structured
    from dune_ddm_amd.solver import TwoLevelSchwarz

    def barrier():
        if world > 1:
            dist.barrier()

    t0 = time.perf_counter()
    C = args.cells
    assert 8 % world == 0
    if args.problem == "dg":
        grid = synth.StructuredDG2D((C, C), (4, 2))
        dec = build_structured(grid, overlap=2, neumann=True)
        cfg = dict(schwarz_type="standard", mode="additive", reduction=1e-8, tol=1e-5, nev=args.nev, local=args.local_solver)
        workload = (f"Q1-DG convection-diffusion {C}x{C} cells = {grid.nglobal} DoF, 8 overlapping subdomains (4x2, overlap 2), local solver "
                    f"'{args.local_solver}', GenEO nev {args.nev} on the symmetric part")
    else:
        # BASELINE.json configs[4] (examples/linearelasticity.cc + .ini): P1 elasticity on the 80 x 8 x 12 bar, refine levels,
        # 8 subdomains, overlap 1, restricted Schwarz, multiplicative coarse level, `cholmod` local solves, GenEO with B = A_neu
        grid = synth.StructuredElasticity(refine=args.refine, parts=8)
        dec = build_structured(grid, overlap=1, neumann=True, second_region="all")
        local = "cholmod" if args.local_solver == "umfpack" else args.local_solver
        cfg = dict(schwarz_type="restricted", mode="multiplicative", reduction=1e-6, tol=1e-6, nev=min(args.nev, 12), local=local)
        workload = (f"P1 linear elasticity on the 80x8x12 bar, refine {args.refine} = {grid.nglobal} DoF, 8 overlapping subdomains (overlap 1), local solver "
                    f"'{local}', GenEO nev {cfg['nev']} (B = A_neu), restricted Schwarz")
    t_host = time.perf_counter() - t0
    t1 = time.perf_counter()
    tl = TwoLevelSchwarz(dec, rank, world, local_rank, comm, schwarz_type=cfg["schwarz_type"], mode=cfg["mode"], coarse="none", subdomain_solver=cfg["local"])
    basis = geneo_basis(tl, nev=cfg["nev"], tol=cfg["tol"], verbose=(rank == 0 and os.environ.get("DDM_VERBOSE") == "1"))
    tl.set_coarse_basis(basis)
    tl.rebuild_combined(cfg["mode"])
    tl.ctx.sync()
    t_dev = time.perf_counter() - t1
    gi = tl.geneo_info
    log(rank, f"{grid.nglobal} DoF; host setup {t_host:.1f} s, device setup {t_dev:.1f} s (GenEO {gi['iterations']} block iterations, "
              f"{gi['setup_s'] + gi['iterate_s']:.1f} s); local solver '{cfg['local']}', engine {tl.schwarz.engine()}, K = {tl.K}")

    solve_info, gpu_hist = None, None
    solves_before = 0
    if args.profile_counts:
        tl.ctx.timing(True)
        tl.ctx.timing_reset()
    if not args.no_solve:
        res, hist, x = tl.solve(reduction=cfg["reduction"], maxit=1000, solver="restartedgmressolver", restart=args.restart)
        gpu_hist = np.asarray(hist, dtype=float)
        solve_info = {"iterations": int(res.iterations), "converged": bool(res.converged), "reduction": float(res.reduction), "solve_s": float(res.elapsed_s),
                      "reduction_target": cfg["reduction"]}
        log(rank, f"full solve: {res.iterations} GMRES iterations to {cfg['reduction']:g}, {res.elapsed_s:.3f} s")
        del x

    # ---- timed region: exactly K GMRES iterations (reduction 0: the solver runs to maxit) ------------------------------------------
    b = tl.to_device(tl.rl.b)
    x = tl.zeros(tl.rl.n_o)
    gmres_solve(tl.ctx, tl.op, tl.prec, x, b, 1e-300, args.warmup, args.restart, False)
    x.zero_()
    if args.profile_counts:
        tl.ctx.sync()
        solves_before = tl.ctx.timer("Schwarz/local solve")[1]
    tl.ctx.timing(True)
    tl.ctx.timing_reset()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r_t, _ = gmres_solve(tl.ctx, tl.op, tl.prec, x, b, 1e-300, args.steps, args.restart, False)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    tl.ctx.timing(False)
    tl.prec.check_status()
    assert int(r_t.iterations) == args.steps, (r_t.iterations, args.steps)
    local_ms, local_cnt = tl.ctx.timer("Schwarz/local solve")
    timers = {name: tl.ctx.timer(name) for name in ("Operator/apply", "Schwarz/local solve", "GalerkinPrec/apply", "CombinedPreconditioner/apply")}
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    its_per_s = args.steps / elapsed

    # ---- roofline of the dominant kernel: the local solve (two triangular sweeps over the factor) -------------------------------
    zf, n = tl.schwarz.factor_nnz(), tl.rl.n
    engine = tl.schwarz.engine()
    # CSR factors: value + column index of every entry once; supernodal panels (device engine): every panel entry once per sweep, no indices
    alg_bytes = (16.0 * zf if engine == "supernodal" else 12.0 * zf) + 40.0 * n
    avg_ms = local_ms / max(local_cnt, 1)
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
    traffic, traffic_source = None, None
    try:   # HBM bytes of ONE local solve (all its level launches) from the committed rocprofv3 --pmc passes of this command: NOT measured in this run
        src = f"profiles/r04_pmc_traffic_{args.problem}.json"
        if not os.path.exists(os.path.join(ROOT, src)):
            src = f"profiles/r03_pmc_traffic_{args.problem}.json"
        pmc = json.load(open(os.path.join(ROOT, src)))
        if pmc.get("cells") == (C if args.problem == "dg" else None) and pmc.get("refine") == (None if args.problem == "dg" else args.refine) and cfg["local"] == pmc.get("local_solver") and pmc.get("engine", "levels") == engine:
            traffic, traffic_source = pmc["local_solve_hbm_bytes_corrected"] / world, src
    except Exception:
        traffic = None
    roofline = {"bound": "hbm", "kernel": "local solve: " + ("ILU(0) triangular solve" if cfg["local"] == "ilu0" else
                                                 "device supernodal factor: one launch per bottom tree level and sweep (sn::k_sn_fwd1, k_sn_bwd1_partial, k_sn_bwd1_diag), one persistent launch for the top levels (sn::k_sn_top1)" if engine == "supernodal" else
                                                 "sparse direct factor, level-scheduled CSR kernels with supernodal blocks (k_trsv_csr_level)"),
                "engine": engine,
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": avg_ms, "launches_timed": int(local_cnt)}

    cpu = None
    if rank == 0 and world == 1 and args.cpu_iters > 0 and gpu_hist is not None:
        from oracle import apply_oracle as ao
        from tests.oracle_bridge import oracle_objects
        ao.set_threads(min(8, os.cpu_count() or 1))
        op, sp_, prec, sch, gal = oracle_objects(dec, coarse=tl.host_basis(), schwarz_type=cfg["schwarz_type"], mode=cfg["mode"],
                                                 local_solver="ilu0" if cfg["local"] == "ilu0" else "direct")
        xo = [np.zeros(sd.n_o) for sd in dec.subs]
        bo = [sd.b.copy() for sd in dec.subs]
        tc = time.perf_counter()
        it_cpu, _, ho = ao.gmres_solve(op, sp_, prec, xo, bo, 0.0, args.cpu_iters, args.restart)
        t_cpu = time.perf_counter() - tc
        ao.set_threads(1)
        ho = np.asarray(ho, dtype=float)
        m = min(len(ho), len(gpu_hist))
        dev = np.abs(gpu_hist[:m] - ho[:m]) / ho[:m]
        cpu = {"value": it_cpu / t_cpu, "unit": "iterations/s", "cores": min(8, os.cpu_count() or 1), "kind": "port",
               "sample": f"{it_cpu} GMRES iterations of the same problem (setup excluded); local solves by scipy's SuperLU in the oracle" if args.local_solver != "ilu0"
                         else f"{it_cpu} GMRES iterations of the same problem (setup excluded)",
               "parity_first_iterations": {"iterations_checked": int(m - 1), "max_rel_dev_residual_norm": float(dev.max()),
                                           "tolerance": "| ||r_k||(hip) - ||r_k||(oracle) | <= 1e-7 ||r_k|| + 1e-11 ||r_0|| (GMRES, direct local solves 1e-9: DESIGN.md section 6)",
                                           "ok": bool(np.all(np.abs(gpu_hist[:m] - ho[:m]) <= 1e-7 * ho[:m] + 1e-11 * ho[0]))}}
        log(rank, f"cpu_baseline: {it_cpu} iterations in {t_cpu:.1f} s; parity over {m - 1} iterations: max rel. dev {dev.max():.2e}")

    if rank == 0:
        out = {"metric": "preconditioned GMRES iterations/sec (two-level Schwarz + GenEO), " + ("convection-diffusion DG 1M DoF" if args.problem == "dg" else "3D linear elasticity"),
               "value": its_per_s, "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "backend": "none (single rank)" if world == 1 else ("nccl (RCCL)" if backend == "nccl" else backend), "exchange": tl.exchange,
               "rccl_comm_size": tl.ctx.rccl_size(), "ranks_share_devices": bool(shared),
               "config": {"workload": workload + f" (K = {tl.K}), {cfg['mode']}, GMRES({args.restart})",
                          "problem": args.problem, "cells": C if args.problem == "dg" else None, "refine": args.refine if args.problem == "elasticity" else None,
                          "local_solver": cfg["local"], "subdomains_per_gpu": 8 // world, "parallelism": f"dd{world}"},
               "dof_iters_per_sec": grid.nglobal * its_per_s, "solve": solve_info, "setup_s": {"host": t_host, "device": t_dev},
               "geneo": {k: gi[k] for k in ("iterations", "converged", "worst_residual", "used_direct", "setup_s", "iterate_s", "nev")},
               "roofline": roofline, "local_solves_in_run": (int(solves_before + local_cnt) if args.profile_counts else None), "phase_ms_per_iteration": {nm: (v[0] / max(v[1], 1)) for nm, v in timers.items()}, "cpu_baseline": cpu}
        print(json.dumps(out), flush=True)
    tl.ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
