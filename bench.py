#!/usr/bin/env python3
"""bench.py -- preconditioned CG iterations/s of the two-level additive Schwarz + coarse-space path.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on; it fits one GPU):
3-D Q1 Poisson on 216^3 = 10 077 696 DoF, 2x2x2 = 8 overlapping subdomains (overlap 2, distance
partition of unity), ILU(0) subdomain solves, coarse space of 8 x k vectors, additive combination, CG.
The 8 subdomains are distributed 8/N per GPU, so the SAME problem (same iteration count) runs at
N = 1, 2, 4, 8  ->  strong scaling.  A "step" is one CG iteration (operator apply with halo sum,
preconditioner apply, 2 dots + 1 norm, vector updates).  Inputs are resident in HBM before the
timed region; setup (assembly, ILU(0), coarse basis, R A R^T) is reported separately.

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)
PIPE_UNBLOCKED_STEP_US = 1.05   # unblocked in-task step of k_trsv_pipe (measured)


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def launch_workers(n, script=None):
    """Parent of `python bench.py --gpus N`: N worker processes via torch.distributed.run (children; no exec, no GPU call here)."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "1"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(script or __file__)] + sys.argv[1:]
    print("[bench] launching", " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env, cwd=ROOT)


def init_distributed(gpus):
    """Worker side of `--gpus N`: (rank, world, device index, TorchComm | None, backend, ranks_share_devices, visible devices, dist).
    One process per GPU over RCCL.  With fewer devices than ranks (rehearsal of the N > 1 path on a one-GPU box) the ranks share
    devices: RCCL refuses two ranks on one device, so the exchange is staged through gloo and the local solves use one launch per
    level (the single-launch engines need the whole GPU); the JSON line says so ("backend", "ranks_share_devices")."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != gpus:
        raise SystemExit(f"--gpus {gpus} but WORLD_SIZE={world}")
    import torch
    import __graft_entry__ as ge
    pkg = ge.import_package()
    pkg.load_library()                      # fails loudly if the HIP extension is missing
    ndev = torch.cuda.device_count()        # (does not initialise the GPU)
    assert ndev > 0 and torch.cuda.is_available(), "the benchmark needs a HIP device: the hot path has no CPU fallback"
    shared = ndev < world
    backend = os.environ.get("DDM_BACKEND", "gloo" if shared else "nccl")
    device = local_rank % ndev
    torch.cuda.set_device(device)
    comm, dist = None, None
    if world > 1:
        import torch.distributed as dist
        from dune_ddm_amd.solver import TorchComm
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend=backend)
        comm = TorchComm()
    return rank, world, device, comm, backend, shared, ndev, dist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--grid", type=int, default=216, help="nodes per axis (216 -> 10M DoF)")
    ap.add_argument("--overlap", type=int, default=2)
    ap.add_argument("--parts", type=int, default=2, help="subdomains per axis (2 -> 8 subdomains; 1 = BASELINE config 2: one subdomain, ILU(0)-CG)")
    ap.add_argument("--coarse", default="auto", choices=["auto", "geneo", "pou", "none"])
    ap.add_argument("--nev", type=int, default=20)
    ap.add_argument("--local-solver", default="ilu0", choices=["ilu0", "cholmod"],
                    help="[schwarz.subdomain_solver] type: ilu0 (the benchmark's configuration) or cholmod (the reference's shipped .ini: sparse Cholesky, "
                         "factorised on the device from 5e11 multiply-adds on)")
    ap.add_argument("--geneo-preconditioner", default="auto", choices=["auto", "ilu0", "cholesky"],
                    help="preconditioner of the GenEO block eigensolver: auto = sparse Cholesky (device engine) when it fits the flop / memory limits, else ILU(0)")
    ap.add_argument("--no-solve", action="store_true", help="skip the full solve to 1e-10 (iteration count / residual check)")
    ap.add_argument("--cpu-iters", type=int, default=60, help="CG iterations of the CPU oracle timed for cpu_baseline (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=8)
    ap.add_argument("--no-order-leg", action="store_true", help="skip the second oracle run (other summation order) of the cpu leg")
    ap.add_argument("--no-geneo-check", action="store_true", help="skip the host (scipy) residual check of the device GenEO eigenpairs")
    ap.add_argument("--emulate-rank-of", type=int, default=0, metavar="N",
                    help="time the RANK-LOCAL workload of an N-GPU run on this one GPU: the first 8 / N subdomains of the decomposition, every phase, "
                         "exchanges with the absent ranks cut (problem.restrict_decomposition); no solve, no CPU leg; not a benchmark of the metric")
    ap.add_argument("--no-secondary", action="store_true", help="skip the second workload (configs[3], bench_convdiff.py --problem dg) appended as `secondary`")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as the driver may call it: this process becomes the launcher.  It starts the N workers (one
        # process per GPU) through torch.distributed.run BEFORE anything here touches the GPU, passes their output through (rank 0
        # prints the one JSON line) and exits with their return code.
        sys.exit(launch_workers(args.gpus))

    import numpy as np
    import torch
    import __graft_entry__ as ge
    pkg = ge.import_package()
    rank, world, local_rank, comm, backend, shared, ndev, dist = init_distributed(args.gpus)
    from dune_ddm_amd import CgIteration, synth
    from dune_ddm_amd.problem import build_structured
    from dune_ddm_amd.solver import TwoLevelSchwarz

    def barrier():
        if world > 1:
            dist.barrier()

    # ---- setup (host): what DUNE/PDELab + dune-ddm's L3 hand to the hot path -------------------
    t_setup0 = time.perf_counter()
    G = args.grid
    P = args.parts
    assert (P ** 3) % world == 0, "number of subdomains must be a multiple of the number of GPUs"
    grid = synth.StructuredPoisson((G, G, G), (P, P, P))
    coarse = args.coarse
    if coarse == "auto":
        coarse = "geneo" if hasattr(pkg, "GENEO_AVAILABLE") and pkg.GENEO_AVAILABLE else "pou"
    if P == 1:
        coarse = "none"     # a single subdomain has no overlap region: ILU(0)-preconditioned CG
    dec = build_structured(grid, overlap=args.overlap, pou_type="distance", shrink=0, neumann=(coarse == "geneo"))
    emulated = None
    if args.emulate_rank_of:
        from dune_ddm_amd.problem import restrict_decomposition
        assert world == 1 and dec.nsub % args.emulate_rank_of == 0
        keep = list(range(dec.nsub // args.emulate_rank_of))
        dec = restrict_decomposition(dec, keep)
        emulated = {"of_n_gpus": args.emulate_rank_of, "subdomains_on_this_rank": len(keep),
                    "what": "rank 0's share of the decomposition on one GPU; halo pairs and reductions towards the other ranks are cut, everything else (local "
                            "solves, operator, restriction / prolongation, GenEO setup of these subdomains) is the rank's real work"}
        args.no_solve, args.cpu_iters, args.no_secondary, args.no_geneo_check = True, 0, True, True
    t_host = time.perf_counter() - t_setup0
    geneo_check = None
    log(rank, f"host setup (assembly, overlap extension, POU): {t_host:.1f} s")
    t1 = time.perf_counter()
    if coarse == "geneo":
        from dune_ddm_amd.geneo import geneo_basis
        tl = TwoLevelSchwarz(dec, rank, world, local_rank, comm, schwarz_type="standard", mode="additive", coarse="none", subdomain_solver=args.local_solver)
        basis = geneo_basis(tl, nev=args.nev, verbose=(rank == 0 and os.environ.get("DDM_VERBOSE") == "1"), preconditioner=args.geneo_preconditioner)
        gi = tl.geneo_info      # geneo_basis raises if the eigensolver did not converge
        log(rank, f"GenEO: {gi['iterations']} block iterations, converged={gi['converged']} (worst residual {gi['worst_residual']:.2e}), "
                  f"preconditioner {'sparse Cholesky' if gi['used_direct'] else 'ILU(0)'}, setup {gi['setup_s']:.1f} s + iterations {gi['iterate_s']:.1f} s, "
                  f"lambda range of subdomain {tl.rl.local[0]}: {gi['eigenvalues'][tl.rl.local[0]][[0, -1]]}")
        if rank == 0 and world == 1 and not args.no_geneo_check:
            # independent host check (scipy only) of ALL device-built eigenpairs at the benchmark's size: the basis the oracle leg
            # below is handed is pinned here, not assumed (dune_ddm_amd.geneo.host_eigenpair_residuals)
            from concurrent.futures import ThreadPoolExecutor
            from dune_ddm_amd.geneo import host_eigenpair_residuals
            t_chk = time.perf_counter()
            with ThreadPoolExecutor(min(8, os.cpu_count() or 1)) as ex:
                chk = list(ex.map(lambda sd: host_eigenpair_residuals(sd, basis[sd.id], gi["eigenvalues"][sd.id]), tl.rl.subs))
            geneo_check = {"pairs": int(sum(len(c[0]) for c in chk)), "worst_residual": float(max(c[0].max() for c in chk)),
                           "worst_rayleigh_quotient_mismatch": float(max(c[1].max() for c in chk)), "seconds": time.perf_counter() - t_chk,
                           "what": "host (scipy) check of every device eigenpair: ||A_neu x - lambda D B_neu D x||_2 / ||lambda D B_neu D x||_2, x recovered from the finalised vector"}
            log(rank, f"GenEO host check: {geneo_check['pairs']} pairs, worst residual {geneo_check['worst_residual']:.2e}, "
                      f"worst Rayleigh-quotient mismatch {geneo_check['worst_rayleigh_quotient_mismatch']:.2e} ({geneo_check['seconds']:.1f} s)")
            assert geneo_check["worst_residual"] < 1e-4, "device GenEO eigenpairs fail the host residual check"
        tl.set_coarse_basis(basis)
        tl.rebuild_combined("additive")
    else:
        tl = TwoLevelSchwarz(dec, rank, world, local_rank, comm, schwarz_type="standard", mode="additive", coarse=coarse, subdomain_solver=args.local_solver)
    tl.schwarz.wait_setup()     # the pipe schedule is built in the background (beside the GenEO iterations): its end belongs to the setup
    tl.ctx.sync()
    t_dev = time.perf_counter() - t1 - (geneo_check["seconds"] if geneo_check else 0.0)
    log(rank, "device setup phases (s): " + ", ".join(f"{k} {v:.2f}" for k, v in tl.setup_times.items()))
    log(rank, f"device setup (upload, ILU(0), level schedule, coarse space '{coarse}', R A R^T): {t_dev:.1f} s; "
              f"ILU levels L/U = {tl.schwarz_levels()}")

    # ---- full solve: iteration count and final residual (validity of the configuration) --------
    solve_info = None
    gpu_hist = None
    if not args.no_solve:
        res, hist, x = tl.solve(reduction=1e-10, maxit=1000, history=True)
        gpu_hist = np.asarray(hist, dtype=float)
        solve_info = {"iterations": int(res.iterations), "converged": bool(res.converged), "reduction": float(res.reduction),
                      "solve_s": float(res.elapsed_s)}
        log(rank, f"full solve: {res.iterations} iterations, ||r||/||r0|| = {res.reduction:.3e}, {res.elapsed_s:.3f} s")
        del x

    # ---- timed region: exactly K CG iterations ---------------------------------------------------
    x = tl.zeros(tl.rl.n_o)
    b = tl.to_device(tl.rl.b)
    cg = CgIteration(tl.ctx, tl.op, tl.prec, x, b)
    cg.steps(args.warmup)
    tl.ctx.timing(True)
    tl.ctx.timing_reset()
    barrier()
    torch.cuda.synchronize()
    cc0 = tl.ctx.comm_counts()
    t0 = time.perf_counter()
    cg.steps(args.steps)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    cc1 = tl.ctx.comm_counts()
    collectives = {"allreduce_launches": (cc1[0] - cc0[0]) / args.steps, "allreduce_doubles": (cc1[1] - cc0[1]) / args.steps,
                   "halo_send_recv_groups": (cc1[2] - cc0[2]) / args.steps,
                   "what": "RCCL launches per CG iteration as a multi-GPU run issues them (counted in the library, also at N = 1 where nothing is sent): "
                           "<p,q>, <r,z> and the coarse defect with the previous iteration's defect norm riding on it; three grouped halo exchanges"}
    tl.ctx.timing(False)
    deff = cg.defect()
    tl.prec.check_status()      # a timed-out single-launch local solve would invalidate the timing (raises)
    local_ms, local_cnt = tl.ctx.timer("Schwarz/local solve")
    timers = {name: tl.ctx.timer(name) for name in ("Operator/apply", "Schwarz/get defect", "Schwarz/local solve",
                                                    "Schwarz/add solution", "GalerkinPrec/apply", "CombinedPreconditioner/apply")}
    cg.end()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert np.isfinite(deff), "defect became non-finite inside the timed region"

    its_per_s = args.steps / elapsed
    ndof = grid.nglobal
    # ---- roofline of the dominant kernel: the level-scheduled ILU(0) triangular solve -----------
    # unit = one local solve x = (LU)^-1 d over the rank's subdomains (one HIP-graph launch of the
    # level kernels); algorithmic bytes = 12 B per stored factor entry (f64 value + int32 column)
    # + 40 B per row (d read, x written, x read + written by the backward sweep, inverse pivot).
    z, n = tl.A_dir.nnz, tl.rl.n
    alg_bytes = 12.0 * z + 40.0 * n
    if args.local_solver != "ilu0":      # direct factor: its stored entries once per sweep (8 B each in the device engine's panels)
        alg_bytes = 16.0 * tl.schwarz.factor_nnz() + 40.0 * n
    roofline = None
    if local_cnt > 0:
        avg_ms = local_ms / local_cnt
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        engine = tl.schwarz.engine() if args.local_solver == "ilu0" else "direct"
        kname = {"levels": "k_trsv_lower_level + k_trsv_upper_level + k_trsv_small_levels (one launch per level, HIP graph)",
                 "xcd2": "k_trsv_xcd2", "direct": "sparse direct factor (sn_chol.hpp panel solves or CSR level solves)",
                 "box": "k_box_sweep<lower> + k_box_sweep<upper> (structured boxes of the subdomains) + k_trsv_pipe on the overlap shell rows (nested factor) + "
                        "k_box_shell_rhs / k_box_products / k_box_shell_out"}.get(engine, "k_trsv_pipe (+ k_pipe_permute_out)")
        traffic, traffic_source = None, None
        try:   # HBM bytes per launch: NOT measured in this run -- read from the committed rocprofv3 --pmc passes of this command
            #    (profiles/, separate FETCH_SIZE / WRITE_SIZE runs, gfx950-corrected: 2 x FETCH_SIZE + WRITE_SIZE); the file is named in the line
            traffic_source = "profiles/r04_pmc_traffic_grid216_geneo.json"
            if not os.path.exists(os.path.join(ROOT, traffic_source)):
                traffic_source = "profiles/r03_pmc_traffic_grid216_geneo.json"
            pmc = json.load(open(os.path.join(ROOT, traffic_source)))
            if G == 216 and P == 2 and engine in pmc.get("engine_kernels", {}):
                traffic = pmc["kernels"][pmc["engine_kernels"][engine]]["hbm_bytes_per_dispatch_corrected"] / world
        except Exception:
            traffic = None
        roofline = {"bound": "hbm", "kernel": f"{'ILU(0)' if args.local_solver == 'ilu0' else args.local_solver} triangular solve: {kname}",
                    "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": traffic, "traffic_source": traffic_source if traffic is not None else None, "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": avg_ms,
                    "launches_timed": int(local_cnt)}
        if engine == "pipe":
            # the engine's own floor: a task advances one dependency level per in-task step, both sweeps run back to back;
            # the step is the UNBLOCKED step of the stamped trace at this size (p10 over 54 000 steps of tools/pipe_trace.py
            # joined with the schedule by tools/pipe_hops.py) -- a measured constant, not timed in this run
            lev = tl.schwarz.num_levels()
            roofline["latency_floor_ms"] = (lev[0] + lev[1]) * PIPE_UNBLOCKED_STEP_US * 1e-3
            roofline["latency_floor"] = {"levels_lower": lev[0], "levels_upper": lev[1], "unblocked_step_us": PIPE_UNBLOCKED_STEP_US,
                                         "step_source": "stamped trace of round 4 (DESIGN.md section 3)", "frac_of_floor": (lev[0] + lev[1]) * PIPE_UNBLOCKED_STEP_US * 1e-3 / avg_ms}
    # whole-iteration algorithmic traffic: 12(z_o+z) + 16 k n + 56 n + 170 n_o
    k = 0 if tl.galerkin is None else max(tl.k_all)
    it_bytes = 12.0 * (tl.A.nnz + z) + 16.0 * k * n + 56.0 * n + 170.0 * tl.rl.n_o
    iteration = {"algorithmic_bytes": it_bytes, "achieved_GBs": it_bytes / (elapsed / args.steps) / 1e9,
                 "frac_of_hbm_peak": it_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                 "phase_ms_per_iteration": {nm: (v[0] / max(v[1], 1)) for nm, v in timers.items()}}

    # ---- CPU baseline: the oracle (port of the reference's CPU path) on the host cores ----------
    cpu = None
    if rank == 0 and world == 1 and args.cpu_iters > 0 and args.local_solver == "ilu0":   # (the oracle's exact local solver is SuperLU: minutes at these sizes)
        from oracle import apply_oracle as ao
        from tests.oracle_bridge import oracle_time_iterations
        threads = max(1, min(args.cpu_threads, os.cpu_count() or 1, dec.nsub))
        ao.set_threads(threads)
        basis_o = coarse if coarse in ("pou", "none") else tl.host_basis()
        t_cpu, it_cpu = oracle_time_iterations(dec, args.cpu_iters, coarse=basis_o, schwarz_type="standard", mode="additive")
        ho_other = None
        if not args.no_order_leg:
            # the same oracle iterations once more with nothing changed but the ORDER of the additions inside the global dot products
            # (pairwise instead of index by index; oracle/kernels.c orc_masked_dot_order): how far two correct FP64 CG runs drift
            # apart -- the yardstick for the HIP-vs-oracle deviation of the late iterations (tests/test_oracle_order_sensitivity.py)
            ho_ref = np.array(oracle_time_iterations.last_history)
            ao.set_dot_order(2)
            try:
                oracle_time_iterations(dec, args.cpu_iters, coarse=basis_o, schwarz_type="standard", mode="additive")
                ho_other = np.array(oracle_time_iterations.last_history)
            finally:
                ao.set_dot_order(0)
            oracle_time_iterations.last_history = ho_ref
        ao.set_threads(1)
        cpu = {"value": it_cpu / t_cpu, "unit": "iterations/s", "cores": threads, "kind": "port",
               "sample": f"{it_cpu} CG iterations of the same {G}^3 / {P ** 3}-subdomain problem (setup excluded), one host thread per subdomain"}
        log(rank, f"cpu_baseline: {it_cpu} iterations in {t_cpu:.1f} s on {threads} threads")
        # full-size parity: the oracle's residual norms of these iterations against the HIP solve above (same problem, same start)
        ho = getattr(oracle_time_iterations, "last_history", None)
        if gpu_hist is not None and ho is not None:
            ho_all = ho
            ho = ho[:min(len(ho), len(gpu_hist))]          # common prefix (the device solve stops at the reduction)
            dev = np.abs(gpu_hist[:len(ho)] - ho) / ho
            kcheck = min(30, len(ho) - 1)
            cpu["parity_first_iterations"] = {"iterations_checked": int(kcheck), "max_rel_dev_residual_norm": float(dev[:kcheck + 1].max()),
                                              "tolerance": "checked here: | ||r_k||(hip) - ||r_k||(oracle) | <= 1e-8 * ||r_k|| for k <= 30 of the iterations the oracle ran; the oracle "
                                                           "gets the device-built GenEO basis; NOT checked here: the iteration count to 1e-10 and the late iterations, where CG "
                                                           "amplifies rounding-level differences (full-length comparison: tests/test_gpu_fullsize.py at 96^3; DESIGN.md section 6)",
                                              "ok": bool(np.all(dev[:kcheck + 1] <= 1e-8)),
                                              "rel_dev_at": {str(k): float(dev[k]) for k in (1, 10, 20, 30, 40, 50, 60) if k < len(ho)}}
            if ho_other is not None:
                mo = min(len(ho), len(ho_other))
                dev_oo = np.abs(ho_other[:mo] - ho[:mo]) / ho[:mo]
                env_oo = np.maximum.accumulate(dev_oo)
                ratio = dev[:mo] / np.maximum(1e-8, env_oo)
                cpu["parity_first_iterations"]["oracle_vs_oracle_other_summation_order"] = {
                    "what": "relative deviation of ||r_k|| between two ORACLE runs that differ only in the order of the additions inside the global dots "
                            "(running maximum): the drift any two correct FP64 CG implementations show; hip_over_envelope = max_k dev_hip[k] / max(1e-8, envelope[k])",
                    "envelope_at": {str(k): float(env_oo[k]) for k in (1, 10, 20, 30, 40, 50, 60) if k < mo},
                    "hip_over_envelope": float(ratio.max()), "hip_within_30x_envelope": bool((dev[:mo] <= np.maximum(1e-8, 30.0 * env_oo)).all())}
            # with --cpu-iters beyond the iteration count (e.g. 320) the oracle history reaches the reduction as well: full-length
            # comparison at BASELINE size in the form tests/test_gpu_fullsize.py asserts at 96^3
            red = float(solve_info["reduction_target"]) if solve_info and "reduction_target" in solve_info else 1e-10
            hit = np.nonzero(ho_all <= red * ho_all[0])[0]
            if len(hit):
                early = ho >= 2e-3 * ho[0]
                cpu["parity_full_length"] = {"oracle_iterations": int(hit[0]), "hip_iterations": None if not solve_info else int(solve_info["iterations"]),
                                             "max_rel_dev_while_rk_ge_2e-3_r0": float(dev[early].max()), "max_rel_dev_common_prefix": float(dev.max()),
                                             "oracle_final_reduction": float(ho_all[int(hit[0])] / ho_all[0])}
                # the two counts may differ by the iterate at which the smaller count stops: there one run is just below the reduction,
                # the other just above (the only exception tests/test_gpu_fullsize.py allows at 96^3)
                kl = min(int(hit[0]), len(gpu_hist) - 1)
                cpu["parity_full_length"]["at_last_common_iterate"] = {
                    "k": kl, "hip_reduction": float(gpu_hist[kl] / gpu_hist[0]), "oracle_reduction": float(ho_all[kl] / ho_all[0]), "target": red,
                    "straddles_target": bool((gpu_hist[kl] / gpu_hist[0] <= red) != (ho_all[kl] / ho_all[0] <= red))}
            prof = ", ".join(f"k={k}: {dev[k]:.1e} (r_k/r_0 {ho[k] / ho[0]:.1e})" for k in (1, 5, 10, 20, 30, 40, 50, len(ho) - 1) if k < len(ho))
            log(rank, f"full-size parity vs oracle over {len(ho) - 1} iterations: max rel. deviation of ||r_k|| = {dev.max():.2e}; {prof}")
            if tl.galerkin is not None and getattr(tl, "a0", None) is not None:
                log(rank, f"coarse matrix: K = {tl.K}, cond = {np.linalg.cond(tl.a0):.3e}")

    if rank == 0:
        out = {
            "metric": "preconditioned CG iterations/sec (two-level additive Schwarz), 3D Poisson 10M DoF",
            "value": its_per_s, "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "backend": "none (single rank)" if world == 1 else ("nccl (RCCL)" if backend == "nccl" else backend),
            "exchange": tl.exchange, "rccl_comm_size": tl.ctx.rccl_size(), "ranks_share_devices": bool(shared), "visible_devices": ndev,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"3D Q1 Poisson {G}^3 = {ndof} DoF, {P ** 3} overlapping subdomains ({P}x{P}x{P}, overlap {args.overlap}), "
                                   f"{'ILU(0)' if args.local_solver == 'ilu0' else args.local_solver} subdomain solves, coarse space '{coarse}' (K = {0 if tl.galerkin is None else tl.K}), additive, CG",
                       "subdomains_per_gpu": (P ** 3) // world, "parallelism": f"dd{world}"},
            "dof_iters_per_sec": ndof * its_per_s,
            "solve": solve_info,
            "setup_s": {"host": t_host, "device": t_dev},
            "geneo": None if getattr(tl, "geneo_info", None) is None else dict({k: tl.geneo_info[k] for k in ("iterations", "converged", "worst_residual", "used_direct", "setup_s", "iterate_s", "nev")}, host_check=geneo_check),
            "roofline": roofline, "iteration_traffic": iteration, "collectives_per_iteration": collectives, "cpu_baseline": cpu,
        }
        if emulated:
            out["metric"] = "RANK-LOCAL workload of an N-GPU run timed on one GPU (not the benchmark metric): CG iterations/s of the rank's share"
            out["emulated_rank_local"] = emulated
            out["phase_ms_per_iteration"] = {nm: (v[0] / max(v[1], 1)) for nm, v in timers.items()}
            out["setup_phases_s"] = {k: float(v) for k, v in tl.setup_times.items()}
        # Second workload of `north_star` where the driver sees it: BASELINE configs[3] (Q1-DG convection-diffusion 512^2, GMRES, GenEO,
        # `umfpack`-type local solves) run as a CHILD process with the same contract (bench_convdiff.py) once this process has
        # released the GPU memory of the headline problem; its whole JSON line is embedded.  N = 1 only; --no-secondary skips it.
        out["secondary"] = None
        if world == 1 and not args.no_secondary and args.grid == 216:
            import subprocess
            try:
                tl.ctx.close()
                del tl
                torch.cuda.empty_cache()
                t_sec = time.perf_counter()
                p = subprocess.run([sys.executable, os.path.join(ROOT, "bench_convdiff.py"), "--problem", "dg", "--steps", str(args.steps), "--warmup", str(args.warmup)],
                                   capture_output=True, text=True, timeout=600)
                lines = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")]
                if p.returncode == 0 and lines:
                    out["secondary"] = json.loads(lines[-1])
                    out["secondary"]["wall_s"] = time.perf_counter() - t_sec
                    log(rank, f"secondary workload (configs[3], DG convection-diffusion): {out['secondary']['value']:.1f} GMRES it/s, local solve {out['secondary']['roofline']['avg_launch_ms']:.2f} ms")
                else:
                    out["secondary"] = {"error": (p.stderr or p.stdout)[-800:], "returncode": p.returncode}
            except Exception as e:   # the headline number above stands on its own
                out["secondary"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
