"""Condenses the rocprofv3 passes of tools/gpu_prof_r0N_workloads.sh for ONE workload of bench_convdiff.py into
profiles/r0N_pmc_traffic_<problem>.json: HBM bytes of one LOCAL SOLVE = sum over the single-right-hand-side triangular-solve
kernels (all level launches of both sweeps) of 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction, MI355X_MICROARCH.md), divided by the
number of local solves the profiled run performed (bench_convdiff.py --profile-counts).
usage: python tools/make_pmc_json_workload.py <problem> <dir with stats/ pmc1/ pmc2/ bench.json> <out.json>"""
import collections
import csv
import glob
import json
import sys

problem, d, out = sys.argv[1:4]
bench = json.loads([ln for ln in open(d + "/bench.json") if ln.startswith("{")][-1])
nsolves = bench["local_solves_in_run"]


def is_solve_kernel(name):
    n = name.replace("void ", "")
    if bench["roofline"].get("engine") == "supernodal":   # device engine: the single-vector kernels of sn_chol.hpp + the two permutations around them
        return (n.startswith("sn::k_sn_fwd1") or n.startswith("sn::k_sn_bwd1_") or n.startswith("sn::k_sn_top") or n.startswith("ddm::k_perm_gather")
                or n.startswith("ddm::k_perm_scatter"))   # (round 4: + the persistent kernels of the top levels, sn_solve1.hpp)
    return (n.startswith("ddm::k_trsv_") and "multi" not in n) or n.startswith("ddm::k_pipe_permute") or n.startswith("ddm::k_w_permute")


tot = {}
per_kernel = collections.defaultdict(lambda: {"FETCH_SIZE_KiB": 0.0, "WRITE_SIZE_KiB": 0.0, "dispatches": 0})
for tag, sub in (("FETCH_SIZE", "pmc1"), ("WRITE_SIZE", "pmc2")):
    f = glob.glob(f"{d}/{sub}/*counter_collection.csv")[0]
    s = 0.0
    for row in csv.DictReader(open(f)):
        if row.get("Counter_Name") != tag or not is_solve_kernel(row["Kernel_Name"]):
            continue
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        v = float(row["Counter_Value"])
        s += v
        per_kernel[k][tag + "_KiB"] += v
        if tag == "FETCH_SIZE":
            per_kernel[k]["dispatches"] += 1
    tot[tag] = s
stats = {}
f = glob.glob(f"{d}/stats/*kernel_stats.csv")
if f:
    for row in csv.DictReader(open(f[0])):
        if is_solve_kernel(row["Name"]):
            stats[row["Name"].split("(")[0].replace("void ", "")] = {"calls": int(row["Calls"]), "total_ms": float(row["TotalDurationNs"]) / 1e6, "average_us": float(row["AverageNs"]) / 1e3}
doc = {"note": "rocprofv3 --pmc passes (separate runs, --kernel-trace only) of `python3 bench_convdiff.py --problem %s --steps 10 --warmup 3 --cpu-iters 0 --no-solve "
               "--profile-counts`; FETCH_SIZE / WRITE_SIZE in KiB as reported; gfx950: read bytes = 2 x FETCH_SIZE" % problem,
       "problem": problem, "cells": bench["config"].get("cells"), "refine": bench["config"].get("refine"), "local_solver": bench["config"].get("local_solver"), "engine": bench["roofline"].get("engine", "levels"),
       "local_solves_in_run": nsolves, "solve_kernels": {k: dict(v, **stats.get(k, {})) for k, v in per_kernel.items()},
       "local_solve_hbm_bytes_corrected": (2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0 / nsolves,
       "local_solve_kernel_ms": sum(v["total_ms"] for v in stats.values()) / nsolves if stats else None,
       "algorithmic_bytes_per_local_solve": bench["roofline"]["algorithmic_bytes_per_launch"],
       "bench_under_rocprof": {k: bench[k] for k in ("value", "ms_per_step", "roofline")}}
doc["traffic_over_algorithmic"] = doc["local_solve_hbm_bytes_corrected"] / doc["algorithmic_bytes_per_local_solve"]
json.dump(doc, open(out, "w"), indent=1)
print(out, "solves", nsolves, "traffic GB", round(doc["local_solve_hbm_bytes_corrected"] / 1e9, 3), "alg GB", round(doc["algorithmic_bytes_per_local_solve"] / 1e9, 3),
      "kernel ms per solve", doc["local_solve_kernel_ms"])
