#!/bin/bash
# round 3: DG workload profile with the default (device) engine, host-engine DG for the setup comparison, kernel stats of the default bench incl. GenEO setup
set -e
mkdir -p gpurun_out/r03g
DDM_DIRECT_ENGINE=host python bench_convdiff.py --problem dg --cpu-iters 0 > gpurun_out/r03g/dg_host.json 2> gpurun_out/r03g/dg_host.log || { tail -20 gpurun_out/r03g/dg_host.log; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/r03g/dg_host.json"))
print("dg host engine: it/s", round(d["value"],1), "local solve ms", round(d["roofline"]["avg_launch_ms"],3), "setup", d["setup_s"], "geneo", d["geneo"]["setup_s"], d["geneo"]["iterate_s"], "solve", d["solve"]["solve_s"])
PY
PROBLEMS="dg" bash tools/gpu_prof_r03_workloads.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof /tmp/prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o run -- python3 bench.py --steps 10 --warmup 3 --cpu-iters 0 --no-solve --no-geneo-check > gpurun_out/prof/bench.json 2> gpurun_out/prof/bench.log || { tail -20 gpurun_out/prof/bench.log; exit 1; }
cp $(find /tmp/prof -name "run_kernel_stats.csv" | head -1) gpurun_out/prof/run_kernel_stats.csv
head -24 gpurun_out/prof/run_kernel_stats.csv | cut -c1-150
grep -i geneo gpurun_out/prof/bench.log | head
