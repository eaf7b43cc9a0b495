#!/bin/bash
# round 4: kernel times of the box engine inside the headline bench (rocprofv3 kernel statistics)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04q
rm -rf gpurun_out/r04q/prof
rocprofv3 --kernel-trace --stats -d gpurun_out/r04q/prof -o run --output-format csv -- python3 bench.py --steps 10 --warmup 3 --cpu-iters 0 --no-solve --no-geneo-check --no-secondary > gpurun_out/r04q/bench.json 2> gpurun_out/r04q/bench.err || { tail -20 gpurun_out/r04q/bench.err; exit 1; }
f=$(find gpurun_out/r04q/prof -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/r04q/kernel_stats.csv
grep -E "k_box|k_trsv_pipe|k_pipe_permute|k_pipe_prologue" gpurun_out/r04q/kernel_stats.csv | cut -d, -f1-4,6-7 | cut -c1-200
rm -rf gpurun_out/r04q/prof
