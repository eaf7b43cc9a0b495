#!/bin/bash
# kernel statistics of the default (GenEO) configuration: setup + 13 iterations; only the summary travels back
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_geneo /tmp/prof_geneo
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_geneo -o run -- python3 bench.py --steps 10 --warmup 3 --cpu-iters 0 --no-solve > gpurun_out/prof_geneo/bench.json 2> gpurun_out/prof_geneo/bench.log || { tail -20 gpurun_out/prof_geneo/bench.log; exit 1; }
cp $(find /tmp/prof_geneo -name "run_kernel_stats.csv" | head -1) gpurun_out/prof_geneo/run_kernel_stats.csv
grep -E "k_trsv_pipe|k_coarse|k_spmv_stream|k_pipe_permute" gpurun_out/prof_geneo/run_kernel_stats.csv | cut -c1-160
