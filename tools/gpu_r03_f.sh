#!/bin/bash
# round 3: the elasticity / DG workloads with the default engine policy: tests that touch the direct solvers, then kernel stats + PMC traffic
set -e
mkdir -p gpurun_out/r03f
export DDM_PIPE_VERBOSE=1
timeout -k 10 900 python -m pytest tests/test_gpu_sn_chol.py tests/test_gpu_fullsize.py tests/test_golden_configs.py tests/test_cpp_adaptor.py -x -q -m gpu > gpurun_out/r03f/tests.log 2>&1 || { tail -60 gpurun_out/r03f/tests.log; exit 1; }
tail -4 gpurun_out/r03f/tests.log
for P in dg elasticity; do
  python bench_convdiff.py --problem $P > gpurun_out/r03f/bench_$P.json 2> gpurun_out/r03f/bench_$P.log || { tail -20 gpurun_out/r03f/bench_$P.log; exit 1; }
  tail -c 1500 gpurun_out/r03f/bench_$P.json
done
PROBLEMS="elasticity" bash tools/gpu_prof_r03_workloads.sh
