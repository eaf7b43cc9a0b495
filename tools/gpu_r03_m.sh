#!/bin/bash
# round 3: MFMA kernels with the accumulators kept in VGPRs -- every test that runs one, factorisation rates, then the profiled bench
set -e
mkdir -p gpurun_out/r03m
export DDM_PIPE_VERBOSE=1
timeout -k 10 900 python -m pytest tests/test_gpu_blockvec.py tests/test_gpu_geneo.py tests/test_gpu_sn_chol.py tests/test_gpu_coarse_spaces.py -x -q -m gpu -s > gpurun_out/r03m/tests.log 2>&1 || { tail -60 gpurun_out/r03m/tests.log; exit 1; }
tail -2 gpurun_out/r03m/tests.log
grep "numeric factorisation\|sn 64" gpurun_out/r03m/tests.log | sort | uniq | tail -6
unset DDM_PIPE_VERBOSE
python bench.py --grid 128 --cpu-iters 0 --no-order-leg --no-geneo-check > gpurun_out/r03m/bench128.json 2> gpurun_out/r03m/bench128.log || { tail -20 gpurun_out/r03m/bench128.log; exit 1; }
grep -i "GenEO:\|device setup" gpurun_out/r03m/bench128.log | head -4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p /tmp/prof_m
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_m -o run -- python3 bench.py --steps 10 --warmup 3 --cpu-iters 0 --no-solve --no-geneo-check > gpurun_out/r03m/bench.json 2> gpurun_out/r03m/bench.log || { tail -20 gpurun_out/r03m/bench.log; exit 1; }
cp $(find /tmp/prof_m -name "run_kernel_stats.csv" | head -1) gpurun_out/r03m/run_kernel_stats.csv
grep -i "GenEO:\|device setup" gpurun_out/r03m/bench.log | head
head -12 gpurun_out/r03m/run_kernel_stats.csv | cut -c1-50,110-210
