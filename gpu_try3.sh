#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_pipe.py -x -q > gpurun_out/try_tests.log 2>&1 || { tail -30 gpurun_out/try_tests.log; exit 1; }
tail -2 gpurun_out/try_tests.log
timeout -k 10 300 python tools/trsv_engines_bench.py 216 2 2 2 pipe 10 2>&1 | grep engine | tee gpurun_out/try_bench.log
