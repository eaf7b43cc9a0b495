#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_pipe.py -x -q > gpurun_out/pipe_tests.log 2>&1
echo "pytest exit $?" >> gpurun_out/pipe_tests.log
tail -3 gpurun_out/pipe_tests.log
grep -q "pytest exit 0" gpurun_out/pipe_tests.log || exit 1
timeout -k 10 900 python tools/trsv_engines_bench.py 216 2 2 2 pipe 10 > gpurun_out/pipe_bench216.log 2>&1
grep engine gpurun_out/pipe_bench216.log
