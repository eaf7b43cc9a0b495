#!/bin/bash
# first GPU run of the pipe engine: parity tests, then engine timings at 128^3
mkdir -p gpurun_out
export DDM_PIPE_VERBOSE=1
timeout -k 10 400 python -m pytest tests/test_gpu_pipe.py -x -q > gpurun_out/pipe_tests.log 2>&1
echo "pytest exit $?" >> gpurun_out/pipe_tests.log
tail -15 gpurun_out/pipe_tests.log
grep -q "pytest exit 0" gpurun_out/pipe_tests.log && timeout -k 10 400 python tools/trsv_engines_bench.py 128 2 2 2 pipe,pipe:LAZY=0,xcd2 10 > gpurun_out/pipe_bench128.log 2>&1
tail -8 gpurun_out/pipe_bench128.log
