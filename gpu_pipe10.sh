#!/bin/bash
mkdir -p gpurun_out
export DDM_PIPE_VERBOSE=1
timeout -k 10 400 python -m pytest tests/test_gpu_pipe.py -x -q > gpurun_out/pipe_tests.log 2>&1
echo "pytest exit $?" >> gpurun_out/pipe_tests.log
tail -3 gpurun_out/pipe_tests.log
grep -q "pytest exit 0" gpurun_out/pipe_tests.log || exit 1
timeout -k 10 400 python tools/trsv_engines_bench.py 128 2 2 2 ${ENGINES:-pipe,pipe:LAZY=0} 10 > gpurun_out/pipe_bench128.log 2>&1
grep engine gpurun_out/pipe_bench128.log
timeout -k 10 300 python tools/pipe_trace.py 128 2 2 2 > gpurun_out/pipe_trace128.log 2>&1
echo "exit $?" >> gpurun_out/pipe_trace128.log
grep -E "kernel span|exit|^group 0" gpurun_out/pipe_trace128.log
