#!/bin/bash
mkdir -p gpurun_out
export DDM_PIPE_VERBOSE=1
for n in 40 64 96 128; do
timeout -k 10 200 python tools/pipe_trace.py $n 2 2 2 > gpurun_out/pipe_trace$n.log 2>&1
echo "exit $?" >> gpurun_out/pipe_trace$n.log
grep -E "out-of-range|kernel span|exit" gpurun_out/pipe_trace$n.log
grep -q "exit 0" gpurun_out/pipe_trace$n.log || exit 1
done
