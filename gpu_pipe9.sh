#!/bin/bash
mkdir -p gpurun_out
export DDM_PIPE_VERBOSE=1
timeout -k 10 300 python tools/pipe_trace.py ${N:-128} 2 2 2 > gpurun_out/pipe_trace${N:-128}.log 2>&1
echo "exit $?" >> gpurun_out/pipe_trace${N:-128}.log
grep -E "kernel span|exit|^group 0" gpurun_out/pipe_trace${N:-128}.log
