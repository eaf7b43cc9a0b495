#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python tools/pipe_trace.py 128 2 2 2 > gpurun_out/pipe_trace128.log 2>&1
echo "exit $?" >> gpurun_out/pipe_trace128.log
grep -E "kernel span|exit|^group [01]" gpurun_out/pipe_trace128.log
