#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 400 python tools/pipe_trace.py 216 2 2 2 > gpurun_out/pipe_trace216.log 2>&1
grep -E "^tasks|^group" gpurun_out/pipe_trace216.log | cut -c1-1200
