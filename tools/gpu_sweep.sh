#!/bin/bash
# robustness sweep of the pipe engine over subdomain counts / placement modes (each line: bit-identical repeat solves, status 0)
set -e
mkdir -p gpurun_out
: > gpurun_out/pipe_sweep_final.log
for cfg in "64 2 2 2 pipe,pipe:SPREAD=1" "100 2 2 1 pipe,pipe:SPREAD=1" "150 1 1 1 pipe,pipe:SPREAD=0" "120 3 3 3 pipe" "96 4 4 2 pipe" "216 2 1 1 pipe"; do
  set -- $cfg
  timeout -k 10 400 python tools/trsv_engines_bench.py $1 $2 $3 $4 $5 8 2>&1 | grep -E "problem|engine" | tee -a gpurun_out/pipe_sweep_final.log
done
