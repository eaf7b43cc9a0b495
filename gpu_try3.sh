#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 900 python tools/trsv_engines_bench.py 216 2 2 2 pipe,pipe:DELTA=16,pipe:DELTA=32,pipe:DELTA=48,pipe:SPAN=192,pipe:SPAN=384,pipe:DELTA=32:SPAN=384 10 2>&1 | grep engine | tee gpurun_out/pipe_bench_sweep2.log
