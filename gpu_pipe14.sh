#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python tools/trsv_engines_bench.py 216 1 1 1 pipe:SPREAD=1,pipe 10 > gpurun_out/pipe_bench_1sub.log 2>&1
grep engine gpurun_out/pipe_bench_1sub.log
timeout -k 10 600 python tools/trsv_engines_bench.py 100 1 1 1 pipe:SPREAD=1,pipe 10 > gpurun_out/pipe_bench_1sub100.log 2>&1
grep engine gpurun_out/pipe_bench_1sub100.log
timeout -k 10 600 python tools/trsv_engines_bench.py 216 2 1 1 pipe:SPREAD=1,pipe 10 > gpurun_out/pipe_bench_2sub.log 2>&1
grep engine gpurun_out/pipe_bench_2sub.log
