#!/bin/bash
mkdir -p gpurun_out
export DDM_PIPE_VERBOSE=1
timeout -k 10 1000 python tools/trsv_engines_bench.py ${N:-216} 2 2 2 pipe,pipe:DELTA=8,pipe:DELTA=32,pipe:DELTA=48:SPAN=320,pipe:DELTA=24:SPAN=256 10 > gpurun_out/pipe_bench_sweep.log 2>&1
echo "exit $?" >> gpurun_out/pipe_bench_sweep.log
grep -E "engine|pipe schedule|exit" gpurun_out/pipe_bench_sweep.log | cut -c1-260
