#!/bin/bash
# round 4: setup phases of the headline run in detail (host generator in C++, device phases verbose)
set -e
mkdir -p gpurun_out/r04m
DDM_VERBOSE=1 DDM_PIPE_VERBOSE=1 timeout -k 10 600 python bench.py --steps 10 --warmup 3 --cpu-iters 0 --no-geneo-check --no-secondary > gpurun_out/r04m/bench.json 2> gpurun_out/r04m/bench.err || { tail -30 gpurun_out/r04m/bench.err; exit 1; }
grep -v "^\[geneo\] it" gpurun_out/r04m/bench.err | cut -c1-400 | tail -60
