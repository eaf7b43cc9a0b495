#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 500 python tools/spmv_bench.py 216 2>&1 | grep -v amdgpu.ids | tee gpurun_out/spmv_bench.log
