#!/bin/bash
# round 3, end: GenEO tests, then kernel stats + PMC traffic passes of the default bench (tools/gpu_prof_r03.sh)
set -e
mkdir -p gpurun_out/r03n
timeout -k 10 900 python -m pytest tests/test_gpu_blockvec.py tests/test_gpu_geneo.py tests/test_gpu_coarse_spaces.py -x -q -m gpu > gpurun_out/r03n/tests.log 2>&1 || { tail -60 gpurun_out/r03n/tests.log; exit 1; }
tail -2 gpurun_out/r03n/tests.log
bash tools/gpu_prof_r03.sh
grep -i "GenEO:\|device setup" gpurun_out/prof/bench.log | head
