#!/bin/bash
# round 3: GenEO eigensolver work -- block-kernel tests, then the default bench under rocprofv3 --stats with the eigensolver's log
set -e
mkdir -p gpurun_out/r03h
timeout -k 10 900 python -m pytest tests/test_gpu_blockvec.py tests/test_gpu_geneo.py -x -q -m gpu > gpurun_out/r03h/tests.log 2>&1 || { tail -60 gpurun_out/r03h/tests.log; exit 1; }
tail -3 gpurun_out/r03h/tests.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p /tmp/prof_h
export DDM_VERBOSE=1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_h -o run -- python3 bench.py --steps 10 --warmup 3 --cpu-iters 0 --no-solve --no-geneo-check ${BENCH_EXTRA} > gpurun_out/r03h/bench.json 2> gpurun_out/r03h/bench.log || { tail -20 gpurun_out/r03h/bench.log; exit 1; }
cp $(find /tmp/prof_h -name "run_kernel_stats.csv" | head -1) gpurun_out/r03h/run_kernel_stats.csv
head -14 gpurun_out/r03h/run_kernel_stats.csv | cut -c1-60,150-260
grep -i "geneo\] setup\|GenEO:\|device setup" gpurun_out/r03h/bench.log | head
grep "ddm geneo\] it" gpurun_out/r03h/bench.log | awk 'NR%6==0' | cut -c1-200
