#!/bin/bash
# round 4, fifth GPU call: the persistent top kernel with register prefetch and fused forward phases; slots in target-major order
set -e
mkdir -p gpurun_out/r04e
timeout -k 10 700 python -m pytest tests/test_gpu_sn_chol.py -x -q -m gpu -s > gpurun_out/r04e/tests_sn.log 2>&1 || { tail -60 gpurun_out/r04e/tests_sn.log; exit 1; }
grep "\[sn" gpurun_out/r04e/tests_sn.log | cut -c1-250; tail -1 gpurun_out/r04e/tests_sn.log
for p in dg elasticity poisson64; do
  echo "== $p" >> gpurun_out/r04e/probe.log
  timeout -k 10 300 python tools/sn_solve_probe.py $p >> gpurun_out/r04e/probe.log 2>&1 || { tail -30 gpurun_out/r04e/probe.log; exit 1; }
done
grep "==\|50 solve\|single-vector\|residual" gpurun_out/r04e/probe.log | cut -c1-220
