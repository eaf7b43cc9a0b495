#!/bin/bash
# round 4, fourth GPU call: where does the DG iteration lose 2.5 ms?  Single-vector solve probes with and without the persistent top kernel
set -e
mkdir -p gpurun_out/r04d
for p in dg elasticity poisson64; do
  for top in 32 0; do
    echo "== $p DDM_SN_TOP_MAX=$top" >> gpurun_out/r04d/probe.log
    timeout -k 10 300 python tools/sn_solve_probe.py $p DDM_SN_TOP_MAX=$top >> gpurun_out/r04d/probe.log 2>&1 || { tail -30 gpurun_out/r04d/probe.log; exit 1; }
  done
done
grep -v "amdgpu.ids" gpurun_out/r04d/probe.log | grep "==\|solve(s)\|single-vector\|residual\|factor:" 
