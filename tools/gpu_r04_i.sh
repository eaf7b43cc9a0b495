#!/bin/bash
set -e
mkdir -p gpurun_out/r04i
rm -f gpurun_out/r04i/probe.log
for p in dg elasticity; do
  for top in 32 128 512 2048; do
    echo "== $p DDM_SN_TOP_MAX=$top" >> gpurun_out/r04i/probe.log
    timeout -k 10 300 python tools/sn_solve_probe.py $p DDM_SN_TOP_MAX=$top >> gpurun_out/r04i/probe.log 2>&1 || { tail -30 gpurun_out/r04i/probe.log; exit 1; }
  done
done
grep "==\|50 solve\|chains on\|levels 0" gpurun_out/r04i/probe.log | cut -c1-260
