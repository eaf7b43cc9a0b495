#!/bin/bash
set -e
mkdir -p gpurun_out/r04g
DDM_SN_TOP_STAMPS=1 timeout -k 10 300 python tools/sn_solve_probe.py elasticity > gpurun_out/r04g/stamps_elasticity.log 2>&1 || { tail -30 gpurun_out/r04g/stamps_elasticity.log; exit 1; }
grep "barrier " gpurun_out/r04g/stamps_elasticity.log | head -150
