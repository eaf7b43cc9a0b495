#!/bin/bash
# round 4, third GPU call: the deterministic device direct solver (coloured update phases, slot-ordered forward sweeps, iterative
# refinement fixed per factor by a probe) and the persistent kernel for the top tree levels of the single-vector solve
set -e
mkdir -p gpurun_out/r04c
timeout -k 10 700 python -m pytest tests/test_gpu_sn_chol.py -x -q -m gpu -s > gpurun_out/r04c/tests_sn.log 2>&1 || { tail -80 gpurun_out/r04c/tests_sn.log; exit 1; }
grep "\[sn" gpurun_out/r04c/tests_sn.log | head -20; tail -2 gpurun_out/r04c/tests_sn.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_coarse_spaces.py tests/test_gpu_geneo.py tests/test_golden_configs.py tests/test_cpp_adaptor.py -x -q -m gpu > gpurun_out/r04c/tests_rest.log 2>&1 || { tail -80 gpurun_out/r04c/tests_rest.log; exit 1; }
tail -2 gpurun_out/r04c/tests_rest.log
for p in dg elasticity; do
  timeout -k 10 300 python bench_convdiff.py --problem $p > gpurun_out/r04c/bench_$p.json 2> gpurun_out/r04c/bench_$p.err || { tail -30 gpurun_out/r04c/bench_$p.err; exit 1; }
  python - "$p" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r04c/bench_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], {k: d.get(k) for k in ("value", "ms_per_step", "roofline")})
PY
done
