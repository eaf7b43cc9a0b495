#!/bin/bash
# round 4: sn tests, solve probes, then the two GMRES workloads (bench_convdiff.py) with the chain kernel
set -e
mkdir -p gpurun_out/r04j
timeout -k 10 700 python -m pytest tests/test_gpu_sn_chol.py -x -q -m gpu > gpurun_out/r04j/tests_sn.log 2>&1 || { tail -60 gpurun_out/r04j/tests_sn.log; exit 1; }
tail -1 gpurun_out/r04j/tests_sn.log
rm -f gpurun_out/r04j/probe.log
for p in dg elasticity poisson64; do
  echo "== $p" >> gpurun_out/r04j/probe.log
  timeout -k 10 300 python tools/sn_solve_probe.py $p >> gpurun_out/r04j/probe.log 2>&1 || { tail -30 gpurun_out/r04j/probe.log; exit 1; }
done
grep "==\|50 solve" gpurun_out/r04j/probe.log | cut -c1-200
for p in dg elasticity; do
  timeout -k 10 300 python bench_convdiff.py --problem $p > gpurun_out/r04j/bench_$p.json 2> gpurun_out/r04j/bench_$p.err || { tail -30 gpurun_out/r04j/bench_$p.err; exit 1; }
  python - "$p" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r04j/bench_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["solve"], d["setup_s"], (d.get("cpu_baseline") or {}).get("parity_first_iterations"))
PY
done
