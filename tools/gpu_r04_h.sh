#!/bin/bash
# round 4: full GPU suite on the state after the sn-solver work, then the headline bench (with the secondary workload) and the
# rank-local emulation of N = 2, 4, 8
set -e
mkdir -p gpurun_out/r04h
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r04h/tests.log 2>&1 || { tail -80 gpurun_out/r04h/tests.log; exit 1; }
tail -3 gpurun_out/r04h/tests.log
timeout -k 10 500 python bench.py > gpurun_out/r04h/bench.json 2> gpurun_out/r04h/bench.err || { tail -30 gpurun_out/r04h/bench.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04h/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "setup_s", "collectives_per_iteration")}, d["roofline"]["frac"])
s = d.get("secondary") or {}
print("secondary:", {k: s.get(k) for k in ("value", "ms_per_step", "error")}, (s.get("roofline") or {}).get("avg_launch_ms"))
PY
grep "device setup phases\|GenEO:" gpurun_out/r04h/bench.err | cut -c1-400
for N in 2 4 8; do
  timeout -k 10 400 python bench.py --emulate-rank-of $N > gpurun_out/r04h/rank_local_N$N.json 2> gpurun_out/r04h/rank_local_N$N.err || { tail -30 gpurun_out/r04h/rank_local_N$N.err; exit 1; }
  python - $N <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r04h/rank_local_N{sys.argv[1]}.json").read().strip().splitlines()[-1])
print("N", sys.argv[1], d["ms_per_step"], d["phase_ms_per_iteration"], d["geneo"], d["setup_s"])
PY
done
