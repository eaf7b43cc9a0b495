#!/bin/bash
# round 4: full GPU suite, the driver's bench command, then the full-length parity run (oracle to convergence at 216^3)
set -e
mkdir -p gpurun_out/r04l
timeout -k 10 1000 python -m pytest tests -m gpu -x -q -s > gpurun_out/r04l/tests.log 2>&1 || { tail -80 gpurun_out/r04l/tests.log; exit 1; }
tail -1 gpurun_out/r04l/tests.log
timeout -k 10 600 python bench.py > gpurun_out/r04l/bench.json 2> gpurun_out/r04l/bench.err || { tail -30 gpurun_out/r04l/bench.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04l/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"], (d.get("secondary") or {}).get("value"), d.get("setup_s"))
PY
timeout -k 10 900 python bench.py --cpu-iters 320 --no-secondary > gpurun_out/r04l/bench_full.json 2> gpurun_out/r04l/bench_full.err || { tail -30 gpurun_out/r04l/bench_full.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04l/bench_full.json").read().strip().splitlines()[-1])
print(d["cpu_baseline"].get("parity_full_length"))
PY
