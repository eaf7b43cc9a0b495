#!/bin/bash
# round 4: full GPU suite, the driver's bench command, the full-length parity run -- on the code of the end of the round
set -e
mkdir -p gpurun_out/r04t
rm -f gpurun_out/r04t/*
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -s > gpurun_out/r04t/tests.log 2>&1 || { tail -80 gpurun_out/r04t/tests.log | cut -c1-300; exit 1; }
tail -1 gpurun_out/r04t/tests.log
timeout -k 10 600 python bench.py > gpurun_out/r04t/bench.json 2> gpurun_out/r04t/bench.err || { tail -30 gpurun_out/r04t/bench.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04t/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("latency_floor_ms"), (d.get("secondary") or {}).get("value"), d.get("setup_s"), d["cpu_baseline"]["value"])
PY
timeout -k 10 900 python bench.py --cpu-iters 320 --no-secondary > gpurun_out/r04t/bench_full.json 2> gpurun_out/r04t/bench_full.err || { tail -30 gpurun_out/r04t/bench_full.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04t/bench_full.json").read().strip().splitlines()[-1])
print(d["cpu_baseline"].get("parity_full_length"))
PY
