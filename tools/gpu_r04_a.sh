#!/bin/bash
# round 4, first GPU call: full GPU suite on the round's first commits (ADVICE fixes, PDELab-facing adaptor), then the default bench
set -e
mkdir -p gpurun_out/r04a
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r04a/tests.log 2>&1 || { tail -80 gpurun_out/r04a/tests.log; exit 1; }
tail -3 gpurun_out/r04a/tests.log
timeout -k 10 400 python bench.py > gpurun_out/r04a/bench.json 2> gpurun_out/r04a/bench.err || { tail -30 gpurun_out/r04a/bench.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04a/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "roofline")})
PY
