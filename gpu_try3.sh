#!/bin/bash
set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
python bench.py --cpu-iters 0 > gpurun_out/bench_try.json 2> gpurun_out/bench_try.log || { tail -30 gpurun_out/bench_try.log; exit 1; }
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_try.json"))
print("it/s", round(d["value"],2), "ms/step", round(d["ms_per_step"],3), d["solve"], d["iteration_traffic"]["phase_ms_per_iteration"])
PY
