#!/bin/bash
# full GPU test suite with pipe as the default engine, then the 216^3 bench (POU coarse space: short setup) with phase timers
mkdir -p gpurun_out
export DDM_PIPE_VERBOSE=1
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1
echo "pytest exit $?" >> gpurun_out/gpu_tests.log
tail -4 gpurun_out/gpu_tests.log
grep -q "pytest exit 0" gpurun_out/gpu_tests.log || exit 1
timeout -k 10 900 python bench.py --grid 216 --steps 20 --warmup 5 --cpu-iters 0 --coarse pou > gpurun_out/bench_pou.json 2> gpurun_out/bench_pou.log
echo "bench exit $?"
grep -E "pipe schedule|full solve" gpurun_out/bench_pou.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_pou.json"))
print("it/s", round(d["value"], 2), "ms/step", round(d["ms_per_step"], 3), "local solve ms", round(d["roofline"]["avg_launch_ms"], 3), "GB/s", round(d["roofline"]["achieved"], 1))
print(d["iteration_traffic"]["phase_ms_per_iteration"])
print(d["solve"])
PY
