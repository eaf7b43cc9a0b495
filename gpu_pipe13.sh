#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_pipe.py tests/test_gpu_parity.py -x -q > gpurun_out/pipe_tests.log 2>&1
echo "pytest exit $?" >> gpurun_out/pipe_tests.log
tail -3 gpurun_out/pipe_tests.log
grep -q "pytest exit 0" gpurun_out/pipe_tests.log || exit 1
python bench.py --grid 100 --parts 1 --cpu-threads 1 --cpu-iters 0 > gpurun_out/bench_cfg2.json 2> gpurun_out/bench_cfg2.log || { tail -30 gpurun_out/bench_cfg2.log; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_cfg2.json"))
print("cfg2 it/s", round(d["value"], 2), "ms/step", round(d["ms_per_step"], 3), "local solve ms", round(d["roofline"]["avg_launch_ms"], 3), d["solve"])
PY
timeout -k 10 400 python tools/trsv_engines_bench.py 216 1 1 1 pipe 10 > gpurun_out/pipe_bench_1sub.log 2>&1
grep engine gpurun_out/pipe_bench_1sub.log
