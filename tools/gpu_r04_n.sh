#!/bin/bash
# round 4: first runs of the box triangular-solve engine: its tests, the engine parity test, then the headline bench
set -e
mkdir -p gpurun_out/r04n
DDM_PIPE_VERBOSE=1 timeout -k 10 400 python -m pytest tests/test_gpu_box.py -x -q -s > gpurun_out/r04n/tests_box.log 2>&1 || { tail -60 gpurun_out/r04n/tests_box.log; exit 1; }
tail -3 gpurun_out/r04n/tests_box.log
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipe.py -x -q > gpurun_out/r04n/tests_parity.log 2>&1 || { tail -60 gpurun_out/r04n/tests_parity.log; exit 1; }
tail -1 gpurun_out/r04n/tests_parity.log
DDM_PIPE_VERBOSE=1 timeout -k 10 600 python bench.py --cpu-iters 0 --no-geneo-check --no-secondary > gpurun_out/r04n/bench.json 2> gpurun_out/r04n/bench.err || { tail -30 gpurun_out/r04n/bench.err; exit 1; }
grep "box engine\|ILU(0) setup\|full solve" gpurun_out/r04n/bench.err | cut -c1-400
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04n/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["solve"], d["setup_s"])
PY
