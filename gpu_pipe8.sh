#!/bin/bash
mkdir -p gpurun_out
export DDM_PIPE_VERBOSE=1
timeout -k 10 400 python -m pytest tests/test_gpu_pipe.py -x -q > gpurun_out/pipe_tests.log 2>&1
echo "pytest exit $?" >> gpurun_out/pipe_tests.log
tail -3 gpurun_out/pipe_tests.log
grep -q "pytest exit 0" gpurun_out/pipe_tests.log || exit 1
timeout -k 10 400 python tools/trsv_engines_bench.py 128 2 2 2 ${ENGINES:-pipe,pipe:LAZY=0} 10 > gpurun_out/pipe_bench128.log 2>&1
grep engine gpurun_out/pipe_bench128.log
timeout -k 10 900 python bench.py --grid 216 --steps 20 --warmup 5 --cpu-iters 0 --coarse pou --no-solve > gpurun_out/bench_pou.json 2> gpurun_out/bench_pou.log
echo "bench exit $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_pou.json"))
print("it/s", round(d["value"], 2), "ms/step", round(d["ms_per_step"], 3), "local solve ms", round(d["roofline"]["avg_launch_ms"], 3), "GB/s", round(d["roofline"]["achieved"], 1))
print(d["iteration_traffic"]["phase_ms_per_iteration"])
PY
