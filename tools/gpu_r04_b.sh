#!/bin/bash
# round 4, second GPU call: the forward tasks read the right-hand side in natural order (k_pipe_permute_in is gone): pipe parity tests,
# default bench, stamped per-step trace of the 216^3 solve (gpurun_out/pipe_trace216.npz; joined with the schedule by tools/pipe_hops.py)
set -e
mkdir -p gpurun_out/r04b
timeout -k 10 600 python -m pytest tests/test_gpu_pipe.py tests/test_gpu_parity.py tests/test_gpu_sn_chol.py -x -q -m gpu > gpurun_out/r04b/tests.log 2>&1 || { tail -60 gpurun_out/r04b/tests.log; exit 1; }
tail -2 gpurun_out/r04b/tests.log
timeout -k 10 400 python bench.py > gpurun_out/r04b/bench.json 2> gpurun_out/r04b/bench.err || { tail -30 gpurun_out/r04b/bench.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04b/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "roofline")})
PY
timeout -k 10 300 python tools/pipe_trace.py 216 2 2 2 > gpurun_out/r04b/trace.txt 2>&1 || { tail -20 gpurun_out/r04b/trace.txt; exit 1; }
grep "kernel span\|sweep" gpurun_out/r04b/trace.txt | head -12
