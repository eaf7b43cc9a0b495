#!/bin/bash
# round 4: box engine (polls by the helper wave): tests, per-plane stamps, then the bench with the engine on (opt-in) under rocprofv3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04s
rm -f gpurun_out/r04s/*
timeout -k 10 400 python -m pytest tests/test_gpu_box.py tests/test_gpu_parity.py -x -q > gpurun_out/r04s/tests.log 2>&1 || { tail -40 gpurun_out/r04s/tests.log | cut -c1-300; exit 1; }
tail -1 gpurun_out/r04s/tests.log
echo "== 216^3, 8 subdomains" >> gpurun_out/r04s/log.txt
DDM_BOX_CHECK=1 BOX_PROBE_NOCHECK=1 BOX_PROBE_REPS=2 timeout -k 10 600 python tools/box_probe.py 216 216 216 2 2 2 >> gpurun_out/r04s/log.txt 2>&1 || { tail -20 gpurun_out/r04s/log.txt; exit 1; }
grep -v "amdgpu.ids" gpurun_out/r04s/log.txt | cut -c1-330
export DDM_TRSV_MODE=box
rocprofv3 --kernel-trace --stats -d gpurun_out/r04s/prof -o run --output-format csv -- python3 bench.py --steps 10 --warmup 3 --cpu-iters 0 --no-geneo-check --no-secondary > gpurun_out/r04s/bench.json 2> gpurun_out/r04s/bench.err || { tail -20 gpurun_out/r04s/bench.err; exit 1; }
f=$(find gpurun_out/r04s/prof -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/r04s/kernel_stats.csv
rm -rf gpurun_out/r04s/prof
grep -E "k_box|k_trsv_pipe" gpurun_out/r04s/kernel_stats.csv | cut -d, -f1-4,6-7 | cut -c1-200
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04s/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["solve"])
PY
