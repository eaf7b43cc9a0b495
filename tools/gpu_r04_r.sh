#!/bin/bash
# round 4: box engine with a three-step request distance: probes, tests, then the bench under rocprofv3 (kernel statistics)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04r
rm -f gpucore.* gpurun_out/r04r/*
fail() { grep -v "amdgpu.ids\|^\s" gpurun_out/r04r/probe.log | cut -c1-250 | tail -30; rm -f gpucore.*; exit 1; }
for args in "9 8 7 1 1 1" "26 24 22 2 2 2"; do
  echo "== $args" >> gpurun_out/r04r/probe.log
  timeout -k 10 120 python tools/box_probe.py $args >> gpurun_out/r04r/probe.log 2>&1 || fail
done
grep -v "amdgpu.ids\|^\s" gpurun_out/r04r/probe.log | cut -c1-250 | tail -10
grep -q "mismatches [1-9]" gpurun_out/r04r/probe.log && exit 1
timeout -k 10 400 python -m pytest tests/test_gpu_box.py -x -q > gpurun_out/r04r/tests_box.log 2>&1 || { tail -40 gpurun_out/r04r/tests_box.log | cut -c1-300; exit 1; }
tail -1 gpurun_out/r04r/tests_box.log
rocprofv3 --kernel-trace --stats -d gpurun_out/r04r/prof -o run --output-format csv -- python3 bench.py --steps 10 --warmup 3 --cpu-iters 0 --no-geneo-check --no-secondary > gpurun_out/r04r/bench.json 2> gpurun_out/r04r/bench.err || { tail -20 gpurun_out/r04r/bench.err; exit 1; }
f=$(find gpurun_out/r04r/prof -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/r04r/kernel_stats.csv
rm -rf gpurun_out/r04r/prof
grep -E "k_box|k_trsv_pipe" gpurun_out/r04r/kernel_stats.csv | cut -d, -f1-4,6-7 | cut -c1-200
grep "full solve" gpurun_out/r04r/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04r/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["solve"])
PY
