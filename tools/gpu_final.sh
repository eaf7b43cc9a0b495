set -e
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()"
( time python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.log ) 2> gpurun_out/bench_default.time || { tail -30 gpurun_out/bench_default.log; exit 1; }
grep -v "^\[geneo\]" gpurun_out/bench_default.log | tail -8; cat gpurun_out/bench_default.time
python bench.py --grid 100 --parts 1 --cpu-threads 1 > gpurun_out/bench_cfg2.json 2> gpurun_out/bench_cfg2.log || { tail -30 gpurun_out/bench_cfg2.log; exit 1; }
tail -4 gpurun_out/bench_cfg2.log
python - <<'PY'
import json
for f in ("gpurun_out/bench_default.json","gpurun_out/bench_cfg2.json"):
    d=json.load(open(f))
    print(f, "it/s", round(d["value"],2), "ms/step", round(d["ms_per_step"],3), "solve", d["solve"], "roofline", round(d["roofline"]["achieved"],1), d["roofline"]["frac"], "cpu", d["cpu_baseline"])
PY
