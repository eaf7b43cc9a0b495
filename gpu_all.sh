set -e
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
python bench.py --grid ${GRID:-216} --steps 20 --warmup 5 --cpu-iters 0 --coarse pou > gpurun_out/bench_pou.json 2> gpurun_out/bench_pou.log || { tail -30 gpurun_out/bench_pou.log; exit 1; }
grep -E "full solve|levels" gpurun_out/bench_pou.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_pou.json"))
print("it/s", d["value"], "ms/step", d["ms_per_step"], "roofline", d["roofline"]["achieved"], d["roofline"]["avg_launch_ms"], d["solve"])
print(d["iteration_traffic"]["phase_ms_per_iteration"])
PY
