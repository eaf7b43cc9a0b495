mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_pipe.py tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/r02n_tests.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r02n_tests.log
python bench.py --cpu-iters 0 > gpurun_out/r02n_bench.json 2> gpurun_out/r02n_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02n_bench.json'))
print(d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["iteration_traffic"]["phase_ms_per_iteration"])
PY
