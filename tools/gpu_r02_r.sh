mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_geneo.py tests/test_gpu_coarse_spaces.py tests/test_golden_configs.py -m gpu -q -x > gpurun_out/r02r_tests.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r02r_tests.log
python bench.py --cpu-iters 0 --no-solve --steps 5 --warmup 2 > gpurun_out/r02r_bench.json 2> gpurun_out/r02r_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02r_bench.json'))
print(d["value"], d["geneo"], d["setup_s"])
PY
