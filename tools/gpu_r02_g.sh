# fused additive combination (one halo add): correctness and the bench; variants: two-pass order, coarse chain on a side stream
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_golden_configs.py tests/test_gpu_geneo.py tests/test_multirank.py tests/test_cpp_adaptor.py -m gpu -q > gpurun_out/r02g_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02g_tests.log
python bench.py --cpu-iters 0 > gpurun_out/r02g_bench_fused.json 2> gpurun_out/r02g_bench_fused.log; tail -2 gpurun_out/r02g_bench_fused.log
DDM_FUSE_LEVELS=0 python bench.py --cpu-iters 0 --no-solve > gpurun_out/r02g_bench_twopass.json 2> gpurun_out/r02g_bench_twopass.log
python3 - <<'PY'
import json
for f in ("gpurun_out/r02g_bench_fused.json", "gpurun_out/r02g_bench_twopass.json"):
    d = json.load(open(f))
    print(f, "it/s", round(d["value"], 2), "ms/step", round(d["ms_per_step"], 3), "solve", d["solve"], "local ms", round(d["roofline"]["avg_launch_ms"], 3),
          {k: round(v, 3) for k, v in d["iteration_traffic"]["phase_ms_per_iteration"].items()})
PY
