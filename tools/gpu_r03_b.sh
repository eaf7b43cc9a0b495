set -e
mkdir -p gpurun_out/r03b
python -m pytest tests -m gpu -x -q -s > gpurun_out/r03b/gpu_tests.log 2>&1 || { tail -60 gpurun_out/r03b/gpu_tests.log; exit 1; }
tail -3 gpurun_out/r03b/gpu_tests.log
grep "96^3 GenEO" gpurun_out/r03b/gpu_tests.log || true
( time python bench.py > gpurun_out/r03b/bench_default.json 2> gpurun_out/r03b/bench_default.log ) 2> gpurun_out/r03b/bench_default.time || { tail -30 gpurun_out/r03b/bench_default.log; exit 1; }
grep -v "^\[geneo\]" gpurun_out/r03b/bench_default.log | tail -12; cat gpurun_out/r03b/bench_default.time
