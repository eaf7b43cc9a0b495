mkdir -p gpurun_out
python -m pytest tests -m gpu -q -rA -s > gpurun_out/r02f_gpu_tests.log 2>&1; echo "pytest rc=$?"
grep -E "^\[|passed|failed|FAILED|ERROR" gpurun_out/r02f_gpu_tests.log | cut -c1-400 | tail -30
