mkdir -p gpurun_out
python -m pytest tests -m gpu -q -rA > gpurun_out/r02j_gpu_tests.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|^FAILED|^ERROR|Error" gpurun_out/r02j_gpu_tests.log | tail -12
python bench.py > gpurun_out/r02j_bench.json 2> gpurun_out/r02j_bench.err; echo "bench rc=$?"; cat gpurun_out/r02j_bench.json
