# round 2: adaptor tests (device-resident solver, GenEO adaptor), RCCL self test
mkdir -p gpurun_out
python -m pytest tests/test_cpp_adaptor.py tests/test_multirank.py -m gpu -q -rA > gpurun_out/r02e_gpu_tests.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|FAILED|ERROR|SKIPPED|Error|assert" gpurun_out/r02e_gpu_tests.log | tail -30
