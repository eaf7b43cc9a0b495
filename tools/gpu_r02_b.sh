# round 2, GPU pass b: MFMA block kernels, native GenEO (all geneo tests + configs[3]/[4]), probes
mkdir -p gpurun_out
python -m pytest tests/test_gpu_blockvec.py tests/test_gpu_geneo.py tests/test_golden_configs.py -m gpu -q -rA > gpurun_out/r02b_gpu_tests.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|PASSED|FAILED|ERROR|^\[|Error|error|assert" gpurun_out/r02b_gpu_tests.log | tail -40
