mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_golden_configs.py tests/test_cpp_adaptor.py -m gpu -q -k "bicgstab or twolevel or gmres or device" > gpurun_out/r02i_tests.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error|assert" gpurun_out/r02i_tests.log | tail -12
