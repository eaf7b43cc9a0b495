mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_cpp_adaptor.py tests/test_gpu_coarse_spaces.py -m gpu -q -k "remaining or svd" > gpurun_out/r02k_tests.log 2>&1; echo "pytest rc=$?"; tail -40 gpurun_out/r02k_tests.log
