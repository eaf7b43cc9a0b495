mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_multirank.py -m gpu -q -x -k "distributed or shares" > gpurun_out/r02k_tests.log 2>&1; echo "pytest rc=$?"; tail -30 gpurun_out/r02k_tests.log
