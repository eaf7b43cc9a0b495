mkdir -p gpurun_out
python -m pytest tests/test_gpu_fullsize.py -m gpu -q -rA -s -k "parity" > gpurun_out/r02f_a.log 2>&1; grep -E "^\[96|passed|failed" gpurun_out/r02f_a.log | cut -c1-1500
DDM_NO_SHARED_DEFECT=1 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -rA -s -k "parity" > gpurun_out/r02f_b.log 2>&1; grep -E "^\[96|passed|failed" gpurun_out/r02f_b.log | cut -c1-1500
