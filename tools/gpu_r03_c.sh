set -e
mkdir -p gpurun_out/r03c
timeout -k 10 900 python -m pytest tests/test_gpu_sn_chol.py -x -q -s > gpurun_out/r03c/sn.log 2>&1 || { tail -60 gpurun_out/r03c/sn.log; exit 1; }
tail -15 gpurun_out/r03c/sn.log
