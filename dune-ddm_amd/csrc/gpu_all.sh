set -e
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
GRID=100 bash gpu_prof.sh | grep -E "trsv|spmv" | cut -c1-200
GRID=216 CPUITERS=5 bash gpu_bench.sh
