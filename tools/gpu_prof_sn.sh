set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_sn /tmp/sn_prof
export DDM_PIPE_VERBOSE=1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sn_prof -o run -- python3 -m pytest tests/test_gpu_sn_chol.py -x -q -s -k 64_cubed > gpurun_out/r03_sn/test.log 2>&1 || { tail -30 gpurun_out/r03_sn/test.log; exit 1; }
cp $(find /tmp/sn_prof -name "run_kernel_stats.csv" | head -1) gpurun_out/r03_sn/run_kernel_stats.csv
grep "sn 64" gpurun_out/r03_sn/test.log
head -14 gpurun_out/r03_sn/run_kernel_stats.csv | cut -c1-200
