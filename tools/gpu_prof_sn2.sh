set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_sn2 /tmp/sn2
export DDM_DIRECT_ENGINE=device
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sn2 -o run -- python3 bench_convdiff.py --problem elasticity --cpu-iters 0 --no-solve --steps 30 > gpurun_out/r03_sn2/bench.json 2> gpurun_out/r03_sn2/bench.log || { tail -30 gpurun_out/r03_sn2/bench.log; exit 1; }
cp $(find /tmp/sn2 -name "run_kernel_stats.csv" | head -1) gpurun_out/r03_sn2/run_kernel_stats.csv
head -16 gpurun_out/r03_sn2/run_kernel_stats.csv | cut -c1-190
