set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o run -- python3 bench.py --grid ${GRID:-100} --steps 10 --warmup 3 --cpu-iters 0 --no-solve > gpurun_out/prof/bench.json 2> gpurun_out/prof/bench.log || { tail -20 gpurun_out/prof/bench.log; exit 1; }
ls -R gpurun_out/prof | head -30
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
echo "== $f"; head -30 $f
