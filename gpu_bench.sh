set -e
mkdir -p gpurun_out
python bench.py --grid ${GRID:-100} --steps 20 --warmup 5 --cpu-iters ${CPUITERS:-5} > gpurun_out/bench_${GRID:-100}.json 2> gpurun_out/bench_${GRID:-100}.log || { tail -30 gpurun_out/bench_${GRID:-100}.log; exit 1; }
cat gpurun_out/bench_${GRID:-100}.log
cat gpurun_out/bench_${GRID:-100}.json
