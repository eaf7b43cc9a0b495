set -e
mkdir -p gpurun_out/r03a
python -m pytest tests/test_multirank.py -m gpu -x -q > gpurun_out/r03a/multirank.log 2>&1 || { tail -40 gpurun_out/r03a/multirank.log; exit 1; }
tail -3 gpurun_out/r03a/multirank.log
( time python bench.py --gpus 2 --cpu-iters 0 > gpurun_out/r03a/bench_gpus2.json 2> gpurun_out/r03a/bench_gpus2.log ) 2> gpurun_out/r03a/bench_gpus2.time || { tail -40 gpurun_out/r03a/bench_gpus2.log; exit 1; }
grep -v "^\[geneo\]" gpurun_out/r03a/bench_gpus2.log | tail -8; cat gpurun_out/r03a/bench_gpus2.time; cat gpurun_out/r03a/bench_gpus2.json
