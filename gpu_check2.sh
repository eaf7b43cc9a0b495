set -e
mkdir -p gpurun_out
python bench.py --grid 48 --steps 5 --warmup 2 --cpu-iters 20 > gpurun_out/bench_48.json 2> gpurun_out/bench_48.log || { tail -30 gpurun_out/bench_48.log; exit 1; }
grep -E "parity|full solve|cpu_baseline" gpurun_out/bench_48.log
python -c "
import json; d=json.load(open('gpurun_out/bench_48.json')); print(d['cpu_baseline'])"
( time python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.log ) 2> gpurun_out/bench_default.time || { tail -30 gpurun_out/bench_default.log; exit 1; }
grep -v "^\[geneo\]" gpurun_out/bench_default.log | tail -6
python -c "
import json; d=json.load(open('gpurun_out/bench_default.json')); print(d['value'], d['ms_per_step'], d['cpu_baseline'])"
