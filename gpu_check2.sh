set -e
mkdir -p gpurun_out
( time python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.log ) 2> gpurun_out/bench_default.time || { tail -30 gpurun_out/bench_default.log; exit 1; }
grep -v "^\[geneo\]" gpurun_out/bench_default.log | tail -6
python -c "
import json; d=json.load(open('gpurun_out/bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['cpu_baseline'])"
