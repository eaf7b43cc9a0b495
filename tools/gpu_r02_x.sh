mkdir -p gpurun_out
python bench.py --grid 112 --parts 1 --cpu-iters 0 > gpurun_out/r02x_bench_112_1sub.json 2> gpurun_out/r02x.err; echo "rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02x_bench_112_1sub.json'))
print(round(d["value"],2), round(d["ms_per_step"],3), {k: round(v,3) for k,v in d["iteration_traffic"]["phase_ms_per_iteration"].items()}, d["solve"])
PY
