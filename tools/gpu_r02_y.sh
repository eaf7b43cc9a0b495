mkdir -p gpurun_out
DDM_PIPE_SPREAD=1 python bench.py --cpu-iters 0 --no-solve > gpurun_out/r02y_bench_spread.json 2> gpurun_out/r02y.err; echo "rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02y_bench_spread.json'))
print(round(d["value"],2), round(d["ms_per_step"],3), {k: round(v,3) for k,v in d["iteration_traffic"]["phase_ms_per_iteration"].items()})
PY
