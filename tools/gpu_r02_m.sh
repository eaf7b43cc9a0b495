mkdir -p gpurun_out
python bench.py --cpu-iters 320 > gpurun_out/r02m_bench_fullparity.json 2> gpurun_out/r02m_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02m_bench_fullparity.json'))
print(d["value"], d["solve"], d["cpu_baseline"].get("parity_full_length"), d["cpu_baseline"]["parity_first_iterations"]["ok"], d["cpu_baseline"]["value"])
PY
