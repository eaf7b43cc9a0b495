set -e
mkdir -p gpurun_out/r03d
export DDM_PIPE_VERBOSE=1
timeout -k 10 600 python -m pytest tests/test_gpu_sn_chol.py -x -q -s -k "lu" > gpurun_out/r03d/lu.log 2>&1 || { tail -60 gpurun_out/r03d/lu.log; exit 1; }
tail -6 gpurun_out/r03d/lu.log
for E in host device; do
  DDM_DIRECT_ENGINE=$E python bench_convdiff.py --problem elasticity --cpu-iters 0 > gpurun_out/r03d/elast_$E.json 2> gpurun_out/r03d/elast_$E.log || { tail -20 gpurun_out/r03d/elast_$E.log; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r03d/elast_$E.json"))
print("$E", "it/s", round(d["value"],1), "ms/step", round(d["ms_per_step"],3), "local solve ms", round(d["roofline"]["avg_launch_ms"],3), "setup", d["setup_s"], "geneo", d["geneo"]["iterations"], d["geneo"]["setup_s"], d["geneo"]["iterate_s"], "solve", d["solve"])
PY
done
