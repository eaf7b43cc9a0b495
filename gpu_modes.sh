for mode in ${MODES:-xcd}; do
DDM_TRSV_MODE=$mode python bench.py --grid ${GRID:-216} --steps 20 --warmup 5 --cpu-iters 0 --coarse pou --no-solve > gpurun_out/bench_pou.json 2> gpurun_out/bench_pou.log || { tail -30 gpurun_out/bench_pou.log; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/bench_pou.json"))
print("$mode", "it/s", round(d["value"],2), "ms/step", round(d["ms_per_step"],3), "local solve ms", round(d["roofline"]["avg_launch_ms"],3), "GB/s", round(d["roofline"]["achieved"],1), flush=True)
PY
done
