mkdir -p gpurun_out
for g in 0 32 64 128; do
  if [ $g = 0 ]; then unset DDM_OVERLAP_COARSE; else export DDM_OVERLAP_COARSE=1 DDM_OVERLAP_GRID=$g; fi
  python bench.py --cpu-iters 0 --no-solve > gpurun_out/r02o_bench_$g.json 2> gpurun_out/r02o_bench_$g.err || { echo "bench $g failed"; tail -5 gpurun_out/r02o_bench_$g.err; exit 1; }
  python - <<PY
import json
d=json.load(open('gpurun_out/r02o_bench_$g.json'))
print("grid $g:", round(d["value"],2), round(d["ms_per_step"],3), round(d["roofline"]["avg_launch_ms"],3), {k: round(v,3) for k,v in d["iteration_traffic"]["phase_ms_per_iteration"].items()})
PY
done
