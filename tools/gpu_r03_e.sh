set -e
mkdir -p gpurun_out/r03e
export DDM_PIPE_VERBOSE=1
for PC in auto ilu0; do
  python bench.py --grid 128 --cpu-iters 0 --no-geneo-check --geneo-preconditioner $PC > gpurun_out/r03e/bench128_$PC.json 2> gpurun_out/r03e/bench128_$PC.log || { tail -20 gpurun_out/r03e/bench128_$PC.log; exit 1; }
  grep -E "GenEO:|device supernodal|full solve|device setup" gpurun_out/r03e/bench128_$PC.log | cut -c1-330
done
