#!/bin/bash
# round 4 profiles: kernel statistics and HBM traffic (separate --pmc passes, --kernel-trace only) of the default bench
# configuration (216^3, GenEO via ddm_geneo_basis: the run contains the eigensolver's MFMA kernels as well as the Krylov loop)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof gpurun_out/pmc1 gpurun_out/pmc2 /tmp/prof /tmp/pmc1 /tmp/pmc2
ARGS="--steps 10 --warmup 3 --cpu-iters 0 --no-solve --no-secondary --no-geneo-check"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o run -- python3 bench.py $ARGS > gpurun_out/prof/bench.json 2> gpurun_out/prof/bench.log || { tail -20 gpurun_out/prof/bench.log; exit 1; }
cp $(find /tmp/prof -name "run_kernel_stats.csv" | head -1) gpurun_out/prof/run_kernel_stats.csv
head -22 gpurun_out/prof/run_kernel_stats.csv | cut -c1-150
echo "pass FETCH_SIZE"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc1 -o run -- python3 bench.py $ARGS > gpurun_out/pmc1/bench.json 2> gpurun_out/pmc1/bench.log || { tail -20 gpurun_out/pmc1/bench.log; exit 1; }
echo "pass WRITE_SIZE"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc2 -o run -- python3 bench.py $ARGS > gpurun_out/pmc2/bench.json 2> gpurun_out/pmc2/bench.log || { tail -20 gpurun_out/pmc2/bench.log; exit 1; }
cp $(find /tmp/pmc1 -name "*counter_collection.csv" | head -1) gpurun_out/pmc1/run_counter_collection.csv
cp $(find /tmp/pmc2 -name "*counter_collection.csv" | head -1) gpurun_out/pmc2/run_counter_collection.csv
python3 tools/make_pmc_json.py gpurun_out/r04_pmc_traffic_grid216_geneo.json pipe "ddm::k_trsv_pipe"
rm -f gpurun_out/pmc1/run_counter_collection.csv gpurun_out/pmc2/run_counter_collection.csv
