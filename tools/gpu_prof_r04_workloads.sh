#!/bin/bash
# round 4: kernel statistics + HBM traffic (separate --pmc passes, --kernel-trace only) of the second and third workload
# (bench_convdiff.py --problem dg | elasticity: BASELINE configs[3] / configs[4], sparse direct local solves)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for P in ${PROBLEMS:-dg elasticity}; do
  O=gpurun_out/r04_prof_$P
  mkdir -p $O/stats $O/pmc1 $O/pmc2 /tmp/st_$P /tmp/p1_$P /tmp/p2_$P
  ARGS="--problem $P --steps 10 --warmup 3 --cpu-iters 0 --no-solve --profile-counts"
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$P -o run -- python3 bench_convdiff.py $ARGS > $O/bench.json 2> $O/bench.log || { tail -20 $O/bench.log; exit 1; }
  cp $(find /tmp/st_$P -name "run_kernel_stats.csv" | head -1) $O/stats/run_kernel_stats.csv
  head -14 $O/stats/run_kernel_stats.csv | cut -c1-170
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p1_$P -o run -- python3 bench_convdiff.py $ARGS > $O/pmc1/bench.json 2> $O/pmc1/bench.log || { tail -20 $O/pmc1/bench.log; exit 1; }
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/p2_$P -o run -- python3 bench_convdiff.py $ARGS > $O/pmc2/bench.json 2> $O/pmc2/bench.log || { tail -20 $O/pmc2/bench.log; exit 1; }
  cp $(find /tmp/p1_$P -name "*counter_collection.csv" | head -1) $O/pmc1/run_counter_collection.csv
  cp $(find /tmp/p2_$P -name "*counter_collection.csv" | head -1) $O/pmc2/run_counter_collection.csv
  python3 tools/make_pmc_json_workload.py $P $O $O/r04_pmc_traffic_$P.json
  rm -f $O/pmc1/run_counter_collection.csv $O/pmc2/run_counter_collection.csv
done
