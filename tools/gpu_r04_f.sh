#!/bin/bash
# round 4: kernel-level breakdown of the single-vector supernodal solve (rocprofv3 --kernel-trace --stats on tools/sn_solve_probe.py)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04f
for p in dg elasticity; do
  rm -rf /tmp/snp_$p
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/snp_$p -o run -- python3 tools/sn_solve_probe.py $p > gpurun_out/r04f/probe_$p.log 2>&1 || { tail -20 gpurun_out/r04f/probe_$p.log; exit 1; }
  cp $(find /tmp/snp_$p -name "run_kernel_stats.csv" | head -1) gpurun_out/r04f/kernel_stats_$p.csv
  echo "== $p"; head -14 gpurun_out/r04f/kernel_stats_$p.csv | cut -c1-160
done
