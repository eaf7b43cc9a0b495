#!/bin/bash
# k_rotate_mfma / k_gram2_sym at the headline block size (tools/rotate_probe.py) under rocprofv3 --stats
set -e
mkdir -p gpurun_out/rot
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/rp && mkdir -p /tmp/rp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp -o run -- python3 tools/rotate_probe.py > gpurun_out/rot/log.txt 2>&1 || { tail -20 gpurun_out/rot/log.txt; exit 1; }
grep -E "k_rotate_mfma|k_gram2_sym" $(find /tmp/rp -name "run_kernel_stats.csv" | head -1) | cut -c1-40,100-200
