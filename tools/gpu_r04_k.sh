#!/bin/bash
# round 4: rocprofv3 kernel statistics + PMC traffic passes of the headline bench and of the two GMRES workloads
set -e
bash tools/gpu_prof_r04.sh
bash tools/gpu_prof_r04_workloads.sh
