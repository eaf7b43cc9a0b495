mkdir -p gpurun_out
export DDM_PIPE_VARIANT=2
timeout -k 10 400 python tools/pipe_trace.py 216 2 2 2 > gpurun_out/r02u_trace.log 2>&1; echo "trace rc=$?"; grep -E "^tasks|^group|^    task" gpurun_out/r02u_trace.log | cut -c1-230
