mkdir -p gpurun_out
timeout -k 10 400 python tools/pipe_trace.py 216 2 2 2 > gpurun_out/r02l_trace_nc1.log 2>&1; echo "nc1 rc=$?"
DDM_HIP_LIBRARY=$PWD/dune-ddm_amd/libddm_hip_nc2.so timeout -k 10 400 python tools/pipe_trace.py 216 2 2 2 > gpurun_out/r02l_trace_nc2.log 2>&1; echo "nc2 rc=$?"
grep -E "^tasks|^group" gpurun_out/r02l_trace_nc1.log | cut -c1-900
grep -E "^tasks|^group" gpurun_out/r02l_trace_nc2.log | cut -c1-900
