mkdir -p gpurun_out
timeout -k 10 300 python tools/pipe_trace_plain.py 111 > gpurun_out/r02s_nc1.log 2>&1; echo "nc1 rc=$?"; tail -1 gpurun_out/r02s_nc1.log
DDM_HIP_LIBRARY=$PWD/dune-ddm_amd/libddm_hip_nc2.so timeout -k 10 300 python tools/pipe_trace_plain.py 111 > gpurun_out/r02s_nc2.log 2>&1; echo "nc2 rc=$?"; tail -1 gpurun_out/r02s_nc2.log
