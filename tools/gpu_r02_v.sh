mkdir -p gpurun_out
export DDM_PIPE_VARIANT=2
for c in pf nopf; do
  if [ $c = nopf ]; then export DDM_HIP_LIBRARY=$PWD/dune-ddm_amd/libddm_hip_p2_nopf.so; fi
  timeout -k 10 300 python -m pytest tests/test_gpu_pipe.py -m gpu -x -q > gpurun_out/r02v_tests_$c.log 2>&1; echo "tests $c rc=$?"; tail -2 gpurun_out/r02v_tests_$c.log
  timeout -k 10 300 python tools/pipe_trace_plain.py 111 > gpurun_out/r02v_plain_$c.log 2>&1; echo "plain $c rc=$?"; tail -1 gpurun_out/r02v_plain_$c.log
  timeout -k 10 400 python tools/pipe_trace.py 216 2 2 2 > gpurun_out/r02v_trace_$c.log 2>&1; echo "trace $c rc=$?"; grep -E "^tasks|^group|^    task    0" gpurun_out/r02v_trace_$c.log | cut -c1-200
done
