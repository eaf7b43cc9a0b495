# round 2, first GPU pass: the whole -m gpu suite after the engine clean-up + the new configs[3]/[4] tests, then the two probes
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -rA > gpurun_out/r02a_gpu_tests.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|PASSED|FAILED|ERROR|^\[" gpurun_out/r02a_gpu_tests.log | tail -60
DDM_PIPE_VERBOSE=1 timeout -k 10 300 python tools/config_probe.py dg 128 > gpurun_out/r02a_probe_dg128.log 2>&1; tail -12 gpurun_out/r02a_probe_dg128.log
DDM_PIPE_VERBOSE=1 timeout -k 10 300 python tools/config_probe.py elasticity 0 > gpurun_out/r02a_probe_el0.log 2>&1; tail -12 gpurun_out/r02a_probe_el0.log
