# round 2, GPU pass: new tests + full-size probes of configs[3]/[4] with ILU(0) and with the direct local solvers
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_geneo.py tests/test_golden_configs.py -m gpu -q -rA -k "direct" > gpurun_out/r02b_gpu_tests.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|FAILED|ERROR|^\[|Error|assert" gpurun_out/r02b_gpu_tests.log | tail -30
DDM_LOCAL_SOLVER=umfpack timeout -k 10 500 python tools/config_probe.py dg 512 > gpurun_out/r02b_probe_dg512_lu.log 2>&1; tail -9 gpurun_out/r02b_probe_dg512_lu.log
DDM_LOCAL_SOLVER=cholmod timeout -k 10 500 python tools/config_probe.py elasticity 1 > gpurun_out/r02b_probe_el1_chol.log 2>&1; tail -9 gpurun_out/r02b_probe_el1_chol.log

