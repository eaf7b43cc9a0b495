# supernodal transformation of the direct factors: parity tests, then the two full-size probes
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_cpp_adaptor.py tests/test_gpu_geneo.py tests/test_golden_configs.py -m gpu -q -k "direct or device or eigenpairs or golden" > gpurun_out/r02h_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02h_tests.log
DDM_PIPE_VERBOSE=1 DDM_LOCAL_SOLVER=umfpack timeout -k 10 500 python tools/config_probe.py dg 512 > gpurun_out/r02h_probe_dg512_lu.log 2>&1; grep -E "direct factor|local solve|GMRES|GenEO" gpurun_out/r02h_probe_dg512_lu.log | cut -c1-250
DDM_PIPE_VERBOSE=1 DDM_LOCAL_SOLVER=cholmod timeout -k 10 500 python tools/config_probe.py elasticity 1 > gpurun_out/r02h_probe_el1_chol.log 2>&1; grep -E "direct factor|local solve|GMRES|GenEO" gpurun_out/r02h_probe_el1_chol.log | cut -c1-250
