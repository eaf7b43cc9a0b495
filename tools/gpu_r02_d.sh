# round 2: default bench (216^3, GenEO via ddm_geneo_basis) + the whole GPU suite
mkdir -p gpurun_out
( time DDM_VERBOSE=1 python bench.py > gpurun_out/r02d_bench_default.json 2> gpurun_out/r02d_bench_default.log ) 2> gpurun_out/r02d_bench_default.time; echo "bench rc=$?"
grep -v "^\[ddm geneo\]" gpurun_out/r02d_bench_default.log | tail -12; grep "ddm geneo" gpurun_out/r02d_bench_default.log | tail -3; cat gpurun_out/r02d_bench_default.time
python -m pytest tests -m gpu -q -rA > gpurun_out/r02d_gpu_tests.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|FAILED|ERROR" gpurun_out/r02d_gpu_tests.log | tail -10
