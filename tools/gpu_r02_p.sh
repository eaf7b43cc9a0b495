mkdir -p gpurun_out
python -m pytest tests -m gpu -q -rA > gpurun_out/r02p_gpu_tests.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r02p_gpu_tests.log | tail -8
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r02p_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r02p_smoke.log
python bench.py > gpurun_out/r02p_bench.json 2> gpurun_out/r02p_bench.err; echo "bench rc=$?"; cut -c1-700 gpurun_out/r02p_bench.json
python bench.py --grid 100 --parts 1 > gpurun_out/r02p_bench_cfg1.json 2> gpurun_out/r02p_bench_cfg1.err; echo "bench cfg1 rc=$?"; cut -c1-500 gpurun_out/r02p_bench_cfg1.json
