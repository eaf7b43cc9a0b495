mkdir -p gpurun_out
python bench_convdiff.py > gpurun_out/r02t_convdiff_umfpack.json 2> gpurun_out/r02t_convdiff_umfpack.err; echo "rc=$?"; tail -4 gpurun_out/r02t_convdiff_umfpack.err; cut -c1-900 gpurun_out/r02t_convdiff_umfpack.json
