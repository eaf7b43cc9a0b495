mkdir -p gpurun_out
timeout -k 10 600 python tools/geneo_parity_probe.py 96 20 > gpurun_out/geneo_probe.log 2>&1
echo "exit $?" >> gpurun_out/geneo_probe.log
grep -v "^\[geneo\]" gpurun_out/geneo_probe.log | tail -16
