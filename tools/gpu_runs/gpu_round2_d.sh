mkdir -p gpurun_out
DPGO_PERSIST_VERBOSE=1 timeout 600 python tools/ml_probe.py grid625 sphere > gpurun_out/t4_probe.log 2>&1
grep -v "^dpgo_hip" gpurun_out/t4_probe.log | tail -8
grep "^dpgo_hip" gpurun_out/t4_probe.log | grep " 50 iter" | head -8
