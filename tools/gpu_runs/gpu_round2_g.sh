mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -k "persistent" > gpurun_out/t7_new.log 2>&1
tail -6 gpurun_out/t7_new.log
DPGO_PERSIST_VERBOSE=1 timeout 600 python tools/ml_probe.py grid625 sphere > gpurun_out/t7_probe.log 2>&1
grep -v "^dpgo_hip" gpurun_out/t7_probe.log | tail -8
grep "^dpgo_hip" gpurun_out/t7_probe.log | grep " 50 iter" | sort | uniq -c | sort -rn | head -6
grep "^dpgo_hip" gpurun_out/t7_probe.log | grep -v " 50 iter" | head -4
