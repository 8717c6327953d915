set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -k "persistent or full_size" > gpurun_out/t3_new.log 2>&1
tail -8 gpurun_out/t3_new.log
DPGO_PERSIST_VERBOSE=1 timeout 600 python tools/ml_probe.py grid625 sphere grid6250 grid100k > gpurun_out/t3_probe.log 2>&1
grep -v "^dpgo_hip" gpurun_out/t3_probe.log | tail -20
grep "^dpgo_hip" gpurun_out/t3_probe.log | sort | uniq -c | head
