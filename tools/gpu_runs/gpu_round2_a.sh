set -x
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --timeout 900 -k "dense_spd or multilevel or polar or full_size" > gpurun_out/t1_new.log 2>&1
tail -5 gpurun_out/t1_new.log
python tools/ml_probe.py sphere slab grid100k > gpurun_out/t1_probe.log 2>&1
tail -20 gpurun_out/t1_probe.log
python -m pytest tests -m gpu -q --timeout 900 -k "not (dense_spd or multilevel or polar or full_size)" > gpurun_out/t1_rest.log 2>&1
tail -5 gpurun_out/t1_rest.log
