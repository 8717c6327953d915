mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -k "persistent" > gpurun_out/t9_new.log 2>&1
tail -3 gpurun_out/t9_new.log
bash tools/profile_extra.sh r02_v3 > gpurun_out/t9_extra.log 2>&1
tail -24 gpurun_out/t9_extra.log
