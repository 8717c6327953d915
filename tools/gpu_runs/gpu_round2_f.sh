mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -m gpu -q --timeout 1200 --durations=15 ) > gpurun_out/t6_all.log 2>&1
tail -25 gpurun_out/t6_all.log
bash tools/profile_round.sh r02_v2 > gpurun_out/t6_profile.log 2>&1
tail -5 gpurun_out/t6_profile.log
