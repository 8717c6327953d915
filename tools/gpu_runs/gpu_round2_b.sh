set -x
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --timeout 1200 -k "full_size or kitti_00-3-None or rccl or cxx_shim or bench_prints or multi_agent_rbcd or follows_Q" > gpurun_out/t2_new.log 2>&1
tail -8 gpurun_out/t2_new.log
bash tools/profile_round.sh r02_v1 > gpurun_out/t2_profile.log 2>&1
tail -30 gpurun_out/t2_profile.log
