mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -k "persistent or multilevel_preconditioner_matches" > gpurun_out/t8_new.log 2>&1
tail -4 gpurun_out/t8_new.log
timeout 600 python tools/ml_probe.py grid100k slab > gpurun_out/t8_probe.log 2>&1
tail -6 gpurun_out/t8_probe.log
bash tools/profile_round.sh r02_v3 > gpurun_out/t8_profile.log 2>&1
head -14 gpurun_out/r02_v3_bench_kernel_stats.csv | cut -d, -f1-4 | sed 's/(.*"/"/' | cut -c1-100
bash tools/profile_extra.sh r02_v3 > gpurun_out/t8_extra.log 2>&1
tail -22 gpurun_out/t8_extra.log
( DPGO_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --warmup 1 --workload grid:24x24x16 --no-cpu-baseline --no-secondary ) > gpurun_out/t8_n2.log 2>&1
tail -3 gpurun_out/t8_n2.log | cut -c1-600
