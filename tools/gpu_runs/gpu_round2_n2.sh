#!/bin/bash
# the N > 1 bench path end to end on one GPU (both ranks share device 0, host-staged gloo exchange)
mkdir -p gpurun_out/t
for n in 2; do
DPGO_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus $n --steps 4 --warmup 1 > gpurun_out/t/bench_n$n.json 2> gpurun_out/t/bench_n$n.err
echo "rc=$?"; tail -1 gpurun_out/t/bench_n$n.json | cut -c1-300; grep -i "error\|Traceback" gpurun_out/t/bench_n$n.err | head -5
done
