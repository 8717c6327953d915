#!/bin/bash
# round-5 GPU run 9: the peer store ordered on the device (two processes on one GPU), against the host-barrier scheme
export GPU_OUT=r5i
OUT=gpurun_out/$GPU_OUT
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests -q -m gpu -x -k "two_processes_exchange or two_ranks_on_one_gpu" > $OUT/tests.log 2>&1
tail -4 $OUT/tests.log
for hb in 0 1; do
  port=$((29600 + hb))
  DPGO_IPC_HOST_BARRIERS=$hb timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $port \
    bench.py --gpus 2 --transport ipc --steps 50 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/ipc_hb$hb.out 2> $OUT/ipc_hb$hb.err
  grep '^{' $OUT/ipc_hb$hb.out | tail -1 > $OUT/ipc_hb$hb.json
  python - $OUT/ipc_hb$hb.json $hb <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1]))
    print("host barriers=%s: %.3f ms per sweep (%d agents), exchange %.3f ms per sweep | %s" % (sys.argv[2], j["ms_per_step"], j["config"]["agents"], j["quality"]["exchange_ms_per_step_rank0"], j["config"]["schedule"][-90:]))
except Exception as e:
    print("host barriers=%s failed: %r" % (sys.argv[2], e))
PY
  tail -3 $OUT/ipc_hb$hb.err | cut -c1-300
done
