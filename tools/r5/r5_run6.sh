#!/bin/bash
# round-5 GPU run 6: ONE rank's share of the N = 8 / 4 / 2 configurations on this GPU: two agents (one per colour), the
# stream-ordered sweep, every exchange through the 1-rank RCCL communicator
export GPU_OUT=r5f
OUT=gpurun_out/$GPU_OUT
mkdir -p $OUT
for wl in 25x25x20 50x50x10 50x50x20; do for pc in auto jacobi additive; do
  if [ $wl == 50x50x20 ] && [ $pc == additive ]; then continue; fi
  timeout 300 python bench.py --workload grid:$wl --loopback --agents-per-gpu 2 --precond $pc --no-cpu-baseline --no-secondary --steps 100 --warmup 5 2>$OUT/rank_${wl}_$pc.err | grep "^{" | tail -1 > $OUT/rank_${wl}_$pc.json
  python - $OUT/rank_${wl}_$pc.json $wl $pc <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1]))
    print("grid:%s 2 agents %-8s: %.3f ms per sweep (%.0f it/s), %.1f products, exchange %.3f ms, used %s, selection sweeps %s" % (
        sys.argv[2], sys.argv[3], j["ms_per_step"], j["value"], j["products_per_step"], j["quality"]["exchange_ms_per_step_rank0"],
        j["config"]["precond_used_in_timed_steps"], j["config"]["selection_sweeps_before_timing"]))
except Exception as e:
    print("grid:%s %s failed: %r" % (sys.argv[2], sys.argv[3], e))
PY
done; done
