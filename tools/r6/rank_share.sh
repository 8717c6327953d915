#!/bin/bash
# usage: tools/r6/rank_share.sh  -- what DESIGN section 7 predicts N = 2 / 4 / 8 from: (1) one rank's share of the N-GPU job
# (two agents, one per colour, stream-ordered sweep, 1-rank RCCL exchange) and (2) the loop-back sweeps with the solves
# one after the other (--sequential: the sum of the solo solves), by preconditioner
for W in grid:25x25x20 grid:50x25x20 grid:50x50x20; do for PC in auto jacobi additive; do
timeout 300 python bench.py --workload $W --loopback --agents-per-gpu 2 --precond $PC --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{' | tail -1 > /tmp/b.json
python - <<PY
import json
try:
    j=json.load(open("/tmp/b.json"))
    print("rank share %-14s %-8s: %.3f ms per sweep, %.1f products, ran %s, exchange %s ms" % ("$W", "$PC", j["ms_per_step"], j["products_per_step"], j["config"].get("precond_used_in_timed_steps"), j["config"].get("exchange_ms_per_step")))
except Exception as e:
    print("rank share $W $PC: failed (%s)" % e)
PY
done; done
for apg in 16 8; do for PC in auto jacobi additive; do
timeout 300 python bench.py --loopback --agents-per-gpu $apg --sequential --precond $PC --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{' | tail -1 > /tmp/b.json
python - <<PY
import json
try:
    j=json.load(open("/tmp/b.json"))
    print("sequential agents %2d %-8s: %.3f ms per sweep = %.3f ms per solve, %.1f products, ran %s" % ($apg, "$PC", j["ms_per_step"], j["ms_per_step"]/$apg, j["products_per_step"], j["config"].get("precond_used_in_timed_steps")))
except Exception as e:
    print("sequential $apg $PC: failed (%s)" % e)
PY
done; done
