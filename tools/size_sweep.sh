#!/bin/bash
# usage: tools/size_sweep.sh  (on the GPU box): mid-size kernel / iteration times for DPGO_SPLIT variants
for sp in 1 2 4; do for w in grid:50x50x5 grid:25x25x5 sphere2500; do
DPGO_SPLIT=$sp timeout 200 python bench.py --workload $w --steps 10 --warmup 0 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > /tmp/b.json
python - <<PY
import json
j=json.load(open("/tmp/b.json"))
t=j["quality"]["tcg_iterations_per_step_rank0"]
print("split=$sp %-14s n=%6d it/s %8.1f ms/step %7.3f tcg/step %6.1f hess us %6.2f spmm us %6.2f us/tcg-it %6.1f"%("$w",j["config"]["poses_per_agent"],j["value"],j["ms_per_step"],t,j["roofline"]["avg_launch_us"],j["roofline"]["spmm_only"]["avg_launch_us"],1e3*j["ms_per_step"]/max(t,1)))
PY
done; done
