#!/bin/bash
# usage: tools/ab.sh <workload> lib1.so lib2.so ...   -- interleaved A/B of kernel builds on one box
W=$1; shift
for rep in 1 2 3; do for L in "$@"; do
DPGO_LIB=$PWD/$L timeout 300 python bench.py --workload $W --steps 10 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > /tmp/b.json
python - <<PY
import json
j=json.load(open("/tmp/b.json")); t=j["quality"]["tcg_iterations_per_step_rank0"]
print("rep $rep %-36s it/s %8.1f ms/step %7.3f hess us %6.2f spmm us %6.2f us/tcg-it %6.1f"%("$L",j["value"],j["ms_per_step"],j["roofline"]["avg_launch_us"],j["roofline"]["spmm_only"]["avg_launch_us"],1e3*j["ms_per_step"]/max(t,1)))
PY
done; done
