#!/bin/bash
# usage: tools/grid_sweep.sh "U1 H1" "U2 H2" ...  -- sweeps the launch caps of the two tCG kernels on one box (100k grid)
for cfg in "$@"; do
set -- $cfg
DPGO_GRID_UPDATE=$1 DPGO_GRID_HESS=$2 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > /tmp/b.json
python - <<PY
import json
j=json.load(open("/tmp/b.json")); t=j["quality"]["tcg_iterations_per_step_rank0"]
print("update cap %5s hess cap %5s  it/s %8.1f ms/step %7.3f hess us %6.2f us/tcg-it %6.1f"%("$1","$2",j["value"],j["ms_per_step"],j["roofline"]["avg_launch_us"],1e3*j["ms_per_step"]/max(t,1)))
PY
done
