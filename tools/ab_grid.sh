#!/bin/bash
# usage: tools/ab_grid.sh "lib U H" ...  -- interleaved same-box A/B of builds x launch caps (100k grid)
for rep in 1 2; do for cfg in "$@"; do
set -- $cfg
DPGO_LIB=$PWD/$1 DPGO_GRID_UPDATE=$2 DPGO_GRID_HESS=$3 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > /tmp/b.json
python - <<PY
import json
j=json.load(open("/tmp/b.json")); t=j["quality"]["tcg_iterations_per_step_rank0"]
print("rep $rep %-28s U %5s H %5s  it/s %8.1f ms/step %7.3f hess us %6.2f us/tcg-it %6.1f"%("$1","$2","$3",j["value"],j["ms_per_step"],j["roofline"]["avg_launch_us"],1e3*j["ms_per_step"]/max(t,1)))
PY
done; done
