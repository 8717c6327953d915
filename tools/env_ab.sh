#!/bin/bash
# usage: tools/env_ab.sh <reps> "ENV=val ..." "ENV=val ..." ...   -- interleaved bench A/B of environment settings on one box
# ("-" = no setting).  Prints it/s and ms per step of the default workload (100k grid, library-default preconditioner).
REPS=$1; shift
for rep in $(seq 1 $REPS); do for cfg in "$@"; do
  if [ "$cfg" = "-" ]; then E=""; else E="$cfg"; fi
  env $E timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 200 2>/dev/null | tail -1 > /tmp/env_ab.json
  python - "$cfg" <<'PY'
import json, sys
j = json.load(open("/tmp/env_ab.json"))
print("%-44s it/s %6.1f  ms/step %.3f  products/step %s" % (sys.argv[1][-44:], j["value"], j["ms_per_step"], j["config"].get("products_per_step")), flush=True)
PY
done; done
