#!/bin/bash
# usage: tools/r6/env_ab.sh VAR val1 val2 ...   -- interleaved A/B of one environment switch on the default bench (3 rounds)
V=$1; shift
for rep in 1 2 3; do for val in "$@"; do
env $V=$val timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{' | tail -1 > /tmp/b.json
python - <<PY
import json
j=json.load(open("/tmp/b.json"))
print("rep $rep $V=%-6s it/s %7.1f ms/step %6.3f us/product %6.1f products %.0f" % ("$val", j["value"], j["ms_per_step"], j["us_per_product"], j["products_per_step"]))
PY
done; done
