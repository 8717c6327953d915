#!/bin/bash
# usage: tools/r6/loopback_precond.sh lib.so [lib2.so ...]  -- the settled 8 x 12 500 sweep and the 16-agent loop-back sweep
# with the preconditioner forced / chosen by the rule, interleaved, 2 rounds
for rep in 1 2; do for L in "$@"; do for apg in 8 16; do for PC in jacobi additive auto; do
DPGO_LIB=$PWD/$L timeout 300 python bench.py --loopback --agents-per-gpu $apg --precond $PC --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{' | tail -1 > /tmp/b.json
python - <<PY
import json
j=json.load(open("/tmp/b.json"))
print("rep $rep %-28s agents %2d %-8s: %.3f ms per sweep, %.1f products, ran %s" % ("$L", $apg, "$PC", j["ms_per_step"], j["products_per_step"], j["config"].get("precond_used_in_timed_steps")))
PY
done; done; done; done
