#!/bin/bash
# usage: tools/r6/loopback_ab.sh lib1.so lib2.so ...   -- interleaved A/B of library builds on the 16-agent loop-back sweep
# and on the settled 8 x 12 500 sweep (bench.py --loopback), 3 rounds
for rep in 1 2 3; do for L in "$@"; do for apg in 16 8; do
DPGO_LIB=$PWD/$L timeout 300 python bench.py --loopback --agents-per-gpu $apg --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{' | tail -1 > /tmp/b.json
python - <<PY
import json
j=json.load(open("/tmp/b.json"))
q=j.get("quality") or {}
print("rep $rep %-28s agents %2d: %.3f ms per sweep, %.1f products, exchange %.3f ms, ran %s" % ("$L", $apg, j["ms_per_step"], j["products_per_step"], (j["config"].get("exchange_ms_per_step") if "exchange_ms_per_step" in j["config"] else q.get("exchange_ms_per_step_rank0")) or 0, j["config"].get("local_solver","")[-30:] if "precond_used_in_timed_steps" not in j["config"] else j["config"]["precond_used_in_timed_steps"]))
PY
done; done; done
