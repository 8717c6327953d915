#!/bin/bash
# usage: tools/r6/additive_poll.sh lib.so  -- DPGO_POLL_FIRST_PAY sweep on the additive one-launch solve (in-kernel phase times)
L=$1
for W in sphere2500 grid:25x25x20 grid:25x25x10; do for FP in 0 30 44 56 70 90; do
DPGO_LIB=$PWD/$L DPGO_POLL_FIRST_PAY=$FP DPGO_PERSIST_VERBOSE=1 timeout 300 python bench.py --workload $W --precond additive --steps 30 --warmup 3 --no-cpu-baseline --no-secondary 2>/tmp/err.txt | grep '^{' | tail -1 > /tmp/b.json
PH=$(grep 'persistent tCG' /tmp/err.txt | tail -1 | sed 's/.*per iteration (us): //')
python - <<PY
import json
j=json.load(open("/tmp/b.json"))
print("%-14s first_pay %3d %8.1f it/s %5.1f products  %6.2f us/product | $PH" % ("$W", $FP, j["value"], j["products_per_step"], j.get("us_per_product") or 0))
PY
done; done
