#!/bin/bash
# usage: tools/r6/poll_first_sweep.sh  -- DPGO_POLL_FIRST sweep (sleep before the first sweep of an in-kernel reduction) on the
# one-launch solve after round 6's shorter phases: additive and block-Jacobi, 12 500-pose slab / 6 250-pose grid / sphere2500
for PC in additive jacobi; do for W in grid:25x25x20 grid:25x25x10 sphere2500 torus3D; do for FP in 12 20 28 36 44 52; do
DPGO_POLL_FIRST=$FP DPGO_PERSIST_VERBOSE=1 timeout 300 python bench.py --workload $W --precond $PC --steps 30 --warmup 3 --no-cpu-baseline --no-secondary 2>/tmp/err.txt | grep '^{' | tail -1 > /tmp/b.json
PH=$(grep 'persistent tCG' /tmp/err.txt | tail -1 | sed 's/.*per iteration (us): //')
WG=$(grep 'persistent tCG' /tmp/err.txt | tail -1 | sed 's/.*persistent tCG: //; s/,.*//')
python - <<PY
import json
j=json.load(open("/tmp/b.json"))
print("%-9s %-14s first %3d %8.1f it/s %5.1f products  %6.2f us/product | $WG | $PH" % ("$PC", "$W", $FP, j["value"], j["products_per_step"], j.get("us_per_product") or 0))
PY
done; done; done
