#!/bin/bash
# usage: tools/r6/additive_ab.sh lib1.so lib2.so ...   -- interleaved A/B of library builds on the one-launch solve with the
# additive two-level preconditioner (PC=jacobi: block-Jacobi): in-kernel phase times (DPGO_PERSIST_VERBOSE), sphere2500 / 12 500-pose slab / torus3D, 3 rounds
for rep in 1 2 3; do for L in "$@"; do for W in sphere2500 grid:25x25x20 torus3D; do
DPGO_LIB=$PWD/$L DPGO_PERSIST_VERBOSE=1 timeout 300 python bench.py --workload $W --precond ${PC:-additive} --steps 30 --warmup 3 --no-cpu-baseline --no-secondary 2>/tmp/err.txt | grep '^{' | tail -1 > /tmp/b.json
PH=$(grep 'persistent tCG' /tmp/err.txt | tail -1 | sed 's/.*per iteration (us): //')
python - <<PY
import json
j=json.load(open("/tmp/b.json"))
print("rep $rep %-26s %-14s %8.1f it/s %.3f ms/step %5.1f products  %6.2f us/product | $PH" % ("$L", "$W", j["value"], j["ms_per_step"], j["products_per_step"], j.get("us_per_product") or 0))
PY
done; done; done
