#!/bin/bash
# round-5 GPU run 15: A/B of library builds on the headline (arguments: library file names under dpgo_amd/)
export GPU_OUT=${GPU_OUT:-r5o}
OUT=gpurun_out/$GPU_OUT
mkdir -p $OUT
for rep in 1 2 3; do for L in "$@"; do
  DPGO_LIB=$PWD/dpgo_amd/$L timeout 400 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{' | tail -1 > $OUT/h_$L.$rep.json
  python - $OUT/h_$L.$rep.json $L <<'PY'
import json, sys
j = json.load(open(sys.argv[1])); L = sys.argv[2]
rf = j["roofline"]
print("%-18s %.1f it/s  %.3f ms/step  products/step %s | %s %.2f us frac %.3f  tail %.2f" % (L, j["value"], j["ms_per_step"], j.get("products_per_step"), (rf.get("kernel") or "")[:22], rf.get("avg_launch_us") or 0, rf["frac"], rf.get("cycle_tail_us") or 0))
PY
done; done
