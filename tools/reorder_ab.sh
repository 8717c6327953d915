#!/bin/bash
# GPU box: A/B of the agent-level pose renumbering on the headline workload (bench.py, 100k grid, 1 agent).
# usage: bash tools/reorder_ab.sh "off 1 16 64" [extra bench args]
OUT=gpurun_out/${GPU_OUT:-run}; mkdir -p $OUT
modes=$1; shift
for m in $modes; do
  if [ "$m" == "off" ]; then export DPGO_REORDER=0; else export DPGO_REORDER=1 DPGO_REORDER_RUN=$m; fi
  timeout 900 python bench.py --no-cpu-baseline --no-secondary --steps 50 "$@" 2> $OUT/ab_$m.err | grep '^{' | tail -1 > $OUT/ab_$m.json
  python - $OUT/ab_$m.json $m <<'PY'
import json, sys
j = json.load(open(sys.argv[1])); rf = j["roofline"]
print("run=%-4s %.1f it/s %.3f ms  us/product %.1f products %.0f | %s cold %.2f us (frac %.3f, own %.3f) warm %.2f | spmm_sym cold %.2f plain %.2f | plain hess cold %s" % (
    sys.argv[2], j["value"], j["ms_per_step"], j["quality"]["us_per_tcg_iteration_rank0"], j["products_per_step"], rf["kernel"][:22],
    rf["avg_launch_us"], rf["frac"], rf["frac_own_bytes"], rf["warm"]["avg_launch_us"], (rf.get("spmm_symmetric") or {}).get("avg_launch_us", 0),
    rf["spmm_only"]["avg_launch_us"], (rf.get("plain_storage") or {}).get("avg_launch_us")))
print("      kernels", [(k["kernel"][:14], round(k["avg_launch_us"], 1)) for k in rf["kernels"]], "sizes", (rf.get("multilevel") or {}).get("sizes"))
PY
done
