#!/bin/bash
# A/B: fused tCG-step kernel on the plain block-CSR arrays vs the symmetric storage
mkdir -p gpurun_out/o
for sym in 0 1; do
  for pc in multilevel jacobi; do
    DPGO_TCG_SYM=$sym timeout 600 python bench.py --no-cpu-baseline --precond $pc > gpurun_out/o/bench_${sym}_${pc}.json 2> gpurun_out/o/bench_${sym}_${pc}.err
  done
done
python - <<'PY'
import json
for sym in (0, 1):
    for pc in ("multilevel", "jacobi"):
        try:
            d = json.loads(open("gpurun_out/o/bench_%d_%s.json" % (sym, pc)).read().strip().splitlines()[-1])
        except Exception as e:
            print(sym, pc, "failed", e); continue
        r = d["roofline"]
        print("sym", sym, pc, "value %.2f" % d["value"], "hess rot %.1f us frac %.3f | warm %.1f us frac %.3f" % (
            r["avg_launch_us"], r["frac"], r["warm"]["avg_launch_us"], r["warm"]["frac"]))
        print("   kernels", json.dumps(r.get("kernels"))[:600])
        print("   also", json.dumps(d.get("also"))[:900])
PY


