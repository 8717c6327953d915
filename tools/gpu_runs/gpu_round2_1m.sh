#!/bin/bash
# 1M-pose grid with the final code: symmetric storage selected by size, three-level hierarchy
mkdir -p gpurun_out/p
for pc in jacobi multilevel; do
timeout 600 python bench.py --workload grid:100x100x100 --no-cpu-baseline --no-secondary --steps 3 --warmup 1 --spmm-reps 50 --precond $pc > gpurun_out/p/final_$pc.json 2> gpurun_out/p/final_$pc.err
python - <<PY
import json
d = json.loads(open("gpurun_out/p/final_$pc.json").read().strip().splitlines()[-1]); r = d["roofline"]
print("$pc", "value %.2f it/s %.2f ms/step" % (d["value"], d["ms_per_step"]), r.get("spmm_storage_selected"), "hess cold %.1f us %.3f (%s) plain %.1f | warm %.1f %.3f" % (r["avg_launch_us"], r["frac"], r.get("cold_kernel"), r["cold_plain_storage"]["avg_launch_us"], r["warm"]["avg_launch_us"], r["warm"]["frac"]), (r.get("multilevel") or {}).get("ks"), (r.get("multilevel") or {}).get("path"))
PY
done
