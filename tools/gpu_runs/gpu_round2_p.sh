#!/bin/bash
# 1M-pose grid (Q = 0.9 GB: really HBM-bound): plain block-CSR vs the symmetric storage the size switch selects
mkdir -p gpurun_out/p
for sym in 0 1; do
  for pc in jacobi multilevel; do
    DPGO_SPMM_SYMMETRIC=$sym timeout 900 python bench.py --workload grid:100x100x100 --no-cpu-baseline --no-secondary --steps 3 --warmup 1 --spmm-reps 50 --precond $pc > gpurun_out/p/bench_${sym}_${pc}.json 2> gpurun_out/p/bench_${sym}_${pc}.err
    echo "rc=$? sym=$sym $pc"; tail -2 gpurun_out/p/bench_${sym}_${pc}.err
  done
done
timeout 300 python bench.py --workload grid:100x100x100 --no-cpu-baseline --no-secondary --steps 3 --warmup 1 --spmm-reps 50 > gpurun_out/p/bench_auto.json 2> gpurun_out/p/bench_auto.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/p/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "failed", e); continue
    r = d["roofline"]
    print(f, "value %.3f it/s, %.1f ms/step" % (d["value"], d["ms_per_step"]), "selected", r.get("spmm_storage_selected"),
          "| hess cold %.1f us frac %.3f (%s) plain %.1f us | warm %.1f us frac %.3f" % (
              r["avg_launch_us"], r["frac"], r.get("cold_kernel"), r["cold_plain_storage"]["avg_launch_us"],
              r["warm"]["avg_launch_us"], r["warm"]["frac"]))
    print("   spmm plain", r["spmm_only"]["avg_launch_us"], r["spmm_only"]["frac"], "sym", (r.get("spmm_symmetric") or {}).get("avg_launch_us"), (r.get("spmm_symmetric") or {}).get("frac"))
    print("   config", json.dumps(d["config"])[:400])
PY
