#!/bin/bash
mkdir -p gpurun_out/u
for sym in 0 1; do
  DPGO_SPMM_SYMMETRIC=$sym timeout 600 python bench.py --no-cpu-baseline --no-secondary --precond multilevel > gpurun_out/u/bench_${sym}.json 2> gpurun_out/u/bench_${sym}.err
done
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/u/bench_auto.json 2> gpurun_out/u/bench_auto.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/u/bench_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f, "value %.2f it/s  %.2f ms/step" % (d["value"], d["ms_per_step"]), "frac", round(r["frac"], 3), "warm", round(r["warm"]["frac"], 3), r.get("spmm_storage_selected"))
    for k in r["kernels"]:
        print("    %-90s %7.1f us  %.3f" % (k["kernel"][:90], k["avg_launch_us"], k["frac"]))
    if d.get("also"):
        print("    also", json.dumps(d["also"].get("time_to_tolerance"))[:1800])
PY
