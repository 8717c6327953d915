#!/bin/bash
# fp32 storage of the coarsest inverse: parity tests of the multilevel path, then the bench
mkdir -p gpurun_out/r
timeout 1800 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "multilevel or hierarchy or default_precond or full_size or multi_agent" > gpurun_out/r/tests.log 2>&1; tail -6 gpurun_out/r/tests.log
for pc in auto multilevel; do
timeout 900 python bench.py --no-cpu-baseline --precond $pc > gpurun_out/r/bench_$pc.json 2> gpurun_out/r/bench_$pc.err
done
python - <<'PY'
import json
for pc in ("auto", "multilevel"):
    d = json.loads(open("gpurun_out/r/bench_%s.json" % pc).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(pc, "value %.2f it/s  %.2f ms/step" % (d["value"], d["ms_per_step"]), "frac", round(r["frac"], 3), "warm", round(r["warm"]["frac"], 3))
    for k in r["kernels"]:
        print("    %-90s %7.1f us  %.3f" % (k["kernel"][:90], k["avg_launch_us"], k["frac"]))
    print("    tail", r["cycle_tail_us"], "ml", r["multilevel"])
    print("    also", json.dumps(d.get("also"))[:1500])
PY
