#!/bin/bash
# round-5 GPU run 3: fp32 operator copies of the cycle -- parity (switch test) and the A/B on the headline workload
export GPU_OUT=r5c
OUT=gpurun_out/$GPU_OUT
mkdir -p $OUT
bash tools/gpu_run.sh tests "kernel_selecting_switches"
tail -3 $OUT/tests.log
for rep in 1 2; do for b in 32 64; do
  DPGO_ML_OPERATOR_BITS=$b timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-secondary 2>$OUT/ab_$b.err | tail -1 > $OUT/ab_${b}_$rep.json
  python - $OUT/ab_${b}_$rep.json $b <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
ks = j["roofline"]["kernels"]
print("OPERATOR_BITS=%s: %.1f it/s  %.3f ms/step  %.1f us per product  products %.1f | %s" % (sys.argv[2], j["value"], j["ms_per_step"], j["quality"]["us_per_tcg_iteration_rank0"], j["products_per_step"], " ".join("%.1f" % k["avg_launch_us"] for k in ks)))
PY
done; done
DPGO_ML_OPERATOR_BITS=32 timeout 600 python bench.py --steps 100 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench32_full.json
python - <<'PY'
import json
j = json.load(open("gpurun_out/r5c/bench32_full.json"))
for k, v in j["quality"]["to_tolerance"].items():
    print(k, {a: v.get(a) for a in ("products", "ms", "hierarchy_setup_ms", "reached")})
print("value", j["value"], j["time_to_tolerance_ms"])
PY
