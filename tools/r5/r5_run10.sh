#!/bin/bash
# round-5 GPU run 10: the dense level in fp32 with the rest of the cycle's storage -- parity and A/B
export GPU_OUT=r5j
OUT=gpurun_out/$GPU_OUT
mkdir -p $OUT
bash tools/gpu_run.sh tests "kernel_selecting_switches or whole_solve_at_full_size or bench_prints_one_valid"
tail -3 $OUT/tests.log
for rep in 1 2 3; do for b in 32 64; do
  DPGO_ML_DENSE_BITS=$b timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-secondary 2>$OUT/ab_$b.err | grep '^{' | tail -1 > $OUT/ab_${b}_$rep.json
  python - $OUT/ab_${b}_$rep.json $b <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
ks = j["roofline"]["kernels"]
print("DENSE_BITS=%s: %.1f it/s  %.3f ms/step  %.1f us per product | %s | %s" % (sys.argv[2], j["value"], j["ms_per_step"], j["quality"]["us_per_tcg_iteration_rank0"], " ".join("%.1f" % k["avg_launch_us"] for k in ks), j["config"]["cycle_storage"][-60:]))
PY
done; done
