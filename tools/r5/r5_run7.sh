#!/bin/bash
# round-5 GPU run 7: the cycle's internal vectors in fp32 -- parity and A/B
export GPU_OUT=r5g
OUT=gpurun_out/$GPU_OUT
mkdir -p $OUT
bash tools/gpu_run.sh tests "kernel_selecting_switches or whole_solve_at_full_size"
tail -3 $OUT/tests.log
for rep in 1 2 3; do for b in 32 64; do
  DPGO_ML_VECTOR_BITS=$b timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-secondary 2>$OUT/ab_$b.err | grep '^{' | tail -1 > $OUT/ab_${b}_$rep.json
  python - $OUT/ab_${b}_$rep.json $b <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
ks = j["roofline"]["kernels"]
print("VECTOR_BITS=%s: %.1f it/s  %.3f ms/step  %.1f us per product  products %.1f | %s" % (sys.argv[2], j["value"], j["ms_per_step"], j["quality"]["us_per_tcg_iteration_rank0"], j["products_per_step"], " ".join("%.1f" % k["avg_launch_us"] for k in ks)))
PY
done; done
