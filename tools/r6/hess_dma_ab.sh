#!/bin/bash
# A/B of the LDS-DMA variants of the symmetric-storage tCG-step kernel (DPGO_HESS_DMA = 0 / 1 / 2 / 3: MODES="0 1 3"; one library, one box):
# rotating-operand launch time, back-to-back launch time, loop time per product.  usage: bash tools/r6/hess_dma_ab.sh [reps]
for rep in $(seq 1 ${1:-3}); do for mode in ${MODES:-0 1 2}; do
DPGO_HESS_DMA=$mode timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^DETAIL {' | tail -1 | cut -c8- > /tmp/b.json
python - <<PY
import json
j=json.load(open("/tmp/b.json")); t=j["quality"]["tcg_iterations_per_step_rank0"]; rf=j["roofline"]
print("rep $rep DPGO_HESS_DMA=$mode  hess rotating %.2f us (frac %.3f)  back-to-back %.2f us  us/product %.1f  it/s %.1f  grid %s" % (
    rf["avg_launch_us"], rf["frac"], rf["warm"]["avg_launch_us"], 1e3*j["ms_per_step"]/max(t,1), j["value"], j["config"].get("poses_per_agent")))
PY
done; done
