#!/bin/bash
# round-5 GPU run 14: where the symmetric storage starts to win under --precond jacobi (working set of the loop against the
# 256 MB Infinity Cache): grids of 27k .. 100k poses, DPGO_SPMM_SYMMETRIC / DPGO_STREAM_NT = auto, 0, 1
export GPU_OUT=r5n
OUT=gpurun_out/$GPU_OUT
mkdir -p $OUT
for G in 30x30x30 40x40x25 40x40x40 50x40x40 50x50x40; do for M in auto 0 1; do
  if [ $M = auto ]; then E=""; else E="DPGO_SPMM_SYMMETRIC=$M DPGO_STREAM_NT=$M"; fi
  env $E timeout 400 python bench.py --workload grid:$G --precond jacobi --steps 40 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{' | tail -1 > $OUT/j_${G}_$M.json
  python - $OUT/j_${G}_$M.json $G $M <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
rf = j["roofline"]
print("grid %-9s sym/nt %-4s %.1f it/s  %.3f ms/step  products/step %s  %.2f us/product | %s %.2f us" % (sys.argv[2], sys.argv[3], j["value"], j["ms_per_step"], j.get("products_per_step"), 1e3 * j["ms_per_step"] / max(j.get("products_per_step") or 1, 1), (rf.get("kernel") or "")[:24], rf.get("avg_launch_us") or 0))
PY
done; done
