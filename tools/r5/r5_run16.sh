#!/bin/bash
# round-5 GPU run 16: k_tcg_hess_sym (batch 4, 2 waves per SIMD, 512 resident workgroups, 1 563 tiles): grid sizes against the
# tile quantization (3.05 tiles per workgroup)
export GPU_OUT=${GPU_OUT:-r5p}
OUT=gpurun_out/$GPU_OUT
mkdir -p $OUT
L=${1:-libdpgo_hip.so}
for G in 0 392 448 480 512 528 784 1563; do
  DPGO_GRID_HESS_SYM=$G DPGO_LIB=$PWD/dpgo_amd/$L timeout 400 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{' | tail -1 > $OUT/g_$G.json
  python - $OUT/g_$G.json $G <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
rf = j["roofline"]
print("grid %-5s %.1f it/s  %.3f ms/step | %s %.2f us frac %.3f warm %.2f us" % (sys.argv[2], j["value"], j["ms_per_step"], (rf.get("kernel") or "")[:22], rf.get("avg_launch_us") or 0, rf["frac"], (rf.get("warm") or {}).get("avg_launch_us") or 0))
PY
done
