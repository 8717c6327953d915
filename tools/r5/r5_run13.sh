#!/bin/bash
# round-5 GPU run 13: k_tcg_hess_sym's gather with the loads of 4 blocks in flight (DPGO_HESS_BATCH=4, 2 waves per SIMD):
# the symmetric-storage parity tests, then A/B against the previous library on the headline, --precond jacobi and torus3D
export GPU_OUT=r5m
OUT=gpurun_out/$GPU_OUT
mkdir -p $OUT
bash tools/gpu_run.sh tests "symmetric or whole_solve or auto_cost or switches or grid or torus"
tail -3 $OUT/tests.log
for rep in 1 2; do for L in libdpgo_hip.so libdpgo_prev.so; do
  for W in "h --steps 100" "j --steps 100 --precond jacobi" "t --workload torus3D --steps 60"; do
    set -- $W; tag=$1; shift
    DPGO_LIB=$PWD/dpgo_amd/$L timeout 400 python bench.py "$@" --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{' | tail -1 > $OUT/${tag}_$L.$rep.json
    python - $OUT/${tag}_$L.$rep.json $L <<'PY'
import json, sys
j = json.load(open(sys.argv[1])); L = sys.argv[2]
rf = j["roofline"]
print("%-18s %-28s %.1f it/s  %.3f ms/step  products/step %s | %s %.2f us frac %.3f" % (L, j["config"]["workload"][:28], j["value"], j["ms_per_step"], j.get("products_per_step"), (rf.get("kernel") or "")[:22], rf.get("avg_launch_us") or 0, rf["frac"]))
PY
  done
done; done
