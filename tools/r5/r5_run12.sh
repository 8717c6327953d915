#!/bin/bash
# round-5 GPU run 12: the symmetric gather issues the loads of NB blocks before the first FMA (DPGO_SYM_BATCH) -- A/B of
# library builds b<NB>w<waves per SIMD k_tcg_hess_sym is compiled for> on the headline and on torus3D
export GPU_OUT=r5l
OUT=gpurun_out/$GPU_OUT
mkdir -p $OUT
for rep in 1 2; do for L in libdpgo_hip.so libdpgo_b2w3.so libdpgo_b2w2.so libdpgo_b4w2.so; do
  DPGO_LIB=$PWD/dpgo_amd/$L timeout 400 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{' | tail -1 > $OUT/h_$L.$rep.json
  python - $OUT $L $rep <<'PY'
import json, sys
o, L, rep = sys.argv[1:4]
j = json.load(open("%s/h_%s.%s.json" % (o, L, rep)))
rf = j["roofline"]
print("%-18s %s %.1f it/s  %.3f ms/step  products/step %s | %s frac %.3f  %s" % (L, j["config"]["workload"], j["value"], j["ms_per_step"], j.get("products_per_step"), (rf.get("kernel") or "")[:24], rf["frac"], {k: rf[k] for k in rf if "us" in k}))
PY
done; done
