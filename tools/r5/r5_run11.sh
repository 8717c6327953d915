#!/bin/bash
# round-5 GPU run 11: the one-launch solve reports through its commit kernel (no read-back copies): tests, A/B against the
# previous library on the small-block regimes
export GPU_OUT=r5k
OUT=gpurun_out/$GPU_OUT
mkdir -p $OUT
bash tools/gpu_run.sh tests "persistent or additive or stream_ordered or concurrent or multi_agent_rbcd or salt_wraparound or greedy or inactive"
tail -3 $OUT/tests.log
for rep in 1 2 3; do for L in libdpgo_hip.so libdpgo_prev.so; do
  DPGO_LIB=$PWD/dpgo_amd/$L timeout 300 python bench.py --workload sphere2500 --steps 300 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{' | tail -1 > $OUT/s_$L.$rep.json
  DPGO_LIB=$PWD/dpgo_amd/$L timeout 300 python bench.py --loopback --agents-per-gpu 16 --sequential --precond jacobi --steps 30 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{' | tail -1 > $OUT/q_$L.$rep.json
  DPGO_LIB=$PWD/dpgo_amd/$L timeout 300 python bench.py --loopback --agents-per-gpu 16 --precond jacobi --steps 60 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{' | tail -1 > $OUT/c_$L.$rep.json
  python - $OUT $L $rep <<'PY'
import json, sys
o, L, rep = sys.argv[1:4]
s = json.load(open("%s/s_%s.%s.json" % (o, L, rep))); q = json.load(open("%s/q_%s.%s.json" % (o, L, rep))); c = json.load(open("%s/c_%s.%s.json" % (o, L, rep)))
print("%-18s sphere2500 %.0f it/s (%.3f ms)   16 x 6250 sequential %.3f ms   concurrent %.3f ms" % (L, s["value"], s["ms_per_step"], q["ms_per_step"], c["ms_per_step"]))
PY
done; done
