#!/bin/bash
# round-5 GPU run 2: the whole GPU suite on the split library, then the iteration-graph A/B on the headline workload
export GPU_OUT=r5b
OUT=gpurun_out/$GPU_OUT
mkdir -p $OUT
bash tools/gpu_run.sh tests
tail -5 $OUT/tests.log
grep -E "^FAILED|^ERROR" $OUT/tests.log | head
for rep in 1 2; do for g in 1 0; do
  DPGO_ITER_GRAPH=$g timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-secondary 2>$OUT/ab_$g.err | tail -1 > $OUT/ab_${g}_$rep.json
  python - $OUT/ab_${g}_$rep.json $g <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print("ITER_GRAPH=%s: %.1f it/s  %.3f ms/step  %.1f us per product  products %.1f" % (sys.argv[2], j["value"], j["ms_per_step"], j["quality"]["us_per_tcg_iteration_rank0"], j["products_per_step"]))
PY
done; done
DPGO_PERSIST_VERBOSE=1 timeout 120 python -c "
import bench, dpgo_amd, torch
from dpgo_amd.agent import DeviceAgent, ExchangePlan, build_pose_graphs
meas, n, X0, desc = bench.make_workload('grid100k', 5)
ranges, graphs = build_pose_graphs(meas, n, 1, 5)
ag = DeviceAgent(graphs, ExchangePlan(graphs), 0, X0, dpgo_amd.ROptParameters())
ag.update(); ag.update()
print(ag.problem.describe())
" 2>&1 | grep -v amdgpu | head -60 > $OUT/describe.txt
head -12 $OUT/describe.txt
