#!/bin/bash
# usage: tools/ab.sh <workload> lib1.so lib2.so ...   -- interleaved A/B of kernel builds on one box (3 rounds)
# (variant builds: make -C dpgo_amd/csrc OUT=../libdpgo_hip_X.so OBJDIR=build_X EXTRA=-DDPGO_...=...)
W=$1; shift
for rep in 1 2 3; do for L in "$@"; do
DPGO_LIB=$PWD/$L timeout 300 python bench.py --workload $W --steps 10 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^DETAIL {' | tail -1 | cut -c8- > /tmp/b.json
python - <<PY
import json
j=json.load(open("/tmp/b.json")); t=j["quality"]["tcg_iterations_per_step_rank0"]; rf=j["roofline"]
ks=" ".join("%s %.1f" % (k["kernel"].split()[0].replace("k_ml_","").replace("k_tcg_",""), k["avg_launch_us"]) for k in rf.get("kernels") or [])
print("rep $rep %-34s it/s %7.1f ms/step %6.3f hess us %5.2f spmm us %5.2f us/product %6.1f | %s | tail %.1f"%("$L",j["value"],j["ms_per_step"],rf["avg_launch_us"],(rf.get("spmm_symmetric") or rf["spmm_only"])["avg_launch_us"],1e3*j["ms_per_step"]/max(t,1),ks,rf.get("cycle_tail_us") or 0))
PY
done; done
