#!/bin/bash
# round-5 GPU run 4: the breadth-first tile walk of the symmetric-storage kernels: parity, time (rotating probe, bench A/B), PMC
export GPU_OUT=r5d
R=$PWD
OUT=gpurun_out/$GPU_OUT
mkdir -p $OUT
bash tools/gpu_run.sh tests "kernel_selecting_switches or symmetric_storage or spmm_properties"
tail -3 $OUT/tests.log
for rep in 1 2; do for w in 1 0; do
  DPGO_TILE_WALK=$w timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-secondary 2>$OUT/ab_$w.err | tail -1 > $OUT/ab_${w}_$rep.json
  python - $OUT/ab_${w}_$rep.json $w <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
rf = j["roofline"]
print("TILE_WALK=%s: %.1f it/s  %.1f us per product | hess_sym rotating %.2f us warm %.2f | spmm_sym rot %.2f | kernels %s" % (
    sys.argv[2], j["value"], j["quality"]["us_per_tcg_iteration_rank0"], rf["avg_launch_us"], rf["warm"]["avg_launch_us"],
    (rf.get("spmm_symmetric") or {}).get("avg_launch_us", 0), " ".join("%.1f" % k["avg_launch_us"] for k in rf["kernels"])))
PY
done; done
cd /tmp && export TMPDIR=/tmp
for w in 1 0; do for c in FETCH_SIZE WRITE_SIZE; do
  DPGO_TILE_WALK=$w rocprofv3 --pmc $c --output-format csv -d $R/$OUT/pmc_${w}_$c -o b -- \
    python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --spmm-reps 5 > $R/$OUT/pmc_${w}_$c.log 2>&1
done
cd $R
python tools/summarize_prof.py pmc $OUT/pmc_walk$w.json FETCH_SIZE=$OUT/pmc_${w}_FETCH_SIZE WRITE_SIZE=$OUT/pmc_${w}_WRITE_SIZE
rm -rf $OUT/pmc_${w}_FETCH_SIZE $OUT/pmc_${w}_WRITE_SIZE
python - $OUT/pmc_walk$w.json $w <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
for k in j["FETCH_SIZE_KB"]:
    if any(s in k for s in ("tcg_hess_sym", "spmm_sym", "ml_restrict", "k_grad", "k_hess<")):
        f, w = j["FETCH_SIZE_KB"][k]["max"], j["WRITE_SIZE_KB"].get(k, {}).get("max", 0)
        print("walk=%s %-60s fetch x2 + write = %.1f MB" % (sys.argv[2], k[:60], (2 * f + w) * 1024 / 1e6))
PY
cd /tmp
done
