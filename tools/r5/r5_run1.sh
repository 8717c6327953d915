#!/bin/bash
# round-5 GPU run 1: launch-cost lab, the new selection tests, loop-back sweeps per preconditioner, the default bench line
export GPU_OUT=r5a
OUT=gpurun_out/$GPU_OUT
mkdir -p $OUT
( for g in 768 1563 512 256; do timeout 120 tools/launch_lab $g 300; echo; done ) > $OUT/launch_lab.txt 2>&1
tail -32 $OUT/launch_lab.txt
bash tools/gpu_run.sh tests "auto_cost_rule or default_preconditioner_selection or bench_secondary or bench_loopback or additive_preconditioner_selection"
for pc in auto jacobi additive; do
  bash tools/gpu_run.sh bench lb16_$pc --loopback --agents-per-gpu 16 --precond $pc --no-cpu-baseline --no-secondary --steps 60 --warmup 3
  bash tools/gpu_run.sh bench lb16seq_$pc --loopback --agents-per-gpu 16 --sequential --precond $pc --no-cpu-baseline --no-secondary --steps 30 --warmup 3
done
bash tools/gpu_run.sh bench default --steps 200
python - <<'PY'
import json
j=json.load(open("gpurun_out/r5a/default.json"))
cb=j["cpu_baseline"]
print("value", j["value"], "ms", j["ms_per_step"], "ttt", j["time_to_tolerance_ms"], "setup", j["hierarchy_setup_ms"], j["hierarchy_values_only_ms"])
print("cpu settled", cb["value"], cb["tcg_iterations"], "gpu", cb["gpu_same_work"]["value"], cb["gpu_same_work"]["seconds_per_sweep"], cb["gpu_same_work"]["tcg_iterations"], cb["gpu_same_work"]["preconditioners"], cb["gpu_same_work"]["selection_sweeps"])
print("setup per block", cb["gpu_same_work"]["hierarchy_setup_ms"])
print("auto", [(a["state"],a["units_jacobi"],a["units_additive"],a["backoff"]) for a in cb["gpu_same_work"]["auto_rule"]])
print("also", {k:(v.get("it_per_s"),v.get("us_per_product"),v.get("in_kernel_us_per_iteration",{}) and v["in_kernel_us_per_iteration"].get("total")) for k,v in (j.get("also") or {}).items()})
print("us/tcg", j["quality"]["us_per_tcg_iteration_rank0"])
PY
