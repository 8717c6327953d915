#!/bin/bash
# One parametrised GPU-box script (replaces the per-call scripts of earlier rounds).  From the repo root on the box:
#   gpurun --timeout S -- 'bash tools/gpu_run.sh <task> [args] [-- <task> [args]] ...'
# Tasks (each writes under gpurun_out/<out>/, prints a short summary):
#   tests [k-expr]          pytest -m gpu (optionally -k <expr>)
#   smoke                   __graft_entry__.smoke()
#   bench <name> [args...]  python bench.py args...  -> gpurun_out/<out>/<name>.json
#   probe <name> [ENV=val ...] names...   tools/ml_probe.py (per-product cost: multilevel / jacobi / persistent)
#   profile <tag>           tools/profile_round.sh + tools/profile_extra.sh (rocprofv3 summaries for profiles/)
#   py <name> <script> [args...]   any python script under tools/
# GPU_OUT=<dir> selects the output directory name (default: run).
OUT=gpurun_out/${GPU_OUT:-run}
mkdir -p "$OUT"
run_task() {
  local task=$1; shift
  case "$task" in
    tests)
      if [ -n "$1" ]; then timeout 1500 python -m pytest tests -q -m gpu -x -k "$1" > "$OUT/tests.log" 2>&1
      else timeout 1500 python -m pytest tests -q -m gpu -x > "$OUT/tests.log" 2>&1; fi
      grep -E "passed|failed|error|Error" "$OUT/tests.log" | tail -8 ;;
    smoke)
      python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2 ;;
    bench)
      local name=$1; shift
      timeout 1200 python bench.py "$@" > "$OUT/$name.out" 2> "$OUT/$name.err"
      grep '^{' "$OUT/$name.out" | tail -1 > "$OUT/$name.json"
      grep '^DETAIL {' "$OUT/$name.out" | tail -1 | cut -c8- > "$OUT/${name}_detail.json"
      python - "$OUT/$name.json" "$OUT/${name}_detail.json" <<'PY'
import json, sys
try:
    line = open(sys.argv[1]).read()
    json.loads(line)
    j = json.load(open(sys.argv[2]))
    print("bench line: %d bytes" % len(line.strip()))
except Exception as e:
    print("bench: no JSON line (%r)" % (e,)); sys.exit(0)
rf, q = j.get("roofline") or {}, j.get("quality") or {}
print("bench %s: %.2f it/s  %.3f ms/step  products/step %s  agents %s  exch %.3f ms  | %s frac %.3f warm %.3f" % (
    sys.argv[1], j["value"], j["ms_per_step"], j.get("products_per_step"), j["config"].get("agents"),
    q.get("exchange_ms_per_step_rank0") or 0.0, (rf.get("kernel") or "")[:28], rf.get("frac") or 0, (rf.get("warm") or {}).get("frac") or 0))
cb = j.get("cpu_baseline") or {}
if cb.get("gpu_same_work"):
    g = cb["gpu_same_work"]
    print("  same work: cpu %.3f it/s (gn %.1f) | gpu %.1f it/s (gn %.1f, %d products)" % (
        cb["value"], cb.get("gradnorm_after", 0), g["value"], g["gradnorm_after"], g["tcg_iterations"]))
for k, v in (q.get("to_tolerance") or {}).items():
    print("  to_tol %-40s %s" % (k, {a: v.get(a) for a in ("products", "ms", "reached")}))
PY
      tail -3 "$OUT/$name.err" | cut -c1-300 ;;
    probe)
      local name=$1; shift
      local envs=()
      while [ $# -gt 0 ] && [[ "$1" == *=* ]]; do envs+=("$1"); shift; done
      env "${envs[@]}" timeout 900 python tools/ml_probe.py "$@" > "$OUT/$name.log" 2>&1
      grep -v "^dpgo_hip" "$OUT/$name.log" | cut -c1-400 | tail -40 ;;
    profile)
      bash tools/profile_round.sh "$1" > "$OUT/profile_round.log" 2>&1
      bash tools/profile_extra.sh "$1" > "$OUT/profile_extra.log" 2>&1
      tail -5 "$OUT/profile_round.log" | cut -c1-300 ;;
    py)
      local name=$1; shift
      timeout 900 python "$@" > "$OUT/$name.log" 2>&1
      tail -40 "$OUT/$name.log" | cut -c1-400 ;;
    *) echo "unknown task $task" ;;
  esac
}
args=()
for a in "$@"; do
  if [ "$a" == "--" ]; then run_task "${args[@]}"; args=(); else args+=("$a"); fi
done
[ ${#args[@]} -gt 0 ] && run_task "${args[@]}"
