#!/bin/bash
# symmetric-storage product: parity tests + bench line with plain / symmetric rotating figures
mkdir -p gpurun_out/n
timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "symmetric_storage or spmm_properties" > gpurun_out/n/tests.log 2>&1
tail -5 gpurun_out/n/tests.log
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/n/bench.json 2> gpurun_out/n/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/n/bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value", d["value"], "frac", r["frac"], "warm", r["warm"]["frac"])
print("plain", {k: r["spmm_only"][k] for k in ("avg_launch_us", "frac")}, "warm", r["spmm_only"]["warm"])
print("sym", r.get("spmm_symmetric"))
print("selected", r.get("spmm_storage_selected"))
PY
tail -3 gpurun_out/n/bench.err
