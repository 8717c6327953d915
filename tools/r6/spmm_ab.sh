#!/bin/bash
# usage: tools/r6/spmm_ab.sh lib1.so lib2.so ...  -- interleaved A/B of k_spmm_sym builds: rotating-operand and back-to-back time
for rep in 1 2 3; do for L in "$@"; do
DPGO_LIB=$PWD/$L timeout 300 python - <<PY
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import torch, dpgo_amd, bench
meas, n, X0, desc = bench.make_workload("grid100k", 5)
pg = dpgo_amd.PoseGraph(0, 5, 3); pg.setMeasurements(meas)
prob = dpgo_amd.QuadraticProblem(pg)
lib = dpgo_amd.lib.load()
assert prob.setSpmmVariant("symmetric") == "symmetric"
rot, warm, sb = C.c_double(0), C.c_double(0), C.c_double(0)
dpgo_amd.lib.check(lib.dpgo_bench_spmm_rotating(prob.handle, 6, 200, 10, C.byref(rot), C.byref(sb)))
dpgo_amd.lib.check(lib.dpgo_bench_spmm(prob.handle, 200, 10, C.byref(warm)))
print("rep $rep %-30s k_spmm_sym rotating %.2f us (%.3f of 8 TB/s on 123.08 MB, %.3f on its own %.1f MB)  back-to-back %.2f us" % (
    "$L", rot.value * 1e3, 123.084 / rot.value / 8000, sb.value / 1e6 / rot.value / 8000, sb.value / 1e6, warm.value * 1e3), flush=True)
PY
done; done
