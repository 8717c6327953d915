"""Only the rotating-operand launches of bench.py's roofline protocol (every operand of the launch cycles through
> 256 MB of private copies), so that a rocprofv3 --kernel-trace --stats of THIS command shows the HBM-only average
duration of k_tcg_hess_span / k_spmm by itself (in a profile of bench.py they are mixed with the in-loop launches).
usage: python tools/rotating_probe.py [reps]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
import dpgo_amd  # noqa: E402
import bench  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
r = 5
meas, n, X0, desc = bench.make_workload("grid100k", r)
pg = dpgo_amd.PoseGraph(0, r, meas.d)
pg.setMeasurements(meas)
prob = dpgo_amd.QuadraticProblem(pg)
lib = dpgo_amd.lib.load()
nnzb = len(pg.quadraticMatrix()[1])
d = meas.d
set_b = nnzb * (8 * (d + 1) ** 2 + 4) + 2 * 8 * r * (d + 1) * n
nsets = int(min(512, max(3, -(-3 * 256 * 2 ** 20 // max(set_b, 1)) // 2 + 1)))
hsets = max(3, nsets // 2 + 1)
# the storage / kernel instance bench.py's timed loop runs: one solve with the default preconditioner first, so that the
# working-set rule of the library (Q, vectors and what the preconditioner streams) has decided
X = torch.tensor(X0, device="cuda", dtype=torch.float64)
dpgo_amd.QuadraticOptimizer(prob, dpgo_amd.ROptParameters()).optimizeDevice(X)
prob.setSpmmVariant("auto")
ki = prob.tcgKernelInfo()
hsets += 1 if ki["symmetric"] else 0
nsets *= 2 if ki["symmetric"] else 1  # (the symmetric storage's sets are half the size: as many bytes in rotation as bench.py)
ms_h, ms_s, sb = C.c_double(0.0), C.c_double(0.0), C.c_double(0.0)
dpgo_amd.lib.check(lib.dpgo_bench_hess_rotating(prob.handle, hsets, reps, 10, C.byref(ms_h)))
dpgo_amd.lib.check(lib.dpgo_bench_spmm_rotating(prob.handle, nsets, reps, 10, C.byref(ms_s), C.byref(sb)))
hb, spb = bench.hess_bytes(n, nnzb, d, r), bench.spmm_bytes(n, nnzb, d, r)
print("tCG-step kernel instance: %s" % ki)
print("%s: k_tcg_hess rotating over %d sets: %.2f us = %.0f GB/s (%.3f of 8 TB/s); k_spmm rotating over %d sets: %.2f us = "
      "%.0f GB/s (%.3f)" % (desc, hsets, 1e3 * ms_h.value, hb / ms_h.value / 1e6, hb / ms_h.value / 1e6 / 8000, nsets,
                            1e3 * ms_s.value, spb / ms_s.value / 1e6, spb / ms_s.value / 1e6 / 8000))
